"""SURVEY 8(e) / VERDICT r1 item 5 on the device: the baffled steckler room and a randomly relabelled box, partitioned by the
product's RCB / graph-growing partitioners into 2 and 4 sub-domains, solved by the HIP library in its ghost-cell form
(ffm_ldu_create_ext + ffm_ldu_set_ghost_exchange + pair tags) -- one process per rank sharing cuda:0 through the host (gloo)
transport.  The assembled Amul equals the global operator (1e-14), the converged fields equal the serial oracle solve to 1e-10
at tolerance 1e-12 (block-Jacobi DIC / DILU differ from the serial preconditioner only in the iteration path)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import part_cases
from common import rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("meshName,partitioner,world,solver,precond,asym", [
    ("steckler", "rcb", 4, "PCG", "DIC", 0.0), ("steckler", "graph", 2, "PBICGSTAB", "DILU", 0.3),
    ("dag_random", "graph", 4, "PCG", "DIC", 0.0), ("dag_random", "rcb", 2, "SMOOTH", "SYMGS", 0.3)])
def test_partitioned_mesh_on_the_device(O, ffm, ctx, meshName, partitioner, world, solver, precond, asym):
    N, l, u, centres, diag, up, lo, source = part_cases.build(O, meshName, asym)
    Ao = O.Ldu(N, l, u).set_coeffs(diag, up, lo)
    ref, perf = Ao.solve(getattr(O, solver), getattr(O, precond), np.zeros(N), source, tolerance=1e-12, maxIter=1000)
    yref = Ao.amul(O.hash_u(0xF4, np.arange(N)))
    port = 29300 + (os.getpid() % 150) + 5 * world + (0 if meshName == "steckler" else 40)
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "workers", "part_rank.py"), "gpu", str(r), str(world), str(port),
                                   meshName, partitioner, solver, precond, str(asym), tmp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                 for r in range(world)]
        try:
            outs = [p.communicate(timeout=120) for p in procs]
        finally:
            for p in procs:                      # a rank that failed leaves the others waiting in gloo: never leave them behind
                if p.poll() is None:
                    p.kill()
        assert [p.returncode for p in procs] == [0] * world, [o[1][-800:] for o in outs]
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
        ys = [np.load(os.path.join(tmp, "amul%d.npy" % r)) for r in range(world)]
    assert len({int(p["nIter"]) for p in parts}) == 1
    full, y = np.empty(N), np.empty(N)
    for p, yy in zip(parts, ys):
        full[p["gcell"]] = p["psi"]; y[p["gcell"]] = yy
    assert rel_l2(y, yref) < 1e-14
    assert rel_l2(full, ref) < (1e-10 if solver != "SMOOTH" else 1e-9)
