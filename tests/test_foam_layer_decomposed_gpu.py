"""SURVEY 8(b)+(e): the source-level Foam layer (include/ffmFoam.H) on a DECOMPOSED mesh.  examples/b1_demo.C -- rhoEqn, one
YiEqn with relax, UEqn with LUST and the reconstructed buoyancy / pressure source, one pEqn corrector with constrainHbyA,
constrainPressure, ddtCorr, flux and the velocity correction -- runs unchanged on every rank's sub-domain: processor boundaries are
ghost cells + cut faces (ffm_ldu_create_ext), each operator that gathers neighbour-cell values refreshes the ghost entries of its
result (fvMesh::haloRefresh, OpenFOAM's processorFvPatchField evaluate), reductions run over owned cells, the solvers exchange
and all-reduce through the communicator.  2 and 3 ranks (RCB and graph-growing partitions of the product's partitioner, unequal
message sizes) share cuda:0 through the host (gloo) transport; every field on the owned cells against the single-rank run of the
same code (which tests/test_foam_layer_gpu.py compares with the oracle): to rounding for what involves no linear solve beyond the
diagonal one, 1e-8 after the Krylov solves (tolerance 1e-10; block-Jacobi DILU / DIC take a different iteration path)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import foam_case
from common import rel_l2, free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,partitioner", [(2, "rcb"), (3, "graph")])
def test_b1_demo_on_a_decomposed_mesh(O, ffm, ctx, world, partitioner):
    from oracle import plume
    n = (9, 8, 7)
    m = plume.make_mesh(n, h=0.1)
    I = foam_case.inputs(O, m)
    ref, cells, nit1 = foam_case.run_b1_demo(ffm, ctx, m, I)
    assert np.array_equal(cells, np.arange(m.nCells))
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "workers", "foam_rank.py"), str(r), str(world), str(port)] + [str(v) for v in n]
                                  + [partitioner, tmp], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in range(world)]
        try:
            outs = [p.communicate(timeout=150) for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        assert [p.returncode for p in procs] == [0] * world, [o[1][-1500:] for o in outs]
        parts = [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(world)]
    assert sorted(np.concatenate([p["cells"] for p in parts]).tolist()) == list(range(m.nCells))          # a partition
    assert all(int(p["nGhost"]) > 0 for p in parts)
    full = {k: np.full_like(v, np.nan) for k, v in ref.items()}
    for p in parts:
        for k in full:
            full[k][..., p["cells"]] = p[k]
    assert len({tuple(p["nit"].tolist()) for p in parts}) == 1                       # every rank saw the same (global) solver performance
    assert rel_l2(full["rho"], ref["rho"]) < 1e-15              # diagonal solve; fvc::div(phi) sums a cell's faces in the rank's own face order
    assert rel_l2(full["rAU"], ref["rAU"]) < 1e-14
    for k in ("Yi", "K"):
        assert rel_l2(full[k], ref[k]) < 1e-8, k
    for k in ("U", "HbyA", "Uc"):
        for c in range(3):
            assert rel_l2(full[k][c], ref[k][c]) < (1e-7 if k == "HbyA" else 1e-8), (k, c)
    assert np.linalg.norm(full["p"] - ref["p"]) / np.linalg.norm(ref["p"] - ref["p"].mean()) < 1e-7


@pytest.mark.parametrize("world,partitioner", [(2, "rcb"), (4, "graph")])
def test_fvdom_with_reflecting_walls_on_a_decomposed_mesh(O, ffm, ctx, world, partitioner):
    """The fvDOM handle (fvDOM::calculate's iteration, grey-diffusive walls with emissivities < 1 -- the rays are coupled through the walls
    of EVERY rank -- 16 rays, three iterations per call, two calls) on 2 and 4 ranks against the single-rank run of the same code
    (which tests/test_fvdom_gpu.py compares with the oracle): the wall fluxes qin are sums over rays of wall-face values only, the rays
    cross the rank boundaries through the ghost cells of fvm::div; every ray solved to 1e-12, so the block-Jacobi DILU's different
    iteration path does not show: intensities, G and qin to 1e-8."""
    from oracle import plume
    n = (10, 9, 8)
    m = plume.make_mesh(n, h=0.1)
    T, Tb, E, emis = foam_case.fvdom_inputs(m)
    (I1, G1, q1), cells, its1 = foam_case.run_b1_fvdom(ffm, ctx, m, T, Tb, E, emis)
    assert its1 == [3, 3] and np.array_equal(cells, np.arange(m.nCells))
    port = free_port()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "workers", "foam_rank.py"), str(r), str(world), str(port)] + [str(v) for v in n]
                                  + [partitioner, tmp, "fvdom"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in range(world)]
        try:
            outs = [p.communicate(timeout=150) for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        assert [p.returncode for p in procs] == [0] * world, [o[1][-1500:] for o in outs]
        parts = [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(world)]
    assert all(p["its"].tolist() == [3, 3] for p in parts) and all(int(p["nGhost"]) > 0 for p in parts)
    I = np.full_like(I1, np.nan); G = np.full_like(G1, np.nan)
    for p in parts:
        I[:, p["cells"]] = p["I"]; G[p["cells"]] = p["G"]
    for i in range(I.shape[0]):
        assert rel_l2(I[i], I1[i]) < 1e-8, (i, rel_l2(I[i], I1[i]))
    assert rel_l2(G, G1) < 1e-8
    for pt in m.patches:
        full = np.full(pt.size, np.nan)
        for p in parts:
            full[p["pos_" + pt.name]] = p["qin_" + pt.name]
        assert np.abs(full - q1[pt.name][1]).max() <= 1e-8 * max(np.abs(q1[pt.name][1]).max(), 1e-300), pt.name
