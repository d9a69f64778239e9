"""GPU probe: the tile / Amul plans on the baffled steckler room refined r x (config 2: r = 4) with FFM_VERBOSE=1, and PCG iteration times.
usage: FFM_VERBOSE=1 python scripts/room_probe.py [refine]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from ffm_import import ffm
from oracle import oracle as O, steckler
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
O.build()
m = steckler.build_mesh(refine=r)
l, u = m.l.astype(np.int32), m.u.astype(np.int32)
hint = ffm.tile_hint_from_centres(m.C.T.copy())
cOrd, fOrd = ffm.renumber_levels(m.nCells, l, u, groupHint=hint)
l2, u2, _ = ffm.hexmesh.apply_renumbering(m.nCells, l, u, cOrd, fOrd)
ctx = ffm.Context(0)
A = ffm.lduMatrix(ctx, m.nCells, l2, u2, groupHint=hint[cOrd])
print("cells", m.nCells, "sweep mode", A.sweep_mode, "levels", A.nLevels, flush=True)
import common
diag, up, _ = common.laplacian_like(O, m.nCells, l2, u2, seed=3, shift=1e-4)
A.set_coeffs(diag, up)
b = ctx.to_device(O.hash_u(9, np.arange(m.nCells)) - 0.5)
for rep in range(2):
    psi = ctx.zeros(m.nCells)
    ctx.sync(); t = time.time()
    pf = A.solve(psi, b, solver="PCG", preconditioner="DIC", tolerance=1e-10, relTol=0.0, maxIter=200)
    ctx.sync(); dt = time.time() - t
    print("PCG %d iterations, %.3f ms per iteration" % (pf["nIterations"], dt / max(pf["nIterations"], 1) * 1e3), flush=True)
