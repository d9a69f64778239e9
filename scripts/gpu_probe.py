"""First GPU visit: SpMV bandwidth and DIC-PCG timing at a few sizes (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
ctx = ffm.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [100, 200]:
    t0 = time.time()
    blk = H.HexBlock((n, n, n))
    s = H.synth_p_rgh(blk)
    T = int(os.environ.get("FFM_TILE", "16"))
    hint = (blk.j // T) + 10000 * (blk.k // T)          # 2-D tiles of cell columns (used by the tile/pipe sweeps)
    cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u, groupHint=hint)
    l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
    t1 = time.time()
    A = ffm.lduMatrix(ctx, blk.nCells, l2, u2, groupHint=hint[cOrd])
    A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd])
    t2 = time.time()
    N, F = blk.nCells, blk.nFaces
    x = ctx.to_device(s["x"][cOrd])
    ms = A.bench_Amul(x, reps=20)
    alg = 24 * N + 16 * F
    print("n=%d N=%d F=%d levels=%d native=%s  mesh %.1fs create %.1fs" % (n, N, F, A.nLevels, A.native_order, t1 - t0, t2 - t1))
    print("  Amul %.4f ms  -> %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (ms, alg / ms / 1e6, alg / ms / 1e6 / 80))
    b = ctx.to_device(s["source"][cOrd])
    for rep in range(2):
        psi = ctx.zeros(N)
        t3 = time.time()
        p = A.solve(psi, b, solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0.01)
        t4 = time.time()
        print("  DICPCG relTol 0.01: %d it, %.1f ms total, %.3f ms/it  res %.3e" % (p["nIterations"], (t4 - t3) * 1e3, (t4 - t3) * 1e3 / max(p["nIterations"], 1), p["finalResidual"]))
    psi = ctx.zeros(N)
    t3 = time.time()
    p = A.solve(psi, b, solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0.0)
    t4 = time.time()
    print("  DICPCG tol 1e-6: %d it, %.1f ms total, %.3f ms/it" % (p["nIterations"], (t4 - t3) * 1e3, (t4 - t3) * 1e3 / max(p["nIterations"], 1)))
    psi = ctx.zeros(N)
    t3 = time.time()
    p = A.solve(psi, b, solver="PCG", preconditioner="diagonal", tolerance=1e-6, relTol=0.0, maxIter=50)
    t4 = time.time()
    print("  diag-PCG 50 it cap: %d it, %.3f ms/it" % (p["nIterations"], (t4 - t3) * 1e3 / max(p["nIterations"], 1)))
    A.close()
    del x, b, psi
