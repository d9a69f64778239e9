"""GAMG vs PCG on an n^3 synthetic p_rgh matrix: set-up time, solve time, V-cycles.   usage: gamg_probe.py n [tolerance]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
n = int(sys.argv[1]); tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
ctx = ffm.Context(0)
blk = H.HexBlock((n, n, n)); s = H.synth_p_rgh(blk)
d = blk.u.astype(np.int64) - blk.l
axis = np.where(d == 1, 0, np.where(d == n, 1, 2))
Sf = np.zeros((len(blk.l), 3)); Sf[np.arange(len(blk.l)), axis] = 0.05 * 0.05
t0 = time.perf_counter(); A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u); t1 = time.perf_counter()
G = ffm.GAMG(ctx, A, blk.l, blk.u, Sf=Sf); t2 = time.perf_counter()
print("n %d: ldu create %.1f s, gamg create %.1f s, %d coarse levels, coarsest %d cells" % (n, t1 - t0, t2 - t1, G.nLevels, G.level_size(G.nLevels)[0]))
dg, up = ctx.to_device(s["diag"]), ctx.to_device(s["upper"]); src = ctx.to_device(s["source"])
ctx.sync(); t0 = time.perf_counter(); G.set_matrix(dg, up); ctx.sync(); print("agglomerateMatrix (all levels) %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(2):
    psi = ctx.to_device(np.zeros(blk.nCells)); ctx.sync(); t0 = time.perf_counter()
    pf = G.solve(psi, src, smoother="GaussSeidel", tolerance=tol); ctx.sync(); tg = time.perf_counter() - t0
    print("GAMG GaussSeidel: %d V-cycles, %.1f ms (%.1f ms per cycle), final %.2e" % (pf["nIterations"], tg * 1e3, tg * 1e3 / max(pf["nIterations"], 1), pf["finalResidual"]))
B = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"])
for rep in range(2):
    ref = ctx.to_device(np.zeros(blk.nCells)); ctx.sync(); t0 = time.perf_counter()
    pk = B.solve(ref, src, "PCG", "DIC", tolerance=tol); ctx.sync(); tp = time.perf_counter() - t0
    print("PCG+DIC: %d iterations, %.1f ms" % (pk["nIterations"], tp * 1e3))
print("rel diff", float(np.linalg.norm(psi.cpu().numpy() - ref.cpu().numpy()) / np.linalg.norm(ref.cpu().numpy())))
