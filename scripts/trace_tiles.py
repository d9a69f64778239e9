"""Print, for the backward sweep of an n^3 box, when each tile's first entry became ready (us), as a (k-tile, j-tile) matrix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
ctx = ffm.Context(0)
n = int(sys.argv[1])
blk = H.HexBlock((n, n, n)); s = H.synth_p_rgh(blk)
cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u)
l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
A = ffm.lduMatrix(ctx, blk.nCells, l2, u2)
A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd]); A.reciprocalD("DIC")
r = ctx.to_device(s["source"][cOrd]); w = ctx.empty(blk.nCells)
L = ffm.lib()
ap = lambda: L.ffm_precond_apply(A.h, 1, 0, C.c_void_p(r.data_ptr()), C.c_void_p(w.data_ptr()))
for _ in range(3): ap()
ctx.sync()
G = L.ffm_debug_tile_trace(A.h, None, 0)
for _ in range(2): ap()
ctx.sync()
buf = np.zeros(4 * G, np.uint64); L.ffm_debug_tile_trace(A.h, buf.ctypes.data_as(C.c_void_p), 4 * G)
tr = buf.reshape(G, 4).astype(np.float64); t0 = tr[:, 0].min()
T = (n + 15) // 16
first = ((tr[:, 1] - t0) / 100.0).reshape(T, T)       # groups are ranked (k-tile, j-tile) lexicographically by the topological heap
end = ((tr[:, 2] - t0) / 100.0).reshape(T, T)
np.set_printoptions(linewidth=250, precision=0, suppress=True)
print("first-entry-ready time (us), backward sweep:"); print(first)
print("end time (us):"); print(end)
d = np.abs(np.diff(first, axis=1)); print("mean hop in j: %.2f us  mean hop in k: %.2f us" % (d.mean(), np.abs(np.diff(first, axis=0)).mean()))
