"""Time one DIC preconditioner application (forward + backward sweep) on an nx*ny*nz box.
   usage: sweep_probe.py nx ny nz [reps]      env: FFM_SWEEP, FFM_TILE, FFM_TILE_KB"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
ctx = ffm.Context(0)
nx, ny, nz = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
blk = H.HexBlock((nx, ny, nz))
s = H.synth_p_rgh(blk)
T = int(os.environ.get("FFM_TILE", "16"))
hint = (blk.j // T) + 10000 * (blk.k // T)
cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u, groupHint=hint)
l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
A = ffm.lduMatrix(ctx, blk.nCells, l2, u2, groupHint=hint[cOrd])
A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd])
A.reciprocalD("DIC")
r = ctx.to_device(s["source"][cOrd])
w = ctx.empty(blk.nCells)
L = ffm.lib()
def apply():
    rc = L.ffm_precond_apply(A.h, 1, 0, C.c_void_p(r.data_ptr()), C.c_void_p(w.data_ptr()))
    assert rc == 0, L.ffm_last_error()
for _ in range(3):
    apply()
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    apply()
ctx.sync()
ms = (time.perf_counter() - t0) / reps * 1e3
N = blk.nCells
print("box %dx%dx%d N=%d levels=%d sweep=%s tile=%d: %.4f ms per apply (2 sweeps) = %.3f us/level/sweep, %.1f GB/s of 2x60 B/cell"
      % (nx, ny, nz, N, A.nLevels, os.environ.get("FFM_SWEEP", "levels"), T, ms, ms * 1e3 / 2 / max(A.nLevels, 1), 120.0 * N / ms / 1e6))
