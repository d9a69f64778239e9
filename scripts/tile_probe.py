"""Time the tiled kernels on an n^3 box: one DIC application (forward + backward sweep) and one symmetric Amul.
   usage: tile_probe.py n [reps]      env: FFM_TILE, FFM_TILE_EDGE_ORDER, FFM_TILE_ROW_ORDER, FFM_AMUL_SEG"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
ctx = ffm.Context(0)
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
blk = H.HexBlock((n, n, n))
s = H.synth_p_rgh(blk)
cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u)            # the library's own tiling of a detected box
l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
A = ffm.lduMatrix(ctx, blk.nCells, l2, u2)
assert A.native_order and A.sweep_mode == 2
A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd])
A.reciprocalD("DIC")
r = ctx.to_device(s["source"][cOrd]); w = ctx.empty(blk.nCells)
L = ffm.lib()
def apply():
    assert L.ffm_precond_apply(A.h, 1, 0, C.c_void_p(r.data_ptr()), C.c_void_p(w.data_ptr())) == 0, L.ffm_last_error()
for _ in range(3): apply()
ctx.sync(); t0 = time.perf_counter()
for _ in range(reps): apply()
ctx.sync(); ms = (time.perf_counter() - t0) / reps * 1e3
sp = C.c_double()
assert L.ffm_bench_spmv(A.h, C.c_void_p(r.data_ptr()), C.c_void_p(w.data_ptr()), reps, C.byref(sp)) == 0, L.ffm_last_error()
print("n %d edge_order %s: DIC apply %.3f ms, Amul %.3f ms" % (n, os.environ.get("FFM_TILE_EDGE_ORDER", "1"), ms, sp.value))
