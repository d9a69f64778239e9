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
occ = (C.c_int * 3)()
if hasattr(L, "ffm_debug_tile_occupancy") and L.ffm_debug_tile_occupancy(occ) == 0:
    print("runtime occupancy estimate (workgroups per CU): fwd %d bwd %d amul %d" % tuple(occ))
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
if os.environ.get("FFM_TRACE"):
    G = L.ffm_debug_tile_trace(A.h, None, 0)          # switch tracing on
    if G > 0:
        for which in ("bwd",):                        # the trace holds the last launch of an apply = the backward sweep
            apply(); ctx.sync()
            buf = np.zeros(4 * G, np.uint64)
            L.ffm_debug_tile_trace(A.h, buf.ctypes.data_as(C.c_void_p), 4 * G)
            tr = buf.reshape(G, 4).astype(np.float64)
            t0_ = tr[:, 0].min()
            st, fe, en, sp = (tr[:, 0] - t0_) / 100.0, (tr[:, 1] - t0_) / 100.0, (tr[:, 2] - t0_) / 100.0, tr[:, 3]
            print("  trace %s: G=%d  kernel span %.1f us; group start min/median/max %.1f/%.1f/%.1f us; first-entry wait median %.1f max %.1f us;"
                  " run time (end-first entry) min/median/max %.1f/%.1f/%.1f us; re-loads total %d (max per group %d)"
                  % (which, G, en.max(), st.min(), np.median(st), st.max(), np.median(fe - st), (fe - st).max(),
                     (en - fe).min(), np.median(en - fe), (en - fe).max(), int(sp.sum()), int(sp.max())))
            ev = sorted([(t, 1) for t in st] + [(t, -1) for t in en])
            cur = peak = 0
            for _, d in ev:
                cur += d; peak = max(peak, cur)
            print("  resident groups: %d started within 5 us, peak concurrently running %d" % (int((st < 5).sum()), peak))
            order = np.argsort(st)
            for q in (0, G // 4, G // 2, 3 * G // 4, G - 1):
                g = order[q]
                print("    group %4d: start %.1f first %.1f end %.1f reloads %d" % (g, st[g], fe[g], en[g], sp[g]))
print("box %dx%dx%d N=%d levels=%d sweep=%s tile=%d: %.4f ms per apply (2 sweeps) = %.3f us/level/sweep, %.1f GB/s of 2x60 B/cell"
      % (nx, ny, nz, N, A.nLevels, os.environ.get("FFM_SWEEP", "levels"), T, ms, ms * 1e3 / 2 / max(A.nLevels, 1), 120.0 * N / ms / 1e6))
ctx.sync()
wh = w.cpu().numpy()
print("result: sum %.17g, xor of the bit patterns %016x" % (wh.sum(), int(np.bitwise_xor.reduce(wh.view(np.uint64)))))
