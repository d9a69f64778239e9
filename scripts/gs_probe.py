"""Time symGaussSeidel sweeps (smoothSolver) on an n^3 box.  usage: gs_probe.py n [env FFM_SWEEP]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
H = ffm.hexmesh
ctx = ffm.Context(0)
for n in [int(a) for a in sys.argv[1:]]:
    blk = H.HexBlock((n, n, n)); s = H.synth_p_rgh(blk)
    cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u)
    l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, blk.nCells, l2, u2)
    A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd])
    b = ctx.to_device(s["source"][cOrd]); psi = ctx.zeros(blk.nCells)
    A.smooth(psi, b, nSweeps=1, smoother="symGaussSeidel"); ctx.sync()
    t0 = time.perf_counter()
    p = A.smooth(psi, b, nSweeps=10, smoother="symGaussSeidel"); ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    print("n=%d: symGaussSeidel %.3f ms per sweep pair (%s)" % (n, dt * 1e3, os.environ.get("FFM_SWEEP", "auto")))
    A.close()
