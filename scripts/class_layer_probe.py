"""Time (and, under rocprofv3, trace) the class-layer path on the bench's case: the compiled plume case's start state advanced by the
reference's unchanged solver/*.H over include/ffmFoam.H (libffm_refsnippets.so), next to the compiled driver's step.
usage: class_layer_probe.py EDGE [steps] [tail-file]   -- writes the wall time of the LAST class-layer step (ms) to tail-file, so that
scripts/step_breakdown.py --tail-ms can cut that step out of a kernel trace."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffm_import import ffm
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = ffm.Context(0)
case = ffm.Plume(ctx, (n, n, n), h=0.05, deltaT=1e-3)
snap = ffm.snippets.FromPlume(case)
slib = ffm.snippets.load()
os.environ["FFM_FOAM_QUIET"] = "1"
sync = lambda: (ctx.sync(), torch.cuda.synchronize())
if os.environ.get("PROBE_COMPILED", "1") == "1":
    case.step(); sync()
    t = time.perf_counter(); case.step(); case.step(); sync()
    print("compiled driver: %.2f ms per step" % ((time.perf_counter() - t) / 2 * 1e3), flush=True)
solver = slib.firefoam_snippets_create(ctx.h, case.ldu_handle(), case.mesh().h, C.byref(snap.cs))
slib.firefoam_snippets_advance(solver, C.byref(snap.cs), 0); sync()
for s in range(steps):
    t = time.perf_counter()
    nsol = slib.firefoam_snippets_advance(solver, C.byref(snap.cs), 0); sync()
    ms = (time.perf_counter() - t) * 1e3
    print("class layer step %d: %.2f ms, iterations %s" % (s, ms, list(snap.nit[:nsol])), flush=True)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("%.3f\n" % ms)
slib.firefoam_snippets_destroy(solver)
