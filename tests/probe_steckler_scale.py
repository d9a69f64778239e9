"""The REAL physics of cases/steckler (janaf thermo, LES kEqn, EDC, fvDOM by GAMG, the burner / baffle conditions) through the reference's
unchanged equation files at sizes towards the metric's (the 30 x 15 x 20 room refined r x r x r): start-up and time steps on the device,
wall time per step.  No oracle run at these sizes; the checks are the ones of test_the_case_at_the_size_of_baseline_config_2.
usage: tests/probe_steckler_scale.py r [steps] [tail-file: wall time of the last step in ms, for scripts/step_breakdown.py --tail-ms]      (lives under tests/ because the mesh comes from the oracle's mesh builder)"""
import ctypes as C, os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")      # the oracle's OpenMP workers (mesh builder) must not spin next to the HIP runtime's threads
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from ffm_import import ffm
import test_steckler_case_gpu as TS

r = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = ffm.Context(0)
t0 = time.time()
K_ = TS._case(ffm, ctx, True, refine=r, tileCells=16)
lib, m, N, cOrd, A, mesh, G, cs, out, nit, res, nm, sp = (K_[k] for k in ("lib", "m", "N", "cOrd", "A", "mesh", "G", "cs", "out", "nit", "res", "nm", "sp"))
print("refine %d: %d cells, sweep mode %d, set-up (oracle mesh + device tables + GAMG agglomeration) %.1f s" % (r, N, A.sweep_mode, time.time() - t0), flush=True)
os.environ["FFM_FOAM_QUIET"] = "1"
dp = C.POINTER(C.c_double)
nS = C.c_int()
t0 = time.perf_counter()
S = lib.firefoam_steckler_create(ctx.h, A.h, mesh.h, C.byref(cs), C.byref(nS))
print("start-up (hydrostatic initialisation, %d solves, iterations %s): %.2f s" % (nS.value, list(nit[:nS.value]), time.perf_counter() - t0), flush=True)
co = np.zeros(2)
if os.environ.get("PROBE_SLEEP"):
    time.sleep(float(os.environ["PROBE_SLEEP"]))
dt = 1.0 / 15.0 / max(1, r // 4)
for k in range(steps):
    lib.firefoam_steckler_set_delta_t(S, dt)
    t0 = time.perf_counter()
    n = lib.firefoam_steckler_advance(S, C.byref(cs), 1 if k == steps - 1 else 0)
    ms = 1e3 * (time.perf_counter() - t0)
    names = [nm.raw[16 * i:16 * i + 16].split(b"\0")[0].decode() for i in range(n)]
    its = {nme: nit[i] for i, nme in enumerate(names) if not nme.startswith("ILambda")}
    assert all(np.isfinite(res[:2 * n]))
    lib.firefoam_steckler_courant(S, co.ctypes.data_as(dp))
    print("step %d%s: %.0f ms, deltaT %.4g, Courant max %.3f, %d solves, iterations %s" % (k, " (+ download of every field to the host)" if k == steps - 1 else "", ms, dt, co[1], n, its), flush=True)
    dt = min(dt * min(0.9 / (co[1] + 1e-15), 1.2), 0.1)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("%.3f\n" % ms)          # (the last step: its trace ends with the downloads)
back = lambda a: (lambda q: (q.__setitem__(cOrd, a), q)[1])(np.empty(N))
Y = np.stack([back(y) for y in out["Y"]]); T = back(out["T"])
print("species sum - 1: %.1e, Y min %.1e, T in [%.2f, %.2f]" % (np.abs(Y.sum(axis=0) - 1.0).max(), Y.min(), T.min(), T.max()))
assert np.abs(Y.sum(axis=0) - 1.0).max() < 1e-12 and Y.min() >= 0.0 and 298.0 < T.min()
lib.firefoam_steckler_destroy(S)
