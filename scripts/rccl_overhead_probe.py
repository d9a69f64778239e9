"""Cost of routing every reduction of the plume step through ncclAllReduce (one-rank communicator) on one GPU.  usage: n"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffm_import import ffm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = {}
for mode in ("plain", "rccl"):
    if mode == "rccl":
        os.environ["FFM_FORCE_COMM"] = "1"
    ctx = ffm.Context(0)
    if mode == "rccl":
        ctx.comm_init_rccl(0, 1, ffm.Context.comm_unique_id()); del os.environ["FFM_FORCE_COMM"]
    P = ffm.Plume(ctx, (n, n, n))
    for _ in range(2):
        P.step()
    t0 = time.perf_counter()
    for _ in range(3):
        P.step()
    res[mode] = (time.perf_counter() - t0) / 3
    nsolve = sum(pf["nIterations"] for _, pf in P.solves())
    P.close(); ctx.close()
print("n=%d: step %.1f ms without communicator, %.1f ms with every reduction through ncclAllReduce (+%.1f ms, %d solver iterations per step)"
      % (n, 1e3 * res["plain"], 1e3 * res["rccl"], 1e3 * (res["rccl"] - res["plain"]), nsolve))
