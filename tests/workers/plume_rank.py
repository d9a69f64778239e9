"""One rank of a decomposed plume run sharing cuda:0 with the other ranks (host transport over gloo).
usage: plume_rank.py rank world port gx gy gz bx by bz nSteps outdir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
glob = tuple(int(v) for v in sys.argv[4:7]); grid = tuple(int(v) for v in sys.argv[7:10])
nSteps, outdir = int(sys.argv[10]), sys.argv[11]
from ffm_import import ffm  # noqa: E402
gloo_comm = ffm.gloo_comm
gloo_comm.init(rank, world, port)
ctx = ffm.Context(0)
ctx.comm_init_host(rank, world, gloo_comm.allreduce, gloo_comm.exchange)
lo, hi, nbr = ffm.hexmesh.block_of_rank(glob, grid, rank)
case = ffm.Plume(ctx, glob, lo=lo, hi=hi, nbrRank=nbr)
if os.environ.get("FFM_TEST_TIGHT"):
    case.set_tight(True)
iters = []
for _ in range(nSteps):
    case.step()
    iters.append([(n, p["nIterations"]) for n, p in case.solves()])
names = ["rho", "p", "p_rgh", "T", "Ux", "Uy", "Uz", "O2", "C3H8", "CO2", "ph_rgh"]
if os.environ.get("FFM_PLUME_RADIATION"):
    names += ["G", "I0", "I13", "I31"]
out = {name: case.field(name) for name in names}
np.savez(os.path.join(outdir, "rank%d.npz" % rank), lo=np.array(lo), hi=np.array(hi), iters=np.array(iters, dtype=object), **out)
case.close(); ctx.close()
