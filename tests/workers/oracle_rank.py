"""One rank of a decomposed ORACLE solve over gloo (CPU only): the rank-local C solver of oracle/ffo_solvers.c with
its processor-interface exchange and reductions routed through torch.distributed.
usage: oracle_rank.py rank world port gx gy gz bx by bz solver precond outdir"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O  # noqa: E402
from ffm_import import ffm  # noqa: E402   (host-side decomposition tooling only; no GPU is touched)
gloo_comm = ffm.gloo_comm

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
glob = tuple(int(v) for v in sys.argv[4:7]); grid = tuple(int(v) for v in sys.argv[7:10])
solver, precond, outdir = sys.argv[10], sys.argv[11], sys.argv[12]
gloo_comm.init(rank, world, port)
H = ffm.hexmesh
blocks, nbrRank = H.decompose(glob, grid)
blk = blocks[rank]
s = H.synth_p_rgh(blk)
A = O.Ldu(blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"])
A.set_interfaces([i["faceCells"] for i in s["interfaces"]], s["bouCoeffs"])
A.set_global_cells(glob[0] * glob[1] * glob[2])
ranks = nbrRank[rank]


def _allreduce(user, vals, n):
    a = np.ctypeslib.as_array(vals, shape=(n,))
    gloo_comm.allreduce(a, 0)


def _exchange(user, nIf, size, send, recv):
    sizes = [size[p] for p in range(nIf)]
    offs = np.concatenate(([0], np.cumsum(sizes)))[:-1].tolist()
    sb = np.concatenate([np.ctypeslib.as_array(send[p], shape=(sizes[p],)) for p in range(nIf)]) if nIf else np.zeros(0)
    rb = np.empty_like(sb)
    gloo_comm.exchange(sizes, ranks, offs, sb, rb)
    for p in range(nIf):
        np.ctypeslib.as_array(recv[p], shape=(sizes[p],))[:] = rb[offs[p]:offs[p] + sizes[p]]


cb = (O.ALLREDUCE_FN(_allreduce), O.EXCHANGE_FN(_exchange))
A.comm = O.Comm(None, rank, world, cb[0], cb[1])
psi, perf = A.solve(getattr(O, solver), getattr(O, precond), np.zeros(blk.nCells), s["source"], tolerance=1e-12)
np.savez(os.path.join(outdir, "rank%d.npz" % rank), psi=psi, gcell=blk.gcell, nIter=perf["nIterations"],
         initialResidual=perf["initialResidual"])
