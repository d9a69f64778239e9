"""Host-side communicator over torch.distributed (gloo) for the multi-rank tests: implements the
callbacks of ffm_comm_init_host / the oracle's ffo_comm with CPU tensors."""
import numpy as np
import torch
import torch.distributed as dist


GROUP = None        # process group to use (None: the default group); bench.py sets a gloo group here when it falls back from RCCL


def init(rank, world, port):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)


def allreduce(vals, op=0):
    t = torch.from_numpy(np.array(vals, dtype=np.float64, copy=True))
    dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[op], group=GROUP)
    vals[:] = t.numpy()


def exchange(sizes, ranks, offs, send, recv):
    """pairwise exchange, equal counts: lower rank sends first to avoid head-of-line deadlock with blocking ops"""
    me = dist.get_rank()
    reqs, bufs = [], []
    for n, r, o in zip(sizes, ranks, offs):
        if n == 0:
            continue
        ts = torch.from_numpy(np.ascontiguousarray(send[o:o + n]).copy())
        tr = torch.empty(n, dtype=torch.float64)
        reqs.append(dist.isend(ts, r, group=GROUP)); reqs.append(dist.irecv(tr, r, group=GROUP))
        bufs.append((o, n, tr, ts))
    for q in reqs:
        q.wait()
    for o, n, tr, _ in bufs:
        recv[o:o + n] = tr.numpy()
