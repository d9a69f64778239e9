"""Host-side communicator over torch.distributed (gloo): the callbacks of ffm_comm_init_host (several ranks sharing one GPU,
the fall-back transport of bench.py when no RCCL communicator can be made) and of the oracle's ffo_comm, on CPU tensors."""
import numpy as np


GROUP = None        # process group to use (None: the default group); bench.py sets a gloo group here when it falls back from RCCL
_SEQ = [0]          # running message tag: every rank makes the same sequence of exchange calls, so call number k uses tag k on all of
                    # them (back-to-back exchanges between the same pair -- the ghost refresh of gx, gy, gz -- cannot be confused,
                    # and a rank that leaves the common sequence blocks instead of silently pairing the wrong messages)


def _tag():
    _SEQ[0] = (_SEQ[0] + 1) % 30000
    return _SEQ[0]


def _dist():
    import torch.distributed as dist
    return dist


def init(rank, world, port):
    dist = _dist()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)


def allreduce(vals, op=0):
    import torch
    dist = _dist()
    t = torch.from_numpy(np.array(vals, dtype=np.float64, copy=True))
    dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[op], group=GROUP)
    vals[:] = t.numpy()


def exchange(sizes, ranks, offs, send, recv):
    """pairwise exchange, equal counts: lower rank sends first to avoid head-of-line deadlock with blocking ops"""
    import torch
    dist = _dist()
    reqs, bufs = [], []
    tag = _tag()
    for n, r, o in zip(sizes, ranks, offs):
        if n == 0:
            continue
        ts = torch.from_numpy(np.ascontiguousarray(send[o:o + n]).copy())
        tr = torch.empty(n, dtype=torch.float64)
        reqs.append(dist.isend(ts, r, group=GROUP, tag=tag)); reqs.append(dist.irecv(tr, r, group=GROUP, tag=tag))
        bufs.append((o, n, tr, ts))
    for q in reqs:
        q.wait()
    for o, n, tr, _ in bufs:
        recv[o:o + n] = tr.numpy()


def exchange_var(ranks, sends, recvs):
    """one message each way per neighbour, any counts: all sends and receives posted, then waited for"""
    import torch
    dist = _dist()
    reqs, keep = [], []
    tag = _tag()
    for r, s, rv in zip(ranks, sends, recvs):
        if len(s):
            ts = torch.from_numpy(np.ascontiguousarray(s).copy()); keep.append(ts)
            reqs.append(dist.isend(ts, r, group=GROUP, tag=tag))
        if len(rv):
            tr = torch.empty(len(rv), dtype=torch.float64); keep.append((tr, rv))
            reqs.append(dist.irecv(tr, r, group=GROUP, tag=tag))
    for q in reqs:
        q.wait()
    for k in keep:
        if isinstance(k, tuple):
            k[1][:] = k[0].numpy()
