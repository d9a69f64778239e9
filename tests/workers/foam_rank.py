"""One rank of a decomposed run of examples/b1_demo.C (the Foam layer of include/ffmFoam.H on a sub-domain with ghost cells),
all ranks sharing cuda:0 through the host (gloo) transport.   usage: foam_rank.py rank world port nx ny nz partitioner outdir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ffm_import import ffm  # noqa: E402
from oracle import oracle as O, plume  # noqa: E402   (mesh builder and hash only)
import foam_case  # noqa: E402

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = tuple(int(v) for v in sys.argv[4:7]); partitioner, outdir = sys.argv[7], sys.argv[8]
gloo = ffm.gloo_comm
gloo.init(rank, world, port)
m = plume.make_mesh(n, h=0.1)
part = ffm.decompose.partition_rcb(m.C, world) if partitioner == "rcb" else ffm.decompose.partition_graph(m.nCells, m.l, m.u, world)
sub = ffm.decompose.SubDomain(m.nCells, m.l, m.u, part, world, rank)
ctx = ffm.Context(0)
ctx.comm_init_host(rank, world, gloo.allreduce, gloo.exchange, gloo.exchange_var)
if len(sys.argv) > 9 and sys.argv[9] == "fvdom":
    # the fvDOM handle (iteration + grey-diffusive walls) on this rank's sub-domain: tests/test_foam_layer_decomposed_gpu.py
    T, Tb, E, emis = foam_case.fvdom_inputs(m)
    (I, G, qp), cells, its = foam_case.run_b1_fvdom(ffm, ctx, m, T, Tb, E, emis, sub=sub, part=part)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), cells=cells, I=I, G=G, its=np.array(its), nGhost=sub.nGhost,
             **{"pos_" + k: v[0] for k, v in qp.items()}, **{"qin_" + k: v[1] for k, v in qp.items()})
    ctx.close()
    sys.exit(0)
res, cells, nit = foam_case.run_b1_demo(ffm, ctx, m, foam_case.inputs(O, m), sub=sub, part=part)
np.savez(os.path.join(outdir, "rank%d.npz" % rank), cells=cells, nit=np.array(nit), nGhost=sub.nGhost, **res)
ctx.close()
