"""PMC pass over one steady time step of the plume case: run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 scripts/pmc_step.py EDGE
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -- python3 scripts/pmc_step.py EDGE
(separate passes: TCC counter slots, MI355X_MICROARCH 'rocprofv3 PMC slots').  The calibration kernel k_reduce1<0> reads
exactly 8 N bytes (gfx950 reports half of wide coalesced reads in FETCH_SIZE; the summary doubles FETCH_SIZE by that calibration).
Then: scripts/pmc_summary.py FETCH.csv WRITE.csv -> per-kernel bytes of the LAST step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = ffm.Context(0)
case = ffm.Plume(ctx, (n, n, n))
for _ in range(steps):
    case.step()
N = case.nCells
x = ctx.to_device(ffm.hexmesh.hash_u(0xF4, np.arange(N)))
for _ in range(3):
    ctx.gSum(x)                      # calibration: reads 8*N bytes, 8 B per lane
print("N", N, "F", case.nFaces)
case.close(); ctx.close()
