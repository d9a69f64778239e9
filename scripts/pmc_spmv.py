"""PMC pass helper: builds the plume case, then runs (a) a calibration read of known size (k_reduce1<0>: 8 B per lane
streaming read of nCal doubles) and (b) `reps` launches of the Amul kernel k_rows<0,false,W>.  Run under
  rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d OUT -- python scripts/pmc_spmv.py EDGE
  rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d OUT -- python scripts/pmc_spmv.py EDGE
(separate passes: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH 'rocprofv3 PMC slots')."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = ffm.Context(0)
case = ffm.Plume(ctx, (n, n, n))
case.step()
L = ffm.lib()
N = L.ffm_ldu_ncells(case.ldu_handle())
x = ctx.to_device(ffm.hexmesh.hash_u(0xF4, np.arange(N)))
y = ctx.empty(N)
for _ in range(3):
    ctx.gSum(x)                      # calibration: reads 8*N bytes, 8 B per lane
ms = C.c_double()
L.ffm_bench_spmv(case.ldu_handle(), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 5, C.byref(ms))
print("N", N, "F", case.nFaces, "spmv ms", ms.value)
case.close(); ctx.close()
