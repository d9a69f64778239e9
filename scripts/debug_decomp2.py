import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffm_import import ffm
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
glob, grid, nSteps = (20, 36, 34), (2, 1, 1), int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["FFM_PLUME_TIGHT"] = "1"
ctx = ffm.Context(0)
ref = ffm.Plume(ctx, glob)
for _ in range(nSteps): ref.step()
world = 2; port = 29611
with tempfile.TemporaryDirectory() as tmp:
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "workers", "plume_rank.py"), str(r), str(world), str(port), *map(str, glob), *map(str, grid), str(nSteps), tmp], env=dict(os.environ)) for r in range(world)]
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    parts = [np.load(os.path.join(tmp, "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
nx, ny, nz = glob
for name in ["rho", "T", "Ux", "Uy", "Uz", "O2", "p_rgh"]:
    full = np.empty((nz, ny, nx))
    for pt in parts:
        lo, hi = pt["lo"], pt["hi"]
        full[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] = pt[name].reshape(hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0])
    b = ref.field(name).reshape(nz, ny, nx)
    d = np.abs(full - b)
    k, j, i = np.unravel_index(d.argmax(), d.shape)
    print("%-6s rel_l2 %.2e  max|d| %.2e at (i,j,k)=(%d,%d,%d)  |b|max %.2e ; mean|d| by i: %s" % (name, np.linalg.norm(full - b) / max(np.linalg.norm(b), 1e-300), d.max(), i, j, k, np.abs(b).max(), np.array2string(d.mean(axis=(0, 1)), precision=1)))
print([ (n, p["nIterations"]) for n, p in ref.solves()][:12])
