import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffm_import import ffm
glob = (12, 16, 12); grid = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 1, 1)
nSteps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
world = grid[0]*grid[1]*grid[2]
ctx = ffm.Context(0)
ref = ffm.Plume(ctx, glob); ref.set_tight(True)
for _ in range(nSteps): ref.step()
with tempfile.TemporaryDirectory() as tmp:
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests/workers/plume_rank.py"), str(r), str(world), "29611", *map(str, glob), *map(str, grid), str(nSteps), tmp], env=dict(os.environ, FFM_TEST_TIGHT="1")) for r in range(world)]
    print([p.wait(timeout=300) for p in procs])
    parts = [np.load(os.path.join(tmp, "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
nx, ny, nz = glob
for name in ["ph_rgh", "rho", "p", "T", "Ux", "Uy", "Uz", "O2", "C3H8", "CO2", "p_rgh"]:
    full = np.empty((nz, ny, nx))
    for pt in parts:
        lo, hi = pt["lo"], pt["hi"]
        full[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] = pt[name].reshape(hi[2]-lo[2], hi[1]-lo[1], hi[0]-lo[0])
    b = ref.field(name).reshape(nz, ny, nx)
    d = np.abs(full - b)
    w = np.unravel_index(d.argmax(), d.shape)
    if name in ("Ux","Uy","Uz") and d.max()>1e-6:
        bad=np.argwhere(d>1e-3*d.max()); print("   bad cells (k,j,i):", bad[:12].tolist(), "count", len(bad))
    print("%-7s rel %.3e  maxabs %.3e at (k,j,i)=%s  ref %.6g" % (name, np.linalg.norm(full-b)/max(np.linalg.norm(b),1e-300), d.max(), w, b[w]))
print("iters ref:", [(n, p["nIterations"]) for n, p in ref.solves()])
pass
