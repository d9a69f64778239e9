"""The five hydrostatic-initialisation DICPCG solves of the steckler geometry refined r x r x r (r = 4: BASELINE config 2,
576 000 cells, baffles + doorway), on the GPU: tiled sweeps (hint from the cell centres) vs one launch per level."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffm_import import ffm
from oracle import steckler
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = steckler.build_mesh(refine=r)
ctx = ffm.Context(0)
l, u = m.l.astype(np.int32), m.u.astype(np.int32)
for mode in ("tile", "levels"):
    os.environ["FFM_SWEEP"] = mode if mode == "levels" else "auto"
    hint = ffm.tile_hint_from_centres(m.C.T.copy()) if mode == "tile" else None
    cOrd, fOrd = ffm.renumber_levels(m.nCells, l, u, groupHint=hint)
    l2, u2, o2n = ffm.hexmesh.apply_renumbering(m.nCells, l, u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, m.nCells, l2, u2, groupHint=None if hint is None else hint[cOrd])
    tsolve = [0.0]
    def gpu_solve(mesh, diag, upper, source, psi0):
        A.set_coeffs(diag[cOrd], upper[fOrd])
        psi = ctx.to_device(psi0[cOrd]); b = ctx.to_device(source[cOrd]); ctx.sync()
        t0 = time.perf_counter()
        perf = A.solve(psi, b, solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0.01)
        ctx.sync(); tsolve[0] += time.perf_counter() - t0
        out = np.empty(m.nCells); out[cOrd] = psi.cpu().numpy()
        return out, perf
    recs, ph = steckler.hydrostatic_initialisation(gpu_solve, mesh=m)
    its = [q["nIterations"] for q in recs]
    print("r=%d N=%d %-6s sweep_mode=%d levels=%d iterations %s  solve time %.1f ms (%.3f ms/iteration)"
          % (r, m.nCells, mode, A.sweep_mode, A.nLevels, its, tsolve[0] * 1e3, tsolve[0] * 1e3 / max(sum(its), 1)))
    A.close()
