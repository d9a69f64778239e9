"""GPU probe: oracle/fvdom.py's iteration (linearUpwind, maxIter 3) with every ray system solved by BOTH the device GAMG + DILU and the
oracle's, on a 2-D mesh in level-major numbering.  usage: python scripts/ray_gamg_probe2.py nx ny"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from ffm_import import ffm
from oracle import oracle as O, plume, fv, fvdom, gamg

nx, ny = int(sys.argv[1]), int(sys.argv[2])
O.build()
m = plume.make_mesh((nx, ny, 1), h=1.42 / nx, empty=("zmin", "zmax"))
N = m.nCells
cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
l2, u2, o2n = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
ctx = ffm.Context(0)
A = ffm.lduMatrix(ctx, N, l2, u2)
G = ffm.GAMG(ctx, A, l2, u2, Sf=m.Sf[fOrd])
agg = gamg.Agglomeration(N, l2, u2, gamg.face_area_pair_weights(m.Sf[fOrd]), nCellsInCoarsestLevel=10, mergeLevels=1)


def solve(name, d, upper, lower, s, psi0):
    dg2, up2, lo2, s2, p2 = d[cOrd], upper[fOrd], lower[fOrd], s[cOrd], psi0[cOrd]
    psi = ctx.to_device(p2)
    G.set_matrix(ctx.to_device(dg2), ctx.to_device(up2), ctx.to_device(lo2))
    pd = G.solve(psi, ctx.to_device(s2), smoother="DILU", tolerance=1e-4, relTol=0.0, maxIter=25)
    xo, po = gamg.GAMGSolver(agg, dg2, up2, lo2, smoother="DILU").solve(p2.copy(), s2, tolerance=1e-4, relTol=0.0, maxIter=25)
    xd = psi.cpu().numpy()
    print("%s: dev %d cycles (%.2e -> %.2e)  oracle %d cycles (%.2e -> %.2e)  finite src %s  |s| %.2e  rel diff %.2e"
          % (name, pd["nIterations"], pd["initialResidual"], pd["finalResidual"], po["nIterations"], po["initialResidual"], po["finalResidual"],
             np.isfinite(s2).all(), np.abs(s2).max(), np.linalg.norm(xd - xo) / max(np.linalg.norm(xo), 1e-300)), flush=True)
    out = np.empty_like(xo); out[cOrd] = xo
    return out, po


dom = fvdom.FvDOM(m, 2, 2, solve, maxIter=3, tolerance=1e-3, divScheme="linearUpwind", solutionD=(1, 1, -1),
                  emissivity=[np.full(p.size, 0.17 if p.name == "inlet" else 1.0) for p in m.patches])
T = np.full(N, 298.15); Tb = [np.full(p.size, 600.0 if p.name == "inlet" else 298.15) for p in m.patches]
dom.calculate(T, Tb, 0.0, np.zeros(N))
print("iterations", dom.nIterations)
