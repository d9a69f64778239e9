"""GPU probe: DILU application and GAMG + DILU solves of upwind ray matrices on a 2-D mesh (level-major numbering) against the oracle.
usage: python scripts/ray_gamg_probe.py nx ny"""
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
print("cells", N, "levels", A.nLevels, "sweep mode", A.sweep_mode, flush=True)
G = ffm.GAMG(ctx, A, l2, u2, Sf=m.Sf[fOrd])
agg = gamg.Agglomeration(N, l2, u2, gamg.face_area_pair_weights(m.Sf[fOrd]), nCellsInCoarsestLevel=10, mergeLevels=1)
rays = fvdom.ray_set(2, 2, (1, 1, -1))
Tb = [np.full(p.size, 900.0 if p.name == "inlet" else 320.0) for p in m.patches]
for i, (d, dAve, omega) in enumerate(rays):
    Ji = (dAve[0] * m.Sf[:, 0] + dAve[1] * m.Sf[:, 1]) + dAve[2] * m.Sf[:, 2]
    Jib = [(dAve[0] * p.Sf[:, 0] + dAve[1] * p.Sf[:, 1]) + dAve[2] * p.Sf[:, 2] for p in m.patches]
    bc = fv.MixedBC(m, f=[1.0 - fv.pos0(jb) for jb in Jib], ref=[fvdom.SIGMA_SB * tb ** 4 / np.pi for tb in Tb])
    M = fv.fvm_div(m, Ji, Jib, fv.pos0(Ji), [bc])
    dg, s = M.solve_system()
    dg2, up2, lo2, s2 = dg[cOrd], M.upper[fOrd], M.lower[fOrd], s[cOrd]
    Ao = O.Ldu(N, l2, u2).set_coeffs(dg2, up2, lo2)
    A.set_coeffs(dg2, up2, lo2)
    r = O.hash_u(7 + i, np.arange(N)) - 0.5
    w_ref = Ao.dilu_precondition(Ao.dilu_rD(), r)
    w_dev = A.precondition("DILU", ctx.to_device(r)).cpu().numpy()
    t = time.time()
    psi = ctx.zeros(N)
    G.set_matrix(ctx.to_device(dg2), ctx.to_device(up2), ctx.to_device(lo2))
    pd = G.solve(psi, ctx.to_device(s2), smoother="DILU", tolerance=1e-4, relTol=0.0, maxIter=30)
    td = time.time() - t
    xo, po = gamg.GAMGSolver(agg, dg2, up2, lo2, smoother="DILU").solve(np.zeros(N), s2, tolerance=1e-4, relTol=0.0, maxIter=30)
    print("ray %d d=(%.2f,%.2f): DILU apply max|dev-oracle| %.2e (|w| %.2e); GAMG cycles dev %d (final %.2e, %.2fs) oracle %d (final %.2e); x rel diff %.2e"
          % (i, d[0], d[1], np.abs(w_dev - w_ref).max(), np.abs(w_ref).max(), pd["nIterations"], pd["finalResidual"], td, po["nIterations"], po["finalResidual"],
             np.linalg.norm(psi.cpu().numpy() - xo) / np.linalg.norm(xo)), flush=True)
