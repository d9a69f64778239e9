"""T3 of SURVEY 8(c): manufactured Poisson problem laplacian(p) = -f on the unit box, Dirichlet p = 0 on the top (y = 1),
zero normal gradient elsewhere, p = cos(pi x) cos(pi z) sin(pi (1 - y)/2).  At n = 8, 16, 32 the GPU PCG + DIC takes exactly
the oracle's iteration count, agrees with it to 1e-8, and the discretisation error against the analytic solution falls as h^2."""
import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


def test_manufactured_poisson_iteration_identity_and_order(O, ffm, ctx):
    from oracle import fv
    errs = []
    for n in (8, 16, 32):
        m = fv.HexMesh((n, n, n), (0, 0, 0), (1, 1, 1))
        m.set_patches([("top", ["ymax"]), ("walls", ["xmin", "xmax", "ymin", "zmin", "zmax"])])
        N = m.nCells
        x, y, z = m.C.T
        exact = np.cos(np.pi * x) * np.cos(np.pi * z) * np.sin(0.5 * np.pi * (1 - y))
        f = 2.25 * np.pi ** 2 * exact
        bc = fv.MixedBC(m, f=[np.ones(m.patches[0].size), np.zeros(m.patches[1].size)])
        M = fv.fvm_laplacian(m, np.ones(m.nFaces), [np.ones(p.size) for p in m.patches], [bc])
        M.add_su(-f)                                                   # fvm::laplacian(p) == -f
        d, s = M.solve_system(0)
        ctl = dict(tolerance=1e-10, relTol=0.0, maxIter=1000)
        ref, pr = O.Ldu(N, m.l, m.u).set_coeffs(d, M.upper, None).solve(O.PCG, O.DIC, np.zeros(N), s, **ctl)
        A = ffm.lduMatrix(ctx, N, m.l.astype(np.int32), m.u.astype(np.int32))
        A.set_coeffs(d, M.upper)
        psi = ctx.zeros(N)
        pg = A.solve(psi, ctx.to_device(s), solver="PCG", preconditioner="DIC", **ctl)
        A.close()
        assert pg["converged"] == 1 and pg["nIterations"] == pr["nIterations"]
        assert rel_l2(psi.cpu().numpy(), ref) < 1e-8
        errs.append(np.sqrt(np.mean((psi.cpu().numpy() - exact) ** 2)))
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5      # second order
