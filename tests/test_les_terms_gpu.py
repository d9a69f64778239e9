"""SURVEY 8(f) N2, LES kEqn terms on the device: the explicit part of turbulence->divDevRhoReff(U) (solver/UEqn.H:12),
fvc::div((rho*nuEff)*dev2(T(fvc::grad(U)))), and the production term G = nut*(gradU && dev(twoSymm(gradU))) of kEqn::correct,
from the nine cell gradients of the fused gradient pass -- against oracle/steckler_case.py's grad_vector / dev2T / div_tensor (the
functions behind the oracle's reproduction of the golden log's Ux / Uy / Uz and k lines) on a mesh with mixed patch conditions."""
import ctypes as C

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


def test_stress_divergence_and_production_term(O, ffm, ctx):
    from oracle import fv, plume, steckler_case as SC
    m = plume.make_mesh((9, 8, 7), h=0.1)
    N = m.nCells
    hu = lambda seed, n: O.hash_u(seed, np.arange(n))
    U = np.stack([hu(20 + d, N) - 0.5 for d in range(3)], axis=1)                        # [N][3]
    Ub = [np.stack([hu(90 + 10 * d + q, p.size) - 0.5 for d in range(3)], axis=1) for q, p in enumerate(m.patches)]
    gam = 2e-5 * (1.0 + hu(5, N)); gamb = [2e-5 * (1.0 + hu(50 + q, p.size)) for q, p in enumerate(m.patches)]
    nut = 1e-3 * hu(6, N)
    gU, gUb = SC.grad_vector(m, U, Ub)
    X = gam[:, None, None] * SC.dev2T(gU); Xb = [g[:, None, None] * SC.dev2T(t) for g, t in zip(gamb, gUb)]
    div_ref = SC.div_tensor(m, X, Xb)
    twoSymm = gU + np.swapaxes(gU, 1, 2)
    tr = (twoSymm[:, 0, 0] + twoSymm[:, 1, 1]) + twoSymm[:, 2, 2]
    dev = twoSymm.copy()
    for a in range(3):
        dev[:, a, a] = dev[:, a, a] - (1.0 / 3.0) * tr
    G_ref = nut * np.einsum("nij,nij->n", gU, dev)
    # device
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    D = lambda a: ctx.to_device(np.ascontiguousarray(a, np.float64))
    Uc = [D(U[cOrd, d]) for d in range(3)]; Ubd = [D(np.concatenate([b[:, d] for b in Ub])) for d in range(3)]
    g = [[ctx.empty(N) for _ in range(3)] for _ in range(3)]                             # g[j] = gradient of component j: (gx, gy, gz)
    mesh.call("fvc_grad_multi", 3, Uc, Ubd, [g[j][0] for j in range(3)], [g[j][1] for j in range(3)], [g[j][2] for j in range(3)])
    inv = np.empty(N, np.int64); inv[cOrd] = np.arange(N)
    for i in range(3):
        for j in range(3):
            assert rel_l2(g[j][i].cpu().numpy()[inv], gU[:, i, j]) < 1e-13                # d_i U_j
    nine = [g[j][i] for i in range(3) for j in range(3)]                                  # [3*i + j]
    out = [ctx.empty(N) for _ in range(3)]
    mesh.call("fvc_div_dev2T_gradU", nine, D(gam[cOrd]), D(np.concatenate(gamb)), Uc, Ubd, out)
    for j in range(3):
        assert rel_l2(out[j].cpu().numpy()[inv], div_ref[:, j]) < 1e-12, j
    Gd = ctx.empty(N)
    mesh.call("les_keqn_G", nine, D(nut[cOrd]), Gd)
    assert rel_l2(Gd.cpu().numpy()[inv], G_ref) < 1e-13
    mesh.close(); A.close()
