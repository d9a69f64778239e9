"""oracle/pyrolysis.py (reactingOneDim::evolveRegion restated for independent columns) -- conservation and
algebra checks.  The reference ships no output of a pyrolysis run (cases/pyrolysis1D/mlr.plot:3 plots ./referenceResult, which
is not in the tree): PARITY UNPINNED by reference data; what can be checked is what the equations imply.
 * the batched Thomas solve equals a dense solve of the assembled tridiagonal matrices;
 * mass: the solid mass every column loses in a step is the gas it releases (phiGas dt), cell by cell d(rho)/dt = -RRg;
 * no reaction below Tcrit: pure conduction; with the back face adiabatic the enthalpy gained equals the heat put in (q A dt);
 * the geometry of the reference's cases: 8 layers (cases/wallFireSpread2D/system/extrudeToRegionMeshDict: nLayers 8), one
   column (cases/pyrolysis1D) and a 40-column panel; heating chars the exposed layers first and the char fraction only grows."""
import numpy as np


def test_thomas_equals_dense():
    from oracle import pyrolysis as PY
    rng = np.random.RandomState(3)
    nC, nL = 7, 8
    lo = -rng.rand(nC, nL); up = -rng.rand(nC, nL); dg = 2.5 + rng.rand(nC, nL); b = rng.rand(nC, nL) - 0.5
    x = PY.thomas(lo, dg, up, b)
    for c in range(nC):
        M = np.diag(dg[c]) + np.diag(up[c, :-1], 1) + np.diag(lo[c, 1:], -1)
        assert np.allclose(x[c], np.linalg.solve(M, b[c]), rtol=1e-13, atol=1e-15)


def test_conduction_without_reaction_conserves_energy():
    from oracle import pyrolysis as PY
    P = PY.Panel(3, 8, thickness=0.0127, area=0.04, T0=300.0)
    q = np.array([0.0, 1.5e3, 3e3])
    E0 = (P.rho * P.h * P.V).sum(axis=1)
    dt, n = 0.05, 40                                         # stays below Tcrit = 400 K
    for _ in range(n):
        P.step(dt, q)
    assert P.T.max() < 400.0 and np.all(P.Yw == 1.0) and np.all(P.massGas == 0.0)
    E1 = (P.rho * P.h * P.V).sum(axis=1)
    assert np.allclose(E1 - E0, q * P.A * dt * n, rtol=1e-12, atol=1e-9)
    assert np.all(P.T[0] == 300.0)                           # no flux, no change
    assert np.all(np.diff(P.T[2]) < 0)                       # heated from the exposed face


def test_pyrolysis_mass_balance_and_char_front():
    from oracle import pyrolysis as PY
    for nCol in (1, 40):                                     # cases/pyrolysis1D: one column; a wall panel
        P = PY.Panel(nCol, 8, thickness=0.0127, area=0.01, T0=298.15)
        q = 3.0e4 * (1.0 + 0.5 * np.arange(nCol) / max(nCol - 1, 1))
        lost = np.zeros(nCol); gasOut = np.zeros(nCol)
        dt = 0.05
        for _ in range(600):
            m0 = (P.rho * P.V).sum(axis=1); yw0 = P.Yw.copy(); rho0 = P.rho.copy()
            out = P.step(dt, q)
            assert np.allclose((P.rho - rho0) / dt, -out["RRg"], rtol=1e-10, atol=1e-12)
            lost += m0 - (P.rho * P.V).sum(axis=1); gasOut += P.massGas * dt
            assert np.all(P.Yw <= yw0 + 1e-15) and np.all(P.Yw >= 0.0)
        assert np.allclose(lost, gasOut, rtol=1e-10) and lost.min() > 0
        assert np.all(P.Yw[:, 0] < P.Yw[:, -1])             # the exposed layer chars first
        if nCol > 1:
            assert lost[-1] > lost[0]                        # more heat, more pyrolysate
        assert np.isfinite(P.T).all() and P.T.max() < 2000.0 and P.Yw.min() < 0.5


def test_coupled_patch_conditions_balance():
    """oracle couple(): the heat flux handed to the solid is what the mixed condition's gradient carries (q = kappa_s refGrad),
    the wall value follows from that gradient over half a layer, and with equal gas / solid temperatures, no irradiation and
    zero emissivity nothing flows."""
    from oracle import pyrolysis as PY
    p = PY.Panel(5, 8, thickness=0.0127, area=0.01, T0=400.0)
    nf = np.tile(np.array([[-1.0, 0.0, 0.0]]), (5, 1))
    one = np.ones(5)
    q, Tw, refT, U = PY.couple(p, 400.0 * one, 400.0 * one, 10.0 * one, 0.0 * one, 0.0, 0.0, one, 0.01 * one, nf, 1.66e7, 4.6e7)
    assert np.all(q == 0) and np.all(Tw == 400.0) and np.all(refT == 400.0) and np.all(U == 0)
    q, Tw, refT, U = PY.couple(p, 400.0 * one, 900.0 * one, 10.0 * one, 3e4 * one, 0.9, 0.8, one, 0.01 * one, nf, 1.66e7, 4.6e7)
    expect = 10.0 * (900.0 - 400.0) + 0.8 * 3e4 - 0.9 * PY.SIGMA_SB * 400.0 ** 4
    assert np.allclose(q, expect, rtol=1e-14)
    assert np.allclose((Tw - 400.0) * (2.0 / p.dx) * p.kappa()[:, 0], q, rtol=1e-12)
