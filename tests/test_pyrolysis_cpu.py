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


WALLFIRE = dict(model="reactingOneDim21", alphaScheme="linear", kappaScheme="harmonic", back=("constH", 0.0, 293.0),
                radiation=dict(v=(0.17, 0.17), char=(0.85, 0.85)))      # cases/wallFireSpread2D: the selections tests/golden/wallfire_case_data.json holds


def test_reactingOneDim21_energy_equation_differs_only_in_its_sources():
    """reactingOneDim21 (config 5's model, reactingOneDim21.C:331-342) against reactingOneDim (:306-353 of the packages/ version) on the
    same state: identical chemistry, continuity, species; the energy matrices differ by V*RRg on the diagonal (the - fvm::Sp(RRg, h) the
    variant drops) and the sources by V*(RRs(0) T Cp0 + RRs(1) T Cp1); below Tcrit the two models are the same."""
    from oracle import pyrolysis as PY
    a = PY.Panel(4, 8, area=0.01, T0=650.0, model="reactingOneDim", kappaScheme="harmonic")
    b = PY.Panel(4, 8, area=0.01, T0=650.0, model="reactingOneDim21", kappaScheme="harmonic")
    q = np.array([0.0, 1e4, 2e4, 3e4])
    ra, rb = a.step(0.05, q), b.step(0.05, q)
    assert np.array_equal(a.rho, b.rho) and np.array_equal(a.Yw, b.Yw) and ra["RRg"].min() > 0
    assert np.allclose(ra["diag"] - rb["diag"], a.V * ra["RRg"], rtol=1e-12)
    T0 = 650.0
    sr = PY.CHAR.rho / PY.WOOD.rho; omega = ra["RRg"] / (1.0 - sr)
    assert np.allclose(rb["src"] - ra["src"], a.V * (-omega * T0 * PY.WOOD.Cp + sr * omega * T0 * PY.CHAR.Cp), rtol=1e-10)
    assert not np.allclose(a.h, b.h, rtol=1e-6)
    c = PY.Panel(4, 8, area=0.01, T0=300.0, model="reactingOneDim"); d = PY.Panel(4, 8, area=0.01, T0=300.0, model="reactingOneDim21")
    c.step(0.05, q); d.step(0.05, q)
    assert np.array_equal(c.h, d.h)


def test_constHTemperature_back_face_and_the_harmonic_conductivity():
    """constHTemperature (constHTemperatureFvPatchScalarField.C:156-180): h = 0 is an insulated back face (valueFraction ~1e-15), a large
    h holds it at Tinf, and a finite h lies in between.  Harmonic interpolation equals linear interpolation in a uniform solid."""
    from oracle import pyrolysis as PY
    q = np.full(2, 2.0e3)
    ins = PY.Panel(2, 8, area=0.02, T0=300.0, kappaScheme="harmonic", alphaScheme="harmonic", back=("constH", 0.0, 250.0))
    adi = PY.Panel(2, 8, area=0.02, T0=300.0, kappaScheme="linear", back=None)
    hot = PY.Panel(2, 8, area=0.02, T0=300.0, back=("constH", 1e12, 250.0)); fix = PY.Panel(2, 8, area=0.02, T0=300.0, back=("fixed", 250.0))
    mid = PY.Panel(2, 8, area=0.02, T0=300.0, back=("constH", 40.0, 250.0))
    E0 = (mid.rho * mid.h * mid.V).sum(axis=1)
    for _ in range(40):
        for p in (ins, adi, hot, fix, mid):
            p.step(0.05, q)
    assert np.allclose(ins.h, adi.h, rtol=1e-12) and np.allclose(hot.h, fix.h, rtol=1e-9)
    assert np.all(mid.T[:, -1] < ins.T[:, -1]) and np.all(mid.T[:, -1] > fix.T[:, -1])
    E1 = (mid.rho * mid.h * mid.V).sum(axis=1)
    assert np.all(E1 - E0 < q * mid.A * 0.05 * 40)                  # heat leaves through the back face
    Ef = (fix.rho * fix.h * fix.V).sum(axis=1)
    assert np.all(Ef < E1)                                         # and more of it when the face is held at Tinf


def test_coupled_evolve_of_the_wallfire_panel():
    """evolve(): the coupled wall condition evaluated inside the step with the surface emissivity / absorptivity of
    greyMeanSolidAbsorptionEmission (volume-fraction weighted: 0.17 for the virgin solid towards 0.85 for char).  The flux handed to
    the energy equation is the condition's -nbrTotalFlux, the stored wall value the cell value plus refGrad over half a layer, and
    what the gas side reads afterwards is the NEW solid state."""
    from oracle import pyrolysis as PY
    p = PY.Panel(6, 8, area=0.01, T0=293.0, **WALLFIRE)
    Tg = np.linspace(900.0, 1400.0, 6); kD = np.full(6, 12.0); qin = np.linspace(2e4, 6e4, 6)
    a0, e0 = p.surface_radiation()
    assert np.allclose(a0, 0.17) and np.allclose(e0, 0.17)
    Tw0, Tc0 = p.Twall.copy(), p.T[:, 0].copy()
    p.evolve(0.05, Tg, kD, qin)
    expect = -(kD * (Tc0 - Tg) - 0.17 * qin + 0.17 * PY.SIGMA_SB * Tw0 ** 4)
    assert np.allclose(p.qSurf, expect, rtol=1e-14)
    assert np.allclose((p.Twall - p.T[:, 0]) * (2.0 / p.dx) * (PY.WOOD.kappa), p.qSurf, rtol=1e-12)      # kappa() of the virgin solid
    for _ in range(1500):
        p.evolve(0.05, Tg, kD, qin)
    a1, e1 = p.surface_radiation()
    assert p.Yw[:, 0].max() < 0.9 and np.all(e1 > 0.5) and np.all(e1 <= 0.85) and np.all(np.diff(e1) > 0)      # charring: emissivity towards char's (volume fractions)
    nf = np.tile(np.array([[0.0, 0.0, -1.0]]), (6, 1))
    refT, U, emis = p.gas_side(np.full(6, 0.5), np.full(6, 0.01), nf, 1.66e7, 4.6e7)
    assert np.array_equal(refT, p.T[:, 0]) and np.array_equal(emis, e1) and np.all((U * nf).sum(axis=1) <= 0)
