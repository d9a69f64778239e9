"""SURVEY 8(f) N3 on the device: ffm_pyro_step (one thread per column: chemistry, continuity, species, the tridiagonal enthalpy
solve, thermo) against oracle/pyrolysis.py, same inputs, on the geometries of the reference's cases -- one column of 8 layers
(cases/pyrolysis1D) and panels of 40 and 5000 columns (a wall refined as in BASELINE config 5) -- over the heating-up, the onset of
the reaction and the charring: rho, Y, h, T to 1e-11 relative (exp / pow of the device maths library against libm), the coupling
outputs (exposed-face temperature, phiGas) likewise, the mass balance on the device itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nCol,nLay,back", [(1, 8, None), (40, 8, None), (5000, 8, 298.15), (33, 5, None)])
def test_pyrolysis_panel_matches_oracle(ffm, ctx, nCol, nLay, back):
    from oracle import pyrolysis as PY
    ref = PY.Panel(nCol, nLay, thickness=0.0127, area=0.01)
    dev = ffm.PyrolysisPanel(ctx, nCol, nLay, thickness=0.0127, area=0.01)
    q = 3.0e4 * (1.0 + 0.7 * np.sin(0.37 * np.arange(nCol)) ** 2)
    qd = ctx.to_device(q)
    dt = 0.05
    gas = np.zeros(nCol); m0 = (ref.rho * ref.V).sum(axis=1)
    for step in range(600):
        ref.step(dt, q, Tback=back); dev.step(dt, qd, Tback=back)
        if step % 100 == 99 or step == 0:
            for name, r in (("rho", ref.rho), ("Yw", ref.Yw), ("T", ref.T), ("h", ref.h)):
                d = dev.field(name)
                assert np.abs(d - r).max() <= 1e-11 * np.abs(r).max(), (step, name, np.abs(d - r).max())
            assert np.abs(dev.field("phiGas") - ref.massGas).max() <= 1e-10 * max(ref.massGas.max(), 1e-300)
            assert np.abs(dev.field("Tsurf") - ref.T[:, 0]).max() <= 1e-11 * ref.T.max()
        gas += dev.field("phiGas") * dt
    assert ref.Yw.min() < 0.5                                   # the run reached charring
    lost = m0 - (dev.field("rho") * ref.V).sum(axis=1)
    assert np.allclose(lost, gas, rtol=1e-9)
    dev.close()


def test_gas_side_coupling_of_the_pyrolysis_panel(ffm, ctx):
    """The mapped patch conditions of lib/fvPatchFieldsPyrolysis between a gas wall patch and the panel (ffm_pyro_couple_d):
    a fire-exposed wall of 300 columns mapped onto a gas patch through a permutation, incident radiation qin and a hot gas cell
    layer; per step couple() -> step_coupled() on the device against oracle couple() -> step().  Heat flux into the solid, wall
    temperature, the gas-side reference temperature and inlet velocity against the oracle; the pyrolysate leaves as an inflow of
    the gas region (U.n < 0, mass flux = phiGas hocPyr/qFuel); only the mapped patch faces are written."""
    import torch
    from oracle import pyrolysis as PY
    nCol, B = 300, 420                                          # the gas patch's faces are a subset of a larger boundary
    ref = PY.Panel(nCol, 8, thickness=0.0127, area=0.01)
    dev = ffm.PyrolysisPanel(ctx, nCol, 8, thickness=0.0127, area=0.01)
    rng = np.random.default_rng(3)
    fmap = rng.permutation(B)[:nCol].astype(np.int32)
    i = np.arange(B)
    Tg = 600.0 + 500.0 * np.sin(0.05 * i) ** 2; kD = 8.0 + 4.0 * np.cos(0.11 * i) ** 2; qin = 4.0e4 * (1.0 + 0.5 * np.sin(0.023 * i))
    rhob = 0.4 + 0.3 * np.cos(0.07 * i) ** 2; magSf = np.full(B, 0.01)
    nf = np.stack([np.full(B, -1.0), np.zeros(B), np.zeros(B)], axis=1)
    e, a, hocSolid, qFuel = 0.9, 0.85, 1.66e7, 4.6e7                       # cases/wallFireSpread2D/0/U:62 hocSolid; propane qFuel
    D = lambda x: ctx.to_device(np.ascontiguousarray(x, np.float64))
    d = dict(Tg=D(Tg), kD=D(kD), qin=D(qin), rhob=D(rhob), magSf=D(magSf), nf=[D(nf[:, c]) for c in range(3)],
             refT=D(np.full(B, -1.0)), U=[D(np.full(B, 7.0)) for _ in range(3)], map=torch.from_numpy(fmap).cuda())
    Tw = np.full(nCol, 298.15)
    dt = 0.05
    for step in range(400):
        q, Tw, refT, U = PY.couple(ref, Tw, Tg[fmap], kD[fmap], qin[fmap], e, a, rhob[fmap], magSf[fmap], nf[fmap], hocSolid, qFuel)
        ref.step(dt, q)
        dev.couple(d["Tg"], d["kD"], d["qin"], e, a, d["rhob"], d["magSf"], d["nf"], hocSolid, qFuel, d["refT"], d["U"], map=d["map"])
        dev.step_coupled(dt)
        if step % 50 == 0 or step == 399:
            assert np.abs(dev.field("qSurf") - q).max() <= 1e-11 * np.abs(q).max(), step
            assert np.abs(dev.field("Twall") - Tw).max() <= 1e-11 * Tw.max(), step
            rT = d["refT"].cpu().numpy(); Ud = np.stack([u.cpu().numpy() for u in d["U"]], axis=1)
            assert np.abs(rT[fmap] - refT).max() <= 1e-11 * refT.max()
            assert np.abs(Ud[fmap] - U).max() <= 1e-10 * max(np.abs(U).max(), 1e-300)
            other = np.setdiff1d(np.arange(B), fmap)
            assert np.all(rT[other] == -1.0) and np.all(Ud[other] == 7.0)
            assert np.abs(dev.field("T") - ref.T).max() <= 1e-10 * ref.T.max()
    assert ref.Yw[:, 0].min() < 0.9 and ref.massGas.max() > 0                 # the exposed layer pyrolyses
    dev.couple(d["Tg"], d["kD"], d["qin"], e, a, d["rhob"], d["magSf"], d["nf"], hocSolid, qFuel, d["refT"], d["U"], map=d["map"])   # of the final state
    Ud = np.stack([u.cpu().numpy() for u in d["U"]], axis=1)
    inflow = (Ud[fmap] * nf[fmap]).sum(axis=1)
    assert inflow.min() < 0 and inflow.max() <= 0                            # into the gas region
    hocPyr = (hocSolid * PY.WOOD.rho - PY.HOC_CHAR * PY.CHAR.rho) / (PY.WOOD.rho - PY.CHAR.rho)
    assert np.allclose(-inflow * rhob[fmap] * magSf[fmap], dev.field("phiGas") * hocPyr / qFuel, rtol=1e-12, atol=0)
    dev.close()


WALLFIRE = dict(model="reactingOneDim21", alphaScheme="linear", kappaScheme="harmonic", back=("constH", 0.0, 293.0),
                radiation=dict(v=(0.17, 0.17), char=(0.85, 0.85)))      # cases/wallFireSpread2D (tests/golden/wallfire_case_data.json)


@pytest.mark.parametrize("sel", [WALLFIRE, dict(model="reactingOneDim", alphaScheme="harmonic", kappaScheme="harmonic", back=("constH", 25.0, 298.15)),
                                 dict(model="reactingOneDim21", alphaScheme="harmonic", kappaScheme="linear", back=("fixed", 320.0))])
def test_pyrolysis_model_selections_match_oracle(ffm, ctx, sel):
    """ffm_pyro_set_model / _set_back: BASELINE config 5's reactingOneDim21 with `Gauss harmonic` conductivity and the
    constHTemperature back face (first case: exactly the selections of cases/wallFireSpread2D), config 1's reactingOneDim with both
    laplacians harmonic (cases/pyrolysis1D/system/panelRegion/fvSchemes:37-38) and a convecting back face, and a mixed selection --
    600 steps through heating, onset and charring with a given surface flux, every field against the oracle."""
    from oracle import pyrolysis as PY
    nCol = 257
    ref = PY.Panel(nCol, 8, thickness=0.0127, area=0.01, T0=293.0, **sel)
    dev = ffm.PyrolysisPanel(ctx, nCol, 8, thickness=0.0127, area=0.01, T0=293.0)
    dev.set_model(**sel)
    q = 3.5e4 * (1.0 + 0.6 * np.sin(0.21 * np.arange(nCol)) ** 2)
    qd = ctx.to_device(q)
    for step in range(600):
        ref.step(0.05, q); dev.step(0.05, qd)
        if step % 100 == 99 or step == 0:
            for name, r in (("rho", ref.rho), ("Yw", ref.Yw), ("T", ref.T), ("h", ref.h), ("alpha", ref.alpha)):
                d = dev.field(name)
                assert np.abs(d - r).max() <= 1e-11 * np.abs(r).max(), (step, name, np.abs(d - r).max())
            assert np.abs(dev.field("phiGas") - ref.massGas).max() <= 1e-10 * max(ref.massGas.max(), 1e-300)
    assert ref.Yw.min() < 0.5
    dev.close()


def test_wallfire_panel_coupled_evolve_matches_oracle(ffm, ctx):
    """ffm_pyro_evolve_d + ffm_pyro_gas_side_d with the selections of cases/wallFireSpread2D: the coupled wall condition inside the
    step (composition-dependent absorptivity / emissivity of greyMeanSolidAbsorptionEmission, kappa() = Cp()*alpha_), then what the gas
    patch reads from the NEW solid state -- wall temperature, pyrolysate inflow, wall emissivity for greyDiffusiveRadiation -- through a
    permutation map, 500 steps from cold to a charred surface, against the oracle."""
    import torch
    from oracle import pyrolysis as PY
    nCol, B = 300, 420
    ref = PY.Panel(nCol, 8, thickness=0.0127, area=0.01, T0=293.0, **WALLFIRE)
    dev = ffm.PyrolysisPanel(ctx, nCol, 8, thickness=0.0127, area=0.01, T0=293.0)
    dev.set_model(**WALLFIRE)
    rng = np.random.default_rng(5)
    fmap = rng.permutation(B)[:nCol].astype(np.int32)
    i = np.arange(B)
    Tg = 700.0 + 700.0 * np.sin(0.05 * i) ** 2; kD = 8.0 + 4.0 * np.cos(0.11 * i) ** 2; qin = 5.0e4 * (1.0 + 0.5 * np.sin(0.023 * i))
    rhob = 0.4 + 0.3 * np.cos(0.07 * i) ** 2; magSf = np.full(B, 0.01)
    nf = np.stack([np.zeros(B), np.zeros(B), np.full(B, -1.0)], axis=1)
    hocSolid, qFuel = 1.66e7, 4.6e7
    D = lambda x: ctx.to_device(np.ascontiguousarray(x, np.float64))
    d = dict(Tg=D(Tg), kD=D(kD), qin=D(qin), rhob=D(rhob), magSf=D(magSf), nf=[D(nf[:, c]) for c in range(3)],
             refT=D(np.full(B, -1.0)), U=[D(np.full(B, 7.0)) for _ in range(3)], emis=D(np.full(B, 1.0)), map=torch.from_numpy(fmap).cuda())
    for step in range(500):
        ref.evolve(0.05, Tg[fmap], kD[fmap], qin[fmap])
        dev.evolve(0.05, d["Tg"], d["kD"], d["qin"], map=d["map"])
        if step % 50 == 0 or step == 499:
            refT, U, emis = ref.gas_side(rhob[fmap], magSf[fmap], nf[fmap], hocSolid, qFuel)
            dev.gas_side(d["rhob"], d["magSf"], d["nf"], hocSolid, qFuel, d["refT"], d["U"], emissivity=d["emis"], map=d["map"])
            assert np.abs(dev.field("qSurf") - ref.qSurf).max() <= 1e-11 * np.abs(ref.qSurf).max(), step
            assert np.abs(dev.field("Twall") - ref.Twall).max() <= 1e-11 * ref.Twall.max(), step
            for name, r in (("rho", ref.rho), ("Yw", ref.Yw), ("T", ref.T), ("h", ref.h), ("alpha", ref.alpha)):
                assert np.abs(dev.field(name) - r).max() <= 1e-10 * np.abs(r).max(), (step, name)
            rT = d["refT"].cpu().numpy(); em = d["emis"].cpu().numpy(); Ud = np.stack([u.cpu().numpy() for u in d["U"]], axis=1)
            assert np.abs(rT[fmap] - refT).max() <= 1e-11 * refT.max()
            assert np.abs(em[fmap] - emis).max() <= 1e-12
            assert np.abs(Ud[fmap] - U).max() <= 1e-10 * max(np.abs(U).max(), 1e-300)
            other = np.setdiff1d(np.arange(B), fmap)
            assert np.all(rT[other] == -1.0) and np.all(Ud[other] == 7.0) and np.all(em[other] == 1.0)
    assert ref.Yw[:, 0].min() < 0.9 and ref.massGas.max() > 0 and ref.surface_radiation()[1].max() > 0.3
    dev.close()
