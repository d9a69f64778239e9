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
