"""Conditioning of the plume oracle itself (CPU): why the coupled-radiation parity test compares later steps at 1e-5."""
import copy

import numpy as np

from common import rel_l2


def test_limited_weights_of_a_noise_field_are_ill_conditioned():
    """After the first step the ambient enthalpy is rounding noise (|h| < 1e-8 J/kg, mostly < 1e-16: T is uniform and G = 4 sigma T^4 up to the ray
    solver's tolerance).  limitedLinear forms r = 2 (d.gradc)/(h_N - h_P) - 1 from that field (NVDTVD.H:96-130 of OpenFOAM-dev) and
    the resulting weights multiply the new enthalpies implicitly, so a perturbation of 1e-19 J/kg (3e-23 of max|h|) moves the next
    step's h by more than 1e-8: two bit-different but equally valid evaluations of the same formulas cannot agree better."""
    from oracle import plume
    ref = plume.Plume((12, 16, 12)); ref.set_radiation(solverFreq=1); ref.set_radiation_model(0.08, 0.3, 0.3)
    ref.step()
    ambient = np.abs(ref.T - ref.T.min()) < 1e-9
    assert ambient.sum() > 0.5 * ref.h.size and np.abs(ref.h[ambient]).max() < 1e-8
    other = copy.deepcopy(ref)
    other.h = other.h + 1e-19 * np.random.default_rng(1).standard_normal(other.h.shape)
    ref.step(); other.step()
    assert 1e-8 < rel_l2(other.h, ref.h) < 1e-5


def test_the_conditioned_start_state_is_well_conditioned():
    """The counterpart: in the conditioned case (oracle/plume.py:conditioned_state -- every specie and h varies smoothly by 1e-3 of
    its value over the box, inflow and ambient values distinct, no exact zeros) the limiters are formed from differences of the
    fields, not of solver noise, and a relative perturbation of 1e-14 of the transported fields after the second step stays at that
    level over the following steps.  Two evaluations of the same formulas that differ in rounding must therefore agree to 1e-8 there
    (tests/test_plume_multistep_gpu.py holds the device to it); a larger difference would be a defect, not conditioning."""
    from oracle import plume
    ref = plume.Plume((12, 16, 12), conditioned=True)
    for Yi in ref.Y:
        assert Yi.min() > 1e-3 and np.ptp(Yi) > 1e-6
    assert np.ptp(ref.h) > 1e-4 * np.abs(ref.h).max()
    ref.step(); ref.step()
    other = copy.deepcopy(ref)
    rng = np.random.default_rng(1)
    other.h = other.h * (1.0 + 1e-14 * rng.standard_normal(other.h.shape))
    other.Y = other.Y * (1.0 + 1e-14 * rng.standard_normal(other.Y.shape))
    for _ in range(4):
        ref.step(); other.step()
        assert [pf["nIterations"] for _, pf in ref.sol.log] == [pf["nIterations"] for _, pf in other.sol.log]
    f, g = ref.fields(), other.fields()
    for name in ("h", "T", "rho", "Ux", "Uy", "Uz", "O2", "H2O", "C3H8", "CO2", "N2"):
        assert rel_l2(g[name], f[name]) < 1e-11, (name, rel_l2(g[name], f[name]))
