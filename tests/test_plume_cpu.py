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
