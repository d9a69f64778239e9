"""SURVEY 8(f) N2, thermo part: oracle/thermo.py (janaf + sutherland + perfectGas + singleStepReactingMixture, restated from
OpenFOAM-dev) against the numbers the reference prints when it constructs that package for the steckler case
(cases/steckler/original/linux64/log.fireFoam:46-52,108): heat of combustion, stoichiometric ratios, maximum product
concentrations, stoichiometric mixture fraction -- every printed digit.  They pin the janaf formation enthalpies (hc from the
low-temperature coefficients at 298.15 K, mass based), the molecular weights and the reaction parsing."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "steckler_first_step.json")))
DATA = json.load(open(os.path.join(HERE, "golden", "steckler_case_data.json")))


def _species():
    from oracle import thermo as TH
    tab = {n: {k: (np.array(v) if isinstance(v, list) else v) for k, v in d.items()} for n, d in DATA["table"].items()}
    sp = TH.Species(DATA["species"], tab)
    rx = TH.SingleStep(sp, [tuple(t) for t in DATA["reaction"]["lhs"]], [tuple(t) for t in DATA["reaction"]["rhs"]], DATA["fuel"], DATA["inertSpecie"])
    return sp, rx


def test_single_step_mixture_reproduces_the_start_up_prints():
    sp, rx = _species()
    g = GOLD["startup"]
    assert "%.8g" % rx.qFuel == "%.8g" % g["qFuel"] == "46357151"
    assert "%.8g" % rx.stoicRatio == "15.571544" and "%.8g" % rx.s == "3.6282945"
    for n, v in g["Yprod0"].items():
        assert "%.8g" % rx.Yprod0[sp.names.index(n)] == "%.8g" % v
    assert "%.8g" % (1.0 / (1.0 + rx.stoicRatio)) == "0.060344407"


def test_case_data_fixture_is_what_the_reference_files_hold():
    if not os.path.isdir("/root/reference/cases/steckler"):
        pytest.skip("reference not mounted (GPU box)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_steckler_case_data.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    fresh = json.loads(json.dumps(mk.build()))
    assert fresh == DATA


def test_janaf_newton_and_transport_identities():
    """T(h) inverts Hs to the Newton tolerance (1e-4 relative step) from a nearby start; Hs(Tstd) = 0 exactly; Cp = dHs/dT;
    the mixture of one specie is that specie; air at 298.15 K has mu ~ 1.84e-5 Pa s, Pr ~ 0.7 (physical sanity of the
    sutherland / modified-Eucken coefficients of the case)"""
    sp, _ = _species()
    air = sp.mixture(np.array([0.23301, 0.0, 0.0, 0.0, 0.76699]))
    assert air.Hs(1e5, np.array([298.15]))[0] == 0.0
    T = np.array([320.0, 600.0, 1400.0])
    h = air.Hs(1e5, T)
    Tn = air.THs(h, 1e5, T * 1.02)
    assert np.all(np.abs(Tn - T) < 2e-4 * T)
    dT = 1e-3
    assert np.allclose((air.Hs(1e5, T + dT) - air.Hs(1e5, T - dT)) / (2 * dT), air.Cp(1e5, T), rtol=1e-6)
    o2 = sp.mixture(np.array([1.0, 0, 0, 0, 0])); s0 = sp.single(0)
    assert o2.W[0] == s0.W and np.array_equal(o2.low[0], s0.low)
    mu = air.mu(1e5, np.array([298.15]))[0]; pr = mu * air.Cp(1e5, np.array([298.15]))[0] / air.kappa(1e5, np.array([298.15]))[0]
    assert 1.80e-5 < mu < 1.88e-5 and 0.68 < pr < 0.76
