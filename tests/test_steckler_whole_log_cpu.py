"""The oracle follows the reference's golden log THROUGH ALL ITS 29 TIME STEPS (cases/steckler/original/linux64/log.fireFoam: the
steckler room from the cold start at t = 0 through ignition of the propane burner to a 1027 K flame at t = 2 s; fixture
tests/golden/steckler_log_steps.json made by tests/golden/make_steckler_log_steps.py):

  * the time-step sequence -- setMultiRegionDeltaT.H + setDeltaT.H + Time::adjustDeltaT: 0.066667, 0.093333 x 10, 0.1, 0.081818 x 3,
    0.065455, 0.073636, ... 0.021477 -- to every printed digit, and the Courant numbers in front of every step to 4-5 digits;
  * all 406 solver lines: the field names in order and EVERY iteration count (smoothSolver + symGaussSeidel for Ux, Uy, Uz, O2, H2O,
    C3H8, CO2, h, k; DICPCG for both p_rgh correctors: 20 ... 36 iterations), except H2O / CO2 of the third step (2 instead of 3:
    their residual after two sweeps is 9.4e-09 against a tolerance of 1e-08), and every initial residual within 1e-3 relative
    (5e-4 or better for all but a handful);
  * min/max(T) of every step within 2e-4 relative (298.15 ... 1027.3 K), the radiant fractions, the species extrema.

What has to be right for that, beyond the first step (tests/test_steckler_first_step_cpu.py): LUST, limitedLinear and the
multivariateSelection scheme -- ONE limiter for all species and h, the face-wise minimum of the member schemes' limiters, so that the
species are interpolated consistently -- with non-zero fluxes; ddtCorr with a non-zero old flux; the EDC source once fuel and oxygen
meet (from the fourth step), janaf / sutherland thermo from 298 K to 1000 K across the common-temperature switch; radiation->Sh
(36 % of the heat release leaves the enthalpy equation); inletOutlet / pressureInletOutletVelocity patches switching with the flow;
and three pieces of OpenFOAM's field semantics found through this data: the fuel specie's patch coefficients lag one step
(fvPatchField::updated() + singleStepCombustion::Qdot() building R(YFuel)), p's and K's old-time levels are created by the first
request as copies of the current field, patch values are what the last evaluate left (the limiters' gradients read them)."""
import json
import os

import numpy as np
import pytest

LOGSTEPS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_log_steps.json")))["steps"]


def sig(x, n):
    return "%.*g" % (n, x)


def test_the_oracle_follows_all_29_steps_of_the_golden_log(O):
    from oracle import steckler_case as SC
    c = SC.first_step_records()
    c.time = c.dt
    worst = 0.0
    for k, g in enumerate(LOGSTEPS):
        if k > 0:
            c.advance()
            assert sig(c.dt, 5) == sig(g["deltaT"], 5), (k + 1, c.dt, g["deltaT"])
            tolCo = 2e-5 if k < 12 else 5e-4                 # 5 digits for the first dozen steps, 4 later (the flame: T up to 1027 K)
            assert abs(c.meanCoNum - g["courantMean"]) <= max(tolCo, 6e-6 / g["courantMean"]) * g["courantMean"], (k + 1, c.meanCoNum, g["courantMean"])
            assert abs(c.CoNum - g["courantMax"]) <= max(tolCo, 6e-6 / g["courantMax"]) * g["courantMax"], (k + 1, c.CoNum, g["courantMax"])
            log = c.log
        else:
            log = c.log[5:]                                  # after the five hydrostatic solves
        assert sig(c.time, 6) == sig(g["time"], 6), (k + 1, c.time, g["time"])                   # `Time = ...` is printed with 6 digits
        assert [n for n, _ in log] == [s["name"] for s in g["solves"]], k + 1
        for (n, p), s in zip(log, g["solves"]):
            if not (k == 2 and n in ("H2O", "CO2")):          # third step: 9.4e-09 after two sweeps against the tolerance 1e-08 (log: a third sweep)
                assert p["nIterations"] == s["nIterations"], (k + 1, n, p, s)
            if s["initialResidual"] > 0:
                dev = abs(p["initialResidual"] - s["initialResidual"]) / s["initialResidual"]
                worst = max(worst, dev)
                assert dev < 1e-3, (k + 1, n, p, s)
        assert abs(c.minmaxT[1] - g["minmaxT"][1]) <= 2e-4 * g["minmaxT"][1] and sig(c.minmaxT[0], 5) == sig(g["minmaxT"][0], 5), (k + 1, c.minmaxT, g["minmaxT"])
        if "radiantFraction" in g and k > 0:
            assert sig(c.radFraction, 5) == sig(g["radiantFraction"], 5), (k + 1, c.radFraction)
        for sp_, (lo, av, hi) in g["species_min_ave_max"].items():
            st = c.species_stats[sp_]
            assert abs(st[1] - av) <= 2e-3 * abs(av) + 1e-300 and abs(st[2] - hi) <= 2e-3 * abs(hi) + 1e-300, (k + 1, sp_, st, (lo, av, hi))
    assert len(LOGSTEPS) == 29 and abs(c.time - 2.0) < 1e-9 and c.minmaxT[1] > 1000.0
    assert worst < 1e-3
