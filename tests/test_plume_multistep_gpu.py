"""Multi-step parity of the compiled time step at north_star's 1e-8, under conditions that CAN fail.

tests/test_plume_gpu.py compares the steps after the first at 1e-5, because the species and h of the quiescent plume case are
uniform up to solver noise away from the plume and the common limiter of `Gauss multivariateSelection` (solver/YEEqn.H:1-10) is
decided by that noise (tests/test_plume_cpu.py shows the sensitivity on the oracle itself).  That leaves the question whether a
real multi-step discrepancy of 1e-6 -- old-time levels, ddtCorr, lagging boundary coefficients -- would hide behind the loosened
tolerance.  Three tests answer it, each holding every transported field to 1e-8 rel-L2 with identical iteration counts over
several steps:
  1. the conditioned case: every specie and h varies smoothly over the box (no field is uniform anywhere, no exact zeros, inflow and
     ambient values distinct; oracle/plume.py:conditioned_state), so the limiter is well-conditioned on every face
     (tests/test_plume_cpu.py::test_the_conditioned_start_state_is_well_conditioned) -- common limiter ON, six steps;
  2. the quiescent case with one limiter per field (FFM_PLUME_INDEPENDENT_LIMITERS=1, the form of round 2's first session): five steps;
  3. the deciding test on the quiescent case itself: the common limiter's face weights are taken from the oracle and handed to the
     device (ffm_plume_override_mv_weights), everything else of the step -- gradients, assembly, boundary coefficients, old-time
     levels, solves, pressure correctors -- is the device's own: 1e-8 on every step.
If any of these fails, the cause is a defect in the step, not the conditioning of the limiter."""
import os

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu

FIELDS = ["rho", "p", "T", "h", "Ux", "Uy", "Uz", "O2", "H2O", "C3H8", "CO2", "N2", "K"]


def _compare(step, gpu, ref, tol=1e-8, p_rgh_tol=1e-7):
    it_ref = [(nme, pf["nIterations"]) for nme, pf in ref.sol.log]
    it_gpu = [(nme, pf["nIterations"]) for nme, pf in gpu.solves()]
    assert it_ref == it_gpu, (step, it_ref, it_gpu)
    f = ref.fields()
    errs = {}
    for name in FIELDS:
        a, b = gpu.field(name), f[name]
        errs[name] = np.abs(a).max() if np.linalg.norm(b) < 1e-30 else rel_l2(a, b)
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (step, bad)
    # p_rgh is a small fluctuation on top of p (which is held to 1e-8 above): against the scale of its own variation
    a, b = gpu.field("p_rgh"), f["p_rgh"]
    e = np.linalg.norm(a - b) / max(np.linalg.norm(b - b.mean()), 1e-30)
    assert e < p_rgh_tol, (step, "p_rgh", e)
    return max(errs.values())


@pytest.mark.parametrize("n", [(12, 16, 12), (16, 20, 14)])
def test_conditioned_case_six_steps_at_1e8_with_the_common_limiter(O, ffm, ctx, n):
    from oracle import plume
    ref = plume.Plume(n, conditioned=True)
    gpu = ffm.Plume(ctx, n)
    m = ref.m
    Y0, h0 = plume.conditioned_state(m, plume.Y_AMB_COND, plume.H_AMB_COND)
    gpu.set_initial_state(Y0, h0, plume.Y_AMB_COND, plume.Y_IN_COND, plume.H_AMB_COND)
    assert [pf["nIterations"] for _, pf in gpu.solves()] == [pf["nIterations"] for _, pf in ref.sol.log]      # the hydrostatic solves
    assert rel_l2(gpu.field("ph_rgh"), ref.ph_rgh) < 1e-8 and rel_l2(gpu.field("rho"), ref.rho) < 1e-13
    worst = 0.0
    for step in range(6):
        ref.step(); gpu.step()
        worst = max(worst, _compare(step, gpu, ref))
    # the case is not trivial: the limiter is active (weights strictly between upwind and linear on a good share of the faces)
    w = ref.w_mv_last
    mixed = ((w > 0.5 + 1e-6) & (w < 1.0 - 1e-6)) | ((w < 0.5 - 1e-6) & (w > 1e-6))
    assert mixed.mean() > 0.02, mixed.mean()
    print("conditioned case, worst rel-L2 over six steps: %.2e" % worst)
    gpu.close()


def test_quiescent_case_five_steps_at_1e8_with_independent_limiters(O, ffm, ctx):
    from oracle import plume
    n = (12, 16, 12)
    ref = plume.Plume(n); ref.mv_selection = False
    os.environ["FFM_PLUME_INDEPENDENT_LIMITERS"] = "1"
    try:
        gpu = ffm.Plume(ctx, n)
    finally:
        del os.environ["FFM_PLUME_INDEPENDENT_LIMITERS"]
    for step in range(5):
        ref.step(); gpu.step()
        _compare(step, gpu, ref, p_rgh_tol=1e-5)
    gpu.close()


def test_quiescent_case_with_the_oracles_limiter_weights_handed_in(O, ffm, ctx):
    """the deciding test: with the one ill-conditioned quantity of the step (the common limiter's weights on round-off-uniform
    fields) taken out of the comparison, the steps after the first agree to 1e-8 as well"""
    from oracle import plume
    n = (12, 16, 12)
    ref = plume.Plume(n)
    gpu = ffm.Plume(ctx, n)
    free = ffm.Plume(ctx, n)
    for step in range(4):
        ref.step()
        gpu.override_mv_weights(ref.w_mv_last)
        gpu.step(); free.step()
        _compare(step, gpu, ref, p_rgh_tol=1e-5)
    # and the free-running limiter stays within the 1e-5 of tests/test_plume_gpu.py of both
    for name in ("h", "O2", "C3H8", "Uy"):
        assert rel_l2(free.field(name), gpu.field(name)) < 1e-5, name
    gpu.close(); free.close()
