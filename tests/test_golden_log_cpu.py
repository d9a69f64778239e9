"""Parity pin of the oracle against the reference's only golden data for the hot path: the five
DICPCG solves of the hydrostatic initialisation in the steckler golden log
(reference cases/steckler/original/linux64/log.fireFoam:92-101).

What is pinned, honestly: the oracle reproduces the golden iteration counts of solves 1, 2, 4 and 5
exactly (29, 32, 0, 0), solve 3 within one iteration (8 vs 7), every residual within 15 %, the
initial residual of exactly 1, and the converged hydrostatic variation to 2e-6 relative.  The
residuals are NOT reproduced digit for digit (first final residual 0.00836 vs 0.00804); the cause
was not found (candidates: face-centre rounding in the doorway selection, an un-shipped
difference between the golden build d773a7a and the current dictionaries).  PCG control flow,
normFactor, DIC ordering, Laplacian and boundary-coefficient assembly are therefore pinned by
iteration counts, not bitwise; every other oracle function is 'parity unpinned'."""
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_ph_rgh.json")))


def test_hydrostatic_initialisation_against_golden_log(O):
    from oracle import steckler
    recs, ph = steckler.hydrostatic_initialisation(steckler.oracle_solve)
    gold = GOLD["solves"]
    assert [r["nIterations"] for r in recs][:2] == [29, 32]
    assert abs(recs[2]["nIterations"] - gold[2]["nIterations"]) <= 1
    assert [r["nIterations"] for r in recs][3:] == [0, 0]
    assert recs[0]["initialResidual"] == 1.0
    for r, g in zip(recs, gold):
        assert abs(r["finalResidual"] - g["finalResidual"]) <= 0.15 * g["finalResidual"]
        assert abs(r["initialResidual"] - g["initialResidual"]) <= 0.15 * g["initialResidual"]
        assert abs(r["variation"] - g["variation"]) <= 5e-4 * g["variation"]
    assert abs(recs[-1]["variation"] - gold[-1]["variation"]) <= 5e-6 * gold[-1]["variation"]


def test_doorway_borderline_choice_is_the_one_that_matches(O):
    """The four readings of the doorway box (faces at z=+-0.5 in or out) give 34/29/33/33 first-solve
    iterations; only 'both in' matches the golden 29."""
    from oracle import steckler
    counts = {}
    for dk in ((8, 11), (7, 12), (7, 11), (8, 12)):
        recs, _ = steckler.hydrostatic_initialisation(steckler.oracle_solve, nCorr=1, mesh=steckler.build_mesh(dk))
        counts[dk] = recs[0]["nIterations"]
    assert counts[(7, 12)] == 29 and all(v != 29 for k, v in counts.items() if k != (7, 12))
