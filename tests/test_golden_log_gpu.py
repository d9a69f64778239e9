"""The same golden-log check with the HIP solver in the loop: the five hydrostatic-initialisation
DICPCG solves of the steckler case (reference cases/steckler/original/linux64/log.fireFoam:92-101)
solved by libffm through the C ABI must reproduce the golden log itself -- every iteration count and every printed
residual / gMax-gMin to 1e-6 relative (the oracle matches the log digit for digit, tests/test_golden_log_cpu.py; the HIP
solver sums its dot products in a different order, which moves the 8th digit); the assembled matrices come from the
oracle's FV operators (assembly kernels have their own parity tests)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_ph_rgh.json")))


@pytest.mark.parametrize("tiled", [False, True])
def test_steckler_hydrostatic_solves_on_gpu(O, ffm, ctx, tiled):
    """tiled = False: the library sees an unstructured mesh (the baffles break the box pattern) and uses the level-scheduled
    sweeps.  tiled = True: the host passes a group hint (8 x 8 tiles of cell columns from the cell indices) and the tiled
    sweeps run; with the baffles the backward dependency order inside a tile is not the mirror image of the forward one, so
    the backward sweep takes its position-space form (ffm_tile_plan::mirror == false)."""
    from oracle import steckler
    m = steckler.build_mesh()
    hint = None
    if tiled:                                      # from the cell centres alone, as an OpenFOAM host would
        hint = ffm.tile_hint_from_centres(m.C.T.copy(), tileCells=8)
        i, j, k = m.ijk
        assert len(set(hint)) == len(set((j // 8 + 100 * (k // 8))))
    A = ffm.lduMatrix(ctx, m.nCells, m.l, m.u, groupHint=hint)      # natural numbering -> internal permutation
    assert A.sweep_mode == (2 if tiled else 0)

    def gpu_solve(mesh, diag, upper, source, psi0):
        A.set_coeffs(diag, upper)
        psi = ctx.to_device(psi0)
        perf = A.solve(psi, ctx.to_device(source), solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0.01)
        return psi.cpu().numpy(), perf

    recs, ph = steckler.hydrostatic_initialisation(gpu_solve, mesh=m)
    ref, phRef = steckler.hydrostatic_initialisation(steckler.oracle_solve, mesh=m)
    assert [r["nIterations"] for r in recs] == [r["nIterations"] for r in ref]
    assert [r["nIterations"] for r in recs] == [g["nIterations"] for g in GOLD["solves"]] == [29, 32, 7, 0, 0]
    for r, o, g in zip(recs, ref, GOLD["solves"]):
        assert abs(r["finalResidual"] - o["finalResidual"]) <= 1e-6 * o["finalResidual"]
        for key in ("initialResidual", "finalResidual", "variation"):
            assert abs(r[key] - g[key]) <= 1e-6 * g[key], (key, r[key], g[key])
    assert np.linalg.norm(ph - phRef) / np.linalg.norm(phRef) < 1e-8
    A.close()
