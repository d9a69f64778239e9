"""Parity pin of the oracle against the reference's only golden data for the hot path: the five
DICPCG solves of the hydrostatic initialisation in the steckler golden log
(reference cases/steckler/original/linux64/log.fireFoam:92-101).

What is pinned: the oracle reproduces the golden log DIGIT FOR DIGIT -- every iteration count (29, 32, 7, 0, 0), all 8
printed digits of every initial and final residual and of every `gMax-gMin` (asserted to 1e-7 relative: the one visible
difference is 9.6500999e-07 against the printed 9.6501e-07, i.e. 1e-8).  That pins, through the reference's own output:
PCG control flow and stopping rule, normFactor, DIC (calcReciprocalD + precondition) in face order, fvm::laplacian and its
boundary coefficients (fixedValue, fixedGradient via constrainPressure), fvc::interpolate, fvc::snGrad, fvc::div /
surfaceIntegrate, the baffle/doorway mesh surgery and the LDU addressing.  Operators the hydrostatic initialisation does not
touch (convection schemes and limiters, ddt, DILU/PBiCGStab/GaussSeidel, H/A/flux, relax) remain 'parity unpinned' by
reference data: they are checked against dense algebra, hand-computed stencils and identities instead.

Two case details carry the match (found by reading the case files, see oracle/steckler.py): the doorway box is inclusive of
the face centres at z = +-0.5, and the boundary mixture on `top` is O2 alone because cases/steckler/0/N2 leaves N2 = 0 on
that patch until the first YEEqn (its density, +10.9 %, is the fixedValue coefficient of the top cell layer)."""
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_ph_rgh.json")))


def test_hydrostatic_initialisation_against_golden_log(O):
    from oracle import steckler
    recs, ph = steckler.hydrostatic_initialisation(steckler.oracle_solve)
    gold = GOLD["solves"]
    assert [r["nIterations"] for r in recs] == [g["nIterations"] for g in gold] == [29, 32, 7, 0, 0]
    assert recs[0]["initialResidual"] == 1.0
    for r, g in zip(recs, gold):
        for key in ("initialResidual", "finalResidual", "variation"):
            assert abs(r[key] - g[key]) <= 1e-7 * g[key], (key, r[key], g[key])
        # and as the log prints them (8 significant digits); the third final residual is the 1e-8 case named above
        assert "%.8g" % r["initialResidual"] == "%.8g" % g["initialResidual"]
        assert "%.8g" % r["variation"] == "%.8g" % g["variation"]


def test_boundary_mixture_of_the_top_patch_is_what_matches(O, monkeypatch):
    """with the interior's mixture on `top` (N2 boundary value taken as 0.76699 instead of the file's 0) the first final
    residual is 0.00836, not the golden 0.0080439052"""
    from oracle import steckler
    real = steckler.fv.interpolate
    recs, _ = steckler.hydrostatic_initialisation(steckler.oracle_solve, nCorr=1)
    assert "%.8g" % recs[0]["finalResidual"] == "0.0080439052"
    W = 1.0 / (0.23301 / 31.9988 + 0.76699 / 28.0134)

    def interior_mixture_on_top(m, rho, rhob):
        rhob = [b * (W / 31.9988) if p.name == "top" else b for p, b in zip(m.patches, rhob)]
        return real(m, rho, rhob)
    monkeypatch.setattr(steckler.fv, "interpolate", interior_mixture_on_top)
    recs, _ = steckler.hydrostatic_initialisation(steckler.oracle_solve, nCorr=1)
    assert recs[0]["nIterations"] == 29 and abs(recs[0]["finalResidual"] - 0.00836) < 5e-5


def test_doorway_borderline_choice_is_the_one_that_matches(O):
    """The four readings of the doorway box (faces at z=+-0.5 in or out) give 34/29/33/33 first-solve
    iterations; only 'both in' matches the golden 29."""
    from oracle import steckler
    counts = {}
    for dk in ((8, 11), (7, 12), (7, 11), (8, 12)):
        recs, _ = steckler.hydrostatic_initialisation(steckler.oracle_solve, nCorr=1, mesh=steckler.build_mesh(dk))
        counts[dk] = recs[0]["nIterations"]
    assert counts[(7, 12)] == 29 and all(v != 29 for k, v in counts.items() if k != (7, 12))


def test_fvdom_ray_solid_angles_against_golden_log(O):
    """the 32 solid angles the reference prints at start-up (log.fireFoam:125-156), digit for digit: pins the ray set of the
    fvDOM stand-in (oracle/plume.py ray_set; theta outer / phi inner, omega = 2 sin(theta) sin(dTheta/2) dPhi)"""
    from oracle import plume
    rays = plume.ray_set(2, 4)
    gold = GOLD["fvDOM_ray_omega"]["omega"]
    assert len(rays) == len(gold) == 32
    for (_, omega), g in zip(rays, gold):
        assert "%.8g" % omega == "%.8g" % g


def test_steckler_geometry_of_the_oracle_follows_the_reference_case_files(O):
    """oracle/steckler.py hard-codes the room as cell index ranges; here the same ranges are derived from the numbers in the
    reference's own files -- blockMeshDict (vertices, convertToMeters, cells), topoSetDictCompartment (compartment and doorway
    boxes) -- and the baffle / doorway face counts of the oracle's mesh are compared with them.  Skipped where the reference is
    not mounted."""
    import re
    import pytest
    case = "/root/reference/cases/steckler"
    if not os.path.exists(os.path.join(case, "system", "topoSetDictCompartment")):
        pytest.skip("reference not mounted")
    from oracle import steckler
    strip = lambda s: re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", s, flags=re.S))
    bm = strip(open(os.path.join(case, "constant", "polyMesh", "blockMeshDict")).read())
    scale = float(re.search(r"convertToMeters\s+([0-9.eE+-]+)", bm).group(1))
    verts = np.array([[float(x) for x in v.split()] for v in re.findall(r"\(\s*(-?[\d.]+\s+-?[\d.]+\s+-?[\d.]+)\s*\)", bm.split("vertices")[1].split("blocks")[0])]) * scale
    n = tuple(int(x) for x in re.search(r"hex\s*\([^)]*\)\s*\(\s*(\d+)\s+(\d+)\s+(\d+)\s*\)", bm).groups())
    lo, hi = verts.min(axis=0), verts.max(axis=0)
    assert n == (30, 15, 20) and np.allclose(lo, (-2, 0, -2)) and np.allclose(hi, (4, 3, 2))
    ts = strip(open(os.path.join(case, "system", "topoSetDictCompartment")).read())
    boxes = [np.array([float(x) for x in (a + " " + b).split()]).reshape(2, 3)
             for a, b in re.findall(r"box\s*\(([^)]*)\)\s*\(([^)]*)\)", ts)]
    room = next(b for b in boxes if np.allclose(b, [(-1.4, 0, -1.4), (1.4, 2.18, 1.4)]))
    door = next(b for b in boxes if np.allclose(b, [(0, 0, -0.5), (10, 1, 0.5)]))
    d = (hi - lo) / np.array(n)
    m = steckler.build_mesh()
    centres = [lo[a] + (np.arange(n[a]) + 0.5) * d[a] for a in range(3)]
    inside = [np.nonzero((centres[a] >= room[0, a]) & (centres[a] <= room[1, a]))[0] for a in range(3)]
    assert (inside[0][0], inside[0][-1], inside[1][-1], inside[2][0], inside[2][-1]) == (3, 16, 10, 3, 16)      # oracle/steckler.py:build_mesh
    ni, nj, nk = (len(x) for x in inside)
    # shell of the room = faces between a room cell and an outside cell (the floor, y = 0, is a boundary, not a baffle)
    shell = 2 * nj * nk + 2 * ni * nj + ni * nk
    eps = 1e-9
    door_j = np.nonzero(centres[1] <= door[1, 1] + eps)[0]                      # face centres of the x = 1.4 wall: same y, z as the cells
    door_k = np.nonzero((centres[2] >= door[0, 2] - eps) & (centres[2] <= door[1, 2] + eps))[0]     # inclusive: z = +-0.5 are in
    assert (door_j[-1], door_k[0], door_k[-1]) == (4, 7, 12)
    baffle = m.patch("baffle1DWall_master").size
    assert baffle == m.patch("baffle1DWall_slave").size == shell - len(door_j) * len(door_k)
