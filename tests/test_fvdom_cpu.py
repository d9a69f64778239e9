"""oracle/fvdom.py (fvDOM with its iteration and grey-diffusive walls).
 * pin: with the steckler selections (32 rays, maxIter 1, every wall emissivity 1, Gauss upwind, GAMG + DILU) it gives the 32 ray
   solves of oracle/steckler_case.py::radiation_correct, which reproduce the golden log (tests/test_steckler_first_step_cpu.py) --
   same iteration counts, same intensities;
 * the 2-D ray set of fvDOM::initialise (fvDOM.C:98-135): 4 nPhi rays in the x-y plane, solid angles summing to 4 pi, sum dAve = 0;
 * reflecting walls (greyDiffusiveRadiationMixedFvPatchScalarField.C:190-221) and the iteration (fvDOM.C:561-584) -- unpinned by
   reference data, checked by what they imply: an isothermal enclosure at T with any wall emissivities converges to the black-body
   field I = sigma T^4/pi, G = 4 sigma T^4, net wall flux 0; a wall with emissivity e reflects (1 - e) of what falls on it; the
   iteration stops at maxIter or when every ray's scaled initial residual is below the tolerance, converged rays are skipped."""
import numpy as np
import pytest

from common import rel_l2


def _pbicgstab(O, m, tol=1e-10):
    def solve(name, d, upper, lower, s, psi0):
        A = O.Ldu(m.nCells, m.l, m.u).set_coeffs(d, upper, lower)
        return A.solve(O.PBICGSTAB, O.DILU, psi0, s, tolerance=tol, relTol=0.0, maxIter=1000)
    return solve


def test_fvdom_reproduces_the_log_pinned_steckler_rays(O):
    from oracle import steckler_case as SC, fvdom, gamg
    c = SC.first_step_records(with_h=False, with_radiation=True)
    m = c.m
    agg = gamg.Agglomeration(m.nCells, m.l, m.u, gamg.face_area_pair_weights(m.Sf), nCellsInCoarsestLevel=10, mergeLevels=1)

    def solve(name, d, upper, lower, s, psi0):
        return gamg.GAMGSolver(agg, d, upper, lower, smoother="DILU").solve(psi0, s, tolerance=1e-4, relTol=0.0)
    dom = fvdom.FvDOM(m, 2, 4, solve, maxIter=1, tolerance=0.0)
    dom.calculate(c.T, c.Tb, 0.0, c.radFraction * c.Qdot)
    rays = [(n, p) for n, p in c.log if n.startswith("ILambda_")]
    assert [p["nIterations"] for _, p in dom.log] == [p["nIterations"] for _, p in rays]
    assert dom.nIterations == 1
    assert max(rel_l2(a, b) for a, b in zip(dom.I, c.I)) < 1e-12 and rel_l2(dom.G, c.G) < 1e-12


@pytest.mark.parametrize("nPhi", [1, 2, 3])
def test_two_dimensional_ray_set(nPhi):
    from oracle import fvdom
    rays = fvdom.ray_set(nPhi, 7, solutionD=(1, 1, -1))
    assert len(rays) == 4 * nPhi
    assert abs(sum(r[2] for r in rays) - 4.0 * np.pi) < 1e-13
    assert np.abs(sum(r[1] for r in rays)).max() < 1e-14
    assert all(abs(r[0][2]) < 1e-15 and abs(r[1][2]) < 1e-15 for r in rays)            # in the x-y plane
    with pytest.raises(ValueError):
        fvdom.ray_set(nPhi, 2, solutionD=(-1, 1, 1))                                  # fvDOM.C:103-109
    assert len(fvdom.ray_set(2, 4)) == 32 and len(fvdom.ray_set(2, 2, solutionD=(1, -1, -1))) == 2


def _box(n, empty=()):
    from oracle import plume
    return plume.make_mesh(n, empty=empty)


@pytest.mark.parametrize("shape,empty,solD", [((6, 7, 5), (), (1, 1, 1)), ((9, 11, 1), ("zmin", "zmax"), (1, 1, -1))])
def test_isothermal_enclosure_with_reflecting_walls_is_a_black_body(O, shape, empty, solD):
    from oracle import fvdom
    m = _box(shape, empty)
    T = np.full(m.nCells, 800.0); Tb = [np.full(p.size, 800.0) for p in m.patches]
    emis = [np.full(p.size, e) for p, e in zip(m.patches, (0.3, 0.85, 1.0, 0.55))]
    dom = fvdom.FvDOM(m, 2, 2, _pbicgstab(O, m), maxIter=5, tolerance=1e-9, solutionD=solD, emissivity=emis)
    # one calculate() per radiation->correct() of the time loop: a ray whose boundary values did not change in an iteration is flagged
    # converged for the rest of THAT call (rayIdConv, fvDOM.C:553,575-578) although the walls' reflected part still lags -- the
    # reference's algorithm reaches the stationary field over successive calls
    for _ in range(60):
        dom.calculate(T, Tb, 0.1, np.zeros(m.nCells))
    Ibb = fvdom.SIGMA_SB * 800.0 ** 4 / np.pi
    for I in dom.I:
        assert np.abs(I - Ibb).max() < 1e-6 * Ibb
    assert np.abs(dom.G - 4.0 * np.pi * Ibb).max() < 1e-6 * 4.0 * np.pi * Ibb
    for q in range(len(m.patches)):
        # what falls on a wall equals what leaves it (emitted + reflected): qem holds the outgoing side with the sign of nAve (< 0)
        assert np.abs(dom.qin[q] + dom.qem[q]).max() < 1e-6 * np.abs(dom.qin[q]).max()
        assert np.abs(dom.qr[q]).max() < 1e-5 * np.abs(dom.qin[q]).max()


def test_cold_reflecting_wall_returns_what_it_does_not_absorb(O):
    """a transparent medium between a hot black wall (floor + inlet, 1000 K) and cold walls (1 K) of which `top` reflects: with
    emissivity e the top wall's outgoing flux is (1 - e) of its incident flux, and with e = 1 nothing comes back; the iteration
    stops at maxIter."""
    from oracle import fvdom
    m = _box((5, 6, 5))
    names = [p.name for p in m.patches]
    T = np.full(m.nCells, 1.0)
    Tb = [np.full(p.size, 1000.0 if p.name in ("inlet", "floor") else 1.0) for p in m.patches]
    res = {}
    for e in (1.0, 0.4):
        emis = [np.full(p.size, e if p.name == "top" else 1.0) for p in m.patches]
        dom = fvdom.FvDOM(m, 2, 2, _pbicgstab(O, m), maxIter=50, tolerance=1e-8, emissivity=emis)
        nIt = 0
        for _ in range(12):
            dom.calculate(T, Tb, 0.0, np.zeros(m.nCells)); nIt += dom.nIterations
        q = names.index("top")
        res[e] = (nIt, dom.qin[q].copy(), dom.qem[q].copy())
    assert np.abs(res[1.0][2]).max() < 1e-9 * res[1.0][1].max()                          # black and cold: nothing leaves
    assert np.allclose(-res[0.4][2], 0.6 * res[0.4][1], rtol=1e-6)                        # grey: (1 - e) of the incident flux leaves
    capped = fvdom.FvDOM(m, 2, 2, _pbicgstab(O, m), maxIter=2, tolerance=1e-12, emissivity=[np.full(p.size, 0.4) for p in m.patches])
    capped.calculate(T, Tb, 0.0, np.zeros(m.nCells))
    assert capped.nIterations == 2 and len(capped.log) == 2 * 16


def test_linear_upwind_ray_convection_and_converged_rays_are_skipped(O):
    """div(Ji,Ii_h) Gauss linearUpwind grad(Ii_h) (cases/wallFireSpread2D/system/fvSchemes:66): the explicit correction uses the
    intensities of the previous solve, so the first iteration equals upwind's and the iteration converges to a different (second-order)
    field; a ray whose scaled initial residual fell below the tolerance is not solved again (rayIdConv)."""
    from oracle import fvdom
    m = _box((9, 11, 1), ("zmin", "zmax"))
    T = 600.0 + 300.0 * np.sin(3.0 * m.C[:, 0]) * np.cos(2.0 * m.C[:, 1]); Tb = [T[p.faceCells] for p in m.patches]
    mk = lambda scheme, it, tol: fvdom.FvDOM(m, 2, 2, _pbicgstab(O, m), maxIter=it, tolerance=tol, divScheme=scheme, solutionD=(1, 1, -1))
    up, lu = mk("upwind", 1, 0.0), mk("linearUpwind", 1, 0.0)
    up.calculate(T, Tb, 0.5, np.zeros(m.nCells)); lu.calculate(T, Tb, 0.5, np.zeros(m.nCells))
    assert rel_l2(lu.G, up.G) < 1e-14
    lu5 = mk("linearUpwind", 5, 1e-3); lu5.calculate(T, Tb, 0.5, np.zeros(m.nCells))       # the case's maxIter 5, convergence 1e-3
    assert 1e-4 < rel_l2(lu5.G, up.G) < 0.2 and 1 < lu5.nIterations <= 5
    assert len(lu5.log) <= 8 * lu5.nIterations
    up3 = mk("upwind", 3, 1e-3); up3.calculate(T, Tb, 0.5, np.zeros(m.nCells))
    assert up3.nIterations == 2 and len(up3.log) == 16                                     # black walls, upwind: the second pass finds every ray converged
