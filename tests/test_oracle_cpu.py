"""CPU tests of the oracle itself (no GPU): dense cross-checks, solver identities and the
in-process multi-domain runner.  The reference has no unit tests; its one golden log pins PCG/DIC/laplacian/snGrad/
interpolate digit for digit (tests/test_golden_log_cpu.py), the remaining operators are checked here against dense algebra."""
import numpy as np
import pytest

from common import laplacian_like, random_dag_mesh, rel_l2


def dense(N, l, u, diag, up, lo=None):
    D = np.zeros((N, N))
    D[np.arange(N), np.arange(N)] = diag
    D[l, u] = up
    D[u, l] = up if lo is None else lo
    return D


@pytest.mark.parametrize("n", [3, 6])
def test_amul_tmul_sumA_residual_vs_dense(O, n):
    N, l, u = O.hex_ldu(n, n + 1, n + 2)
    diag, up, lo = laplacian_like(O, N, l, u, asym=0.4)
    A = O.Ldu(N, l, u).set_coeffs(diag, up, lo)
    D = dense(N, l, u, diag, up, lo)
    x = O.hash_u(0xF4, np.arange(N)); b = O.hash_u(0xF3, np.arange(N))
    assert np.abs(A.amul(x) - D @ x).max() < 1e-13
    assert np.abs(A.tmul(x) - D.T @ x).max() < 1e-13
    assert np.abs(A.sumA() - D.sum(axis=1)).max() < 1e-13
    assert np.abs(A.residual(x, b) - (b - D @ x)).max() < 1e-13


def test_hex_numbering_matches_blockmesh_rule(O):
    nx, ny, nz = 4, 3, 2
    N, l, u = O.hex_ldu(nx, ny, nz)
    assert N == 24 and len(l) == (nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1)
    assert np.all(l < u) and np.all(np.diff(l) >= 0)
    # within one owner the neighbours ascend (upper-triangular order)
    same = l[1:] == l[:-1]
    assert np.all(u[1:][same] > u[:-1][same])
    assert set(np.unique(u - l)) == {1, nx, nx * ny}


def test_dic_is_incomplete_cholesky_exact_on_tridiagonal(O):
    # 1-D chain: IC(0) == exact Cholesky, so one DIC-PCG iteration solves the system
    n = 50
    N, l, u = O.hex_ldu(n, 1, 1)
    diag, up, _ = laplacian_like(O, N, l, u)
    A = O.Ldu(N, l, u).set_coeffs(diag, up)
    b = O.hash_u(3, np.arange(N))
    psi, perf = A.solve(O.PCG, O.DIC, np.zeros(N), b, tolerance=1e-12)
    assert perf["nIterations"] == 1
    assert rel_l2(dense(N, l, u, diag, up) @ psi, b) < 1e-12


@pytest.mark.parametrize("solver,precond", [("PCG", "DIC"), ("PBICGSTAB", "DILU"), ("PBICG", "DILU"),
                                            ("SMOOTH", "SYMGS"), ("SMOOTH", "GS"), ("PCG", "NONE"),
                                            ("PBICGSTAB", "NONE"), ("PCG", "DIAGONALP")])
def test_solvers_converge_to_dense_solution(O, solver, precond):
    n = 7
    N, l, u = O.hex_ldu(n, n, n)
    sym = solver == "PCG"
    diag, up, lo = laplacian_like(O, N, l, u, asym=0.0 if sym else 0.3, shift=0.05)
    A = O.Ldu(N, l, u).set_coeffs(diag, up, lo)
    b = 2 * O.hash_u(0xF3, np.arange(N)) - 1
    psi, perf = A.solve(getattr(O, solver), getattr(O, precond), np.zeros(N), b, tolerance=1e-11, maxIter=2000)
    assert perf["converged"] == 1 and perf["initialResidual"] == 1.0
    ref = np.linalg.solve(dense(N, l, u, diag, up, lo), b)
    assert rel_l2(psi, ref) < 1e-8


def test_diagonal_solver_and_zero_iteration_exit(O):
    N, l, u = O.hex_ldu(3, 3, 3)
    diag = 1.0 + O.hash_u(1, np.arange(N))
    A = O.Ldu(N, l, u).set_coeffs(diag, np.zeros(len(l)))
    b = O.hash_u(2, np.arange(N))
    psi, perf = A.solve(O.DIAGONAL, O.NONE, np.zeros(N), b)
    assert np.array_equal(psi, b / diag) and perf["nIterations"] == 0 and perf["finalResidual"] == 0.0
    # already converged initial guess: PCG must not iterate (golden log: 'No Iterations 0')
    psi2, perf2 = A.solve(O.PCG, O.DIC, psi, b, tolerance=1e-6)
    assert perf2["nIterations"] == 0 and np.array_equal(psi2, psi)


def test_relTol_and_maxIter_semantics(O):
    n = 8
    N, l, u = O.hex_ldu(n, n, n)
    diag, up, _ = laplacian_like(O, N, l, u)
    A = O.Ldu(N, l, u).set_coeffs(diag, up)
    b = 2 * O.hash_u(0xF3, np.arange(N)) - 1
    _, p1 = A.solve(O.PCG, O.DIC, np.zeros(N), b, tolerance=1e-12, relTol=0.01)
    assert p1["finalResidual"] < 0.01 * p1["initialResidual"] and p1["finalResidual"] > 1e-12
    _, p2 = A.solve(O.PCG, O.DIC, np.zeros(N), b, tolerance=1e-30, maxIter=3)
    assert p2["nIterations"] == 3 and p2["converged"] == 0
    _, p3 = A.solve(O.SMOOTH, O.SYMGS, np.zeros(N), b, tolerance=1e-30, maxIter=10, nSweeps=4)
    assert p3["nIterations"] == 12   # counts in steps of nSweeps, as smoothSolver does


def test_renumbered_mesh_gives_same_dic_pcg(O, ffm):
    """Level-major renumbering is a topological order of the same DAG: DIC-PCG on the renumbered
    matrix follows the same iteration history (hex box: identical row accumulation order)."""
    n = 6
    N, l, u = O.hex_ldu(n, n + 2, n + 1)
    diag, up, _ = laplacian_like(O, N, l, u)
    b = 2 * O.hash_u(0xF3, np.arange(N)) - 1
    psi0, perf0 = O.Ldu(N, l, u).set_coeffs(diag, up).solve(O.PCG, O.DIC, np.zeros(N), b, tolerance=1e-10)
    cOrd, fOrd = ffm.renumber_levels(N, l, u)
    l2, u2, _ = ffm.hexmesh.apply_renumbering(N, l, u, cOrd, fOrd)
    psi1, perf1 = O.Ldu(N, l2, u2).set_coeffs(diag[cOrd], up[fOrd]).solve(O.PCG, O.DIC, np.zeros(N), b[cOrd], tolerance=1e-10)
    assert perf1["nIterations"] == perf0["nIterations"]
    assert rel_l2(psi1, psi0[cOrd]) < 1e-13


def _decomposed_case(O, ffm, glob, grid):
    from oracle import multi
    return multi.decomposed_case(ffm.hexmesh, glob, grid)


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])
def test_multi_domain_matches_serial(O, ffm, grid):
    """T8: decomposed (block-Jacobi DIC) vs serial: converged fields agree <= 1e-10 at tolerance 1e-12."""
    H = ffm.hexmesh
    glob = (8, 6, 8)
    whole = H.HexBlock(glob)
    s = H.synth_p_rgh(whole)
    psiS, perfS = O.Ldu(whole.nCells, whole.l, whole.u).set_coeffs(s["diag"], s["upper"]).solve(
        O.PCG, O.DIC, np.zeros(whole.nCells), s["source"], tolerance=1e-12)
    blocks, nbrRank, nbrPatch, ldus, srcs = _decomposed_case(O, ffm, glob, grid)
    psis, perfs = O.solve_multi(ldus, nbrRank, nbrPatch, O.PCG, O.DIC, [np.zeros(b.nCells) for b in blocks], srcs,
                                tolerance=1e-12)
    assert all(p["converged"] for p in perfs)
    assert len({p["nIterations"] for p in perfs}) == 1
    assert abs(perfs[0]["initialResidual"] - perfS["initialResidual"]) < 1e-12
    full = np.empty(whole.nCells)
    for b, p in zip(blocks, psis):
        full[b.gcell] = p
    assert rel_l2(full, psiS) < 1e-10


def test_fvdom_ray_set_properties(O):
    """the 32-ray set of the fvDOM stand-in (fvDOM.C:55-90, radiativeIntensityRay.C:126-143): solid angles tile the sphere, the
    mean directions cancel, and an isothermal box at the ambient temperature is in radiative equilibrium (G = 4 sigma T^4)."""
    from oracle import plume
    rays = plume.ray_set(2, 4)
    assert len(rays) == 32
    assert abs(sum(o for _, o in rays) - 4 * np.pi) < 1e-13
    assert np.abs(sum(d for d, _ in rays)).max() < 1e-14
    P = plume.Plume((6, 8, 6)); P.set_radiation(solverFreq=1)
    P.T[:] = plume.TREF
    P.radiation_correct()
    assert np.abs(P.G / (4 * plume.SIGMA_SB * plume.TREF ** 4) - 1).max() < 1e-4       # solves stop at 1e-4


def test_limiter_uniform_zones_are_linear(O):
    """NVDTVD::r with OpenFOAM's two-valued sign() (s >= 0 ? 1 : -1): in a uniform zone gradf = gradcf = 0 gives r = 1999, the
    limiter is 1 and the weights are the LINEAR ones (a three-valued sign would give r = -1 and upwind weights over most of
    the ambient of the reference cases' species / enthalpy fields); gradf = 0 with gradcf < 0 gives r = -2001 -> upwind."""
    from oracle import fv
    m = fv.HexMesh((6, 1, 1), (0, 0, 0), (6, 1, 1)).set_patches([])
    vf = np.array([0.2, 0.2, 0.2, 0.7, 0.7, 0.7])               # piecewise constant
    g = np.zeros((6, 3)); g[:, 0] = [0.0, 0.0, 0.25, 0.25, 0.0, 0.0]
    for sgn in (1.0, -1.0):
        for scheme in ("limitedLinear", "limitedLinear01"):
            w = fv.limited_weights(m, scheme, sgn * np.ones(5), vf, g, 1.0)
            assert w[0] == 0.5 and w[4] == 0.5, (scheme, sgn, w)   # both cells and the upwind gradient uniform: linear
    g2 = np.zeros((6, 3)); g2[:, 0] = -0.1
    w = fv.limited_weights(m, "limitedLinear", np.ones(5), vf, g2, 1.0)
    assert w[0] == 1.0                                          # gradf = 0, gradcf < 0: r = 2000*(-1)(+1) - 1 < 0 -> upwind


def test_filteredLinear2V_unit_stencils(O):
    """filteredLinear2V k l (cases/wallFireSpread2D/system/fvSchemes:41: 0.2 0.05), hand-computed on a 1-D row of cells, h = 1:
    * a linear profile U_x = a x: df = a^2, tcP = tcN = 2 a^2 -> nothing to limit, limiter = min(1 + l, 1) = 1: linear weights;
    * a uniform field: df = tc = 0 -> limiter = l + 1 - k*0/(0 + SMALL) -> 1: linear weights;
    * a kink: cell gradients (centred) smaller than the face difference -> limiter = 1 + l - k*min(df - tcP, df - tcN)/max|tc|,
      weights blended towards upwind accordingly; the limiter does not depend on the flux direction, the upwind weight does"""
    from oracle import fv
    m = fv.HexMesh((5, 1, 1), (0, 0, 0), (5, 1, 1)).set_patches([])
    U = np.zeros((5, 3)); g = np.zeros((5, 3, 3))
    U[:, 0] = 2.0 * np.arange(5); g[:, 0, 0] = 2.0
    w = fv.filtered_linear2V_weights(m, np.ones(4), U, g, 0.2, 0.05)
    assert np.array_equal(w, np.full(4, 0.5))
    w = fv.filtered_linear2V_weights(m, -np.ones(4), np.ones((5, 3)), np.zeros((5, 3, 3)), 0.2, 0.05)
    assert np.array_equal(w, np.full(4, 0.5))
    # kink in U_y along x: values 0 0 1 1 1, centred gradients d_x U_y = 0 0.5 0.5 0 0
    U = np.zeros((5, 3)); U[:, 1] = [0, 0, 1, 1, 1]
    g = np.zeros((5, 3, 3)); g[:, 0, 1] = [0, 0.5, 0.5, 0, 0]
    w = fv.filtered_linear2V_weights(m, np.ones(4), U, g, 0.2, 0.05)
    # face 1|2: gradfV = (0,1,0), df = 1, tcP = tcN = 2*1*0.5 = 1 -> df - tc = 0 -> limiter 1 -> linear
    assert w[1] == 0.5
    # strengthen the kink: gradients 0.2 -> tc = 0.4, df - tc = 0.6 -> limiter = 1.05 - 0.2*0.6/0.4 = 0.75
    g[:, 0, 1] = [0, 0.2, 0.2, 0, 0]
    w = fv.filtered_linear2V_weights(m, np.ones(4), U, g, 0.2, 0.05)
    lim = 1.05 - 0.2 * 0.6 / (0.4 + 1e-15)
    assert abs(w[1] - (lim * 0.5 + (1 - lim) * 1.0)) < 1e-15
    w2 = fv.filtered_linear2V_weights(m, -np.ones(4), U, g, 0.2, 0.05)
    assert abs(w2[1] - (lim * 0.5 + (1 - lim) * 0.0)) < 1e-15
    # the other components' differences enter the same limiter (one limiter per face for the vector)
    U[:, 2] = [0, 0, 3, 3, 3]
    w3 = fv.filtered_linear2V_weights(m, np.ones(4), U, g, 0.2, 0.05)
    df = 1 + 9; tc = 0.4
    lim3 = max(min(1.05 - 0.2 * (df - tc) / (tc + 1e-15), 1.0), 0.0)
    assert lim3 == 0.0 and w3[1] == 1.0                   # a large unresolved jump: fully upwind
