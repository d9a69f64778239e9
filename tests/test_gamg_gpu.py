"""GAMG on the GPU (csrc/ffm_gamg.hip) against the oracle restatement (oracle/gamg.py): the same agglomeration (every level's
addressing), the same coarse coefficients (bitwise: identical summation order), and the same solve -- V-cycle by V-cycle the
same residuals and iteration count -- for the two selections the reference makes (GaussSeidel on a symmetric p_rgh matrix,
DILU on an asymmetric ray-transport-like matrix) plus the DIC / symGaussSeidel smoothers.  PARITY UNPINNED by reference data
(see the oracle's header)."""
import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


def _box(ffm, n, h=0.05):
    H = ffm.hexmesh
    blk = H.HexBlock(n)
    s = H.synth_p_rgh(blk, h=h)
    d = blk.u.astype(np.int64) - blk.l
    axis = np.where(d == 1, 0, np.where(d == n[0], 1, 2))
    Sf = np.zeros((len(blk.l), 3))
    Sf[np.arange(len(blk.l)), axis] = h * h * np.array([1.0, 1.3, 0.7])[axis]
    return blk, s, Sf


@pytest.mark.parametrize("n", [(14, 12, 10), (40, 36, 30)])
def test_agglomeration_and_coarse_matrices_equal_the_oracle(O, ffm, ctx, n):
    from oracle import gamg
    blk, s, Sf = _box(ffm, n)
    A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    G = ffm.GAMG(ctx, A, blk.l, blk.u, Sf=Sf)
    assert np.array_equal(G.weights, gamg.face_area_pair_weights(Sf))
    agg = gamg.Agglomeration(blk.nCells, blk.l, blk.u, G.weights)
    assert G.nLevels == agg.nLevels
    for lev in range(agg.nLevels + 1):
        l, u = G.level_addressing(lev)
        assert np.array_equal(l, agg.l[lev]) and np.array_equal(u, agg.u[lev]), lev
    lo = s["upper"] * (1.0 + 0.3 * (ffm.hexmesh.hash_u(0xA1, blk.gface) - 0.5))
    for lower in (None, lo):
        G.set_matrix(ctx.to_device(s["diag"]), ctx.to_device(s["upper"]), None if lower is None else ctx.to_device(lower))
        ref = gamg.GAMGSolver(agg, s["diag"], s["upper"], lower)
        for lev in range(agg.nLevels + 1):
            d, up, low = G.level_coeffs(lev)
            rd, ru, rl = ref.coef[lev]
            assert np.array_equal(d, rd) and np.array_equal(up, ru), lev
            assert np.array_equal(low, ru if rl is None else rl), lev
    G.close(); A.close()


@pytest.mark.parametrize("smoother,asym", [("GaussSeidel", False), ("DILU", True), ("DIC", False), ("symGaussSeidel", False), ("GaussSeidel", True)])
def test_gamg_solve_equals_the_oracle(O, ffm, ctx, smoother, asym):
    from oracle import gamg
    blk, s, Sf = _box(ffm, (20, 16, 12))
    lo = s["upper"] * (1.0 + 0.3 * (ffm.hexmesh.hash_u(0xA1, blk.gface) - 0.5)) if asym else None
    A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    G = ffm.GAMG(ctx, A, blk.l, blk.u, Sf=Sf)
    G.set_matrix(ctx.to_device(s["diag"]), ctx.to_device(s["upper"]), None if lo is None else ctx.to_device(lo))
    agg = gamg.Agglomeration(blk.nCells, blk.l, blk.u, G.weights)
    ref = gamg.GAMGSolver(agg, s["diag"], s["upper"], lo, smoother=smoother)
    src = ctx.to_device(s["source"])
    for kw in (dict(tolerance=1e-6, relTol=0.0), dict(tolerance=1e-5, relTol=0.01), dict(tolerance=1e-30, maxIter=3)):
        psi = ctx.to_device(np.zeros(blk.nCells))
        pf = G.solve(psi, src, smoother=smoother, **kw)
        xr, pr = ref.solve(np.zeros(blk.nCells), s["source"], **kw)
        assert pf["nIterations"] == pr["nIterations"], (kw, pf, pr)
        assert abs(pf["initialResidual"] - pr["initialResidual"]) <= 1e-12 * pr["initialResidual"]
        assert abs(pf["finalResidual"] - pr["finalResidual"]) <= 1e-6 * pr["finalResidual"], (pf, pr)
        assert rel_l2(psi.cpu().numpy(), xr) < 1e-10
        cs = G.coarsest_solves()
        # (tolerance 1e-30 drives the coarsest Krylov solve into rounding: its stagnation point may move by an iteration)
        slack = 1 if kw["tolerance"] < 1e-20 else 0
        assert len(cs) == len(ref.coarsest_log)
        assert all(abs(a["nIterations"] - b["nIterations"]) <= slack for a, b in zip(cs, ref.coarsest_log)), (cs, ref.coarsest_log)
    G.close(); A.close()


def test_gamg_on_a_large_box_in_tile_mode(O, ffm, ctx):
    """128^3 (2.1 M cells, the finest level smoothed by the tiled Gauss-Seidel kernels, 17 coarse levels by the level-scheduled
    ones): converges to the PCG solution; the V-cycle count is that of a textbook multigrid (far below PCG's)."""
    blk, s, Sf = _box(ffm, (128, 128, 128))
    A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    assert A.sweep_mode == 2
    G = ffm.GAMG(ctx, A, blk.l, blk.u, Sf=Sf)
    assert G.nLevels >= 15 and G.level_size(G.nLevels)[0] >= 10
    G.set_matrix(ctx.to_device(s["diag"]), ctx.to_device(s["upper"]))
    src = ctx.to_device(s["source"])
    psi = ctx.to_device(np.zeros(blk.nCells))
    pf = G.solve(psi, src, smoother="GaussSeidel", tolerance=1e-8)
    assert pf["converged"] and pf["nIterations"] <= 25, pf
    B = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"])
    ref = ctx.to_device(np.zeros(blk.nCells))
    pk = B.solve(ref, src, "PCG", "DIC", tolerance=1e-11)
    assert pk["nIterations"] > 3 * pf["nIterations"]
    assert rel_l2(psi.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    G.close(); A.close(); B.close()


@pytest.mark.parametrize("smoother", ["GaussSeidel", "DIC"])
def test_solver_GAMG_through_the_foam_layer(O, ffm, ctx, smoother):
    """fvMatrix::solve() with {solver GAMG; smoother ...; agglomerator faceAreaPair} (cases/wallFireSpread2D/system/fvSolution:36-60)
    on a p_rgh-shaped equation written against include/ffmFoam.H (examples/b1_demo.C b1_gamg_solve): the matrix reaches the
    multigrid in the library's native coefficient layout; iteration count, residuals and the field against the oracle's
    assembly (oracle/fv.py) + GAMG (oracle/gamg.py) on the same numbering."""
    import ctypes as C
    import os
    from oracle import fv, plume, gamg
    m = plume.make_mesh((16, 14, 12), h=0.1)
    N, F = m.nCells, m.nFaces
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    G = ffm.GAMG(ctx, A, l2, u2, Sf=m.Sf[fOrd])
    hu = lambda seed, n: O.hash_u(seed, np.arange(n))
    dt = 1e-3
    psi = 1.17e-5 * (0.9 + 0.2 * hu(1, N)); gam = 1e-3 * (0.5 + hu(2, N)); p0 = 10.0 * (hu(3, N) - 0.5); S = 2.0 * hu(4, N) - 1.0
    bc = fv.MixedBC(m, f=[np.full(p.size, 1.0 if p.name == "top" else 0.0) for p in m.patches],
                    ref=[2.0 * hu(40 + q, p.size) - 1.0 for q, p in enumerate(m.patches)])
    # the oracle's evaluation
    gb = [gam[p.faceCells] for p in m.patches]
    gf, gfb = fv.interpolate(m, gam, gb)
    E = fv.fvm_ddt(m, 1.0 / dt, psi, psi, p0)
    E -= fv.fvm_laplacian(m, gf, gfb, [bc])
    E.add_su(S)
    d, s = E.solve_system()
    agg = gamg.Agglomeration(N, l2, u2, gamg.face_area_pair_weights(m.Sf[fOrd]))
    ref = gamg.GAMGSolver(agg, d[cOrd], E.upper[fOrd], None, smoother=smoother)
    xr, pr = ref.solve(p0[cOrd], s[cOrd], tolerance=1e-8, relTol=0.0)
    # the Foam layer
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.__file__), "lib", "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    h = lambda a: np.ascontiguousarray(a, np.float64)
    keep = [h(psi[cOrd]), h(gam[cOrd]), h(p0[cOrd]), h(S[cOrd])] + [h(np.concatenate(x)) for x in (bc.f, bc.ref, bc.refGrad)]
    P = lambda a: a.ctypes.data_as(dp)
    bcp = (dp * 3)(P(keep[4]), P(keep[5]), P(keep[6]))
    out = np.empty(N); res = np.empty(2)
    lib.b1_gamg_solve.restype = C.c_int
    lib.b1_gamg_solve.argtypes = [C.c_void_p] * 4 + [C.c_double, C.c_int, C.c_double, C.c_double] + [dp] * 3 + [C.POINTER(dp)] + [dp] * 3
    sm = {"GaussSeidel": 3, "DIC": 1}[smoother]
    nit = lib.b1_gamg_solve(ctx.h, A.h, mesh.h, G.h, dt, sm, 1e-8, 0.0, P(keep[0]), P(keep[1]), P(keep[2]), bcp, P(keep[3]), P(out), P(res))
    assert nit == pr["nIterations"] and nit >= 2, (nit, pr)
    assert abs(res[0] - pr["initialResidual"]) <= 1e-10 * pr["initialResidual"]
    assert abs(res[1] - pr["finalResidual"]) <= 1e-5 * pr["finalResidual"]
    assert rel_l2(out, xr) < 1e-10
    G.close(); A.close()


@pytest.mark.parametrize("meshName,asym,smoother", [("steckler", 0.0, "GaussSeidel"), ("dag_random", 0.0, "DIC"), ("dag_random", 0.3, "DILU")])
def test_gamg_on_unstructured_addressing(O, ffm, ctx, meshName, asym, smoother):
    """The baffled steckler room (cases/steckler geometry: cells with fewer than six neighbours around the baffles) and a randomly
    relabelled box (an LDU graph without any structure in its numbering) with arbitrary face weights: agglomeration, every coarse
    matrix and the solve against the oracle, as on the hexahedral box."""
    import part_cases
    from oracle import gamg
    N, l, u, centres, diag, up, lo, source = part_cases.build(O, meshName, asym)
    w = 0.5 + O.hash_u(0x5F, np.arange(len(l)))
    A = ffm.lduMatrix(ctx, N, l, u)
    G = ffm.GAMG(ctx, A, l, u, weights=w)
    agg = gamg.Agglomeration(N, l, u, w)
    assert G.nLevels == agg.nLevels and agg.nLevels >= 3
    for lev in range(agg.nLevels + 1):
        gl, gu = G.level_addressing(lev)
        assert np.array_equal(gl, agg.l[lev]) and np.array_equal(gu, agg.u[lev]), lev
    G.set_matrix(ctx.to_device(diag), ctx.to_device(up), None if lo is None else ctx.to_device(lo))
    ref = gamg.GAMGSolver(agg, diag, up, lo, smoother=smoother)
    for lev in range(agg.nLevels + 1):
        d, uu, ll = G.level_coeffs(lev)
        assert np.array_equal(d, ref.coef[lev][0]) and np.array_equal(uu, ref.coef[lev][1]), lev
    psi = ctx.to_device(np.zeros(N))
    pf = G.solve(psi, ctx.to_device(source), smoother=smoother, tolerance=1e-8)
    xr, pr = ref.solve(np.zeros(N), source, tolerance=1e-8)
    assert pf["nIterations"] == pr["nIterations"] and pf["converged"], (pf, pr)
    assert rel_l2(psi.cpu().numpy(), xr) < 1e-9
    G.close(); A.close()


def test_the_agglomeration_direction_carries_over_to_the_next_mesh_of_a_run(O, ffm, ctx):
    """pairGAMGAgglomeration::forward_ is a static upstream: it is toggled by every pair agglomeration of the run, so the hierarchy of a
    second GAMG mesh / region depends on how many levels the first one had.  The context keeps it: a second ffm.GAMG with forward=None
    continues in the direction the first ended with and equals the oracle's Agglomeration(forward=first.forward_after); started afresh it
    equals the forward=True one -- and the two differ."""
    from oracle import gamg
    blk1, s1, Sf1 = _box(ffm, (9, 7, 6)); blk2, s2, Sf2 = _box(ffm, (14, 12, 10))
    A1 = ffm.lduMatrix(ctx, blk1.nCells, blk1.l, blk1.u); A2 = ffm.lduMatrix(ctx, blk2.nCells, blk2.l, blk2.u)
    G1 = ffm.GAMG(ctx, A1, blk1.l, blk1.u, Sf=Sf1)                         # the first agglomeration of the "run"
    agg1 = gamg.Agglomeration(blk1.nCells, blk1.l, blk1.u, G1.weights)
    G2 = ffm.GAMG(ctx, A2, blk2.l, blk2.u, Sf=Sf2, forward=None)           # continues
    w2 = gamg.face_area_pair_weights(Sf2)
    cont = gamg.Agglomeration(blk2.nCells, blk2.l, blk2.u, w2, forward=agg1.forward_after)
    fresh = gamg.Agglomeration(blk2.nCells, blk2.l, blk2.u, w2)
    want = cont if agg1.forward_after is False else fresh
    assert G2.nLevels == want.nLevels
    for lev in range(want.nLevels + 1):
        l, u = G2.level_addressing(lev)
        assert np.array_equal(l, want.l[lev]) and np.array_equal(u, want.u[lev]), lev
    if agg1.forward_after is False:            # an odd number of levels in the first hierarchy: the second one is built the other way round
        assert any(not np.array_equal(cont.l[lev], fresh.l[lev]) or not np.array_equal(cont.u[lev], fresh.u[lev]) for lev in range(1, min(cont.nLevels, fresh.nLevels) + 1))
    G3 = ffm.GAMG(ctx, A2, blk2.l, blk2.u, Sf=Sf2)                         # started afresh
    l, u = G3.level_addressing(1)
    assert np.array_equal(l, fresh.l[1]) and np.array_equal(u, fresh.u[1])
    G1.close(); G2.close(); G3.close(); A1.close(); A2.close()


def test_a_mesh_below_nCellsInCoarsestLevel_is_solved_on_the_fine_matrix(O, ffm, ctx):
    """the first agglomeration already falls below nCellsInCoarsestLevel (a small region): no coarse level; the solve then is the
    coarsest-level solver (PCG + DIC, PBiCGStab + DILU) on the fine matrix, with its iteration count"""
    blk, s, Sf = _box(ffm, (5, 4, 3))
    A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    G = ffm.GAMG(ctx, A, blk.l, blk.u, Sf=Sf, nCellsInCoarsestLevel=50)
    assert G.nLevels == 0
    for lower in (None, s["upper"] * (1.0 + 0.3 * (ffm.hexmesh.hash_u(0xA1, blk.gface) - 0.5))):
        G.set_matrix(ctx.to_device(s["diag"]), ctx.to_device(s["upper"]), None if lower is None else ctx.to_device(lower))
        psi = ctx.zeros(blk.nCells)
        pf = G.solve(psi, ctx.to_device(s["source"]), smoother="GaussSeidel" if lower is None else "DILU", tolerance=1e-9)
        Ao = O.Ldu(blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"], lower)
        ref, pr = Ao.solve(O.PCG if lower is None else O.PBICGSTAB, O.DIC if lower is None else O.DILU, np.zeros(blk.nCells), s["source"], tolerance=1e-9)
        assert pf["nIterations"] == pr["nIterations"] and pf["converged"]
        assert rel_l2(psi.cpu().numpy(), ref) < 1e-10
    G.close(); A.close()
