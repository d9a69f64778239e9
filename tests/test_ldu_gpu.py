"""GPU parity tests (through the C ABI) of the lduMatrix kernels, the level-scheduled
preconditioner sweeps and the solvers against the CPU oracle on the same seeded inputs.

Bars: the row kernels (Amul, Tmul, sumA, residual) and every sweep (DIC/DILU reciprocalD and
precondition, Gauss-Seidel) accumulate each row in the reference's face order with FMA
contraction off, so they must be BIT-EXACT.  Solvers contain dot products (tree sums on the
GPU, serial sums on the CPU), so residual histories agree to ~1e-12 relative; iteration counts
must be identical and converged fields agree to 1e-8 rel-L2 (north_star tolerance)."""
import numpy as np
import pytest

from common import laplacian_like, random_dag_mesh, rel_l2

pytestmark = pytest.mark.gpu


def meshes(O):
    out = {}
    out["hex_natural"] = O.hex_ldu(9, 7, 8)
    out["hex_big"] = O.hex_ldu(23, 37, 41)          # 3 x 3 tiles of 16 x 16 columns, entries up to 256 cells
    out["dag_random"] = random_dag_mesh(O, 8)
    out["chain"] = O.hex_ldu(300, 1, 1)
    out["plane"] = O.hex_ldu(1, 20, 31)
    return out


@pytest.fixture(scope="module", params=["hex_natural", "hex_levelmajor", "dag_random", "chain", "plane",
                                        "hex_levelmajor_t41", "hex_natural_t29", "plane_t17",
                                        "hex_natural_t37", "dag_random_t23", "chain_t64", "plane_t50", "hex_natural_t500", "hex_tiles_t0",
                                        "hex_natural_lv", "hex_levelmajor_lv", "chain_lv", "plane_lv", "hex_big", "hex_big_lv", "hex_baffled_t0"])
def case(request, O, ffm, ctx):
    """`_tNN`: tiled wavefront sweep on chunks of NN cells, so that the cross-workgroup hand-off (mailboxes, LDS ring
    wrap-around) is exercised on small meshes too (t0: a 2-D tile hint).  `_lv`: level-scheduled sweeps.  No suffix: the
    default (tiled sweeps on detected boxes and hinted meshes, level-scheduled otherwise)."""
    import os
    name = request.param
    grp = None
    if name.endswith("_lv"):
        name = name[:-3]
        os.environ["FFM_SWEEP"] = "levels"
    elif "_t" in name:                      # tiled wavefront sweep (ffm_tile.hip); t0 = 2-D tile hint instead of chunks
        name, grp = name.rsplit("_t", 1)
        if grp != "0":
            os.environ["FFM_PIPE_GROUP_CELLS"] = grp
        os.environ["FFM_SWEEP"] = "tile"
    try:
        yield from _make_case(name, grp, O, ffm, ctx)
    finally:
        os.environ.pop("FFM_PIPE_GROUP_CELLS", None)
        os.environ.pop("FFM_SWEEP", None)


def _make_case(name, grp, O, ffm, ctx):
    if name == "hex_levelmajor":
        N, l, u = O.hex_ldu(9, 7, 8)
        cOrd, fOrd = ffm.renumber_levels(N, l, u)
        l, u, _ = ffm.hexmesh.apply_renumbering(N, l, u, cOrd, fOrd)
    elif name in ("hex_tiles", "hex_baffled"):
        nx, ny, nz = (9, 7, 8) if name == "hex_tiles" else (11, 13, 12)
        N, l, u = O.hex_ldu(nx, ny, nz)
        if name == "hex_baffled":       # a fifth of the faces removed (baffles): inside a tile the backward dependency order is
            keep = O.hash_u(0xBAF, np.arange(len(l))) > 0.2          # no longer the mirror image of the forward order
            l, u = l[keep], u[keep]
        c = np.arange(N)
        hint = ((c // nx) % ny) // 3 + 10 * ((c // (nx * ny)) // 3)      # 3x3 tiles of cell columns (j,k)
        A = ffm.lduMatrix(ctx, N, l, u, groupHint=hint)
        assert A.sweep_mode == 2
        yield name, N, l, u, A
        A.close()
        return
    else:
        N, l, u = meshes(O)[name]
    A = ffm.lduMatrix(ctx, N, l, u)
    if grp is None and name != "hex_big":
        assert A.native_order == (name in ("hex_levelmajor", "chain"))
    yield name, N, l, u, A
    A.close()


def _both(O, ctx, case, asym):
    name, N, l, u, A = case
    diag, up, lo = laplacian_like(O, N, l, u, seed=3, asym=asym, shift=0.05)
    A.set_coeffs(diag, up, lo)
    Ao = O.Ldu(N, l, u).set_coeffs(diag, up, lo)
    return N, A, Ao


@pytest.mark.parametrize("asym", [0.0, 0.35])
def test_row_kernels_bit_exact(O, ctx, case, asym):
    N, A, Ao = _both(O, ctx, case, asym)
    x = 2 * O.hash_u(0xF4, np.arange(N)) - 0.7
    b = O.hash_u(0xF3, np.arange(N))
    xd, bd = ctx.to_device(x), ctx.to_device(b)
    assert np.array_equal(A.Amul(xd).cpu().numpy(), Ao.amul(x))
    assert np.array_equal(A.Tmul(xd).cpu().numpy(), Ao.tmul(x))
    assert np.array_equal(A.sumA().cpu().numpy(), Ao.sumA())
    assert np.array_equal(A.residual(xd, bd).cpu().numpy(), Ao.residual(x, b))


def test_dic_bit_exact(O, ctx, case):
    N, A, Ao = _both(O, ctx, case, 0.0)
    r = 2 * O.hash_u(11, np.arange(N)) - 1
    rD = Ao.dic_rD()
    assert np.array_equal(A.reciprocalD("DIC").cpu().numpy(), rD)
    assert np.array_equal(A.precondition("DIC", ctx.to_device(r)).cpu().numpy(), Ao.dic_precondition(rD, r))


def test_sweep_ticket_counter_cannot_wrap(O, ctx, case):
    """the 32-bit group ticket of the tiled sweeps is zeroed on the stream before every sweep, so a counter left
    close to 2^32 (what ~7 million launches would have reached) changes nothing: DIC stays bitwise the serial face loops."""
    N, A, Ao = _both(O, ctx, case, 0.0)
    if A.sweep_mode != 2:
        pytest.skip("level-scheduled sweeps have no ticket")
    r = 2 * O.hash_u(21, np.arange(N)) - 1
    rD = Ao.dic_rD()
    for preset in (2 ** 32 - 1, 2 ** 32 - 3, 2 ** 31 + 5):
        A.debug_set_sweep_ticket(preset)
        assert np.array_equal(A.reciprocalD("DIC").cpu().numpy(), rD)
        A.debug_set_sweep_ticket(preset)
        assert np.array_equal(A.precondition("DIC", ctx.to_device(r)).cpu().numpy(), Ao.dic_precondition(rD, r))
    A.debug_set_sweep_ticket(2 ** 32 - 2)
    psi0 = O.hash_u(13, np.arange(N)); b = O.hash_u(14, np.arange(N))
    got = A.smooth(ctx.to_device(psi0), ctx.to_device(b), nSweeps=2, smoother="symGaussSeidel")
    assert np.array_equal(got.cpu().numpy(), Ao.gs_smooth(psi0, b, nSweeps=2, sym=True))


def test_dilu_bit_exact(O, ctx, case):
    N, A, Ao = _both(O, ctx, case, 0.35)
    r = 2 * O.hash_u(12, np.arange(N)) - 1
    rD = Ao.dilu_rD()
    assert np.array_equal(A.reciprocalD("DILU").cpu().numpy(), rD)
    rd = ctx.to_device(r)
    assert np.array_equal(A.precondition("DILU", rd).cpu().numpy(), Ao.dilu_precondition(rD, r))
    assert np.array_equal(A.precondition("DILU", rd, transpose=True).cpu().numpy(), Ao.dilu_precondition(rD, r, transpose=True))


@pytest.mark.parametrize("sym", [True, False])
def test_gauss_seidel_bit_exact(O, ctx, case, sym):
    N, A, Ao = _both(O, ctx, case, 0.35)
    psi0 = O.hash_u(13, np.arange(N)); b = O.hash_u(14, np.arange(N))
    got = A.smooth(ctx.to_device(psi0), ctx.to_device(b), nSweeps=2, smoother="symGaussSeidel" if sym else "GaussSeidel")
    assert np.array_equal(got.cpu().numpy(), Ao.gs_smooth(psi0, b, nSweeps=2, sym=sym))


SOLVES = [("PCG", "DIC", 0.0), ("PCG", "none", 0.0), ("PCG", "diagonal", 0.0), ("PBiCGStab", "DILU", 0.35),
          ("PBiCGStab", "DILU", 0.0), ("PBiCG", "DILU", 0.35), ("smoothSolver", "symGaussSeidel", 0.35),
          ("smoothSolver", "GaussSeidel", 0.35), ("PBiCGStab", "none", 0.35)]
OR = {"PCG": "PCG", "PBiCGStab": "PBICGSTAB", "PBiCG": "PBICG", "smoothSolver": "SMOOTH", "diagonal": "DIAGONAL"}
OP = {"DIC": "DIC", "DILU": "DILU", "none": "NONE", "diagonal": "DIAGONALP", "symGaussSeidel": "SYMGS", "GaussSeidel": "GS"}


@pytest.mark.parametrize("solver,precond,asym", SOLVES)
def test_solver_parity(O, ctx, case, solver, precond, asym):
    N, A, Ao = _both(O, ctx, case, asym)
    b = 2 * O.hash_u(0xF3, np.arange(N)) - 1
    kw = dict(tolerance=1e-11, relTol=0.0, maxIter=3000)
    ref, pr = Ao.solve(getattr(O, OR[solver]), getattr(O, OP[precond]), np.zeros(N), b, **kw)
    psi = ctx.zeros(N)
    pg = A.solve(psi, ctx.to_device(b), solver=solver, preconditioner=precond, smoother=precond, **kw)
    assert pg["converged"] == 1 and pr["converged"] == 1
    if precond == "none":
        # un-preconditioned Krylov on these matrices takes O(100) iterations and is sensitive to the
        # rounding of the dot products; the count may differ by a few iterations
        assert abs(pg["nIterations"] - pr["nIterations"]) <= max(3, pr["nIterations"] // 8)
    else:
        assert pg["nIterations"] == pr["nIterations"]
    assert abs(pg["initialResidual"] - pr["initialResidual"]) <= 1e-12 * pr["initialResidual"]
    # residual histories drift apart by rounding only (tree-sum vs serial-sum dot products)
    if pg["nIterations"] == pr["nIterations"] and precond != "none":
        # (normalised residuals of a few 1e-12 are at the rounding floor of the larger meshes: absolute slack 2e-12)
        assert abs(pg["finalResidual"] - pr["finalResidual"]) <= 0.05 * pr["finalResidual"] + 2e-12
    assert rel_l2(psi.cpu().numpy(), ref) < 1e-8        # north_star: fields within 1e-8 rel-L2


def test_fixed_iteration_history_matches(O, ctx, case):
    """minIter == maxIter removes the stopping decision: after k iterations the GPU and CPU iterates
    agree to rounding (SURVEY 7 'hard parts': compare at a fixed iteration count)."""
    N, A, Ao = _both(O, ctx, case, 0.0)
    b = 2 * O.hash_u(0xF3, np.arange(N)) - 1
    # on the 1-D chain DIC is the exact Cholesky factor: iterations after the first act on rounding noise
    for k in ((1,) if case[0] == "chain" else (1, 5)):
        ref, pr = Ao.solve(O.PCG, O.DIC, np.zeros(N), b, tolerance=0.0, minIter=k, maxIter=k)
        psi = ctx.zeros(N)
        pg = A.solve(psi, ctx.to_device(b), tolerance=0.0, minIter=k, maxIter=k)
        assert pg["nIterations"] == pr["nIterations"] == k
        assert rel_l2(psi.cpu().numpy(), ref) < 1e-12
        # (absolute slack: on the chain the residual after one exact-Cholesky iteration is pure rounding noise)
        assert abs(pg["finalResidual"] - pr["finalResidual"]) <= 1e-10 * pr["finalResidual"] + 1e-13


def test_solver_edge_cases(O, ctx, ffm):
    # diagonal matrix -> diagonalSolver; converged initial guess -> 0 iterations; host-pointer entry
    N, l, u = O.hex_ldu(4, 4, 4)
    diag = 1.0 + O.hash_u(1, np.arange(N)); b = O.hash_u(2, np.arange(N))
    A = ffm.lduMatrix(ctx, N, l, u).set_coeffs(diag, np.zeros(len(l)))
    psi = ctx.zeros(N)
    p = A.solve(psi, ctx.to_device(b), solver="diagonal")
    assert np.array_equal(psi.cpu().numpy(), b / diag) and p["nIterations"] == 0 and p["finalResidual"] == 0.0
    p2 = A.solve(psi, ctx.to_device(b), solver="PCG", preconditioner="DIC")
    assert p2["nIterations"] == 0 and np.array_equal(psi.cpu().numpy(), b / diag)
    psiH, p3 = A.solve_host(np.zeros(N), b, solver="PCG", preconditioner="DIC", tolerance=1e-12)
    assert p3["converged"] == 1 and rel_l2(psiH, b / diag) < 1e-12
    # PCG on an asymmetric matrix and DIC on an asymmetric matrix are refused, loudly
    A.set_coeffs(diag, np.full(len(l), -0.01), np.full(len(l), -0.02))
    with pytest.raises(ffm.FfmError):
        A.solve(psi, ctx.to_device(b), solver="PCG", preconditioner="DIC")
    A.close()
    # single cell, no faces
    A1 = ffm.lduMatrix(ctx, 1, np.zeros(0, np.int32), np.zeros(0, np.int32)).set_coeffs(np.array([2.0]), np.zeros(0))
    x1 = ctx.zeros(1)
    p4 = A1.solve(x1, ctx.to_device(np.array([4.0])), solver="PCG", preconditioner="DIC", tolerance=1e-12)
    assert abs(float(x1[0]) - 2.0) < 1e-15 and p4["nIterations"] <= 1
    A1.close()


def test_reductions(O, ctx):
    n = 1_000_003
    x = 2 * O.hash_u(21, np.arange(n)) - 1; y = O.hash_u(22, np.arange(n))
    xd, yd = ctx.to_device(x), ctx.to_device(y)
    assert abs(ctx.gSum(xd) - x.sum()) < 1e-9
    assert abs(ctx.gSumProd(xd, yd) - np.dot(x, y)) < 1e-9
    assert abs(ctx.gSumMag(xd) - np.abs(x).sum()) < 1e-9
    assert ctx.gMin(xd) == x.min() and ctx.gMax(xd) == x.max()
    assert ctx.gSum(xd) == ctx.gSum(xd)      # deterministic (no atomics)


@pytest.mark.parametrize("smallLimit,flow", [("0", "0"), ("1000000", None), ("0", None), ("1000000", "all"), (None, None), (None, "0")])
def test_one_launch_per_level_and_single_workgroup_sweeps_agree(O, ffm, ctx, monkeypatch, smallLimit, flow):
    """Level-scheduled sweeps have three forms: one launch per dependency level (FFM_FLOW_SWEEP=0; the round-1 path); for matrices of
    at most FFM_SMALL_SWEEP_CELLS cells ONE workgroup that walks all levels with a barrier per level (csrc/ffm_solve.hip:
    k_small_sweep); and -- the default for every level-scheduled matrix: unstructured meshes, GAMG's coarse levels -- ONE launch in
    which every cell waits for the values it needs (k_flow_sweep: chunks of 256 cells by ticket, values published in sentinel-filled
    arrays).  All must be the
    serial face loops bit for bit: DIC and DILU (calcReciprocalD, precondition, transposed), GaussSeidel and symGaussSeidel, on a
    randomly relabelled DAG mesh (non-contiguous backward levels), a hex box, and larger ones of both (hundreds of chunks)."""
    if smallLimit is None:                      # the defaults: dataflow sweeps for every size; with them off, single-workgroup sweeps up to 131072 cells
        monkeypatch.delenv("FFM_SMALL_SWEEP_CELLS", raising=False)
    else:
        monkeypatch.setenv("FFM_SMALL_SWEEP_CELLS", smallLimit)
    if flow is None:
        monkeypatch.delenv("FFM_FLOW_SWEEP", raising=False)
    else:
        monkeypatch.setenv("FFM_FLOW_SWEEP", flow)
    monkeypatch.setenv("FFM_SWEEP", "levels")
    H = ffm.hexmesh
    for name in ("dag", "hex", "dag-large", "hex-large"):
        if name == "dag":
            N, l, u = random_dag_mesh(O, 14, seed=5)
        elif name == "dag-large":
            N, l, u = random_dag_mesh(O, 36, seed=9)
        elif name == "hex-large":
            blk = H.HexBlock((47, 41, 33)); N, l, u = blk.nCells, blk.l, blk.u
        else:
            blk = H.HexBlock((13, 11, 9)); N, l, u = blk.nCells, blk.l, blk.u
        A = ffm.lduMatrix(ctx, N, l, u)
        assert A.sweep_mode == 0
        for asym in (0.0, 0.35):
            diag, up, lo = laplacian_like(O, N, l, u, seed=3, asym=asym, shift=0.05)
            A.set_coeffs(diag, up, lo)
            Ao = O.Ldu(N, l, u).set_coeffs(diag, up, lo)
            r = 2 * O.hash_u(12, np.arange(N)) - 1
            rd = ctx.to_device(r)
            if asym == 0.0:
                rD = Ao.dic_rD()
                assert np.array_equal(A.reciprocalD("DIC").cpu().numpy(), rD)
                assert np.array_equal(A.precondition("DIC", rd).cpu().numpy(), Ao.dic_precondition(rD, r))
            rD = Ao.dilu_rD()
            assert np.array_equal(A.reciprocalD("DILU").cpu().numpy(), rD)
            assert np.array_equal(A.precondition("DILU", rd).cpu().numpy(), Ao.dilu_precondition(rD, r))
            assert np.array_equal(A.precondition("DILU", rd, transpose=True).cpu().numpy(), Ao.dilu_precondition(rD, r, transpose=True))
            psi0 = O.hash_u(13, np.arange(N)); b = O.hash_u(14, np.arange(N))
            for sym in (True, False):
                got = A.smooth(ctx.to_device(psi0), ctx.to_device(b), nSweeps=2, smoother="symGaussSeidel" if sym else "GaussSeidel")
                assert np.array_equal(got.cpu().numpy(), Ao.gs_smooth(psi0, b, nSweeps=2, sym=sym))
        A.close()
