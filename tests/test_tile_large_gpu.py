"""The tiled kernels (ffm_tile.hip) under real concurrency: on a 160^3 box (100 tiles, 478 dependency levels, every tile
waiting on its neighbours' mailboxes) DIC's reciprocal diagonal, the preconditioner application, DILU on an asymmetric
matrix and Amul must be bitwise equal to the level-scheduled / row kernels, which share no code path with them beyond the
arithmetic.  Repeated applications check that stale mailbox contents of an earlier sweep are never read."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(ffm, ctx, n, mode, asym):
    """Both modes get the blockMesh numbering and renumber inside the library, which keeps the caller's face order in every
    row sum: the two are then bitwise comparable (two ffm_renumber_levels numberings are not -- the public renumbering sorts
    a cell's faces by the NEW neighbour labels, and the tile numbering's edge-first order within a level differs from the
    level numbering's)."""
    H = ffm.hexmesh
    os.environ["FFM_SWEEP"] = mode
    try:
        blk = H.HexBlock((n, n, n))
        s = H.synth_p_rgh(blk)
        A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    finally:
        os.environ.pop("FFM_SWEEP", None)
    assert A.sweep_mode == (2 if mode == "tile" else 0)
    up = s["upper"]
    lo = up * (1.0 + 0.3 * (H.hash_u(0xA1, blk.gface) - 0.5)) if asym else None
    A.set_coeffs(s["diag"], up, lo)
    return blk, s, np.arange(blk.nCells), A


@pytest.mark.parametrize("asym", [False, True])
def test_tiled_kernels_equal_level_kernels_at_scale(ffm, ctx, asym):
    n = 160
    out = {}
    for mode in ("tile", "levels"):
        blk, s, cOrd, A = _build(ffm, ctx, n, mode, asym)
        pre = "DILU" if asym else "DIC"
        N = blk.nCells
        nat = lambda t: (lambda o: (o.__setitem__(cOrd, t.cpu().numpy()), o)[1])(np.empty(N))
        rD = nat(A.reciprocalD(pre))
        r = ctx.to_device(s["source"][cOrd])
        w1 = nat(A.precondition(pre, r))
        r2 = ctx.to_device((s["source"] * (1.0 + s["x"]))[cOrd])
        w2 = nat(A.precondition(pre, r2))                  # second application: mailboxes are re-armed every sweep
        w3 = nat(A.precondition(pre, r))
        y = nat(A.Amul(ctx.to_device(s["x"][cOrd])))
        out[mode] = (rD, w1, w2, w3, y)
        A.close()
    for idx, (a, b) in enumerate(zip(out["tile"], out["levels"])):
        bad = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
        assert len(bad) == 0, (idx, len(bad), bad[:5], a[bad[:5]], b[bad[:5]], int(np.isnan(a).sum()), int(np.isnan(b).sum()))
    assert np.array_equal(out["tile"][1], out["tile"][3])


def test_tiled_sweeps_on_a_baffled_box_equal_the_serial_loops(O, ffm, ctx):
    """96^3 box with a fifth of the faces removed: the backward sweep takes its position-space form (the backward order inside
    a tile is not the mirror image of the forward order); 36 tiles, mailboxes across all of them.  Compared bitwise with the
    oracle's serial face loops ON THE SAME NUMBERING (with baffles the relative order of a cell's neighbours, hence the order
    of its sums, depends on the numbering, so the level-scheduled kernels on their own numbering are no bitwise reference)."""
    H = ffm.hexmesh
    n = 96
    blk = H.HexBlock((n, n, n))
    keep = H.hash_u(0xBAF1, blk.gface) > 0.2
    l, u = blk.l[keep], blk.u[keep]
    N = blk.nCells
    up = -(1e-3 * (0.5 + H.hash_u(0xF1, blk.gface[keep]))) * 0.05
    lo = up * (1.0 + 0.3 * (H.hash_u(0xA1, blk.gface[keep]) - 0.5))
    diag = np.zeros(N); np.add.at(diag, l, -up); np.add.at(diag, u, -lo); diag += 1e-6
    r0 = 2.0 * H.hash_u(0xF3, np.arange(N)) - 1.0
    hint = (blk.j // 16 + 1000 * (blk.k // 16)).astype(np.int32)
    cOrd, fOrd = ffm.renumber_levels(N, l, u, groupHint=hint)
    l2, u2, _ = H.apply_renumbering(N, l, u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2, groupHint=hint[cOrd])
    assert A.native_order and A.sweep_mode == 2
    Ao = O.Ldu(N, l2, u2)
    r = r0[cOrd]
    rd = ctx.to_device(r)
    # DIC
    A.set_coeffs(diag[cOrd], up[fOrd]); Ao.set_coeffs(diag[cOrd], up[fOrd], None)
    rD = Ao.dic_rD()
    assert np.array_equal(A.reciprocalD("DIC").cpu().numpy(), rD)
    ref = Ao.dic_precondition(rD, r)
    assert np.array_equal(A.precondition("DIC", rd).cpu().numpy(), ref)
    assert np.array_equal(A.precondition("DIC", rd).cpu().numpy(), ref)         # mailboxes re-armed
    x = H.hash_u(0xF4, np.arange(N))
    assert np.array_equal(A.Amul(ctx.to_device(x)).cpu().numpy(), Ao.amul(x))   # tiled Amul, cell-space upper coefficients
    # DILU and its transpose
    A.set_coeffs(diag[cOrd], up[fOrd], lo[fOrd]); Ao.set_coeffs(diag[cOrd], up[fOrd], lo[fOrd])
    rD = Ao.dilu_rD()
    assert np.array_equal(A.reciprocalD("DILU").cpu().numpy(), rD)
    assert np.array_equal(A.precondition("DILU", rd).cpu().numpy(), Ao.dilu_precondition(rD, r))
    assert np.array_equal(A.precondition("DILU", rd, transpose=True).cpu().numpy(), Ao.dilu_precondition(rD, r, transpose=True))
    # symGaussSeidel on the asymmetric matrix: tiled forward sweep, position-space reverse sweep
    psi0 = H.hash_u(13, np.arange(N))
    # the baffled synthetic matrix is not diagonally dominant enough for Gauss-Seidel to contract; bitwise equality is the point
    got = A.smooth(ctx.to_device(psi0), rd, nSweeps=2, smoother="symGaussSeidel")
    assert np.array_equal(got.cpu().numpy(), Ao.gs_smooth(psi0, r, nSweeps=2, sym=True))
    A.close()


def test_pcg_with_the_vector_updates_fused_into_the_sweeps(O, ffm, ctx):
    """PCG + DIC in tile mode carries rA -= alpha wA, sum|rA| (forward sweep), wA.rA (backward sweep) and psi += alpha pA (next
    direction update) inside other kernels (ffm_solve.hip pcg(), k_tile<.., FUSE>).  Same per-cell arithmetic: the iteration
    count and every residual equal the unfused loop's and the oracle's (PCG.C), the solutions agree to the rounding of the
    reductions; a maxIter stop and a one-iteration solve leave psi complete (the deferred update is applied)."""
    H = ffm.hexmesh
    n = 96
    blk = H.HexBlock((n, n, n)); s = H.synth_p_rgh(blk)
    A = ffm.lduMatrix(ctx, blk.nCells, blk.l, blk.u)
    assert A.sweep_mode == 2
    A.set_coeffs(s["diag"], s["upper"])
    Ao = O.Ldu(blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"])
    src = ctx.to_device(s["source"])
    for kw in (dict(tolerance=1e-9), dict(tolerance=1e-30, maxIter=7), dict(tolerance=1e-30, maxIter=1)):
        res = {}
        for unfused in ("1", "0"):
            os.environ["FFM_PCG_UNFUSED"] = unfused
            try:
                psi = ctx.to_device(np.zeros(blk.nCells))
                pf = A.solve(psi, src, "PCG", "DIC", **kw)
            finally:
                os.environ.pop("FFM_PCG_UNFUSED", None)
            res[unfused] = (psi.cpu().numpy(), pf)
        psi_o, pf_o = Ao.solve(O.PCG, O.DIC, np.zeros(blk.nCells), s["source"], **kw)
        (xu, pu), (xf, pff) = res["1"], res["0"]
        assert pu["nIterations"] == pff["nIterations"] == pf_o["nIterations"], (kw, pu, pff, pf_o)
        assert abs(pff["finalResidual"] - pu["finalResidual"]) <= 1e-9 * pu["finalResidual"]
        assert abs(pff["initialResidual"] - pu["initialResidual"]) <= 1e-12 * pu["initialResidual"]
        assert np.linalg.norm(xf - xu) <= 1e-11 * np.linalg.norm(xu)
        assert np.linalg.norm(xf - psi_o) <= 1e-10 * np.linalg.norm(psi_o)
    A.close()
