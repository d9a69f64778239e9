"""The tiled kernels (ffm_tile.hip) under real concurrency: on a 160^3 box (100 tiles, 478 dependency levels, every tile
waiting on its neighbours' mailboxes) DIC's reciprocal diagonal, the preconditioner application, DILU on an asymmetric
matrix and Amul must be bitwise equal to the level-scheduled / row kernels, which share no code path with them beyond the
arithmetic.  Repeated applications check that stale mailbox contents of an earlier sweep are never read."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(ffm, ctx, n, mode, asym):
    H = ffm.hexmesh
    os.environ["FFM_SWEEP"] = mode
    try:
        blk = H.HexBlock((n, n, n))
        s = H.synth_p_rgh(blk)
        cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u)
        l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
        A = ffm.lduMatrix(ctx, blk.nCells, l2, u2)
    finally:
        os.environ.pop("FFM_SWEEP", None)
    assert A.native_order and A.sweep_mode == (2 if mode == "tile" else 0)
    up = s["upper"]
    lo = up * (1.0 + 0.3 * (H.hash_u(0xA1, blk.gface) - 0.5)) if asym else None
    A.set_coeffs(s["diag"][cOrd], up[fOrd], None if lo is None else lo[fOrd])
    return blk, s, cOrd, A


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
    for a, b in zip(out["tile"], out["levels"]):
        assert np.array_equal(a, b)
    assert np.array_equal(out["tile"][1], out["tile"][3])
