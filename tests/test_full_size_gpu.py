"""BASELINE.json's full size (400^3 = 64 M cells, SURVEY 8d synthetic p_rgh matrix) through size-independent properties, since the
serial oracle would take minutes there: the tiled Amul equals the row kernel bit for bit; Amul is linear; the DIC preconditioner
is a symmetric operator (<M^-1 r, s> = <r, M^-1 s>); a DIC-PCG solve reports residuals that an independent evaluation of
|b - A psi|_1 reproduces, and meets its tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_size_properties(ffm, ctx):
    import torch
    H = ffm.hexmesh
    n = int(os.environ.get("FFM_FULL_EDGE", "400"))
    blk = H.HexBlock((n, n, n))
    s = H.synth_p_rgh(blk)
    cOrd, fOrd = ffm.renumber_levels(blk.nCells, blk.l, blk.u)
    l2, u2, _ = H.apply_renumbering(blk.nCells, blk.l, blk.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, blk.nCells, l2, u2)
    assert A.native_order and A.sweep_mode == 2
    A.set_coeffs(s["diag"][cOrd], s["upper"][fOrd])
    N = blk.nCells
    x = ctx.to_device(s["x"][cOrd]); b = ctx.to_device(s["source"][cOrd])
    y = ctx.to_device(H.hash_u(0xF6, blk.gcell)[cOrd])
    del blk, l2, u2
    # (1) tiled Amul == row kernel, bitwise
    Ax = A.Amul(x)
    os.environ["FFM_NO_TILE_AMUL"] = "1"
    try:
        Ax_rows = A.Amul(x)
    finally:
        del os.environ["FFM_NO_TILE_AMUL"]
    assert torch.equal(Ax, Ax_rows)
    del Ax_rows
    # (2) linearity
    Ay = A.Amul(y)
    lhs = A.Amul(x + 2.0 * y)
    rhs = Ax + 2.0 * Ay
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) < 1e-14
    del lhs, rhs, Ay, Ax
    # (3) DIC is a symmetric operator
    A.reciprocalD("DIC")
    Mr = A.precondition("DIC", b); Ms = A.precondition("DIC", y)
    d1 = float(torch.dot(Mr, y)); d2 = float(torch.dot(b, Ms))
    assert abs(d1 - d2) <= 1e-11 * max(abs(d1), abs(d2))
    del Mr, Ms
    # (4) DIC-PCG: reported residual ratio = independently evaluated |b - A psi|_1 ratio; tolerance met
    psi = ctx.zeros(N)
    r0 = float((b - A.Amul(psi)).abs().sum())
    perf = A.solve(psi, b, solver="PCG", preconditioner="DIC", tolerance=1e-6, relTol=0.0)
    r1 = float((b - A.Amul(psi)).abs().sum())
    assert perf["converged"] == 1 and perf["finalResidual"] < 1e-6
    assert abs(r1 / r0 - perf["finalResidual"] / perf["initialResidual"]) <= 1e-6 * (r1 / r0)
    A.close()
