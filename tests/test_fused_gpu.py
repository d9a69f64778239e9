"""The fused assembly passes (csrc/ffm_fused.hip: ffm_fvc_grad_multi, ffm_fvm_scalar_transport_multi, ffm_fvm_lust_source3) must
give, bit for bit, what the chain of per-operator entry points gives (each of which is compared with the oracle in
tests/test_fv_operators_gpu.py): same expressions, same order, FMA contraction off.  And the compiled time step built on them
must equal the one built on the per-operator kernels (FFM_PLUME_UNFUSED) in every field, bitwise."""
import os
from ctypes import c_int as C_int

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(O, ffm, ctx):
    from oracle import fv, plume
    m = plume.make_mesh((9, 8, 7), h=0.1)
    N, F = m.nCells, m.nFaces
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())
    yield dict(fv=fv, m=m, A=A, mesh=mesh, cOrd=cOrd, fOrd=fOrd, N=N, F=F, B=sum(p.size for p in m.patches))
    mesh.close(); A.close()


def _fields(s, O, ctx, nf):
    N, F, B, mesh = s["N"], s["F"], s["B"], s["mesh"]
    dev = ctx.to_device
    vf = [dev(0.2 + O.hash_u(31 + 7 * i, np.arange(N))) for i in range(nf)]
    # one field that is constant over most of the mesh (exercises the large-ratio branch of NVDTVD::r)
    a = 0.2 + O.hash_u(31, np.arange(N)); vf[0] = dev(np.where(a > 0.9, a, 0.25))
    vb = [dev(0.1 + O.hash_u(133 + i, np.arange(B))) for i in range(nf)]
    phi = mesh.to_native(0.3 * (O.hash_u(32, np.arange(F)) - 0.5))
    return vf, vb, phi


@pytest.mark.parametrize("nf", [1, 3, 4])
def test_grad_multi_equals_grad(setup, O, ctx, nf):
    s, mesh = setup, setup["mesh"]
    vf, vb, _ = _fields(s, O, ctx, nf)
    g = [[ctx.zeros(s["N"]) for _ in range(3)] for _ in range(nf)]
    mesh.call("fvc_grad_multi", nf, vf, vb, [x[0] for x in g], [x[1] for x in g], [x[2] for x in g])
    for i in range(nf):
        r = [ctx.zeros(s["N"]) for _ in range(3)]
        mesh.call("fvc_grad", vf[i], vb[i], *r)
        for d in range(3):
            assert np.array_equal(g[i][d].cpu().numpy(), r[d].cpu().numpy()), (i, d)


@pytest.mark.parametrize("nf,scheme,with_expl", [(4, 3, False), (1, 2, True), (2, 2, False)])
def test_scalar_transport_multi_equals_the_operator_chain(setup, O, ctx, nf, scheme, with_expl):
    s, mesh = setup, setup["mesh"]
    N, F, B = s["N"], s["F"], s["B"]
    dev = ctx.to_device
    vf, vb, phi = _fields(s, O, ctx, nf)
    if scheme == 3:
        vf = [v * 1.1 - 0.2 for v in vf]                          # some values outside [0, 1]
    rho, rho0 = dev(1.0 + O.hash_u(60, np.arange(N))), dev(1.0 + O.hash_u(61, np.arange(N)))
    vf0 = [dev(O.hash_u(62 + i, np.arange(N))) for i in range(nf)]
    gam = mesh.to_native(0.01 * (1 + O.hash_u(63, np.arange(F)))); gamb = dev(0.01 * (1 + O.hash_u(64, np.arange(B))))
    phib = dev(0.2 * (O.hash_u(70, np.arange(B)) - 0.5))
    f = [dev(np.round(O.hash_u(80 + i, np.arange(B)) * 2) / 2) for i in range(nf)]
    ref = [dev(O.hash_u(84 + i, np.arange(B))) for i in range(nf)]
    rg = [dev(O.hash_u(88 + i, np.arange(B)) - 0.5) for i in range(nf)]
    su = [dev(O.hash_u(90 + i, np.arange(N)) - 0.3) if i % 2 == 0 else None for i in range(nf)]
    expl = [dev(O.hash_u(95 + e, np.arange(N)) - 0.5) for e in range(3)] if with_expl else None
    # a second explicit source and an implicit one (radiation->Sh in the enthalpy equation): on the first field when expl is on
    su2 = [dev(O.hash_u(98, np.arange(N)) - 0.2) if (with_expl and i == 0) else None for i in range(nf)]
    sp = [dev(O.hash_u(99, np.arange(N))) if (with_expl and i == 0) else None for i in range(nf)]
    rdt = 1000.0
    # gradients of the fields (the limiter's input)
    g = [[ctx.zeros(N) for _ in range(3)] for _ in range(nf)]
    for i in range(nf):
        mesh.call("fvc_grad", vf[i], vb[i], *g[i])
    D = [ctx.zeros(N) for _ in range(nf)]; S = [ctx.zeros(N) for _ in range(nf)]
    Up = [ctx.zeros(mesh.nNative) for _ in range(nf)]; Lo = [ctx.zeros(mesh.nNative) for _ in range(nf)]
    mesh.call("fvm_scalar_transport_multi", nf, scheme, 1.0, 0.0, 1.0, rdt, rho, rho0, phi, phib, gam, gamb,
              vf, [x[0] for x in g], [x[1] for x in g], [x[2] for x in g], vf0, f, ref, rg, su, su2, sp,
              (expl + [None] * (3 * (nf - 1))) if with_expl else None, D, Up, Lo, S)
    V = dev(s["m"].V[s["cOrd"]])
    for i in range(nf):
        w = ctx.zeros(mesh.nNative)
        mesh.call("fv_limited_weights", scheme, 1.0, 0.0, 1.0, phi, vf[i], *g[i], w)
        d, up, lo = ctx.zeros(N), ctx.zeros(mesh.nNative), ctx.zeros(mesh.nNative)
        mesh.call("fvm_transport", rdt, rho, phi, w, gam, -1, d, up, lo)
        ic, bc = ctx.zeros(B), ctx.zeros(B)
        mesh.call("fvm_boundary_coeffs", phib, gamb, -1, f[i], ref[i], rg[i], ic, bc)
        src = rdt * rho0 * vf0[i] * V
        if with_expl and i == 0:
            src = ((src - V * expl[0]) - V * expl[1]) - V * expl[2]
        if su[i] is not None:
            src = src + V * su[i]
        if sp[i] is not None:
            d = d + V * sp[i]
        if su2[i] is not None:
            src = src + V * su2[i]
        dO, sO = ctx.zeros(N), ctx.zeros(N)
        mesh.call("fvm_add_boundary", ic, bc, d, src, None, dO, sO)
        assert np.array_equal(Up[i].cpu().numpy(), up.cpu().numpy()), i
        assert np.array_equal(Lo[i].cpu().numpy(), lo.cpu().numpy()), i
        assert np.array_equal(D[i].cpu().numpy(), dO.cpu().numpy()), i
        assert np.array_equal(S[i].cpu().numpy(), sO.cpu().numpy()), i


@pytest.mark.parametrize("nf", [2, 6])
def test_multivariate_weights_equal_the_running_minimum_chain(setup, O, ctx, nf):
    """multivariateSelectionScheme: the weights of ONE limiter, the minimum over the fields' own limitedLinear / limitedLinear01
    limiters.  The one-pass kernel (ffm_fv_multivariate_weights, the compiled time step) = ffm_fv_limited_limiter over the fields
    (running minimum) + ffm_fv_weights_from_limiter (what the Foam layer's convectionScheme calls), bit for bit; and the minimum of the
    fields' own weights can be recovered from it where the limiter is 0 or 1."""
    import ctypes as C
    s, mesh = setup, setup["mesh"]
    N = s["N"]
    vf, vb, phi = _fields(s, O, ctx, nf)
    vf = [v * 1.1 - 0.2 if i % 2 else v for i, v in enumerate(vf)]           # limitedLinear01 fields with values outside [0, 1]
    sch = [2 if i == 0 else 3 for i in range(nf)]
    g = [[ctx.zeros(N) for _ in range(3)] for _ in range(nf)]
    for i in range(nf):
        mesh.call("fvc_grad", vf[i], vb[i], *g[i])
    w1 = ctx.zeros(mesh.nNative)
    mesh.call("fv_multivariate_weights", nf, (C.c_int * nf)(*sch), 1.0, 0.0, 1.0, phi, vf, [x[0] for x in g], [x[1] for x in g], [x[2] for x in g], w1)
    lim, w2 = ctx.zeros(mesh.nNative), ctx.zeros(mesh.nNative)
    for i in range(nf):
        mesh.call("fv_limited_limiter", sch[i], 1.0, 0.0, 1.0, phi, vf[i], *g[i], lim, 0 if i == 0 else 1)
    mesh.call("fv_weights_from_limiter", phi, lim, w2)
    assert np.array_equal(w1.cpu().numpy(), w2.cpu().numpy())
    # against the per-field weights: limiter_i = (w_i - upwind)/(linear - upwind); the common limiter is their minimum (real faces only:
    # the native layout has padding entries)
    fn = mesh.from_native
    up = (fn(phi) >= 0).astype(float)
    wl = ctx.zeros(mesh.nNative); mesh.call("fv_limited_weights", 1, 1.0, 0.0, 1.0, phi, None, None, None, None, wl)
    lims = []
    for i in range(nf):
        wi = ctx.zeros(mesh.nNative)
        mesh.call("fv_limited_weights", sch[i], 1.0, 0.0, 1.0, phi, vf[i], *g[i], wi)
        lims.append((fn(wi) - up) / (fn(wl) - up))
    assert np.allclose(np.min(lims, axis=0), fn(lim), rtol=0, atol=1e-12)
    assert float(np.mean(fn(lim) > 0)) < 0.999 and (nf > 2 or float(np.mean(fn(lim) > 0)) > 0.02)      # (six random fields: the minimum is 0 nearly everywhere)


@pytest.mark.parametrize("nf,with_expl", [(4, False), (1, True)])
def test_scalar_transport_multi_with_given_weights_equals_the_operator_chain(setup, O, ctx, nf, with_expl):
    """ffm_fvm_scalar_transport_multi_w (mvConvection->fvmDiv with the common weights + ddt + laplacian + sources + boundary terms of nf
    equations in one pass) against fvm_transport(w) / fvm_boundary_coeffs / fvm_add_boundary per field, bitwise."""
    s, mesh = setup, setup["mesh"]
    N, F, B = s["N"], s["F"], s["B"]
    dev = ctx.to_device
    _, _, phi = _fields(s, O, ctx, nf)
    w = mesh.to_native(O.hash_u(41, np.arange(F)))                             # any weights in [0, 1]
    rho, rho0 = dev(1.0 + O.hash_u(60, np.arange(N))), dev(1.0 + O.hash_u(61, np.arange(N)))
    vf0 = [dev(O.hash_u(62 + i, np.arange(N))) for i in range(nf)]
    gam = mesh.to_native(0.01 * (1 + O.hash_u(63, np.arange(F)))); gamb = dev(0.01 * (1 + O.hash_u(64, np.arange(B))))
    phib = dev(0.2 * (O.hash_u(70, np.arange(B)) - 0.5))
    f = [dev(np.round(O.hash_u(80 + i, np.arange(B)) * 2) / 2) for i in range(nf)]
    ref = [dev(O.hash_u(84 + i, np.arange(B))) for i in range(nf)]
    rg = [dev(O.hash_u(88 + i, np.arange(B)) - 0.5) for i in range(nf)]
    su = [dev(O.hash_u(90 + i, np.arange(N)) - 0.3) if i % 2 == 0 else None for i in range(nf)]
    expl = [dev(O.hash_u(95 + e, np.arange(N)) - 0.5) for e in range(3)] if with_expl else None
    su2 = [dev(O.hash_u(98, np.arange(N)) - 0.2) if (with_expl and i == 0) else None for i in range(nf)]
    sp = [dev(O.hash_u(99, np.arange(N))) if (with_expl and i == 0) else None for i in range(nf)]
    rdt = 1000.0
    D = [ctx.zeros(N) for _ in range(nf)]; S = [ctx.zeros(N) for _ in range(nf)]
    Up = [ctx.zeros(mesh.nNative) for _ in range(nf)]; Lo = [ctx.zeros(mesh.nNative) for _ in range(nf)]
    mesh.call("fvm_scalar_transport_multi_w", nf, w, rdt, rho, rho0, phi, phib, gam, gamb, vf0, f, ref, rg, su, su2, sp,
              (expl + [None] * (3 * (nf - 1))) if with_expl else None, D, Up, Lo, S)
    V = dev(s["m"].V[s["cOrd"]])
    for i in range(nf):
        d, up, lo = ctx.zeros(N), ctx.zeros(mesh.nNative), ctx.zeros(mesh.nNative)
        mesh.call("fvm_transport", rdt, rho, phi, w, gam, -1, d, up, lo)
        ic, bc = ctx.zeros(B), ctx.zeros(B)
        mesh.call("fvm_boundary_coeffs", phib, gamb, -1, f[i], ref[i], rg[i], ic, bc)
        src = rdt * rho0 * vf0[i] * V
        if with_expl and i == 0:
            src = ((src - V * expl[0]) - V * expl[1]) - V * expl[2]
        if su[i] is not None:
            src = src + V * su[i]
        if sp[i] is not None:
            d = d + V * sp[i]
        if su2[i] is not None:
            src = src + V * su2[i]
        dO, sO = ctx.zeros(N), ctx.zeros(N)
        mesh.call("fvm_add_boundary", ic, bc, d, src, None, dO, sO)
        assert np.array_equal(Up[i].cpu().numpy(), up.cpu().numpy()), i
        assert np.array_equal(Lo[i].cpu().numpy(), lo.cpu().numpy()), i
        assert np.array_equal(D[i].cpu().numpy(), dO.cpu().numpy()), i
        assert np.array_equal(S[i].cpu().numpy(), sO.cpu().numpy()), i


def test_lust_source3_equals_the_operator_chain(setup, O, ctx):
    s, mesh = setup, setup["mesh"]
    N, B = s["N"], s["B"]
    dev = ctx.to_device
    vf, vb, phi = _fields(s, O, ctx, 3)
    rho0 = dev(1.0 + O.hash_u(61, np.arange(N)))
    U0 = [dev(O.hash_u(62 + i, np.arange(N)) - 0.5) for i in range(3)]
    g = [[ctx.zeros(N) for _ in range(3)] for _ in range(3)]
    for i in range(3):
        mesh.call("fvc_grad", vf[i], vb[i], *g[i])
    out = [ctx.zeros(N) for _ in range(3)]
    rdt = 250.0
    mesh.call("fvm_lust_source3", rdt, phi, rho0, U0, [x[0] for x in g], [x[1] for x in g], [x[2] for x in g], out)
    V = dev(s["m"].V[s["cOrd"]])
    zb = ctx.zeros(B)
    for i in range(3):
        corr = ctx.zeros(mesh.nNative)
        mesh.call("fv_lust_correction", phi, *g[i], corr)
        corr = phi * corr
        divc = ctx.zeros(N)
        mesh.call("fvc_surface_integrate", corr, zb, divc)
        ref = rdt * rho0 * U0[i] * V - V * divc
        assert np.array_equal(out[i].cpu().numpy(), ref.cpu().numpy()), i


FIELDS = ["rho", "p", "p_rgh", "T", "h", "Ux", "Uy", "Uz", "O2", "H2O", "C3H8", "CO2", "N2", "K"]


@pytest.mark.parametrize("n", [(12, 16, 12), (40, 36, 33)])
def test_time_step_fused_equals_per_operator(ffm, ctx, n):
    """the compiled time step on the fused passes == the same step on one kernel per operator: every field and every
    iteration count identical after three steps (the second size has several tiles and levels wider than one entry)"""
    fused = ffm.Plume(ctx, n)
    os.environ["FFM_PLUME_UNFUSED"] = "1"
    try:
        plain = ffm.Plume(ctx, n)
    finally:
        os.environ.pop("FFM_PLUME_UNFUSED", None)
    for step in range(3):
        fused.step(); plain.step()
        assert [(a, p["nIterations"]) for a, p in fused.solves()] == [(a, p["nIterations"]) for a, p in plain.solves()]
        for name in FIELDS:
            assert np.array_equal(fused.field(name), plain.field(name)), (step, name)
    fused.close(); plain.close()


@pytest.mark.parametrize("n,nf", [((40, 36, 33), 6), ((24, 40, 20), 3), ((50, 34, 34), 6)])
def test_tiled_multivariate_weights_equal_the_two_pass_form(O, ffm, ctx, n, nf):
    """ffm_fv_multivariate_weights_tiled (k_mv_tile: gradients + common limiter in one pass, cell values staged through LDS on the tile
    numbering) == ffm_fvc_grad_multi + ffm_fv_multivariate_weights, bit for bit, on the tile-numbered mesh of the plume case with hashed
    fields (values outside [0, 1], a field that is constant over most of the mesh, fluxes of both signs and exact zeros)."""
    case = ffm.Plume(ctx, n)
    mesh = case.mesh()
    N, B, nNat = case.nCells, mesh.nBoundary, mesh.nNative
    dev = ctx.to_device
    vf = [dev(1.1 * (0.2 + O.hash_u(31 + 7 * i, np.arange(N))) - 0.2) for i in range(nf)]
    a = 0.2 + O.hash_u(31, np.arange(N)); vf[0] = dev(np.where(a > 0.9, a, 0.25))
    vb = [dev(0.1 + O.hash_u(133 + i, np.arange(B))) for i in range(nf)]
    ph = 0.3 * (O.hash_u(32, np.arange(nNat)) - 0.5); ph[::7] = 0.0
    phi = dev(ph)
    sch = (C_int * nf)(*([2] + [3] * (nf - 1)))
    g = [[ctx.zeros(N) for _ in range(3)] for _ in range(nf)]
    mesh.call("fvc_grad_multi", nf if nf <= 4 else 4, vf[:4], vb[:4], [x[0] for x in g[:4]], [x[1] for x in g[:4]], [x[2] for x in g[:4]])
    if nf > 4:
        mesh.call("fvc_grad_multi", nf - 4, vf[4:], vb[4:], [x[0] for x in g[4:]], [x[1] for x in g[4:]], [x[2] for x in g[4:]])
    two = ctx.zeros(nNat) + 7.0
    mesh.call("fv_multivariate_weights", nf, sch, 1.0, 0.0, 1.0, phi, vf, [x[0] for x in g], [x[1] for x in g], [x[2] for x in g], two)
    one = ctx.zeros(nNat) + 7.0
    mesh.call("fv_multivariate_weights_tiled", nf, sch, 1.0, 0.0, 1.0, phi, vf, vb, one)
    a, b = one.cpu().numpy(), two.cpu().numpy()
    assert np.array_equal(a, b), (np.abs(a - b).max(), int((a != b).sum()))
    real = b[b != 7.0]                                                                  # (padding entries of the native layout keep the fill value)
    assert ((real > 1e-6) & (real < 1 - 1e-6) & (np.abs(real - 0.5) > 1e-6)).any() and (real == 1.0).any() and (real == 0.0).any()
    case.close()
