"""Operator-level parity of the fvc:: / fvm:: kernels through the C ABI against the numpy
restatement oracle/fv.py, on a hex box with four patches and hashed fields.  Bars: interpolate, snGrad, flux,
surfaceIntegrate, grad, limiter weights, matrix coefficients, boundary coefficients, A/H/flux accumulate in the
reference's face order with FMA contraction off -> compared at 1e-14 relative (bitwise where the oracle's numpy
expression order is identical); reconstruct inverts the cell tensor analytically (oracle: LAPACK) -> 1e-13."""
import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(O, ffm, ctx):
    from oracle import fv, plume
    m = plume.make_mesh((7, 6, 5), h=0.1)
    N, F = m.nCells, m.nFaces
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    assert A.native_order
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    yield dict(fv=fv, m=m, A=A, mesh=mesh, cOrd=cOrd, fOrd=fOrd, N=N, F=F, B=sum(p.size for p in m.patches))
    mesh.close(); A.close()


def cellf(s, ctx, a): return ctx.to_device(np.asarray(a)[s["cOrd"]])
def facef(s, a): return s["mesh"].to_native(np.asarray(a)[s["fOrd"]])
def bndf(ctx, lst): return ctx.to_device(np.concatenate(lst))
def back_cell(s, t): out = np.empty(s["N"]); out[s["cOrd"]] = t.cpu().numpy(); return out
def back_face(s, t): out = np.empty(s["F"]); out[s["fOrd"]] = s["mesh"].from_native(t); return out


def fields(s, O):
    N, F, m = s["N"], s["F"], s["m"]
    vf = 0.2 + O.hash_u(31, np.arange(N))
    phi = 0.3 * (O.hash_u(32, np.arange(F)) - 0.5)
    vb = [0.1 + O.hash_u(33 + q, np.arange(p.size)) for q, p in enumerate(m.patches)]
    return vf, phi, vb


def test_fvc_operators(setup, O, ctx):
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    vf, phi, vb = fields(s, O)
    vfd, vbd = cellf(s, ctx, vf), bndf(ctx, vb)
    out_f = ctx.zeros(mesh.nNative)
    mesh.call("fvc_interpolate", None, vfd, out_f)
    assert np.array_equal(back_face(s, out_f), fv.interpolate(m, vf, vb)[0])
    mesh.call("fvc_snGrad", vfd, out_f)
    sg, sgb = fv.snGrad(m, vf, vb)
    assert np.array_equal(back_face(s, out_f), sg)
    out_b = ctx.zeros(s["B"])
    mesh.call("fvc_snGrad_b", vfd, vbd, out_b)
    assert np.array_equal(out_b.cpu().numpy(), np.concatenate(sgb))
    # surfaceIntegrate / surfaceSum
    out_c = ctx.zeros(s["N"])
    ssb = [0.05 * (O.hash_u(40 + q, np.arange(p.size)) - 0.5) for q, p in enumerate(m.patches)]
    mesh.call("fvc_surface_integrate", facef(s, phi), bndf(ctx, ssb), out_c)
    assert rel_l2(back_cell(s, out_c), fv.surface_integrate(m, phi, ssb)) < 1e-15
    mesh.call("fvc_surface_sum", facef(s, phi), bndf(ctx, ssb), out_c)
    assert rel_l2(back_cell(s, out_c), fv.surface_sum(m, phi, ssb)) < 1e-15
    # grad
    g = [ctx.zeros(s["N"]) for _ in range(3)]
    mesh.call("fvc_grad", vfd, vbd, *g)
    ref = fv.grad(m, vf, vb)
    for d in range(3):
        assert rel_l2(back_cell(s, g[d]), ref[:, d]) < 1e-14
    # reconstruct
    mesh.call("fvc_reconstruct", facef(s, phi), bndf(ctx, ssb), *g)
    ref = fv.reconstruct(m, phi, ssb)
    for d in range(3):
        assert rel_l2(back_cell(s, g[d]), ref[:, d]) < 1e-13
    # flux of a vector field
    U = np.stack([O.hash_u(50 + d, np.arange(s["N"])) - 0.5 for d in range(3)])
    mesh.call("fvc_flux", *[cellf(s, ctx, U[d]) for d in range(3)], out_f)
    ref = sum(fv.interpolate(m, U[d], [np.zeros(p.size) for p in m.patches])[0] * m.Sf[:, d] for d in range(3))
    assert rel_l2(back_face(s, out_f), ref) < 1e-15


@pytest.mark.parametrize("scheme,code", [("upwind", 0), ("linear", 1), ("limitedLinear", 2), ("limitedLinear01", 3)])
def test_limited_weights(setup, O, ctx, scheme, code):
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    vf, phi, vb = fields(s, O)
    if scheme == "limitedLinear01":
        vf = vf * 1.1 - 0.2                      # some values outside [0,1] to exercise the bound test
    grad = fv.grad(m, vf, vb)
    w = ctx.zeros(mesh.nNative)
    mesh.call("fv_limited_weights", code, 1.0, 0.0, 1.0, facef(s, phi), cellf(s, ctx, vf), *[cellf(s, ctx, grad[:, d]) for d in range(3)], w)
    ref = fv.limited_weights(m, scheme, phi, vf, grad, 1.0)
    got = back_face(s, w)
    assert np.abs(got - ref).max() < 1e-13
    assert got.min() >= 0.0 and got.max() <= 1.0


def test_limiter_unit_vectors(O):
    """T6: hand-computed 1-D stencils of limitedLinear(1): r = 2*(d.gradP)/(N-P) - 1, limiter = clamp(2r, 0, 1)."""
    from oracle import fv
    m = fv.HexMesh((4, 1, 1), (0, 0, 0), (4, 1, 1)).set_patches([])
    vf = np.array([0.0, 1.0, 2.0, 3.0])                  # linear profile: centred gradient = downwind gradient -> r = 1 -> limiter 1
    g = np.zeros((4, 3)); g[:, 0] = 1.0
    w = fv.limited_weights(m, "limitedLinear", np.array([1.0, 1.0, 1.0]), vf, g, 1.0)
    assert np.allclose(w, 0.5)                           # pure linear
    vf = np.array([0.0, 0.0, 1.0, 1.0]); g[:, 0] = [0.0, 0.5, 0.5, 0.0]
    w = fv.limited_weights(m, "limitedLinear", np.array([1.0, 1.0, 1.0]), vf, g, 1.0)
    # face 1|2: gradf = 1, gradcf = d.gradP = 0.5 -> r = 0 -> limiter 0 -> upwind weight 1
    assert w[1] == 1.0
    w = fv.limited_weights(m, "limitedLinear", np.array([-1.0, -1.0, -1.0]), vf, g, 1.0)
    assert w[1] == 0.0                                   # reversed flux: upwind = neighbour
    # limitedLinear01: value out of [0,1] on the upwind side switches to upwind
    vf = np.array([0.0, 1.2, 1.0, 1.0]); g[:, 0] = 0.3
    w = fv.limited_weights(m, "limitedLinear01", np.array([1.0, 1.0, 1.0]), vf, g, 1.0)
    assert w[1] == 1.0


@pytest.mark.parametrize("scheme,code", [("limitedLinear", 2), ("limitedLinear01", 3)])
def test_limited_weights_piecewise_constant(setup, O, ctx, scheme, code):
    """the same on the device: a field that is constant over most of the mesh (the ambient of a plume) gets linear weights
    there, bitwise the oracle's"""
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    vf, phi, vb = fields(s, O)
    vf = np.where(vf > 0.8, vf, 0.25)                           # ~80 % of the cells at one value
    vb = [np.full(p.size, 0.25) for p in m.patches]
    grad = fv.grad(m, vf, vb)
    w = ctx.zeros(mesh.nNative)
    mesh.call("fv_limited_weights", code, 1.0, 0.0, 1.0, facef(s, phi), cellf(s, ctx, vf), *[cellf(s, ctx, grad[:, d]) for d in range(3)], w)
    ref = fv.limited_weights(m, scheme, phi, vf, grad, 1.0)
    got = back_face(s, w)
    assert np.array_equal(got, ref)
    flat = (vf[m.l] == 0.25) & (vf[m.u] == 0.25) & (np.abs(grad[m.l]).sum(axis=1) == 0) & (np.abs(grad[m.u]).sum(axis=1) == 0)
    assert flat.sum() > 0 and np.all(got[flat] == m.weights[flat])


def test_filteredLinear2V_weights(setup, O, ctx):
    """the vector limiter of cases/wallFireSpread2D/system/fvSchemes:41 on the device against oracle/fv.py (hand-computed stencils:
    tests/test_oracle_cpu.py::test_filteredLinear2V_unit_stencils); smooth and rough fields so that all three outcomes
    (limiter 1, blended, 0) occur"""
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    N, F = s["N"], s["F"]
    _, phi, _ = fields(s, O)
    for rough in (0.05, 1.0):
        U = np.stack([np.sin(1.3 * m.C[:, 0] + d) * np.cos(0.7 * m.C[:, 1] - d) + rough * (O.hash_u(70 + d, np.arange(N)) - 0.5) for d in range(3)], axis=1)
        Ub = [[0.5 * U[p.faceCells, d] for p in m.patches] for d in range(3)]
        g = np.stack([fv.grad(m, U[:, d], Ub[d]) for d in range(3)], axis=2)              # g[c][i][j] = d_i U_j
        ref = fv.filtered_linear2V_weights(m, phi, U, g, 0.2, 0.05)
        w = ctx.zeros(mesh.nNative)
        Ud = [cellf(s, ctx, U[:, d]) for d in range(3)]
        gd = [[cellf(s, ctx, g[:, i, d]) for d in range(3)] for i in range(3)]           # gd[i][d] = d_i U_d
        mesh.call("fv_filtered_linear2V_weights", 0.2, 0.05, facef(s, phi), Ud, gd[0], gd[1], gd[2], w)
        got = back_face(s, w)
        assert np.abs(got - ref).max() < 1e-13
        lim = np.where(phi >= 0, (1.0 - got) / 0.5, got / 0.5)                             # (1-lim) share of upwind: 1 - lim
        assert got.min() >= 0.0 and got.max() <= 1.0
    assert (np.abs(got - 0.5) < 1e-15).any() and (np.abs(got - fv.pos0(phi)) < 1e-15).any()   # rough field: both ends of the limiter


def test_fvm_assembly_and_matrix_ops(setup, O, ctx):
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    N, F, B = s["N"], s["F"], s["B"]
    vf, phi, vb = fields(s, O)
    rho = 1.0 + O.hash_u(60, np.arange(N)); rho0 = 1.0 + O.hash_u(61, np.arange(N)); vf0 = O.hash_u(62, np.arange(N))
    gam = 0.01 * (1 + O.hash_u(63, np.arange(F))); gamb = [0.01 * (1 + O.hash_u(64 + q, np.arange(p.size))) for q, p in enumerate(m.patches)]
    phib = [0.2 * (O.hash_u(70 + q, np.arange(p.size)) - 0.5) for q, p in enumerate(m.patches)]
    w = fv.limited_weights(m, "limitedLinear", phi, vf, fv.grad(m, vf, vb), 1.0)
    bc = fv.MixedBC(m)
    for q, p in enumerate(m.patches):
        bc.f[q] = np.round(O.hash_u(80 + q, np.arange(p.size)) * 2) / 2          # 0, 0.5, 1
        bc.ref[q] = O.hash_u(84 + q, np.arange(p.size)); bc.refGrad[q] = O.hash_u(88 + q, np.arange(p.size)) - 0.5
    rdt = 1000.0
    M = fv.fvm_ddt(m, rdt, rho, rho0, vf0)
    M += fv.fvm_div(m, phi, phib, w, [bc])
    M -= fv.fvm_laplacian(m, gam, gamb, [bc])
    diag, up, lo = ctx.zeros(N), ctx.zeros(mesh.nNative), ctx.zeros(mesh.nNative)
    mesh.call("fvm_transport", rdt, cellf(s, ctx, rho), facef(s, phi), facef(s, w), facef(s, gam), -1, diag, up, lo)
    assert np.array_equal(back_face(s, up), M.upper) and np.array_equal(back_face(s, lo), M.lower)
    assert rel_l2(back_cell(s, diag), M.diag) < 1e-15
    ic, bcc = ctx.zeros(B), ctx.zeros(B)
    fd, rd, gd = bndf(ctx, bc.f), bndf(ctx, bc.ref), bndf(ctx, bc.refGrad)
    mesh.call("fvm_boundary_coeffs", bndf(ctx, phib), bndf(ctx, gamb), -1, fd, rd, gd, ic, bcc)
    assert rel_l2(ic.cpu().numpy(), np.concatenate([a[0] for a in M.internalCoeffs])) < 1e-15
    assert rel_l2(bcc.cpu().numpy(), np.concatenate([a[0] for a in M.boundaryCoeffs])) < 1e-15
    vbd = ctx.zeros(B)
    mesh.call("bc_values", fd, rd, gd, cellf(s, ctx, vf), vbd)
    assert np.array_equal(vbd.cpu().numpy(), np.concatenate(bc.values(m, vf)))
    # solve_system (addBoundaryDiag/Source), A, H, flux
    srcd = cellf(s, ctx, M.source[0])
    d2, s2 = ctx.zeros(N), ctx.zeros(N)
    mesh.call("fvm_add_boundary", ic, bcc, diag, srcd, None, d2, s2)
    dref, sref = M.solve_system()
    assert rel_l2(back_cell(s, d2), dref) < 1e-15 and rel_l2(back_cell(s, s2), sref) < 1e-15
    outA = ctx.zeros(N)
    mesh.call("fvm_A", 1, diag, ic, ic, ic, outA)
    assert rel_l2(back_cell(s, outA), M.A()) < 1e-15
    psi = O.hash_u(90, np.arange(N))
    outH = ctx.zeros(N)
    mesh.call("fvm_H", 1, 0, up, lo, srcd, ic, ic, ic, bcc, cellf(s, ctx, psi), outH)
    assert rel_l2(back_cell(s, outH), M.H(psi[None, :])[0]) < 1e-14
    ff, fb = ctx.zeros(mesh.nNative), ctx.zeros(B)
    mesh.call("fvm_flux", up, lo, ic, bcc, cellf(s, ctx, psi), ff, fb)
    fi, fbr = M.flux(psi)
    assert np.array_equal(back_face(s, ff), fi) and rel_l2(fb.cpu().numpy(), np.concatenate(fbr)) < 1e-15


def test_lust_weights_and_correction(setup, O, ctx):
    """`div(phi,U) Gauss LUST grad(U)` (cases/steckler/system/fvSchemes:32): weights 0.75 linear + 0.25 upwind, explicit
    correction 0.25*(Cf - C_upwind) & grad(vf)_upwind; both bitwise against oracle/fv.py (same expression order)."""
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    vf, phi, vb = fields(s, O)
    phi = phi.copy(); phi[::7] = 0.0                      # the flux == 0 branch differs between weights (>= 0) and correction (> 0)
    mesh.set_face_centres(m.Cf[s["fOrd"]].T.copy())
    w = ctx.zeros(mesh.nNative)
    mesh.call("fv_limited_weights", 4, 1.0, 0.0, 1.0, facef(s, phi), None, None, None, None, w)
    assert np.array_equal(back_face(s, w), fv.lust_weights(m, phi))
    gref = fv.grad(m, vf, vb)
    g = [cellf(s, ctx, gref[:, d]) for d in range(3)]
    corr = ctx.zeros(mesh.nNative)
    mesh.call("fv_lust_correction", facef(s, phi), *g, corr)
    assert np.array_equal(back_face(s, corr), fv.lust_correction(m, phi, gref))
    # linearUpwind (cases/wallFireSpread2D/system/fvSchemes:58): upwind weights, the same correction without the factor 0.25
    mesh.call("fv_limited_weights", 5, 1.0, 0.0, 1.0, facef(s, phi), None, None, None, None, w)
    assert np.array_equal(back_face(s, w), fv.pos0(phi))
    mesh.call("fv_linear_upwind_correction", facef(s, phi), *g, corr)
    assert np.array_equal(back_face(s, corr), fv.linear_upwind_correction(m, phi, gref))


@pytest.mark.parametrize("nc", [1, 3])
def test_fvm_relax(setup, O, ctx, nc):
    """fvMatrix::relax(alpha) on a convection-diffusion matrix that is NOT diagonally dominant everywhere (so that the
    max(|D|, sumMagOffDiag) branch is exercised), scalar and vector (cmptMax / cmptMin of the boundary coefficients)."""
    s, fv, m, mesh = setup, setup["fv"], setup["m"], setup["mesh"]
    N = s["N"]
    vf, phi, vb = fields(s, O)
    bcs = [fv.MixedBC(m, f=[O.hash_u(90 + 7 * c + q, np.arange(p.size)) for q, p in enumerate(m.patches)],
                      ref=[np.full(p.size, 0.3 + c) for p in m.patches]) for c in range(nc)]
    phib = [0.2 * (O.hash_u(60 + q, np.arange(p.size)) - 0.5) for q, p in enumerate(m.patches)]
    M = fv.fvm_div(m, 5.0 * phi, phib, fv.limited_weights(m, "linear", phi, None, None), bcs)       # central differencing: not dominant
    M -= fv.fvm_laplacian(m, np.full(s["F"], 1e-3), [np.full(p.size, 1e-3) for p in m.patches], bcs)
    M += fv.fvm_ddt(m, 0.1, np.ones(N), np.ones(N), np.stack([vf + c for c in range(nc)]))
    psi = np.stack([vf * (1 + 0.1 * c) for c in range(nc)])
    up, lo, dg = facef(s, M.upper), facef(s, M.lower), cellf(s, ctx, M.diag)
    ic = [bndf(ctx, [M.internalCoeffs[q][c] for q in range(len(m.patches))]) for c in range(nc)]
    src = [cellf(s, ctx, M.source[c]) for c in range(nc)]
    ps = [cellf(s, ctx, psi[c]) for c in range(nc)]
    pad = lambda lst: lst + [None] * (3 - len(lst))
    mesh.call("fvm_relax", 0.7, nc, up, lo, *pad(ic), dg, *pad(ps), *pad(src))
    D0 = M.diag.copy()
    M.relax(0.7, psi)
    assert (np.abs(D0) < np.abs(M.diag) * 0.7 - 1e-15).any()          # dominance was enforced somewhere
    assert np.array_equal(back_cell(s, dg), M.diag)
    for c in range(nc):
        assert np.array_equal(back_cell(s, src[c]), M.source[c])


def test_nonorthogonal_correction_on_a_sheared_mesh(O, ffm, ctx):
    """`Gauss linear corrected` / `corrected` snGrad (cases/wallFireSpread2D/system/fvSchemes; zero on the reference's own
    orthogonal meshes): on a sheared hex box the explicit term nonOrthCorrectionVectors & interpolate(grad(vf)) against
    oracle/fv.py, the Gauss gradient on the sheared geometry, and the property that makes the scheme second order: for a linear
    field the corrected surface-normal gradient of an interior face is exact, nf & a."""
    from oracle import fv, plume
    m = plume.make_mesh((7, 6, 5), h=0.1)
    fv.shear(m, [[1.0, 0.35, 0.1], [0.0, 1.0, 0.25], [0.0, 0.0, 1.0]])
    assert np.abs(m.nonOrthCorrectionVectors).max() > 0.1
    N, F = m.nCells, m.nFaces
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_nonorth_correction(m.nonOrthCorrectionVectors[fOrd].T.copy())
    s = dict(mesh=mesh, cOrd=cOrd, fOrd=fOrd, N=N, F=F)
    # hashed field: gradient and correction against the oracle
    vf = 0.2 + O.hash_u(71, np.arange(N)); vb = [0.1 + O.hash_u(72 + q, np.arange(p.size)) for q, p in enumerate(m.patches)]
    g = [ctx.zeros(N) for _ in range(3)]
    mesh.call("fvc_grad", cellf(s, ctx, vf), bndf(ctx, vb), *g)
    gref = fv.grad(m, vf, vb)
    for d in range(3):
        assert rel_l2(back_cell(s, g[d]), gref[:, d]) < 1e-14
    corr = ctx.zeros(mesh.nNative)
    mesh.call("fvc_snGrad_correction", *[cellf(s, ctx, gref[:, d]) for d in range(3)], corr)
    cref = fv.snGrad_correction(m, gref)
    assert np.abs(back_face(s, corr) - cref).max() <= 1e-14 * np.abs(cref).max()
    # linear field with its exact boundary values: Gauss gradient exact, corrected snGrad exact on faces between interior cells
    a = np.array([0.7, -1.3, 2.1])
    lin = m.C @ a + 0.4; linb = [p.Cf @ a + 0.4 for p in m.patches]
    glin = fv.grad(m, lin, linb)
    assert np.abs(glin - a).max() < 1e-12
    mesh.call("fvc_grad", cellf(s, ctx, lin), bndf(ctx, linb), *g)
    mesh.call("fvc_snGrad_correction", *g, corr)
    sg = ctx.zeros(mesh.nNative)
    mesh.call("fvc_snGrad", cellf(s, ctx, lin), sg)
    nf = m.Sf / m.magSf[:, None]
    total = back_face(s, sg) + back_face(s, corr)
    assert np.abs(total - nf @ a).max() < 1e-11
    assert np.abs(back_face(s, sg) - nf @ a).max() > 1e-2            # (the uncorrected one is not)
    # the same through the Foam layer (include/ffmFoam.H with mesh.snGradCorrected / laplacianCorrected): fvc::snGrad(vf) and the
    # explicit source of fvm::laplacian(gamma, vf) = -V*div(gamma_f*magSf*correction)
    import ctypes as C, os
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    gam = 1.0 + O.hash_u(75, np.arange(N))
    B = sum(p.size for p in m.patches)
    vbv = np.concatenate(vb)                                # boundary values = fixedValue conditions
    bcs = [np.ones(B), vbv.copy(), np.zeros(B)]
    arr = (dp * 3)(*[a.ctypes.data_as(dp) for a in bcs])
    src, sgF = np.empty(N), np.empty(F)
    h = lambda a_: np.ascontiguousarray(a_, np.float64)
    vfl, gaml = h(vf[cOrd]), h(gam[cOrd])
    lib.b1_corrected_schemes.argtypes = [C.c_void_p] * 3 + [dp, dp, dp, C.POINTER(dp), dp, dp]
    assert lib.b1_corrected_schemes(ctx.h, A.h, mesh.h, vfl.ctypes.data_as(dp), vbv.ctypes.data_as(dp), gaml.ctypes.data_as(dp), arr,
                                    src.ctypes.data_as(dp), sgF.ctypes.data_as(dp)) == 0
    gf, _ = fv.interpolate(m, gam, [gam[p.faceCells] for p in m.patches])
    src_ref = -m.V * fv.surface_integrate(m, gf * m.magSf * cref, [np.zeros(p.size) for p in m.patches])
    back = np.empty(N); back[cOrd] = src
    assert np.abs(back - src_ref).max() <= 1e-13 * np.abs(src_ref).max()
    sgo = np.empty(F); sgo[fOrd] = sgF
    sg_ref = fv.snGrad(m, vf, vb)[0] + cref
    assert np.abs(sgo - sg_ref).max() <= 1e-13 * np.abs(sg_ref).max()
    mesh.close(); A.close()
