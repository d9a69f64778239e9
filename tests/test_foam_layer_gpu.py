"""B1 subset: solver statements in the reference's style (rhoEqn, one YiEqn with relax, UEqn with LUST and the
reconstructed buoyancy/pressure source, rAU and HbyA of pEqn) written against include/ffmFoam.H -- examples/b1_demo.C,
built as libffm_b1demo.so with the host compiler -- run on the device and are compared with the same equations evaluated by
oracle/fv.py + the C solvers.  Bars: fields after a linear solve (tolerance 1e-10) within 1e-8 rel-L2 with identical
iteration counts; quantities without a solve (rho from the diagonal solver, rAU) to rounding."""
import ctypes as C
import os

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


def test_b1_demo_matches_oracle(O, ffm, ctx):
    from oracle import fv, plume
    m = plume.make_mesh((9, 8, 7), h=0.1)
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())
    hu = lambda seed, n: O.hash_u(seed, np.arange(n))
    pl = lambda seed, scale=1.0, shift=0.0: [shift + scale * hu(seed + q, p.size) for q, p in enumerate(m.patches)]
    dt, alphaY, mu = 2e-3, 0.8, 1.8e-5
    rho_old = 1.0 + 0.2 * hu(1, N); rho_now = rho_old * (1 + 0.01 * (hu(2, N) - 0.5))
    phi = 0.02 * (hu(3, F) - 0.5); phib = pl(10, 0.02, -0.01)
    Yi0 = 0.1 + 0.8 * hu(4, N); dEff = 2e-5 * (1 + hu(5, N)); R = 0.5 * (hu(6, N) - 0.5)
    U0 = np.stack([hu(20 + d, N) - 0.5 for d in range(3)])
    p_rgh = 10.0 * (hu(7, N) - 0.5); ghf = -9.81 * m.Cf[:, 1]; ghfb = [-9.81 * p.Cf[:, 1] for p in m.patches]
    bcY = fv.MixedBC(m, f=pl(30), ref=pl(40, 0.5), refGrad=pl(50, 0.1, -0.05))
    bcU = [fv.MixedBC(m, f=pl(60 + 10 * d), ref=pl(90 + 10 * d, 1.0, -0.5)) for d in range(3)]
    p_b = pl(120, 10.0, -5.0)
    # pEqn inputs: compressibility now / old, gh, pRef; p_rgh: fixedFluxPressure on inlet, floor and sides (gradient set by
    # constrainPressure), fixedValue on the top; U fixes its value on inlet and floor
    psi_now = 1.17e-5 * (0.9 + 0.2 * hu(130, N)); psi_old = psi_now * (1 + 1e-3 * (hu(131, N) - 0.5))
    gh = -9.81 * m.C[:, 1]; pRef = 101325.0
    names = [p.name for p in m.patches]
    fluxMask = [np.full(p.size, 0.0 if p.name == "top" else 1.0) for p in m.patches]
    UfixMask = [np.full(p.size, 1.0 if p.name in ("inlet", "floor") else 0.0) for p in m.patches]
    for d in range(3):                                    # consistent with the mask: fixedValue there
        for q, p in enumerate(m.patches):
            if p.name in ("inlet", "floor"):
                bcU[d].f[q] = np.ones(p.size)
    bcP = fv.MixedBC(m, f=[np.full(p.size, 1.0 if p.name == "top" else 0.0) for p in m.patches], ref=pl(140, 2.0, -1.0))
    rdt = 1.0 / dt
    zb = [np.zeros(p.size) for p in m.patches]
    ctl = dict(tolerance=1e-10, relTol=0.0)

    # ------------------------------------------------------------------ the oracle's evaluation of the same statements
    M = fv.fvm_ddt(m, rdt, np.ones(N), np.ones(N), rho_old); M.add_vol(fv.surface_integrate(m, phi, phib))
    d, s = M.solve_system(0)
    rho_new = s / d
    Yb = bcY.values(m, Yi0)
    w = fv.limited_weights(m, "limitedLinear01", phi, Yi0, fv.grad(m, Yi0, Yb), 1.0)
    dEf, dEb = fv.interpolate(m, dEff, [dEff[p.faceCells] for p in m.patches])
    M = fv.fvm_ddt(m, rdt, rho_new, rho_old, Yi0); M += fv.fvm_div(m, phi, phib, w, [bcY]); M -= fv.fvm_laplacian(m, dEf, dEb, [bcY])
    M.add_su(R)
    M.relax(alphaY, Yi0[None])
    d, s = M.solve_system(0)
    Yi_ref, pfY = O.Ldu(N, m.l, m.u).set_coeffs(d, M.upper, M.lower).solve(O.PBICGSTAB, O.DILU, Yi0, s, **ctl)
    Yi_ref = np.maximum(Yi_ref, 0.0)
    Ub = [bcU[c].values(m, U0[c]) for c in range(3)]
    UEqn = fv.fvm_ddt(m, rdt, rho_new, rho_old, U0)
    divU = fv.fvm_div(m, phi, phib, fv.lust_weights(m, phi), bcU)
    divU.add_vol(np.stack([fv.surface_integrate(m, phi * fv.lust_correction(m, phi, fv.grad(m, U0[c], Ub[c])), zb) for c in range(3)]))
    UEqn += divU
    UEqn -= fv.fvm_laplacian(m, np.full(F, mu), [np.full(p.size, mu) for p in m.patches], bcU)
    sgr, _ = fv.snGrad(m, rho_new, [rho_new[p.faceCells] for p in m.patches])
    sgp, sgpb = fv.snGrad(m, p_rgh, p_b)
    rec = fv.reconstruct(m, (-ghf * sgr - sgp) * m.magSf, [-sb * p.magSf for sb, p in zip(sgpb, m.patches)])
    import copy
    Msolve = copy.deepcopy(UEqn); Msolve.add_su(rec.T)
    U_ref, itU = np.empty((3, N)), []
    for c in range(3):
        d, s = Msolve.solve_system(c)
        U_ref[c], pf = O.Ldu(N, m.l, m.u).set_coeffs(d, UEqn.upper, UEqn.lower).solve(O.PBICGSTAB, O.DILU, U0[c], s, **ctl)
        itU.append(pf["nIterations"])
    K_ref = 0.5 * ((U_ref[0] ** 2 + U_ref[1] ** 2) + U_ref[2] ** 2)
    rAU_ref = 1.0 / UEqn.A()
    HbyA_ref = rAU_ref * UEqn.H(U_ref)
    # ---- pEqn.H, one corrector (the association of the reference's tmp<fvMatrix> algebra: one source update per term)
    rhorAU = rho_new * rAU_ref
    rhorAUf, rhorAUfb = fv.interpolate(m, rhorAU, [rhorAU[p.faceCells] for p in m.patches])
    Ubn = [bcU[c].values(m, U_ref[c]) for c in range(3)]
    HbyAb = [[np.where(UfixMask[q] == 1.0, Ubn[c][q], HbyA_ref[c][p.faceCells]) for q, p in enumerate(m.patches)] for c in range(3)]
    rhob = [rho_new[p.faceCells] for p in m.patches]
    phig = -rhorAUf * ghf * sgr * m.magSf
    rhoH = rho_new * HbyA_ref
    flux = sum(fv.interpolate(m, rhoH[c], [rhob[q] * HbyAb[c][q] for q in range(len(m.patches))])[0] * m.Sf[:, c] for c in range(3))
    fluxb = [(rhob[q] * HbyAb[0][q] * p.Sf[:, 0] + rhob[q] * HbyAb[1][q] * p.Sf[:, 1]) + rhob[q] * HbyAb[2][q] * p.Sf[:, 2] for q, p in enumerate(m.patches)]
    rhoU0 = rho_old * U0
    phiCorr = phi - sum((m.weights * rhoU0[c][m.l] + (1 - m.weights) * rhoU0[c][m.u]) * m.Sf[:, c] for c in range(3))
    coeff = 1.0 - np.minimum(np.abs(phiCorr) / (np.abs(phi) + 1e-15), 1.0)
    phiHbyA = flux + rhorAUf * (coeff * rdt * phiCorr) + phig
    grads = [(fluxb[q] - rhob[q] * ((p.Sf[:, 0] * Ubn[0][q] + p.Sf[:, 1] * Ubn[1][q]) + p.Sf[:, 2] * Ubn[2][q])) / (p.magSf * rhorAUfb[q])
             for q, p in enumerate(m.patches)]
    bcp = fv.MixedBC(m, f=bcP.f, ref=bcP.ref, refGrad=[np.where(fluxMask[q] == 1.0, grads[q], bcP.refGrad[q]) for q in range(len(m.patches))])
    E = fv.fvm_ddt(m, rdt, psi_now, psi_old, p_rgh)
    E.add_vol(rdt * (psi_now * rho_new - psi_old * rho_old) * gh)
    E.add_vol(rdt * (psi_now - psi_old) * pRef)
    E.add_vol(fv.surface_integrate(m, phiHbyA, fluxb))
    E -= fv.fvm_laplacian(m, rhorAUf, rhorAUfb, [bcp])
    d, s = E.solve_system(0)
    p_ref, pfP = O.Ldu(N, m.l, m.u).set_coeffs(d, E.upper, None).solve(O.PCG, O.DIC, p_rgh, s, **ctl)
    fl, flb = E.flux(p_ref)
    phi_ref = phiHbyA + fl
    phib_ref = [a + b for a, b in zip(fluxb, flb)]
    rec2 = fv.reconstruct(m, (fl + phig) / rhorAUf, [b / r for b, r in zip(flb, rhorAUfb)])
    Ucorr_ref = HbyA_ref + rAU_ref * rec2.T

    # ------------------------------------------------------------------ the C++ layer on the device
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    h = lambda a: np.ascontiguousarray(a, np.float64)
    cell = lambda a: h(np.asarray(a)[..., cOrd])
    face = lambda a: h(np.asarray(a)[fOrd])
    bnd = lambda lst: h(np.concatenate(lst))
    keep = []
    def P(a):
        keep.append(a)
        return a.ctypes.data_as(dp)
    def PP(arrs):
        arrs = [h(a) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    out = dict(rho=np.empty(N), Yi=np.empty(N), U=np.empty((3, N)), K=np.empty(N), rAU=np.empty(N), HbyA=np.empty((3, N)),
               p=np.empty(N), phi=np.empty(F), phib=np.empty(B), Uc=np.empty((3, N)))
    nit = (C.c_int * 16)()
    lib.b1_demo.restype = C.c_int
    lib.b1_demo.argtypes = ([C.c_void_p] * 3 + [C.c_double] * 2 + [dp] * 5 + [C.POINTER(dp)] + [dp] * 3 + [C.POINTER(dp)] + [C.c_double] + [dp] * 4
                            + [dp] * 3 + [C.c_double] + [C.POINTER(dp)] + [dp] * 2 + [dp] * 10 + [C.POINTER(C.c_int)])
    bcPp = PP([bnd(bcP.f), bnd(bcP.ref), bnd(bcP.refGrad)])
    bcYp = PP([bnd(bcY.f), bnd(bcY.ref), bnd(bcY.refGrad)])
    bcUp = PP([x for d in range(3) for x in (bnd(bcU[d].f), bnd(bcU[d].ref), bnd(bcU[d].refGrad))])
    ctx._ready()
    ns = lib.b1_demo(ctx.h, A.h, mesh.h, dt, alphaY, P(cell(rho_old)), P(cell(rho_now)), P(face(phi)), P(bnd(phib)),
                     P(cell(Yi0)), bcYp, P(cell(dEff)), P(cell(R)), P(cell(U0)), bcUp, mu, P(face(ghf)), P(bnd(ghfb)),
                     P(cell(p_rgh)), P(bnd(p_b)), P(cell(psi_now)), P(cell(psi_old)), P(cell(gh)), pRef, bcPp, P(bnd(fluxMask)), P(bnd(UfixMask)),
                     P(out["rho"]), P(out["Yi"]), P(out["U"]), P(out["K"]), P(out["rAU"]), P(out["HbyA"]),
                     P(out["p"]), P(out["phi"]), P(out["phib"]), P(out["Uc"]), nit)
    assert ns == 6                                   # rho, Yi, Ux, Uy, Uz, p_rgh
    back = lambda a: (lambda o: (o.__setitem__((Ellipsis, cOrd), a), o)[1])(np.empty_like(a))
    assert np.array_equal(back(out["rho"]), rho_new)
    assert list(nit[:6]) == [0, pfY["nIterations"]] + itU + [pfP["nIterations"]]
    assert rel_l2(back(out["Yi"]), Yi_ref) < 1e-8
    for c in range(3):
        assert rel_l2(back(out["U"])[c], U_ref[c]) < 1e-8
        assert rel_l2(back(out["HbyA"])[c], HbyA_ref[c]) < 1e-7
    assert rel_l2(back(out["K"]), K_ref) < 1e-8
    assert rel_l2(back(out["rAU"]), rAU_ref) < 1e-14
    # pEqn: pressure against the scale of its own variation, fluxes and the corrected velocity
    pg = back(out["p"])
    assert np.linalg.norm(pg - p_ref) / np.linalg.norm(p_ref - p_ref.mean()) < 1e-7
    phig_ = np.empty(F); phig_[fOrd] = out["phi"]
    assert rel_l2(phig_, phi_ref) < 1e-8
    assert rel_l2(out["phib"], np.concatenate(phib_ref)) < 1e-8
    for c in range(3):
        assert rel_l2(back(out["Uc"])[c], Ucorr_ref[c]) < 1e-8
    mesh.close(); A.close()


def test_burner_patch_conditions_on_the_device(O, ffm, ctx):
    """flowRateInletVelocity (cases/steckler/0/U:40-54) and totalFlowRateAdvectiveDiffusive
    (cases/steckler/0/C3H8:44-50) of the Foam layer (include/ffmFoam.H: mixedBC::update) against the formulas the oracle's
    steckler case uses -- the ones that reproduce the golden log's first time step (oracle/steckler_case.py:
    update_burner_velocity, bc_specie): U_b = -mdot(t)/sum(rho_b*magSf) * nf on the burner faces, the other faces untouched;
    value fraction 1/(1 + alphaEff_b*deltaCoeffs*magSf/max(|phi_b|, SMALL)), refValue = massFluxFraction."""
    from oracle import plume
    m = plume.make_mesh((9, 8, 7), h=0.1)
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    lib.b1_burner_bcs.restype = C.c_int
    lib.b1_burner_bcs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, dp, C.c_double] + [dp] * 10
    hu = lambda seed, n: O.hash_u(seed, np.arange(n))
    cat = np.concatenate
    mask = cat([np.full(p.size, 1.0 if p.name == "inlet" else 0.0) for p in m.patches])
    rhob = 1.0 + hu(1, B); alphab = 2e-4 * (1 + hu(2, B))
    phib = 0.02 * (hu(3, B) - 0.5); phib[::3] = 0.0                       # some faces without flux: max(|phi|, SMALL)
    phi = 0.02 * (hu(4, F) - 0.5)
    Uc = np.stack([hu(20 + d, N) - 0.5 for d in range(3)]); Yc = hu(30, N)
    table = np.array([0.0, 0.01, 10.0, 0.03, 20.0, 0.03])                  # 0.01 -> 0.03 kg/s over 10 s, read at t = 4 s
    t, mff = 4.0, 0.7
    Ub = np.zeros((3, B)); Yb = np.zeros(B); fo = np.zeros(B)
    P = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(dp)
    ctx.sync()
    rc = lib.b1_burner_bcs(ctx.h, A.h, mesh.h, t, 3, P(table), mff, P(mask), P(rhob), P(alphab), P(phi[fOrd]), P(phib),
                           P(np.ascontiguousarray(Uc[:, cOrd])), P(Yc[cOrd]), P(Ub), P(Yb), P(fo))
    assert rc == 0
    magSf = cat([p.magSf for p in m.patches]); Sf = cat([p.Sf for p in m.patches]); delta = cat([p.deltaCoeffs for p in m.patches])
    fc = cat([p.faceCells for p in m.patches])
    mdot = 0.01 + 0.02 * t / 10.0
    avgU = -mdot / np.sum((rhob * magSf)[mask > 0.5])
    for d in range(3):
        want = np.where(mask > 0.5, avgU * (Sf[:, d] / magSf), 0.0)        # fixed value 0 elsewhere
        assert np.allclose(Ub[d], want, rtol=1e-14, atol=1e-300), d
    assert Ub[1][mask > 0.5].min() > 0                                       # inflow: against the outward normal (0 -1 0)
    f = np.where(mask > 0.5, 1.0 / (1.0 + alphab * delta * magSf / np.maximum(np.abs(phib), 1e-15)), 0.0)
    assert np.allclose(fo, f, rtol=1e-14, atol=0)
    assert np.allclose(Yb, f * mff + (1.0 - f) * Yc[fc], rtol=1e-14, atol=0)
    assert np.any((mask > 0.5) & (phib == 0.0)) and np.all(fo[(mask > 0.5) & (phib == 0.0)] < 1e-8)   # no flux: zero-gradient
    mesh.close(); A.close()


def test_device_array_value_semantics_under_lazy_evaluation(ffm, ctx):
    """include/ffmFoam.H: dField -- copies that share storage until written, expressions evaluated when first needed (one ffm_field_eval
    pass), views of library arrays, trees larger than one program: the value semantics of OpenFOAM's Field algebra must survive all of it.
    examples/b1_demo.C: b1_dfield_semantics runs the cases on the device and compares with host arithmetic bit for bit; it returns the
    number of the first failing check (1 copy independence, 2 pending expression vs overwritten operand, 3 expression used twice,
    4 oversized / right-deep trees, 5 views, 6 scalar-first operators and constants)."""
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    lib.b1_dfield_semantics.restype = C.c_int
    lib.b1_dfield_semantics.argtypes = [C.c_void_p, C.c_int, dp, dp, C.POINTER(C.c_int)]
    rng = np.random.default_rng(11)
    for n in (1, 777, 300001):
        a = rng.standard_normal(n); b = rng.uniform(0.5, 2.0, n) * rng.choice([-1.0, 1.0], n)
        ev = C.c_int()
        os.environ["FFM_FOAM_QUIET"] = "1"
        assert lib.b1_dfield_semantics(ctx.h, n, a.ctypes.data_as(dp), b.ctypes.data_as(dp), C.byref(ev)) == 0, n
