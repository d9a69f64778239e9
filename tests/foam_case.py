"""The inputs of examples/b1_demo.C's b1_demo (one rhoEqn / YiEqn / UEqn / pEqn pass written against include/ffmFoam.H) on a
plume mesh, and a runner that executes it on the whole mesh or on ONE RANK's sub-domain of a decomposition (ghost-cell form:
firefoam-dev_amd/decompose.py SubDomain) -- shared by tests/test_foam_layer_decomposed_gpu.py and its worker."""
import ctypes as C
import os

import numpy as np


def inputs(O, m):
    from oracle import fv
    N, F = m.nCells, m.nFaces
    hu = lambda seed, n: O.hash_u(seed, np.arange(n))
    pl = lambda seed, scale=1.0, shift=0.0: [shift + scale * hu(seed + q, p.size) for q, p in enumerate(m.patches)]
    I = dict(dt=2e-3, alphaY=0.8, mu=1.8e-5, pRef=101325.0)
    I["rho_old"] = 1.0 + 0.2 * hu(1, N); I["rho_now"] = I["rho_old"] * (1 + 0.01 * (hu(2, N) - 0.5))
    I["phi"] = 0.02 * (hu(3, F) - 0.5); I["phib"] = pl(10, 0.02, -0.01)
    I["Yi0"] = 0.1 + 0.8 * hu(4, N); I["dEff"] = 2e-5 * (1 + hu(5, N)); I["R"] = 0.5 * (hu(6, N) - 0.5)
    I["U0"] = np.stack([hu(20 + d, N) - 0.5 for d in range(3)])
    I["p_rgh"] = 10.0 * (hu(7, N) - 0.5); I["ghf"] = -9.81 * m.Cf[:, 1]; I["ghfb"] = [-9.81 * p.Cf[:, 1] for p in m.patches]
    bcY = fv.MixedBC(m, f=pl(30), ref=pl(40, 0.5), refGrad=pl(50, 0.1, -0.05))
    bcU = [fv.MixedBC(m, f=pl(60 + 10 * d), ref=pl(90 + 10 * d, 1.0, -0.5)) for d in range(3)]
    I["p_b"] = pl(120, 10.0, -5.0)
    I["psi_now"] = 1.17e-5 * (0.9 + 0.2 * hu(130, N)); I["psi_old"] = I["psi_now"] * (1 + 1e-3 * (hu(131, N) - 0.5))
    I["gh"] = -9.81 * m.C[:, 1]
    I["fluxMask"] = [np.full(p.size, 0.0 if p.name == "top" else 1.0) for p in m.patches]
    I["UfixMask"] = [np.full(p.size, 1.0 if p.name in ("inlet", "floor") else 0.0) for p in m.patches]
    for d in range(3):
        for q, p in enumerate(m.patches):
            if p.name in ("inlet", "floor"):
                bcU[d].f[q] = np.ones(p.size)
    bcP = fv.MixedBC(m, f=[np.full(p.size, 1.0 if p.name == "top" else 0.0) for p in m.patches], ref=pl(140, 2.0, -1.0))
    I["bcY"] = [bcY.f, bcY.ref, bcY.refGrad]
    I["bcU"] = [x for d in range(3) for x in (bcU[d].f, bcU[d].ref, bcU[d].refGrad)]
    I["bcP"] = [bcP.f, bcP.ref, bcP.refGrad]
    return I


def run_b1_demo(ffm, ctx, m, I, sub=None, part=None):
    """sub None: the whole mesh on this context.  Otherwise: the sub-domain `sub` of the decomposition `part` (cell -> rank).
    Returns {field: values on the OWNED cells}, the owned cells' global labels and the iteration counts."""
    N, F = m.nCells, m.nFaces
    if sub is None:
        gcell = np.arange(N); nOwn, nGhost = N, 0
        l, u, gface, sign = m.l, m.u, np.arange(F), np.ones(F)
        pmask = [np.ones(p.size, bool) for p in m.patches]
        g2l = np.arange(N)
    else:
        gcell, nOwn, nGhost = sub.gcell, sub.nOwned, sub.nGhost
        l, u, gface = sub.l, sub.u, sub.gface
        sign = np.where(sub.flip.astype(bool), -1.0, 1.0)
        pmask = [part[p.faceCells] == sub.rank for p in m.patches]
        g2l = np.full(N, -1, np.int64); g2l[gcell[:nOwn]] = np.arange(nOwn)
    nLoc = nOwn + nGhost
    cOrd, fOrd = ffm.renumber_levels(nOwn, l, u, nGhost=nGhost)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(nLoc, l, u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, nOwn, l2, u2, nGhost=nGhost)
    if sub is not None:
        A.set_ghost_exchange(sub.nbrRank, sub.sendCount, oldToNew[sub.sendCells], sub.recvCount, tags=sub.tags, globalCells=N)
    flipW = sign < 0
    wgt = np.where(flipW, 1.0 - m.weights[gface], m.weights[gface])
    patches = [(oldToNew[g2l[p.faceCells[k]]].astype(np.int32), p.Sf[k].T.copy(), p.deltaCoeffs[k]) for p, k in zip(m.patches, pmask)]
    mesh = ffm.fvMesh(A, m.V[gcell][cOrd], m.C[gcell][cOrd].T.copy(), (m.Sf[gface] * sign[:, None])[fOrd].T.copy(), m.magSf[gface][fOrd],
                      wgt[fOrd], m.deltaCoeffs[gface][fOrd], patches)
    mesh.set_face_centres(m.Cf[gface][fOrd].T.copy())
    B = sum(int(k.sum()) for k in pmask)
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    h = lambda a: np.ascontiguousarray(a, np.float64)
    cell = lambda a: h(np.asarray(a)[..., gcell][..., cOrd])
    face = lambda a, s=None: h((np.asarray(a)[gface] * (sign if s else 1.0))[fOrd])
    bnd = lambda lst: h(np.concatenate([np.asarray(x)[k] for x, k in zip(lst, pmask)])) if B else np.zeros(1)
    keep = []

    def P(a):
        keep.append(a)
        return a.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [h(a) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    Fl = len(l)
    out = dict(rho=np.empty(nLoc), Yi=np.empty(nLoc), U=np.empty((3, nLoc)), K=np.empty(nLoc), rAU=np.empty(nLoc), HbyA=np.empty((3, nLoc)),
               p=np.empty(nLoc), phi=np.empty(max(Fl, 1)), phib=np.empty(max(B, 1)), Uc=np.empty((3, nLoc)))
    nit = (C.c_int * 16)()
    lib.b1_demo.restype = C.c_int
    lib.b1_demo.argtypes = ([C.c_void_p] * 3 + [C.c_double] * 2 + [dp] * 5 + [C.POINTER(dp)] + [dp] * 3 + [C.POINTER(dp)] + [C.c_double] + [dp] * 4
                            + [dp] * 3 + [C.c_double] + [C.POINTER(dp)] + [dp] * 2 + [dp] * 10 + [C.POINTER(C.c_int)])
    bcPp = PP([bnd(x) for x in I["bcP"]]); bcYp = PP([bnd(x) for x in I["bcY"]]); bcUp = PP([bnd(x) for x in I["bcU"]])
    ctx._ready()
    ns = lib.b1_demo(ctx.h, A.h, mesh.h, I["dt"], I["alphaY"], P(cell(I["rho_old"])), P(cell(I["rho_now"])), P(face(I["phi"], True)), P(bnd(I["phib"])),
                     P(cell(I["Yi0"])), bcYp, P(cell(I["dEff"])), P(cell(I["R"])), P(cell(I["U0"])), bcUp, I["mu"], P(face(I["ghf"])), P(bnd(I["ghfb"])),
                     P(cell(I["p_rgh"])), P(bnd(I["p_b"])), P(cell(I["psi_now"])), P(cell(I["psi_old"])), P(cell(I["gh"])), I["pRef"], bcPp,
                     P(bnd(I["fluxMask"])), P(bnd(I["UfixMask"])),
                     P(out["rho"]), P(out["Yi"]), P(out["U"]), P(out["K"]), P(out["rAU"]), P(out["HbyA"]),
                     P(out["p"]), P(out["phi"]), P(out["phib"]), P(out["Uc"]), nit)
    assert ns == 6
    res = {}
    for k in ("rho", "Yi", "U", "K", "rAU", "HbyA", "p", "Uc"):
        a = out[k]; o = np.empty_like(a); o[..., cOrd] = a
        res[k] = o[..., :nOwn]
    mesh.close(); A.close()
    return res, gcell[:nOwn].copy(), list(nit[:6])


def run_b1_fvdom(ffm, ctx, m, T, Tb, E, emis, sub=None, part=None, nPhi=2, nTheta=2, maxIter=3, tolerance=0.0, scheme=0, a=0.3, IiTol=1e-12, nCalls=2):
    """b1_fvdom of examples/b1_demo.C (the fvDOM handle with its iteration and grey-diffusive walls) on the whole mesh or on one rank's
    sub-domain.  T, E: global cell fields; Tb, emis: per patch.  Returns (I [nRay][owned], G [owned], qin per owned boundary face as a
    dict patch -> (global face positions, values)), the owned cells' global labels and the iterations of every call."""
    N, F = m.nCells, m.nFaces
    if sub is None:
        gcell = np.arange(N); nOwn, nGhost = N, 0
        l, u, gface, sign = m.l, m.u, np.arange(F), np.ones(F)
        pmask = [np.ones(p.size, bool) for p in m.patches]
        g2l = np.arange(N)
    else:
        gcell, nOwn, nGhost = sub.gcell, sub.nOwned, sub.nGhost
        l, u, gface = sub.l, sub.u, sub.gface
        sign = np.where(sub.flip.astype(bool), -1.0, 1.0)
        pmask = [part[p.faceCells] == sub.rank for p in m.patches]
        g2l = np.full(N, -1, np.int64); g2l[gcell[:nOwn]] = np.arange(nOwn)
    nLoc = nOwn + nGhost
    cOrd, fOrd = ffm.renumber_levels(nOwn, l, u, nGhost=nGhost)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(nLoc, l, u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, nOwn, l2, u2, nGhost=nGhost)
    if sub is not None:
        A.set_ghost_exchange(sub.nbrRank, sub.sendCount, oldToNew[sub.sendCells], sub.recvCount, tags=sub.tags, globalCells=N)
    wgt = np.where(sign < 0, 1.0 - m.weights[gface], m.weights[gface])
    patches = [(oldToNew[g2l[p.faceCells[k]]].astype(np.int32), p.Sf[k].T.copy(), p.deltaCoeffs[k]) for p, k in zip(m.patches, pmask)]
    mesh = ffm.fvMesh(A, m.V[gcell][cOrd], m.C[gcell][cOrd].T.copy(), (m.Sf[gface] * sign[:, None])[fOrd].T.copy(), m.magSf[gface][fOrd],
                      wgt[fOrd], m.deltaCoeffs[gface][fOrd], patches)
    mesh.set_face_centres(m.Cf[gface][fOrd].T.copy())
    B = sum(int(k.sum()) for k in pmask)
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    h = lambda x: np.ascontiguousarray(x, np.float64)
    cell = lambda x: h(np.asarray(x)[gcell][cOrd])
    bnd = lambda lst: h(np.concatenate([np.asarray(x)[k] for x, k in zip(lst, pmask)])) if B else np.zeros(1)
    solD = getattr(m, "solutionD", (1, 1, 1))
    nRay = 4 * nPhi * nTheta if min(solD) > 0 else 4 * nPhi
    Tc, Tbb, Ec, emb = cell(T), bnd(Tb), cell(E), bnd(emis)
    IOut, GOut = np.empty((nRay, nLoc)), np.empty(nLoc)
    qin, qem, qr = np.empty(max(B, 1)), np.empty(max(B, 1)), np.empty(max(B, 1))
    iters, nSolves = (C.c_int * nCalls)(), C.c_int()
    lib.b1_fvdom.restype = C.c_int
    lib.b1_fvdom.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_double, C.c_int, C.c_double, C.c_double] + [dp] * 4 + [C.c_int] + [dp] * 5 + [C.POINTER(C.c_int)] * 2
    os.environ["FFM_FOAM_QUIET"] = "1"
    ctx._ready()
    P = lambda x: x.ctypes.data_as(dp)
    n = lib.b1_fvdom(ctx.h, A.h, mesh.h, sum(1 << d for d in range(3) if solD[d] < 0), nPhi, nTheta, maxIter, tolerance, scheme, a, IiTol,
                     P(Tc), P(Tbb), P(Ec), P(emb), nCalls, P(IOut), P(GOut), P(qin), P(qem), P(qr), iters, C.byref(nSolves))
    assert n == nRay
    inv = np.empty(nLoc, np.int64); inv[cOrd] = np.arange(nLoc)
    I = IOut[:, inv][:, :nOwn]; G = GOut[inv][:nOwn]
    # qin per patch on this rank's faces: (positions inside the global patch, values)
    qp, off = {}, 0
    for p, k in zip(m.patches, pmask):
        cnt = int(k.sum())
        qp[p.name] = (np.nonzero(k)[0], qin[off:off + cnt].copy()); off += cnt
    mesh.close(); A.close()
    return (I, G, qp), gcell[:nOwn].copy(), list(iters)


def fvdom_inputs(m):
    """a flame-like temperature / emission field and wall emissivities for run_b1_fvdom"""
    x, y = m.C[:, 0], m.C[:, 1]
    T = 500.0 + 600.0 * np.exp(-((x - x.mean()) ** 2 + (y - 0.3 * y.max()) ** 2) / 0.05)
    E = 2.0e5 * np.exp(-((x - x.mean()) ** 2 + (y - 0.3 * y.max()) ** 2) / 0.03)
    Tb = [np.full(p.size, 900.0 if p.name == "inlet" else 320.0) for p in m.patches]
    emis = [np.full(p.size, e) for p, e in zip(m.patches, (0.3, 0.85, 1.0, 0.55))]
    return T, Tb, E, emis
