"""SURVEY 8(f) N1 on the device: the fvDOM handle of include/fireFoamHandles.H (fvDOM::calculate's iteration, fvDOM.C:547-584, with the
wall condition greyDiffusiveRadiationMixedFvPatchScalarField::updateCoeffs as the kernel ffm_fvdom_wall_coeffs_d, the ray equations
through the Foam layer's fvm::div / fvm::Sp and the LDU solvers) against oracle/fvdom.py: same mesh, temperature, emission, wall
emissivities, ray set, iteration controls and div(Ji,Ii_h) scheme -- every ray's intensity, G, the wall fluxes qin / qem / qr, the
number of iterations of every call and the number of ray solves.  Cases: the 3-D box with reflecting walls; BASELINE config 5's
selections (cases/wallFireSpread2D/constant/radiationProperties:28-46: nPhi 2, nTheta 2 -> 8 rays in the x-y plane of a 2-D mesh, maxIter 5,
convergence 1e-3; system/fvSchemes:66 Gauss linearUpwind; one wall patch with emissivity < 1 as `emissivityMode solidRadiation` gives
it, 0/IDefault:21-35) over several radiation->correct() calls; the steckler selections (maxIter 1, emissivity 1, upwind).  The
emissivity-1 / maxIter-1 path is pinned by the golden log (tests/test_steckler_case_gpu.py); the rest is unpinned by reference data."""
import ctypes as C
import os

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,empty,nPhi,nTheta,maxIter,tol,scheme,nCalls,emis", [
    ((7, 8, 6), (), 2, 2, 4, 1e-6, "upwind", 2, (0.3, 0.85, 1.0, 0.55)),
    ((14, 18, 1), ("zmin", "zmax"), 2, 2, 5, 1e-3, "linearUpwind", 3, (0.17, 1.0, 1.0, 1.0)),
    ((14, 18, 1), ("zmin", "zmax"), 2, 2, 5, 1e-3, "linearUpwind", 3, (0.85, 1.0, 1.0, 1.0)),
    ((6, 7, 5), (), 2, 4, 1, 0.0, "upwind", 1, (1.0, 1.0, 1.0, 1.0))])
def test_fvdom_iteration_with_reflecting_walls(O, ffm, ctx, shape, empty, nPhi, nTheta, maxIter, tol, scheme, nCalls, emis):
    from oracle import plume, fvdom
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    m = plume.make_mesh(shape, empty=empty)
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())
    solD = m.solutionD
    x, y = m.C[:, 0], m.C[:, 1]
    T = 500.0 + 600.0 * np.exp(-((x - x.mean()) ** 2 + (y - 0.3 * y.max()) ** 2) / 0.02)         # a flame-like hot spot
    Tb = [np.full(p.size, 900.0 if p.name == "inlet" else 320.0) for p in m.patches]
    E = 2.0e5 * np.exp(-((x - x.mean()) ** 2 + (y - 0.3 * y.max()) ** 2) / 0.01)
    a = 0.3
    emissivity = [np.full(p.size, e) for p, e in zip(m.patches, emis)]

    def solve(name, d, upper, lower, s, psi0):
        Ao = O.Ldu(N, m.l, m.u).set_coeffs(d, upper, lower)
        return Ao.solve(O.PBICGSTAB, O.DILU, psi0, s, tolerance=1e-9, relTol=0.0, maxIter=1000)
    ref = fvdom.FvDOM(m, nPhi, nTheta, solve, maxIter=maxIter, tolerance=tol, divScheme=scheme, solutionD=solD, emissivity=emissivity)
    its_ref = []
    for _ in range(nCalls):
        ref.calculate(T, Tb, a, E); its_ref.append(ref.nIterations)
    nRay = len(ref.rays)
    dp = C.POINTER(C.c_double)
    P = lambda v: np.ascontiguousarray(v, np.float64)
    Tc, Tbb, Ec, emb = P(T[cOrd]), P(np.concatenate(Tb)), P(E[cOrd]), P(np.concatenate(emissivity))
    IOut, GOut = np.empty((nRay, N)), np.empty(N)
    qin, qem, qr = np.empty(B), np.empty(B), np.empty(B)
    iters, nSolves = (C.c_int * nCalls)(), C.c_int()
    lib.b1_fvdom.restype = C.c_int
    lib.b1_fvdom.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_double, C.c_int, C.c_double, C.c_double] + [dp] * 4 + [C.c_int] + [dp] * 5 + [C.POINTER(C.c_int)] * 2
    os.environ["FFM_FOAM_QUIET"] = "1"
    ed = sum(1 << d for d in range(3) if solD[d] < 0)
    n = lib.b1_fvdom(ctx.h, A.h, mesh.h, ed, nPhi, nTheta, maxIter, tol, 5 if scheme == "linearUpwind" else 0, a, 1e-9,
                     Tc.ctypes.data_as(dp), Tbb.ctypes.data_as(dp), Ec.ctypes.data_as(dp), emb.ctypes.data_as(dp), nCalls,
                     IOut.ctypes.data_as(dp), GOut.ctypes.data_as(dp), qin.ctypes.data_as(dp), qem.ctypes.data_as(dp), qr.ctypes.data_as(dp), iters, C.byref(nSolves))
    assert n == nRay == (4 * nPhi if empty else 4 * nPhi * nTheta)
    assert list(iters) == its_ref, (list(iters), its_ref)
    assert nSolves.value == len(ref.log)
    inv = np.empty(N, np.int64); inv[cOrd] = np.arange(N)
    for i in range(nRay):
        assert rel_l2(IOut[i][inv], ref.I[i]) < 1e-8, (i, rel_l2(IOut[i][inv], ref.I[i]))
    assert rel_l2(GOut[inv], ref.G) < 1e-8
    for name, got, want in (("qin", qin, ref.qin), ("qem", qem, ref.qem), ("qr", qr, ref.qr)):
        w = np.concatenate(want)
        assert np.abs(got - w).max() <= 1e-8 * max(np.abs(w).max(), 1e-300), name
    if min(emis) < 1.0:
        assert max(its_ref) > 1                                                    # the walls' reflection needed the iteration
    A.close()
