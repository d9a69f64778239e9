"""The reference's own solver/rhoEqn.H, UEqn.H, YEEqn.H and pEqn.H, compiled unchanged against include/ffmFoam.H +
include/fireFoamHandles.H (examples/fireFoam_snippets.C, built where the reference is mounted; the library travels to the GPU
box), run one time step of the synthetic plume case on the device and are compared with the oracle's time step
(oracle/plume.py with GeometricField boundary-value semantics for p_rgh: stored_bc).  Two consecutive steps: the second starts
from the first's results (non-zero flow, both correctors' flux updates, dynamic inletOutlet / total-pressure conditions).
Bars: 1e-8 rel-L2 per field with identical iteration counts, as for the compiled plume driver (tests/test_plume_gpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from common import rel_l2

pytestmark = pytest.mark.gpu


class SnippetCase(C.Structure):
    dp, dpp = C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_double))
    _fields_ = ([("deltaT", C.c_double)] + [(k, C.c_double) for k in ("RR", "Cp", "Tref", "pRef", "mu", "Pr", "sO2", "HC", "tau")]
                + [(k, C.c_int) for k in ("nSpecies", "inertIndex", "fuelIndex", "o2Index")]
                + [("W", dp), ("nu", dp)]
                + [("rho", dp), ("U", dp), ("p", dp), ("p_rgh", dp), ("h", dp), ("Y", dpp)]
                + [("K", dp), ("dpdt", dp), ("phiF", dp), ("phiB", dp)]
                + [("gh", dp), ("ghfF", dp), ("ghfB", dp)]
                + [("fU", dp), ("refU", dp), ("fixesU", dp)]
                + [("fY", dp), ("refY", dpp), ("fH", dp), ("refH", dp)]
                + [("fluxMaskP", dp), ("totalMaskP", dp), ("ph_rgh_b", dp), ("p_rghB", dp)]
                + [("rhoOut", dp), ("UOut", dp), ("pOut", dp), ("p_rghOut", dp), ("hOut", dp), ("YOut", dpp), ("TOut", dp), ("KOut", dp)]
                + [("dpdtOut", dp), ("phiOutF", dp), ("phiOutB", dp), ("p_rghBOut", dp), ("nIterOut", C.POINTER(C.c_int)), ("nIterCap", C.c_int)]
                + [("radiationFreq", C.c_int), ("kAbs", C.c_double), ("sigmaSB", C.c_double), ("dAve", dp), ("omega", dp), ("GOut", dp)]
                + [("psiB", dp), ("resOut", dp)]
                + [("adjustTimeStep", C.c_int), ("maxCo", C.c_double), ("maxDeltaT", C.c_double), ("dtOut", dp), ("emptyDirections", C.c_int)]
                + [("wallFireSelection", C.c_int), ("gamg", C.c_void_p)]
                + [("pyro", C.c_void_p), ("pyroCols", C.c_int), ("pyroMap", C.POINTER(C.c_int)), ("pyroQin", dp)]
                + [(k, C.c_double) for k in ("pyroEmissivity", "pyroAbsorptivity", "pyroHocSolid", "pyroQFuel")]
                + [(k, C.c_int) for k in ("fvdomReal", "radNPhi", "radNTheta", "radMaxIter", "radDivScheme")]
                + [(k, C.c_double) for k in ("radTolerance", "radEhrr1", "radEhrr2")] + [("radMlrMask", dp), ("radMlrMask2", dp), ("radEmissivity", dp)]
                + [("qinOut", dp), ("radItersOut", C.POINTER(C.c_int)), ("pyroInStep", C.c_int), ("pyroMaxDi", C.c_double)])


@pytest.mark.parametrize("shape,empty", [((10, 12, 9), ()), ((1, 24, 20), ("xmin", "xmax"))])
def test_reference_equation_files_run_a_time_step(O, ffm, ctx, shape, empty):
    """second case: a 2-D mesh, one cell thick with `empty` x-patches (the shape of cases/wallFireSpread2D, BASELINE config 5)"""
    from oracle import plume
    newPlume = lambda: plume.Plume(shape, mesh=plume.make_mesh(shape, empty=empty))
    so = os.path.join(os.path.dirname(ffm.libpath()), "libffm_refsnippets.so")
    if not os.path.exists(so):
        pytest.skip("libffm_refsnippets.so not built (needs /root/reference at build time)")
    lib = C.CDLL(so)
    lib.firefoam_snippets_step.restype = C.c_int
    lib.firefoam_snippets_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]

    ref = newPlume()
    ref.stored_bc = True
    m = ref.m
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())

    dp = C.POINTER(C.c_double)
    h = lambda a: np.ascontiguousarray(a, np.float64)
    keep = []

    def P(a):
        a = h(a); keep.append(a)
        return a.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [h(a) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    cell = lambda a: np.asarray(a)[..., cOrd]
    face = lambda a: np.asarray(a)[fOrd]
    bnd = lambda lst: np.concatenate(lst)
    per = lambda fn: bnd([fn(p) for p in m.patches])
    is_open = lambda p: p.name not in ("inlet", "floor")
    # boundary conditions in mixed form; f = -1: value fraction 1 - pos0(phi_b) (inletOutlet, tangential pressureInletOutletVelocity)
    fU = np.concatenate([per(lambda p, d=d: np.where(np.abs(p.Sf[:, d]) > 0, 0.0, -1.0) if is_open(p) else np.ones(p.size)) for d in range(3)])
    refU = np.concatenate([per(lambda p, d=d: np.full(p.size, plume.U_IN if (p.name == "inlet" and d == 1) else 0.0)) for d in range(3)])
    fixesU = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
    fY = per(lambda p: np.full(p.size, 1.0 if p.name == "inlet" else (0.0 if p.name == "floor" else -1.0)))
    refY = [per(lambda p, i=i: np.full(p.size, plume.Y_IN[i] if p.name == "inlet" else (plume.Y_AMB[i] if is_open(p) else 0.0))) for i in range(5)]
    fH = per(lambda p: np.full(p.size, -1.0 if is_open(p) else 1.0))
    refH = per(lambda p: np.full(p.size, plume.CP * (plume.T_IN - plume.TREF) if p.name == "inlet" else 0.0))
    fluxMask = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
    totalMask = per(lambda p: np.full(p.size, 1.0 if is_open(p) else 0.0))
    ghfb = bnd([p.Cf @ plume.G - ref.ghRef for p in m.patches])

    # ---- start-up: solver/phrghEqn.H (included unchanged) against the oracle's hydrostatic initialisation
    lib.firefoam_snippets_hydrostatic.restype = C.c_int
    lib.firefoam_snippets_hydrostatic.argtypes = lib.firefoam_snippets_step.argtypes
    Yamb = [np.full(N, plume.Y_AMB[i]) for i in range(5)]
    psi0 = 1.0 / (plume.RR * plume.TREF * sum(plume.Y_AMB[i] / plume.WMOL[i] for i in range(5)))
    o = dict(rho=np.empty(N), p=np.empty(N), p_rgh=np.empty(N), p_rghB=np.empty(B))
    dummyN, dummy3, dummyF, dummyB = np.zeros(N), np.zeros((3, N)), np.zeros(F), np.zeros(B)
    nit = (C.c_int * 32)()
    topMask = per(lambda p: np.full(p.size, 1.0 if p.name == "top" else 0.0))
    cs = SnippetCase(
        deltaT=ref.dt, RR=plume.RR, Cp=plume.CP, Tref=plume.TREF, pRef=plume.PREF, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC,
        tau=plume.TAU, nSpecies=5, inertIndex=plume.INERT, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
        rho=P(np.full(N, psi0 * plume.PREF)), U=P(dummy3), p=P(np.full(N, plume.PREF)), p_rgh=P(dummyN), h=P(dummyN), Y=PP(Yamb),
        K=P(dummyN), dpdt=P(dummyN), phiF=P(dummyF), phiB=P(dummyB), gh=P(cell(ref.gh)), ghfF=P(face(ref.ghf)), ghfB=P(ghfb),
        fU=P(fU), refU=P(np.zeros_like(refU)), fixesU=P(fixesU), fY=P(fY), refY=PP(refY), fH=P(fH), refH=P(refH),      # U_b = 0 at t = 0
        fluxMaskP=P(1.0 - topMask), totalMaskP=P(topMask), ph_rgh_b=P(dummyB), p_rghB=P(dummyB),
        rhoOut=o["rho"].ctypes.data_as(dp), pOut=o["p"].ctypes.data_as(dp), p_rghOut=o["p_rgh"].ctypes.data_as(dp),
        p_rghBOut=o["p_rghB"].ctypes.data_as(dp), nIterOut=nit, nIterCap=32)
    os.environ["FFM_FOAM_QUIET"] = "1"
    n = lib.firefoam_snippets_hydrostatic(ctx.h, A.h, mesh.h, C.byref(cs))
    init = newPlume()
    its0 = [pf["nIterations"] for _, pf in init.sol.log]
    inv0 = np.empty(N, np.int64); inv0[cOrd] = np.arange(N)
    assert list(nit[:n]) == its0 and n == 5, (list(nit[:n]), its0)
    assert rel_l2(o["p_rgh"][inv0], init.ph_rgh) < 1e-8 and rel_l2(o["p"][inv0], init.p) < 1e-13 and rel_l2(o["rho"][inv0], init.rho) < 1e-13
    assert np.abs(o["p_rghB"] - bnd(init.ph_rgh_b)).max() <= 1e-8 * np.abs(bnd(init.ph_rgh_b)).max() + 1e-14

    def new_out():
        return dict(rho=np.empty(N), U=np.empty((3, N)), p=np.empty(N), p_rgh=np.empty(N), h=np.empty(N), Y=[np.empty(N) for _ in range(5)],
                    T=np.empty(N), K=np.empty(N), dpdt=np.empty(N), phi=np.empty(F), phib=np.empty(B), p_rghB=np.empty(B))

    def case_of(ref, out, nit):
        cs = SnippetCase(
            deltaT=ref.dt, RR=plume.RR, Cp=plume.CP, Tref=plume.TREF, pRef=plume.PREF, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC,
            tau=plume.TAU, nSpecies=5, inertIndex=plume.INERT, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
            rho=P(cell(ref.rho)), U=P(cell(ref.U)), p=P(cell(ref.p)), p_rgh=P(cell(ref.p_rgh)), h=P(cell(ref.h)), Y=PP([cell(ref.Y[i]) for i in range(5)]),
            K=P(cell(ref.K)), dpdt=P(cell(ref.dpdt)), phiF=P(face(ref.phi)), phiB=P(bnd(ref.phib)),
            gh=P(cell(ref.gh)), ghfF=P(face(ref.ghf)), ghfB=P(ghfb),
            fU=P(fU), refU=P(refU), fixesU=P(fixesU), fY=P(fY), refY=PP(refY), fH=P(fH), refH=P(refH),
            fluxMaskP=P(fluxMask), totalMaskP=P(totalMask), ph_rgh_b=P(bnd(ref.ph_rgh_b)), p_rghB=P(bnd(ref.p_rgh_b)),
            rhoOut=out["rho"].ctypes.data_as(dp), UOut=out["U"].ctypes.data_as(dp), pOut=out["p"].ctypes.data_as(dp),
            p_rghOut=out["p_rgh"].ctypes.data_as(dp), hOut=out["h"].ctypes.data_as(dp),
            TOut=out["T"].ctypes.data_as(dp), KOut=out["K"].ctypes.data_as(dp), dpdtOut=out["dpdt"].ctypes.data_as(dp),
            phiOutF=out["phi"].ctypes.data_as(dp), phiOutB=out["phib"].ctypes.data_as(dp), p_rghBOut=out["p_rghB"].ctypes.data_as(dp),
            nIterOut=nit, nIterCap=32)
        yout = (dp * 5)(*[a.ctypes.data_as(dp) for a in out["Y"]]); keep.append(yout)
        cs.YOut = yout
        cs.emptyDirections = sum(1 << d for d in range(3) if m.solutionD[d] < 0)      # 2-D case: the x component is not solved
        return cs

    # ---- the state stays on the device: create once, advance three steps, download once
    lib.firefoam_snippets_create.restype = C.c_void_p
    lib.firefoam_snippets_create.argtypes = lib.firefoam_snippets_step.argtypes
    lib.firefoam_snippets_advance.restype = C.c_int
    lib.firefoam_snippets_advance.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
    lib.firefoam_snippets_destroy.argtypes = [C.c_void_p]
    ref3 = newPlume(); ref3.stored_bc = True
    out3, nit3 = new_out(), (C.c_int * 32)()
    cs3 = case_of(ref3, out3, nit3)
    solver = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(cs3))
    for k in range(3):
        n3 = lib.firefoam_snippets_advance(solver, C.byref(cs3), 1 if k == 2 else 0)
        ref3.step()
        assert list(nit3[:n3]) == [pf["nIterations"] for _, pf in ref3.sol.log], k
    lib.firefoam_snippets_destroy(solver)
    f3 = ref3.fields()
    for name, a in (("rho", out3["rho"]), ("T", out3["T"]), ("Ux", out3["U"][0]), ("Uy", out3["U"][1]), ("Uz", out3["U"][2]), ("C3H8", out3["Y"][2])):
        assert rel_l2(a[inv0], f3[name]) < 1e-5, (name, rel_l2(a[inv0], f3[name]))      # three steps; the common limiter on round-off-uniform fields (tests/test_plume_gpu.py header)

    # ---- the body of the reference's time loop with its time-step control: solidRegionDiffusionNo.H and setMultiRegionDeltaT.H
    # (the reference's, unchanged) between this layer's compressibleCourantNo.H and setDeltaT.H; maxCo 0.3 lets deltaT grow by
    # 1.2 x 1.2 per step from 1 ms until the Courant number of the plume limits it
    lib.firefoam_snippets_time_step.restype = C.c_int
    lib.firefoam_snippets_time_step.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
    refT = newPlume(); refT.stored_bc = True; refT.set_time_controls(0.3, 0.05)
    outT, nitT, dtOut = new_out(), (C.c_int * 32)(), np.zeros(1)
    csT = case_of(refT, outT, nitT)
    csT.adjustTimeStep = 1; csT.maxCo = 0.3; csT.maxDeltaT = 0.05; csT.dtOut = dtOut.ctypes.data_as(dp)
    solverT = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(csT))
    dts = []
    for k in range(5):
        nT = lib.firefoam_snippets_time_step(solverT, C.byref(csT), 1 if k == 4 else 0)
        refT.step()
        dts.append(dtOut[0])
        assert abs(dtOut[0] - refT.dt) <= 1e-6 * refT.dt, (k, dtOut[0], refT.dt)       # deltaT follows the Courant number of fields that agree to ~1e-7
        assert list(nitT[:nT]) == [pf["nIterations"] for _, pf in refT.sol.log], k
    lib.firefoam_snippets_destroy(solverT)
    assert dts[0] > 1.43e-3 and max(dts) > dts[0] and any(abs(b / a - 1.44) > 1e-6 for a, b in zip(dts, dts[1:]))   # grew, then was limited
    fT = refT.fields()
    for name, a in (("rho", outT["rho"]), ("T", outT["T"]), ("Uy", outT["U"][1]), ("O2", outT["Y"][0])):
        assert rel_l2(a[inv0], fT[name]) < 1e-5, (name, rel_l2(a[inv0], fT[name]))

    # ---- with the fvDOM stand-in as the radiation handle: radiation->correct() of solver/YEEqn.H:80 solves the 32 rays
    refR = newPlume(); refR.stored_bc = True; refR.set_radiation(solverFreq=1, ordered=False)      # the handle solves every ray iteratively
    outR, nitR, G = new_out(), (C.c_int * 64)(), np.empty(N)
    csR = case_of(refR, outR, nitR)
    csR.nIterCap = 64; csR.radiationFreq = 1; csR.kAbs = plume.K_ABS; csR.sigmaSB = plume.SIGMA_SB
    csR.dAve = P(np.concatenate([d for d, _ in refR.rays])); csR.omega = P(np.array([o for _, o in refR.rays])); csR.GOut = G.ctypes.data_as(dp)
    nR = lib.firefoam_snippets_step(ctx.h, A.h, mesh.h, C.byref(csR))
    refR.step()
    assert list(nitR[:nR]) == [pf["nIterations"] for _, pf in refR.sol.log] and nR == (9 if empty else 10) + 32
    assert rel_l2(G[inv0], refR.G) < 1e-10 and rel_l2(outR["T"][inv0], refR.T) < 1e-10

    if empty:
        # ---- BASELINE config 5's selection on this 2-D mesh (cases/wallFireSpread2D/system): p_rgh by GAMG + GaussSeidel on the
        # mesh's cached faceAreaPair agglomeration, momentum convection by filteredLinear2V 0.2 0.05, U / Yi / h by PBiCG + DILU --
        # two time steps of the reference's unchanged equation files against the oracle with the same selection
        G5 = ffm.GAMG(ctx, A, l2, u2, Sf=m.Sf[fOrd])
        ref5 = plume.Plume(shape, mesh=plume.make_mesh(shape, empty=empty), solvers=plume.WallFireSolvers(cOrd, fOrd, l2, u2, m.Sf[fOrd]))
        ref5.stored_bc = True; ref5.divU_scheme = ("filteredLinear2V", 0.2, 0.05)
        out5, nit5 = new_out(), (C.c_int * 32)()
        cs5 = case_of(ref5, out5, nit5)
        cs5.wallFireSelection = 1; cs5.gamg = G5.h
        solver5 = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(cs5))
        for k in range(2):
            n5 = lib.firefoam_snippets_advance(solver5, C.byref(cs5), 1 if k == 1 else 0)
            ref5.step()
            assert list(nit5[:n5]) == [pf["nIterations"] for _, pf in ref5.sol.log], (k, list(nit5[:n5]), [pf["nIterations"] for _, pf in ref5.sol.log])
        lib.firefoam_snippets_destroy(solver5)
        assert max(pf["nIterations"] for nm, pf in ref5.sol.log if nm == "p_rgh") >= 2                 # V-cycles were needed
        f5 = ref5.fields()
        # (5e-7: PBiCG to 1e-7 / GAMG to 1e-6, and the multivariateSelection limiter -- the minimum over six fields, some of them uniform
        # up to round-off -- need not pick the same weights on faces with negligible flux in the two implementations)
        for name, a in (("rho", out5["rho"]), ("T", out5["T"]), ("Uy", out5["U"][1]), ("Uz", out5["U"][2]), ("O2", out5["Y"][0]), ("h", out5["h"])):
            assert rel_l2(a[inv0], f5[name]) < 5e-7, (name, rel_l2(a[inv0], f5[name]))
        assert np.linalg.norm(out5["p_rgh"][inv0] - f5["p_rgh"]) / np.linalg.norm(f5["p_rgh"] - f5["p_rgh"].mean()) < 1e-4
        G5.close()

    for step in range(2):
        out = new_out()
        nit = (C.c_int * 32)()
        cs = case_of(ref, out, nit)
        os.environ["FFM_FOAM_QUIET"] = "1"
        n = lib.firefoam_snippets_step(ctx.h, A.h, mesh.h, C.byref(cs))
        ref.step()
        its_ref = [pf["nIterations"] for _, pf in ref.sol.log]
        assert list(nit[:n]) == its_ref, (step, list(nit[:n]), its_ref)
        inv = np.empty(N, np.int64); inv[cOrd] = np.arange(N)            # library order -> natural order
        finv = np.empty(F, np.int64); finv[fOrd] = np.arange(F)
        f = ref.fields()
        got = {"rho": out["rho"], "p": out["p"], "T": out["T"], "h": out["h"], "K": out["K"], "Ux": out["U"][0], "Uy": out["U"][1], "Uz": out["U"][2]}
        for i, sname in enumerate(plume.SPECIES):
            got[sname] = out["Y"][i]
        errs = {}
        for name, a in got.items():
            a, b = a[inv], f[name]
            errs[name] = np.abs(a).max() if np.linalg.norm(b) < 1e-30 else rel_l2(a, b)
        errs["p_rgh"] = np.linalg.norm(out["p_rgh"][inv] - f["p_rgh"]) / max(np.linalg.norm(f["p_rgh"] - f["p_rgh"].mean()), 1e-30)
        errs["phi"] = np.linalg.norm(out["phi"][finv] - ref.phi) / max(np.linalg.norm(ref.phi), 1e-30)
        errs["dpdt"] = np.linalg.norm(out["dpdt"][inv] - ref.dpdt) / max(np.linalg.norm(ref.dpdt), 1e-30)
        errs["p_rgh_b"] = np.abs(out["p_rghB"] - bnd(ref.p_rgh_b)).max() / max(np.abs(bnd(ref.p_rgh_b)).max(), 1e-30)
        tol = {"p_rgh": 1e-5, "dpdt": 1e-5, "phi": 1e-7, "p_rgh_b": 1e-5}
        bad = {k: v for k, v in errs.items() if not v < tol.get(k, 1e-8)}
        assert not bad, (step, bad, {k: float("%.3g" % v) for k, v in errs.items()})
    A.close()


@pytest.mark.parametrize("from_polymesh", [False, True])
def test_reference_phrghEqn_reproduces_the_golden_log_on_the_device(O, ffm, ctx, capfd, from_polymesh, tmp_path):
    """The reference's solver/phrghEqn.H, included unchanged, run on the reference's steckler case (30 x 15 x 20 cells, the
    compartment baffles and doorway, ph_rgh fixedValue 0 on `top` and fixedFluxPressure elsewhere, the boundary mixtures of the
    case files: oracle/steckler.py) through the Foam layer on the device: every operator of the file (fvc::interpolate, snGrad,
    constrainPressure, fvm::laplacian, fvc::div, the DICPCG solve, thermo.rho()) is this repository's device code, and the five
    solves reproduce the reference's golden log (cases/steckler/original/linux64/log.fireFoam:92-101): iteration counts
    29, 32, 7, 0, 0 and every printed residual and gMax-gMin to 1e-6."""
    import json
    from oracle import plume, steckler
    so = os.path.join(os.path.dirname(ffm.libpath()), "libffm_refsnippets.so")
    if not os.path.exists(so):
        pytest.skip("libffm_refsnippets.so not built (needs /root/reference at build time)")
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_ph_rgh.json")))["solves"]
    lib = C.CDLL(so)
    lib.firefoam_snippets_hydrostatic.restype = C.c_int
    lib.firefoam_snippets_hydrostatic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]
    m = steckler.build_mesh()
    if from_polymesh:
        # the mesh goes through a case directory: written as constant/polyMesh (tests/polymesh_writer.py, the files blockMesh +
        # createBaffles would leave), read back and turned into finite-volume geometry by the library's reader (ffm_polymesh_*);
        # everything the run below uses -- addressing, V, C, Sf, Cf, weights, deltaCoeffs, patches -- comes from there
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from polymesh_writer import write_polymesh
        from oracle import fv
        d = str(tmp_path / "polyMesh")
        write_polymesh(d, m, (-2.0, 0.0, -2.0), patch_types={"base": "wall", "baffle1DWall_master": "wall", "baffle1DWall_slave": "wall"})
        L = ffm.lib()
        pm = C.c_void_p()
        assert L.ffm_polymesh_read(d.encode(), C.byref(pm)) == 0, L.ffm_last_error()
        nC, nI, nPa = C.c_int(), C.c_int(), C.c_int()
        L.ffm_polymesh_sizes(pm, None, C.byref(nC), None, C.byref(nI), C.byref(nPa))
        ipp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)); dpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

        class _M:
            pass
        r = _M(); r.nCells, r.nFaces = nC.value, nI.value
        r.l, r.u = np.empty(r.nFaces, np.int32), np.empty(r.nFaces, np.int32)
        L.ffm_polymesh_addressing(pm, ipp(r.l), ipp(r.u))
        V, Cc, Sf, Cf = np.empty(r.nCells), np.empty((3, r.nCells)), np.empty((3, r.nFaces)), np.empty((3, r.nFaces))
        mag, w, dl = np.empty(r.nFaces), np.empty(r.nFaces), np.empty(r.nFaces)
        L.ffm_polymesh_geometry(pm, dpp(V), dpp(Cc), dpp(Sf), dpp(Cf), dpp(mag), dpp(w), dpp(dl), None)
        r.V, r.C, r.Sf, r.Cf, r.magSf, r.weights, r.deltaCoeffs = V, Cc.T.copy(), Sf.T.copy(), Cf.T.copy(), mag, w, dl
        r.patches = []
        name, st, n = C.create_string_buffer(64), C.c_int(), C.c_int()
        for q in range(nPa.value):
            L.ffm_polymesh_patch(pm, q, name, None, C.byref(st), C.byref(n))
            fc, pS, pC, pd = np.empty(n.value, np.int32), np.empty((3, n.value)), np.empty((3, n.value)), np.empty(n.value)
            L.ffm_polymesh_patch_geometry(pm, q, ipp(fc), dpp(pS), dpp(pC), dpp(pd))
            r.patches.append(fv.Patch(name.value.decode(), fc, pS.T.copy(), pC.T.copy(), pd))
        L.ffm_polymesh_destroy(pm)
        m = r
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    dp = C.POINTER(C.c_double)
    keep = []

    def P(a):
        a = np.ascontiguousarray(a, np.float64); keep.append(a)
        return a.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [np.ascontiguousarray(a, np.float64) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    per = lambda fn: np.concatenate([fn(p) for p in m.patches])
    # the case data of oracle/steckler.py (each item cites the reference's case file there)
    g = np.array([0.0, -9.81, 0.0]); ghRef = -np.linalg.norm(g) * 3.0
    gh, ghf, ghb = m.C @ g - ghRef, m.Cf @ g - ghRef, per(lambda p: p.Cf @ g - ghRef)
    Wbaffle = 1.0 / (0.232 / 31.9988 + 0.768 / 28.0134)
    psib = per(lambda p: np.full(p.size, 1.0 / ((steckler.RR / Wbaffle) * 300.0) if p.name.startswith("baffle") else 1.0 / ((steckler.RR / 31.9988) * 298.15)))
    Wmix = 1.0 / (0.23301 / 31.9988 + 0.76699 / 28.0134)
    psi = 1.0 / ((steckler.RR / Wmix) * 298.15)
    Y = [np.full(N, v) for v in (0.23301, 0.0, 0.0, 0.0, 0.76699)]
    topMask = per(lambda p: np.full(p.size, 1.0 if p.name == "top" else 0.0))
    zN, z3, zF, zB = np.zeros(N), np.zeros((3, N)), np.zeros(F), np.zeros(B)
    o = dict(rho=np.empty(N), p=np.empty(N), p_rgh=np.empty(N), p_rghB=np.empty(B))
    nit, res = (C.c_int * 8)(), np.zeros(16)
    cs = SnippetCase(
        deltaT=1.0, RR=steckler.RR, Cp=plume.CP, Tref=298.15, pRef=101325.0, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC, tau=plume.TAU,
        nSpecies=5, inertIndex=4, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
        rho=P(np.full(N, psi * 101325.0)), U=P(z3), p=P(np.full(N, 101325.0)), p_rgh=P(zN), h=P(zN), Y=PP(Y), K=P(zN), dpdt=P(zN), phiF=P(zF), phiB=P(zB),
        gh=P(gh[cOrd]), ghfF=P(ghf[fOrd]), ghfB=P(ghb),
        fU=P(np.ones(3 * B)), refU=P(np.zeros(3 * B)), fixesU=P(np.ones(B)), fY=P(zB), refY=PP([zB] * 5), fH=P(zB), refH=P(zB),
        fluxMaskP=P(1.0 - topMask), totalMaskP=P(topMask), ph_rgh_b=P(zB), p_rghB=P(zB),
        rhoOut=o["rho"].ctypes.data_as(dp), pOut=o["p"].ctypes.data_as(dp), p_rghOut=o["p_rgh"].ctypes.data_as(dp), p_rghBOut=o["p_rghB"].ctypes.data_as(dp),
        nIterOut=nit, nIterCap=8, psiB=P(psib), resOut=res.ctypes.data_as(dp))
    os.environ["FFM_FOAM_QUIET"] = "1"
    n = lib.firefoam_snippets_hydrostatic(ctx.h, A.h, mesh.h, C.byref(cs))
    assert n == 5 and list(nit[:5]) == [g_["nIterations"] for g_ in gold] == [29, 32, 7, 0, 0], list(nit[:n])
    for k, g_ in enumerate(gold):
        assert abs(res[2 * k] - g_["initialResidual"]) <= 1e-6 * g_["initialResidual"], (k, res[2 * k], g_["initialResidual"])
        assert abs(res[2 * k + 1] - g_["finalResidual"]) <= 1e-6 * g_["finalResidual"], (k, res[2 * k + 1], g_["finalResidual"])
    variation = o["p_rgh"].max() - o["p_rgh"].min()
    assert abs(variation - gold[-1]["variation"]) <= 1e-6 * gold[-1]["variation"], variation
    # the log the Foam layer prints while the reference's phrghEqn.H runs diffs against the golden one: same lines, solver name,
    # field, iteration count; the 8-digit numbers equal to 1e-6 (SolverPerformance::print)
    del os.environ["FFM_FOAM_QUIET"]
    try:
        capfd.readouterr()
        assert lib.firefoam_snippets_hydrostatic(ctx.h, A.h, mesh.h, C.byref(cs)) == 5
        out = capfd.readouterr().out
    finally:
        os.environ["FFM_FOAM_QUIET"] = "1"
    import re
    pat = re.compile(r"^(\w+):  Solving for (\w+), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)$")
    got = [pat.match(ln).groups() for ln in out.splitlines() if ln.startswith("DICPCG")]
    golden_lines = ["DICPCG:  Solving for ph_rgh, Initial residual = 1, Final residual = 0.0080439052, No Iterations 29",
                    "DICPCG:  Solving for ph_rgh, Initial residual = 0.0010688694, Final residual = 9.4376262e-06, No Iterations 32",
                    "DICPCG:  Solving for ph_rgh, Initial residual = 9.4390676e-06, Final residual = 9.6501e-07, No Iterations 7",
                    "DICPCG:  Solving for ph_rgh, Initial residual = 9.6507488e-07, Final residual = 9.6507488e-07, No Iterations 0",
                    "DICPCG:  Solving for ph_rgh, Initial residual = 9.6507488e-07, Final residual = 9.6507488e-07, No Iterations 0"]
    want = [pat.match(ln).groups() for ln in golden_lines]            # cases/steckler/original/linux64/log.fireFoam:92-100
    assert len(got) == 5
    for g_, w_ in zip(got, want):
        assert (g_[0], g_[1], g_[4]) == (w_[0], w_[1], w_[4])
        assert abs(float(g_[2]) - float(w_[2])) <= 1e-6 * float(w_[2]) and abs(float(g_[3]) - float(w_[3])) <= 1e-6 * float(w_[3])
    assert out.splitlines()[0] == golden_lines[0]                      # the first line: every character
    A.close()


def test_reference_equation_files_hold_1e8_over_six_steps_on_the_conditioned_case(O, ffm, ctx):
    """The multi-step parity question of tests/test_plume_multistep_gpu.py asked of the path the reference's files take: the
    unchanged solver/rhoEqn.H, UEqn.H, YEEqn.H (with its `Gauss multivariateSelection` common limiter) and pEqn.H over the Foam
    layer, six consecutive steps on the device-resident state of the conditioned plume case (every specie and h smoothly varying,
    no exact zeros, inflow and ambient values distinct: the limiter is well-conditioned on every face), every field within 1e-8
    rel-L2 of the oracle's after EVERY step with identical iteration counts."""
    from oracle import plume
    so = os.path.join(os.path.dirname(ffm.libpath()), "libffm_refsnippets.so")
    if not os.path.exists(so):
        pytest.skip("libffm_refsnippets.so not built (needs /root/reference at build time)")
    lib = C.CDLL(so)
    shape = (12, 16, 12)
    ref = plume.Plume(shape, conditioned=True); ref.stored_bc = True
    m = ref.m
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())
    dp = C.POINTER(C.c_double)
    keep = []

    def P(a):
        a = np.ascontiguousarray(a, np.float64); keep.append(a)
        return a.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [np.ascontiguousarray(a, np.float64) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    cell = lambda a: np.asarray(a)[..., cOrd]
    face = lambda a: np.asarray(a)[fOrd]
    bnd = lambda lst: np.concatenate(lst)
    per = lambda fn: bnd([fn(p) for p in m.patches])
    is_open = lambda p: p.name not in ("inlet", "floor")
    fU = np.concatenate([per(lambda p, d=d: np.where(np.abs(p.Sf[:, d]) > 0, 0.0, -1.0) if is_open(p) else np.ones(p.size)) for d in range(3)])
    refU = np.concatenate([per(lambda p, d=d: np.full(p.size, plume.U_IN if (p.name == "inlet" and d == 1) else 0.0)) for d in range(3)])
    fixesU = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
    fY = per(lambda p: np.full(p.size, 1.0 if p.name == "inlet" else (0.0 if p.name == "floor" else -1.0)))
    refY = [per(lambda p, i=i: np.full(p.size, ref.Y_in[i] if p.name == "inlet" else (ref.Y_amb[i] if is_open(p) else 0.0))) for i in range(5)]
    fH = per(lambda p: np.full(p.size, -1.0 if is_open(p) else 1.0))
    refH = per(lambda p: np.full(p.size, plume.CP * (plume.T_IN - plume.TREF) if p.name == "inlet" else (ref.h_amb if is_open(p) else 0.0)))
    fluxMask = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
    totalMask = per(lambda p: np.full(p.size, 1.0 if is_open(p) else 0.0))
    ghfb = bnd([p.Cf @ plume.G - ref.ghRef for p in m.patches])
    out = dict(rho=np.empty(N), U=np.empty((3, N)), p=np.empty(N), p_rgh=np.empty(N), h=np.empty(N), Y=[np.empty(N) for _ in range(5)],
               T=np.empty(N), K=np.empty(N), dpdt=np.empty(N), phi=np.empty(F), phib=np.empty(B), p_rghB=np.empty(B))
    nit = (C.c_int * 32)()
    cs = SnippetCase(
        deltaT=ref.dt, RR=plume.RR, Cp=plume.CP, Tref=plume.TREF, pRef=plume.PREF, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC,
        tau=plume.TAU, nSpecies=5, inertIndex=plume.INERT, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
        rho=P(cell(ref.rho)), U=P(cell(ref.U)), p=P(cell(ref.p)), p_rgh=P(cell(ref.p_rgh)), h=P(cell(ref.h)), Y=PP([cell(ref.Y[i]) for i in range(5)]),
        K=P(cell(ref.K)), dpdt=P(cell(ref.dpdt)), phiF=P(face(ref.phi)), phiB=P(bnd(ref.phib)),
        gh=P(cell(ref.gh)), ghfF=P(face(ref.ghf)), ghfB=P(ghfb),
        fU=P(fU), refU=P(refU), fixesU=P(fixesU), fY=P(fY), refY=PP(refY), fH=P(fH), refH=P(refH),
        fluxMaskP=P(fluxMask), totalMaskP=P(totalMask), ph_rgh_b=P(bnd(ref.ph_rgh_b)), p_rghB=P(bnd(ref.p_rgh_b)),
        rhoOut=out["rho"].ctypes.data_as(dp), UOut=out["U"].ctypes.data_as(dp), pOut=out["p"].ctypes.data_as(dp),
        p_rghOut=out["p_rgh"].ctypes.data_as(dp), hOut=out["h"].ctypes.data_as(dp),
        TOut=out["T"].ctypes.data_as(dp), KOut=out["K"].ctypes.data_as(dp), dpdtOut=out["dpdt"].ctypes.data_as(dp),
        phiOutF=out["phi"].ctypes.data_as(dp), phiOutB=out["phib"].ctypes.data_as(dp), p_rghBOut=out["p_rghB"].ctypes.data_as(dp),
        nIterOut=nit, nIterCap=32)
    yout = (dp * 5)(*[a.ctypes.data_as(dp) for a in out["Y"]]); keep.append(yout)
    cs.YOut = yout
    os.environ["FFM_FOAM_QUIET"] = "1"
    lib.firefoam_snippets_create.restype = C.c_void_p
    lib.firefoam_snippets_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]
    lib.firefoam_snippets_advance.restype = C.c_int
    lib.firefoam_snippets_advance.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
    lib.firefoam_snippets_destroy.argtypes = [C.c_void_p]
    solver = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(cs))
    inv = np.empty(N, np.int64); inv[cOrd] = np.arange(N)
    finv = np.empty(F, np.int64); finv[fOrd] = np.arange(F)
    worst = 0.0
    for step in range(6):
        n = lib.firefoam_snippets_advance(solver, C.byref(cs), 1)
        ref.step()
        its_ref = [pf["nIterations"] for _, pf in ref.sol.log]
        assert list(nit[:n]) == its_ref, (step, list(nit[:n]), its_ref)
        f = ref.fields()
        got = {"rho": out["rho"], "p": out["p"], "T": out["T"], "h": out["h"], "K": out["K"], "Ux": out["U"][0], "Uy": out["U"][1], "Uz": out["U"][2]}
        for i, sname in enumerate(plume.SPECIES):
            got[sname] = out["Y"][i]
        errs = {name: rel_l2(a[inv], f[name]) for name, a in got.items()}
        errs["phi"] = rel_l2(out["phi"][finv], ref.phi)
        bad = {k: v for k, v in errs.items() if not v < 1e-8}
        assert not bad, (step, bad)
        e = np.linalg.norm(out["p_rgh"][inv] - f["p_rgh"]) / np.linalg.norm(f["p_rgh"] - f["p_rgh"].mean())
        assert e < 1e-7, (step, "p_rgh", e)
        worst = max(worst, max(errs.values()))
    lib.firefoam_snippets_destroy(solver)
    print("reference files, conditioned case, worst rel-L2 over six steps: %.2e" % worst)
    A.close()


def test_the_compiled_case_advanced_by_the_reference_files(O, ffm, ctx):
    """bench.py's class-layer measurement: the state of a compiled plume case (ffm.Plume) exported in the library's own orders
    (ffm_plume_get_raw -> firefoam-dev_amd/snippets.py:FromPlume) and advanced by the reference's unchanged equation files on the plume's
    own matrix and mesh, against the compiled driver advanced from the same state: identical iteration counts, fields to 1e-8 after the
    first step (the later ones at the 1e-5 of the free-running common limiter, tests/test_plume_gpu.py)."""
    S = ffm.snippets
    lib = S.load()
    if lib is None:
        pytest.skip("libffm_refsnippets.so not built (needs /root/reference at build time)")
    n = (24, 30, 20)
    case = ffm.Plume(ctx, n)
    snap = S.FromPlume(case)
    out = snap.outputs()
    os.environ["FFM_FOAM_QUIET"] = "1"
    solver = lib.firefoam_snippets_create(ctx.h, case.ldu_handle(), case.mesh().h, C.byref(snap.cs))
    other = ffm.Plume(ctx, n)                      # the compiled driver from the same start state (its own matrix)
    for step in range(3):
        nsol = lib.firefoam_snippets_advance(solver, C.byref(snap.cs), 1)
        other.step()
        its = [p["nIterations"] for _, p in other.solves()]
        assert list(snap.nit[:nsol]) == its, (step, list(snap.nit[:nsol]), its)
        tol = 1e-8 if step == 0 else 1e-5
        for name, a in (("rho", out["rho"]), ("T", out["T"]), ("h", out["h"]), ("p", out["p"]), ("Uy", out["U"][1]), ("O2", out["Y"][0])):
            b = other.raw(name)
            assert rel_l2(a, b) < tol, (step, name, rel_l2(a, b))
    lib.firefoam_snippets_destroy(solver)
    case.close(); other.close()
