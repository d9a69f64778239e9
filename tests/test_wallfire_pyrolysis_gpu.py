"""BASELINE config 5 (cases/wallFireSpread2D: a 2-D gas region, one cell thick with `empty` patches, coupled to a reactingOneDim
pyrolysis region) as ONE time loop on the device: the body of the reference's loop (solver/fireFoam.C:76-121 -- its unchanged
solidRegionDiffusionNo.H / setMultiRegionDeltaT.H, `pyrolysis.evolve()`, rhoEqn.H, UEqn.H, YEEqn.H, pEqn.H) through
examples/fireFoam_snippets.C with
  * the case's solver / scheme selection (cases/wallFireSpread2D/system/fvSolution:36-60,67-152, fvSchemes:41): p_rgh by GAMG +
    GaussSeidel, div(phi,U) Gauss filteredLinear2V 0.2 0.05, U / Yi / h by PBiCG + DILU;
  * a pyrolysing panel behind one wall patch: pyrolysisModelCollection::evolve() (include/fireFoamHandles.H) = the mapped patch
    conditions of lib/fvPatchFieldsPyrolysis the case selects (0/T:61-75 turbulentTemperatureRadiationQinCoupledMixed, 0/U:52-64
    flowRateInletVelocityPyrolysisCoupled: ffm_pyro_couple_d) + reactingOneDim::evolveRegion for every column (ffm_pyro_step).
The solid heats the gas (wall temperature -> the energy patch's fixed value), the gas and the incident radiation heat the solid,
and the pyrolysate enters the gas region as the wall patch's inflow velocity.
Oracle: oracle/plume.py (same selection: WallFireSolvers, filteredLinear2V) and oracle/pyrolysis.py (Panel.step, couple) composed
in the same order.  Synthetic panel state (started hot so that the wood pyrolyses within the steps of the test); parity unpinned by
reference data, as the two halves are."""
import ctypes as C
import os

import numpy as np
import pytest

from common import rel_l2
from test_reference_snippets_gpu import SnippetCase

pytestmark = pytest.mark.gpu


def test_gas_region_and_pyrolysis_panel_in_one_time_loop(O, ffm, ctx):
    """round 2's form: reactingOneDim columns, both sides of the coupled wall condition from the state before the step, a given
    incident radiation"""
    _run(O, ffm, ctx, (1, 24, 20), ("xmin", "xmax"), None, 6)


def test_config5_selections_from_the_case_files(O, ffm, ctx):
    """BASELINE config 5 with the selections of the reference's own case files (tests/golden/wallfire_case_data.json, made by
    tests/golden/make_wallfire_case_data.py from cases/wallFireSpread2D/constant/{pyrolysisZones,radiationProperties}, constant/panelRegion/*,
    system/panelRegion/fvSchemes, 0/panelRegion/T, 0/IDefault, system/controlDict ...; tests/test_dictionary_cpu.py reads the same entries
    with include/ffmDictionary.H where the reference is mounted):
      * pyrolysisModel reactingOneDim21 (lib/regionModels/pyrolysisModels/reactingOneDim21), laplacian(kappa,T) Gauss harmonic, the
        constHTemperature back face, the solids / reaction / layer count / thickness of the case, evolved in the reference's order
        (coupled wall condition inside the solid's step, gas patches from the new solid state), solidRegionDiffNo() and maxDi in the
        time-step control (solver/solidRegionDiffusionNo.H, setMultiRegionDeltaT.H -- the reference's files, unchanged);
      * radiationModel fvDOM with nPhi 2 / nTheta 2 on the 2-D mesh (8 rays in the x-y plane), maxIter 5, convergence 1e-3, solverFreq
        (shortened from the case's 10 to 2 so that the test's steps cross three radiation solves), div(Ji,Ii_h) Gauss linearUpwind, Ii by
        GAMG + DILU, constRadFractionEmission with Ehrr1 0.6 / Ehrr2 0.3 and radScaling over patch1 (no burner in the synthetic geometry)
        and patch2 = the panel patch; the panel patch's wall emissivity `solidRadiation` (0.17 virgin -> 0.85 char by volume
        fractions), every other patch 1; the solid reads the model's qin (neighbourFieldRadiativeName qin).
    Device (the reference's unchanged equation files + the handles of include/fireFoamHandles.H) = oracle (oracle/plume.py +
    oracle/pyrolysis.py + oracle/fvdom.py) step by step.  PARITY UNPINNED by reference data: the reference ships no output of this
    case; what is pinned is each ingredient's restatement (fvDOM's emissivity-1 path and GAMG by the steckler log)."""
    import json
    sel = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wallfire_case_data.json")))
    assert sel["pyrolysis"]["pyrolysisModel"] == "reactingOneDim21" and sel["radiation"]["radiationModel"] == "fvDOM"
    _run(O, ffm, ctx, (20, 24, 1), ("zmin", "zmax"), sel, 7)


@pytest.mark.parametrize("shape,h", [((355, 710, 1), 0.004), ((1420, 2840, 1), 0.001)])
def test_config5_at_its_size_with_the_panel_attached(O, ffm, ctx, shape, h):
    """BASELINE config 5's size -- a 2-D gas region of 1420 x 2840 x 1 = 4.03 M cells (z empty) -- with the case's selections and the
    pyrolysing panel behind the wall patch (710 columns x 8 layers), three time steps of the reference's loop on the device; no oracle
    at this size: property checks (finite and bounded fields, the panel's mass balance, the radiation model irradiating the panel, every
    solve within maxIter).  The start state is the ambient at rest with p = pRef + rho gh."""
    import json
    sel = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wallfire_case_data.json")))
    _run(O, ffm, ctx, shape, ("zmin", "zmax"), sel, 3, compare=False, h=h)


class _NoSolve:
    """stands in for the oracle's solver selection where only the mesh and the start state of oracle/plume.py are wanted"""
    def __init__(self):
        self.log = []

    def solve(self, kind, name, mesh, diag, upper, lower, source, psi0):
        self.log.append((name, dict(nIterations=0, initialResidual=0.0, finalResidual=0.0)))
        return psi0


def _run(O, ffm, ctx, shape, empty, sel, nSteps, compare=True, h=0.05):
    import time
    from oracle import plume, pyrolysis as PY
    t00 = time.time()

    def stage(what):
        if not compare:
            print("  [%7.1f s] %s" % (time.time() - t00, what), flush=True)
    so = os.path.join(os.path.dirname(ffm.libpath()), "libffm_refsnippets.so")
    if not os.path.exists(so):
        pytest.skip("libffm_refsnippets.so not built (needs /root/reference at build time)")
    lib = C.CDLL(so)
    argt = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]
    lib.firefoam_snippets_create.restype = C.c_void_p; lib.firefoam_snippets_create.argtypes = argt
    lib.firefoam_snippets_time_step.restype = C.c_int; lib.firefoam_snippets_time_step.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
    lib.firefoam_snippets_destroy.argtypes = [C.c_void_p]
    gasMesh = plume.make_mesh(shape, h=h, empty=empty)
    m = gasMesh
    N, F = m.nCells, m.nFaces
    B = sum(p.size for p in m.patches)
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    mesh.set_face_centres(m.Cf[fOrd].T.copy())
    stage("device mesh built")
    G5 = ffm.GAMG(ctx, A, l2, u2, Sf=m.Sf[fOrd])
    stage("GAMG agglomeration built (%d levels)" % G5.nLevels if hasattr(G5, "nLevels") else "GAMG agglomeration built")
    # (the size runs start from the conditioned state of oracle/plume.py: in a gas at rest with uniform temperature the rays that leave
    # only ambient walls have a UNIFORM solution, for which OpenFOAM's residual normalisation degenerates -- normFactor ~ round-off --
    # and the ray solves run to maxIter on noise; no fire has such a state after its first instants)
    ref = plume.Plume(shape, h=h, mesh=gasMesh, solvers=plume.WallFireSolvers(cOrd, fOrd, l2, u2, m.Sf[fOrd]) if compare else _NoSolve(),
                      conditioned=not compare)
    ref.stored_bc = True; ref.divU_scheme = ("filteredLinear2V", 0.2, 0.05)

    # ---- the panel: one column behind every face of the wall patch (the plume's `inlet` patch plays the pyrolysing wall)
    names = [p.name for p in m.patches]
    qw = names.index("inlet"); pw = m.patches[qw]
    start = int(np.concatenate([[0], np.cumsum([p.size for p in m.patches])])[qw])
    nCol = pw.size
    fmap = (start + np.arange(nCol)).astype(np.int32)
    area = float(pw.magSf[0]); assert np.allclose(pw.magSf, area)
    T0 = 600.0                                                    # above the reaction's critical temperature: pyrolysis from the first step
    if sel is None:
        solid = PY.Panel(nCol, 8, thickness=0.0127, area=area, T0=T0)
        dev = ffm.PyrolysisPanel(ctx, nCol, 8, thickness=0.0127, area=area, T0=T0)
    else:
        py, so = sel["pyrolysis"], sel["solids"]
        assert (so["v"]["rho"], so["v"]["Cp"], so["v"]["kappa"], so["v"]["Hf"]) == (PY.WOOD.rho, PY.WOOD.Cp, PY.WOOD.kappa, PY.WOOD.Hf)      # the oracle's constants are the case's
        assert (so["char"]["rho"], so["char"]["Cp"], so["char"]["kappa"], so["char"]["Hf"]) == (PY.CHAR.rho, PY.CHAR.Cp, PY.CHAR.kappa, PY.CHAR.Hf)
        assert sel["reaction"] == dict(A=PY.REACTION["A"], Ta=PY.REACTION["Ta"], Tcrit=PY.REACTION["Tcrit"], order=PY.REACTION["n"])
        bk = sel["panelT"]["back"]; assert bk["type"] == "constHTemperature"
        msel = dict(model=py["pyrolysisModel"], alphaScheme=sel["panelSchemes"]["laplacian(thermo:alpha,h)"], kappaScheme=sel["panelSchemes"]["laplacian(kappa,T)"],
                    back=("constH", bk["h"], bk["Tinf"]),
                    radiation=dict(v=(so["v"]["absorptivity"], so["v"]["emissivity"]), char=(so["char"]["absorptivity"], so["char"]["emissivity"])))
        solid = PY.Panel(nCol, py["nLayers"], thickness=py["thickness"], area=area, T0=T0, **msel)
        dev = ffm.PyrolysisPanel(ctx, nCol, py["nLayers"], thickness=py["thickness"], area=area, T0=T0)
        dev.set_model(**msel)
    qin = np.zeros(B); qin[fmap] = 3.0e4 * (1.0 + 0.3 * np.sin(0.7 * np.arange(nCol)))      # incident radiation on the wall faces [W/m2]
    e, a, hocSolid, qFuel = 0.9, 0.85, 1.66e7, 4.6e7              # cases/wallFireSpread2D/0/U:62 hocSolid; propane qFuel
    if sel is not None:
        hocSolid = sel["hocSolid"]
        rd, ct = sel["radiation"], sel["controls"]
        ref.set_fvdom(rd["nPhi"], rd["nTheta"], 2, rd["maxIter"], rd["convergence"], 0.0, rd["Ehrr1"], rd["Ehrr2"], divScheme=rd["div(Ji,Ii_h)"])
        ref.rad_patches = ((), ("inlet",))                        # patch1: the burner (absent here); patch2: the panel patch
        assert len(ref.fvdom.rays) == 8 and rd["wallEmissivity"][".*"] == ["lookup", 1.0] and rd["wallEmissivity"]["region0_to_panelRegion_panel"][0] == "solidRadiation"
        ref.fvdom.emissivity[qw] = solid.surface_radiation()[1]
        ref.set_time_controls(ct["maxCo"], ct["maxDeltaT"])
        ref.dt = ref.dt                                            # the synthetic case starts from its own deltaT (1 ms); maxDeltaT / maxCo / maxDi are the case's
    nfw = pw.Sf / pw.magSf[:, None]
    kDw = (plume.MU / plume.PR * plume.CP) * pw.deltaCoeffs       # kappaEff*deltaCoeffs of the gas side (constant-property stand-in)

    dp = C.POINTER(C.c_double)
    keep = []

    def P(x):
        x = np.ascontiguousarray(x, np.float64); keep.append(x)
        return x.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [np.ascontiguousarray(x, np.float64) for x in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[x.ctypes.data_as(dp) for x in arrs]); keep.append(arr)
        return arr
    cell = lambda x: np.asarray(x)[..., cOrd]
    face = lambda x: np.asarray(x)[fOrd]
    bnd = lambda lst: np.concatenate(lst)
    per = lambda fn: bnd([fn(p) for p in m.patches])
    is_open = lambda p: p.name not in ("inlet", "floor")
    fU = np.concatenate([per(lambda p, d=d: np.where(np.abs(p.Sf[:, d]) > 0, 0.0, -1.0) if is_open(p) else np.ones(p.size)) for d in range(3)])
    fixesU = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
    fY = per(lambda p: np.full(p.size, 1.0 if p.name == "inlet" else (0.0 if p.name == "floor" else -1.0)))
    refY = [per(lambda p, i=i: np.full(p.size, ref.Y_in[i] if p.name == "inlet" else (ref.Y_amb[i] if is_open(p) else 0.0))) for i in range(5)]
    fH = per(lambda p: np.full(p.size, -1.0 if is_open(p) else 1.0))
    Z = per(lambda p: np.full(p.size, ref.h_amb if is_open(p) else 0.0))
    fluxMask = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0)); totalMask = 1.0 - fluxMask
    ghfb = bnd([p.Cf @ plume.G - ref.ghRef for p in m.patches])
    out = dict(rho=np.empty(N), U=np.empty((3, N)), p=np.empty(N), p_rgh=np.empty(N), h=np.empty(N), Y=[np.empty(N) for _ in range(5)],
               T=np.empty(N), K=np.empty(N), dpdt=np.empty(N), phi=np.empty(F), phib=np.empty(B), p_rghB=np.empty(B))
    nit, dtOut = (C.c_int * 32)(), np.zeros(1)
    cs = SnippetCase(
        deltaT=ref.dt, RR=plume.RR, Cp=plume.CP, Tref=plume.TREF, pRef=plume.PREF, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC,
        tau=plume.TAU, nSpecies=5, inertIndex=plume.INERT, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
        rho=P(cell(ref.rho)), U=P(cell(ref.U)), p=P(cell(ref.p)), p_rgh=P(cell(ref.p_rgh)), h=P(cell(ref.h)), Y=PP([cell(ref.Y[i]) for i in range(5)]),
        K=P(cell(ref.K)), dpdt=P(cell(ref.dpdt)), phiF=P(face(ref.phi)), phiB=P(bnd(ref.phib)),
        gh=P(cell(ref.gh)), ghfF=P(face(ref.ghf)), ghfB=P(ghfb),
        # the wall patch starts with U = 0 and the enthalpy of the gas at rest; evolve() overwrites both reference values on its faces
        fU=P(fU), refU=P(np.zeros(3 * B)), fixesU=P(fixesU), fY=P(fY), refY=PP(refY), fH=P(fH), refH=P(Z),
        fluxMaskP=P(fluxMask), totalMaskP=P(totalMask), ph_rgh_b=P(bnd(ref.ph_rgh_b)), p_rghB=P(bnd(ref.p_rgh_b)),
        rhoOut=P(out["rho"]), UOut=P(out["U"]), pOut=P(out["p"]), p_rghOut=P(out["p_rgh"]), hOut=P(out["h"]), TOut=P(out["T"]), KOut=P(out["K"]),
        dpdtOut=P(out["dpdt"]), phiOutF=P(out["phi"]), phiOutB=P(out["phib"]), p_rghBOut=P(out["p_rghB"]), nIterOut=nit, nIterCap=32)
    cs.YOut = PP(out["Y"])
    cs.emptyDirections = sum(1 << d for d in range(3) if m.solutionD[d] < 0)
    cs.wallFireSelection = 1; cs.gamg = G5.h
    cs.adjustTimeStep = 0; cs.maxCo = 0.3; cs.maxDeltaT = 0.05; cs.dtOut = dtOut.ctypes.data_as(dp)
    cs.pyro = dev.h; cs.pyroCols = nCol; cs.pyroMap = fmap.ctypes.data_as(C.POINTER(C.c_int)); cs.pyroQin = P(qin)
    cs.pyroEmissivity = e; cs.pyroAbsorptivity = a; cs.pyroHocSolid = hocSolid; cs.pyroQFuel = qFuel
    radIters, qinDev = C.c_int(0), np.zeros(B)
    if sel is not None:
        emis0 = np.ones(B); emis0[fmap] = solid.surface_radiation()[1]
        maskPanel = np.zeros(B); maskPanel[fmap] = 1.0
        cs.adjustTimeStep = 1; cs.maxCo = ct["maxCo"]; cs.maxDeltaT = ct["maxDeltaT"]
        cs.pyroInStep = 1; cs.pyroMaxDi = ct["maxDi"]
        cs.fvdomReal = 1; cs.radiationFreq = 2; cs.radNPhi = rd["nPhi"]; cs.radNTheta = rd["nTheta"]; cs.radMaxIter = rd["maxIter"]; cs.radTolerance = rd["convergence"]
        cs.radDivScheme = {"upwind": 0, "linearUpwind": 5}[rd["div(Ji,Ii_h)"]]; cs.kAbs = 0.0; cs.sigmaSB = plume.SIGMA_SB
        cs.radEhrr1 = rd["Ehrr1"]; cs.radEhrr2 = rd["Ehrr2"]; cs.radMlrMask = P(np.zeros(B)); cs.radMlrMask2 = P(maskPanel); cs.radEmissivity = P(emis0)
        cs.qinOut = qinDev.ctypes.data_as(dp); cs.radItersOut = C.pointer(radIters); cs.nIterCap = 96
        nit = (C.c_int * 96)(); cs.nIterOut = nit
    os.environ["FFM_FOAM_QUIET"] = "1"
    stage("start state ready")
    solver = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(cs))
    stage("solver object created")
    inv0 = np.empty(N, np.int64); inv0[cOrd] = np.arange(N)
    Tw = np.full(nCol, T0)                                        # the wall value starts at the solid's temperature (ffm_pyro_create)
    radSolves = 0
    if not compare:
        gas, m0 = np.zeros(nCol), (dev.field("rho") * solid.V).sum(axis=1)
        for k in range(nSteps):
            t0 = time.time()
            n = lib.firefoam_snippets_time_step(solver, C.byref(cs), 1 if k == nSteps - 1 else 0)
            print("config 5 at %d cells with the panel: step %d  %.2f s  deltaT %.3g  %d solves, iterations %s" % (N, k, time.time() - t0, dtOut[0], n, list(nit[:n])), flush=True)
            assert 0 < n <= 96 and max(nit[:n]) < 1000
            gas += dev.field("phiGas") * dtOut[0]
        lib.firefoam_snippets_destroy(solver)
        lost = m0 - (dev.field("rho") * solid.V).sum(axis=1)
        assert np.allclose(lost, gas, rtol=1e-9) and lost.min() > 0                       # the panel's mass balance: what it lost entered the gas region
        for name in ("rho", "T", "p", "h"):
            assert np.isfinite(out[name]).all(), name
        assert np.isfinite(out["U"]).all() and all(np.isfinite(y).all() for y in out["Y"])
        assert 0.3 < out["rho"].min() and out["rho"].max() < 1.5 and out["T"].min() > 280.0 and out["T"].max() <= T0 + 5.0
        assert out["T"][inv0][pw.faceCells].max() > plume.TREF + 0.5                       # the hot wall heats the gas cell layer
        assert radIters.value >= 1 and qinDev[fmap].min() >= 0 and qinDev[fmap].max() > 0  # the walls' and the flame's radiation reaches the panel
        assert abs(sum(y for y in out["Y"]) - 1.0).max() < 1e-10
        G5.close(); dev.close(); mesh.close(); A.close()
        return
    for k in range(nSteps):
        if sel is None:
            # oracle: the mapped conditions with the gas state at the start of the step, the columns, then the gas region
            q, Tw, refT, Uw = PY.couple(solid, Tw, ref.T[pw.faceCells], kDw, qin[fmap], e, a, ref.rho[pw.faceCells], pw.magSf, nfw, hocSolid, qFuel)
            solid.step(ref.dt, q)
        else:
            # the reference's loop: deltaT from the Courant number of the gas AND the diffusion number of the solid (maxDi), then
            # pyrolysis.evolve() with the radiation model's qin, then the gas patches from the new solid state
            ref.solid_DiNum, ref.maxDi = solid.diff_no(ref.dt), ct["maxDi"]
            ref.set_delta_t(); ref.adjust = False
            solid.evolve(ref.dt, ref.T[pw.faceCells], kDw, ref.fvdom.qin[qw])
            q = solid.qSurf
            refT, Uw, emisW = solid.gas_side(ref.rho[pw.faceCells], pw.magSf, nfw, hocSolid, qFuel)
            ref.fvdom.emissivity[qw] = emisW
        ref.inlet_U = Uw; ref.inlet_h = plume.CP * (refT - plume.TREF)
        ref.step()
        if sel is not None:
            ref.adjust = True
        n = lib.firefoam_snippets_time_step(solver, C.byref(cs), 1 if (k == nSteps - 1 or sel is not None) else 0)
        assert abs(dtOut[0] - ref.dt) <= 1e-9 * ref.dt, (k, dtOut[0], ref.dt)
        if sel is not None and k % 2 == 0:
            radSolves += 1
            assert radIters.value == ref.fvdom.nIterations, (k, radIters.value, ref.fvdom.nIterations)
            w = ref.fvdom.qin[qw]
            assert np.abs(qinDev[fmap] - w).max() <= 1e-6 * max(np.abs(w).max(), 1e-300), (k, np.abs(qinDev[fmap] - w).max(), np.abs(w).max())
        assert list(nit[:n]) == [pf["nIterations"] for _, pf in ref.sol.log], (k, list(nit[:n]), [pf["nIterations"] for _, pf in ref.sol.log])
        # config 5: the wall flux contains a*qin of rays solved to 1e-4 (GAMG, fvSolution:160-170) in two implementations: 1e-6
        tq = 1e-9 if sel is None else 1e-6
        assert np.abs(dev.field("qSurf") - q).max() <= tq * np.abs(q).max(), k
        assert np.abs(dev.field("T") - solid.T).max() <= tq * solid.T.max(), k
        assert np.abs(dev.field("phiGas") - solid.massGas).max() <= 10 * tq * max(solid.massGas.max(), 1e-300), k
        if sel is not None:
            assert abs(dev.diff_no(ref.dt) - solid.diff_no(ref.dt)) <= 1e-9 * solid.diff_no(ref.dt)
    lib.firefoam_snippets_destroy(solver)
    if sel is not None:
        assert radSolves >= 3 and max(np.abs(b).max() for b in ref.fvdom.qin) > 0          # the flame and the walls irradiate the panel
    # the coupling acted in both directions
    assert solid.massGas.min() > 0 and solid.Yw[:, 0].max() < 1.0                    # the wood pyrolyses ...
    f = ref.fields()
    Uin = (out["U"][:, inv0][:, pw.faceCells] * (-nfw.T)).sum(axis=0)                # ... its gas enters the gas region through the wall cells
    assert f["T"][pw.faceCells].min() > plume.TREF + 1.0                             # and the hot wall heats the gas cell layer
    assert Uin.max() > 0
    # six steps of solves to 1e-7 / 1e-8 (p_rgh: GAMG to 1e-6 of the initial residual) with different summation orders, and the common
    # limiter of the multivariateSelection scheme (on fields that are uniform up to round-off the two implementations need not pick the
    # same weights on faces with negligible flux): velocity and transported scalars agree to 1e-5
    errs = {name: rel_l2(got[inv0], f[name]) for name, got in (("rho", out["rho"]), ("T", out["T"]), ("h", out["h"]), ("Uy", out["U"][1]), ("Uz", out["U"][2]),
                                                               ("O2", out["Y"][0]), ("C3H8", out["Y"][2]))}
    # (config 5: seven steps with deltaT growing 1.44-fold per step towards maxDeltaT; the velocity follows a pressure solved by GAMG to
    # 1e-5 / relTol 0.01, fvSolution:36-47: 5e-5)
    bad = {k: v for k, v in errs.items() if not v < (5e-5 if (sel is not None and k.startswith("U")) else 1e-5)}
    assert not bad, (bad, errs)
    assert np.linalg.norm(out["p_rgh"][inv0] - f["p_rgh"]) / np.linalg.norm(f["p_rgh"] - f["p_rgh"].mean()) < 1e-4
    G5.close(); dev.close(); mesh.close(); A.close()
