/*---------------------------------------------------------------------------*\
  fireFoam_steckler.C -- the reference's steckler room-fire case (cases/steckler) on the device, start-up and time steps, with the
  REFERENCE'S OWN equation files included unchanged from /root/reference/solver (as in fireFoam_snippets.C) and the case's REAL
  physics plug-ins behind the handles of include/fireFoamHandles.H instead of the stand-ins of the synthetic plume:

      thermo        hePsiThermoJanaf      hePsiThermo<reactingMixture<sutherland<janaf<perfectGas>>>>   (constant/thermophysicalProperties:18-27)
      turbulence    kEqnLES               LES kEqn, delta cubeRootVol                                   (constant/turbulenceProperties:18-30)
      combustion    eddyDissipationEDC    the reference's eddyDissipationModel on the single-step mixture (constant/combustionProperties, reactions)
      radiation     fvDOM                 32 rays, constRadFractionEmission, greyDiffusiveRadiation walls, Ii by GAMG + DILU   (constant/radiationProperties)
      patch types   flowRateInletVelocity (0/U:40-54), totalFlowRateAdvectiveDiffusive (0/C3H8:44-50), inletOutlet, pressureInletOutletVelocity,
                    prghTotalHydrostaticPressure, fixedFluxPressure, mixedEnergy / fixedEnergy with compressible::thermalBaffle1D (0/T:51-82)

  This file is what solver/createFields.H + the time loop of solver/fireFoam.C:76-121 are to the reference.  The sequence is the
  reference's: createFields (thermo construction from T, rho = thermo.rho(), hydrostatic initialisation through solver/phrghEqn.H,
  turbulence->validate()), then per step rhoEqn.H, UEqn.H, YEEqn.H, 2 x pEqn.H, turbulence->correct().  Golden data: the first
  time step of cases/steckler/original/linux64/log.fireFoam:163-226 (tests/golden/steckler_first_step.json);
  tests/test_steckler_case_gpu.py runs it and compares every printed solve.

  Built like fireFoam_snippets.C: only where /root/reference exists, with -I/root/reference/solver; nothing is copied.
\*---------------------------------------------------------------------------*/
#include "fireFoamHandles.H"

using namespace Foam;

struct stecklerCaseData      // host arrays: cell fields [N] in the library's cell order, face fields [F] in LDU order, boundary [B]
{
    double deltaT, pRef, RR;
    int nSpecies, inertIndex, fuelIndex, o2Index;
    const char* const* specieNames;
    // thermo table (constant/thermo.compressibleGas) and the reaction's stoichiometric coefficients (negative: reactant)
    const double *W, *Tlow, *Thigh, *Tcommon, *highCpCoeffs, *lowCpCoeffs, *As, *Ts, *stoich;
    // 0/ files: internal values and the stored patch values after construction
    const double* T; const double* TB; const double* const* Y; const double* const* YB;
    double k0, nut0;
    // geometry-derived fields of createFields.H / readGravitationalAcceleration.H: gh, ghf, LES delta (cubeRootVol)
    const double *gh, *ghfF, *ghfB, *delta;
    // patch conditions (mixed form; f = -1: inletOutlet)
    const double *fU, *refU, *fixesU, *flowRateMask, *flowRateNf; double massFlowRate;
    const double* const* fY; const double* const* refY; const double* const* tfradMask;     // per specie; tfradMask[i] may be NULL
    const double *fK, *refK;
    const double *fixesT, *inletOutletT; double Tinlet; long baffleMaster0, baffleSlave0, nBaffle; double baffleThickness, baffleQs, baffleKappa;
    const double *phTopMask, *phFluxMask, *fluxMaskP, *totalMaskP;      // ph_rgh: fixedValue 0 / fixedFluxPressure; p_rgh: fixedFluxPressure / prghTotalHydrostaticPressure
    const double *nutZeroGrad, *alphatZeroGrad;
    // radiation: gamg != NULL selects the case's fvDOM (radiationProperties: nPhi 2, nTheta 4, solverFreq; constRadFractionEmission
    // a = 0, Ehrr1 0.5, Ehrr2 0.22 over the burner faces = mlrMask; Ii by GAMG + DILU on the agglomeration gamg); NULL: noRadiation
    ffm_gamg* gamg; int radiationFreq; const double* mlrMask; double sigmaSB;
    // results
    double *rhoOut, *UOut, *pOut, *p_rghOut, *hOut, *TOut, *kOut, *phiOutF; double* const* YOut;
    int* nIterOut; double* resOut; char* namesOut; int logCap;          // per solve: iterations, {initial, final} residual, field name [16]
    double* contErrOut;                                                  // cumulative continuity error (compressibleContinuityErrs.H)
};

struct stecklerSolver
{
    ffm_ctx* ctx; ffm_mesh* msh;
    fvMesh mesh;
    const label N, B;
    std::vector<double> zeroBh;
    pimpleControl pimple;
    ffm_thermo* th;
    hePsiThermoJanaf thermoObj;
    psiReactionThermo& thermo;
    basicMultiComponentMixture& composition;
    PtrList<volScalarField>& Y;
    const label inertIndex;
    volScalarField& p;
    volScalarField& T;
    const volScalarField& psi;
    volScalarField rho;
    volVectorField U;
    surfaceScalarField phi;
    volScalarField p_rgh, gh;
    surfaceScalarField ghf;
    const dimensionedScalar pRef;
    volScalarField K, dpdt, Qdot;
    multivariateSurfaceInterpolationScheme<scalar>::fieldTable fields;
    autoPtr<compressible::turbulenceModel> turbulence;
    autoPtr<combustionModels::psiCombustionModel> combustion;
    autoPtr<radiation::radiationModel> radiation;
    noParcels parcels;
    noSurfaceFilm surfaceFilm;
    noFvOptions fvOptions;
    noMRF MRF;
    const scalar lewisNo = 1;                                   // solver/readAdditionalThermo.H:32
    dimensionedScalar DM;
    const bool constD = false;
    scalar cumulativeContErr = 0;
    Time runTime;
    struct { bool adjustTimeStep; scalar maxCo, maxDeltaT; } timeControls{true, 0.9, 0.1};      // cases/steckler/system/controlDict:50-56
    pyrolysisModelCollection pyrolysis;                         // selection `none` (log.fireFoam:119)
    const scalar maxDi = 0.25;                                  // controlDict:54
    const bool solvePyrolysisRegion = true, solvePrimaryRegion = true;
    kEqnLES* les = nullptr;

    static pimpleDict pimpleOf()
    {
        pimpleDict pd; pd.nOuterCorrectors = 1; pd.nCorrectors = 2; pd.nNonOrthogonalCorrectors = 0;      // cases/steckler/system/fvSolution:84-92
        pd.hydrostaticInitialization = true; pd.nHydrostaticCorrectors = 5;
        return pd;
    }
    static ffm_thermo* thermoOf(ffm_ctx* c, const stecklerCaseData* cs)
    {
        ffm_thermo* t = nullptr;
        FFM_FOAM_CHK(ffm_thermo_create(c, cs->nSpecies, cs->W, cs->Tlow, cs->Thigh, cs->Tcommon, cs->highCpCoeffs, cs->lowCpCoeffs, cs->As, cs->Ts, cs->RR, &t));
        return t;
    }

    stecklerSolver(ffm_ctx* c, ffm_ldu* ldu, ffm_mesh* m, const stecklerCaseData* cs)
    : ctx(c), msh(m), mesh(c, ldu, m, cs->deltaT), N(mesh.nCells), B(mesh.nBoundary), zeroBh(B, 0.0), pimple(mesh, pimpleOf()),
      th(thermoOf(c, cs)), thermoObj(mesh, th, cs->fixesT), thermo(thermoObj), composition(thermo.composition()), Y(composition.Y()),
      inertIndex(cs->inertIndex), p(thermo.p()), T(thermo.T()), psi(thermo.psi()), rho("rho", mesh), U("U", mesh), phi(mesh),
      p_rgh("p_rgh", mesh), gh("gh", mesh), ghf(mesh), pRef("pRef", cs->pRef), K("K", mesh), dpdt("dpdt", mesh), Qdot("Qdot", mesh),
      parcels(mesh), surfaceFilm(mesh), DM("DM", 0.0), runTime(0, cs->deltaT)
    {
        runTime.link(mesh.deltaT, mesh.timeValue);
        // ---- system/fvSolution:19-62, system/fvSchemes:28-61
        const solverControls smooth6{FFM_SMOOTH, FFM_SYMGS, 1e-6, 0, 0, 10, 1}, smooth8{FFM_SMOOTH, FFM_SYMGS, 1e-8, 0, 0, 10, 1};
        mesh.solvers["rho"] = mesh.solvers["rhoFinal"] = {FFM_DIAGONAL, FFM_NONE, 1e-6, 0, 0, 1000, 1};
        mesh.solvers["U"] = mesh.solvers["UFinal"] = smooth6;
        for (const char* n : {"Yi", "YiFinal", "h", "hFinal", "k", "kFinal"}) mesh.solvers[n] = smooth8;
        mesh.solvers["p_rgh"] = mesh.solvers["ph_rgh"] = {FFM_PCG, FFM_DIC, 1e-6, 0.01, 0, 1000, 1};
        mesh.solvers["p_rghFinal"] = {FFM_PCG, FFM_DIC, 1e-6, 0, 0, 1000, 1};
        mesh.divSchemes["div(phi,U)"] = {4, 1, 0, 1};                                   // Gauss LUST grad(U)
        mesh.divSchemes["div(phi,k)"] = mesh.divSchemes["div(phi,K)"] = {2, 1, 0, 1};   // Gauss limitedLinear 1
        mesh.multivariateSelection["div(phi,Yi_h)"]["h"] = {2, 1, 0, 1};
        // ---- thermo: species, Y, T, p; he = Hs(p, T) on cells and faces (heThermo's constructor); calculate()
        for (label i = 0; i < cs->nSpecies; i++) {
            composition.species_.push_back(cs->specieNames[i]); composition.active_.push_back(true);
            Y.append(new volScalarField(cs->specieNames[i], mesh));
            Y[i].v.assignHost(cs->Y[i]); Y[i].b.assignHost(cs->YB[i]);
            if (i != inertIndex) {
                Y[i].bc = std::make_shared<mixedBC>(ctx, B, cs->fY[i], cs->refY[i], zeroBh.data());
                if (cs->tfradMask && cs->tfradMask[i]) { Y[i].bc->tfradMask = std::make_shared<dField>(ctx, B); Y[i].bc->tfradMask->assignHost(cs->tfradMask[i]); }
            }
            mesh.multivariateSelection["div(phi,Yi_h)"][cs->specieNames[i]] = {3, 1, 0, 1};                // Yi limitedLinear01 1
        }
        T.v.assignHost(cs->T); T.b.assignHost(cs->TB);
        p = volScalarField("p", mesh, cs->pRef);
        thermoObj.initHe();
        thermoObj.installEnergyPatches(cs->inletOutletT, cs->Tinlet, cs->baffleMaster0, cs->baffleSlave0, cs->nBaffle, cs->baffleThickness, cs->baffleQs, cs->baffleKappa);
        thermo.correct();
        rho = thermo.rho();                                         // `calculated` patches: no boundary condition object
        // ---- U, phi, p_rgh, gh, ghf
        for (int d = 0; d < 3; d++) {
            U.bc[d] = std::make_shared<mixedBC>(ctx, B, cs->fU + (size_t)d*B, cs->refU + (size_t)d*B, zeroBh.data());
            U.bc[d]->flowRateMask = std::make_shared<dField>(ctx, B); U.bc[d]->flowRateMask->assignHost(cs->flowRateMask);
            U.bc[d]->flowRateNf = std::make_shared<dField>(ctx, B); U.bc[d]->flowRateNf->assignHost(cs->flowRateNf + (size_t)d*B);
            U.bc[d]->massFlowRate = Function1Table(cs->massFlowRate);
        }
        U.fixesValue = std::make_shared<dField>(ctx, B); U.fixesValue->assignHost(cs->fixesU);
        p_rgh.bc = std::make_shared<mixedBC>(ctx, B, cs->totalMaskP, zeroBh.data(), zeroBh.data());
        p_rgh.bc->totalMask = std::make_shared<dField>(ctx, B); p_rgh.bc->totalMask->assignHost(cs->totalMaskP);
        p_rgh.bc->ph_rgh_b = std::make_shared<dField>(ctx, B, 0.0);
        p_rgh.fixedFluxMask = std::make_shared<dField>(ctx, B); p_rgh.fixedFluxMask->assignHost(cs->fluxMaskP);
        p_rgh.notAssignable = p_rgh.bc->totalMask;                  // fixedValue-derived patches keep their values in `p_rgh = ph_rgh`
        p_rgh.snGradFromGradient = true;
        gh.v.assignHost(cs->gh); gh.b.assignHost(cs->ghfB);
        FFM_FOAM_CHK(ffm_faces_to_native(msh, cs->ghfF, ghf.v.data())); ghf.b.assignHost(cs->ghfB);
        K.lazyOldTime = true; p.lazyOldTime = true;            // created by the first fvc::ddt(rho, K) / fvc::ddt(p), as copies of the current fields
        forAll(Y, i) { fields.add(Y[i]); }
        fields.add(thermo.he());
        mesh.store("phi", phi); mesh.store("rho", rho); mesh.store("U", U);
        // ---- turbulence, combustion, radiation
        les = new kEqnLES(mesh, rho, U, phi, thermoObj.mu(), thermoObj.alpha(), cs->nutZeroGrad, cs->alphatZeroGrad);
        les->k_ = volScalarField("k", mesh, cs->k0);
        les->k_.bc = std::make_shared<mixedBC>(ctx, B, cs->fK, cs->refK, zeroBh.data());
        les->nut_ = volScalarField("nut", mesh, cs->nut0);
        les->delta_.v.assignHost(cs->delta);
        turbulence = autoPtr<compressible::turbulenceModel>(les);
        const singleStepMixture rx(cs->nSpecies, cs->W, cs->lowCpCoeffs, cs->RR, cs->stoich, cs->fuelIndex, cs->o2Index);
        combustion = autoPtr<combustionModels::psiCombustionModel>(new eddyDissipationEDC(thermo, rho, *les, cs->fuelIndex, cs->o2Index, rx.s, rx.qFuel, rx.massCoeffs));
        if (cs->gamg) {
            mesh.gamg = cs->gamg;
            solverControls ii; ii.solver = FFM_GAMG; ii.preconditioner = FFM_DILU; ii.tolerance = 1e-4; ii.relTol = 0;      // fvSolution:63-73
            mesh.solvers["Ii"] = ii;
            mesh.divSchemes["div(Ji,Ii_h)"] = {0, 1, 0, 1};                               // Gauss upwind (fvSchemes:60)
            mesh.store("Qdot", Qdot);
            radiation = autoPtr<radiation::radiationModel>(new fvDOM(mesh, T, 2, 4, cs->radiationFreq, 0.0, cs->sigmaSB, 0.5, 0.22, cs->mlrMask));
        }
        else radiation = autoPtr<radiation::radiationModel>(new noRadiation());
    }
    ~stecklerSolver() { if (th) ffm_thermo_destroy(th); }

    // solver/createFields.H:100-104 -> solver/phrghEqn.H; then turbulence->validate() (solver/fireFoam.C:69)
    void startUp(const double* phTopMask, const double* fluxMask)
    {
        mesh.log.clear();
        {
            std::shared_ptr<volScalarField> f(new volScalarField("ph_rgh", mesh));
            f->bc = std::make_shared<mixedBC>(ctx, B, phTopMask, zeroBh.data(), zeroBh.data());
            f->fixedFluxMask = std::make_shared<dField>(ctx, B); f->fixedFluxMask->assignHost(fluxMask);
            mesh.fieldFiles["ph_rgh"] = f;
        }

        #include "phrghEqn.H"

        // prghTotalHydrostaticPressure refers to the field ph_rgh by name (0/p_rgh: ph_rgh ph_rgh): its patch values
        *p_rgh.bc->ph_rgh_b = mesh.lookupObject<volScalarField>("ph_rgh").b;
        les->correctNut();
        FFM_FOAM_CHK(ffm_ctx_sync(ctx));
    }

    // the body of the reference's time loop, solver/fireFoam.C:76-121: time-step control -- the reference's solidRegionDiffusionNo.H and
    // setMultiRegionDeltaT.H between this layer's restatements of the OpenFOAM headers, Time::setDeltaT adjusting towards the write
    // times -- then the step.  The golden log's deltaT sequence comes out of it (tests/test_steckler_case_gpu.py)
    void timeStep()
    {
        #include "readTimeControls.H"
        #include "compressibleCourantNo.H"
        #include "solidRegionDiffusionNo.H"
        #include "setMultiRegionDeltaT.H"
        #include "setDeltaT.H"

        runTime++;

        parcels.evolve();

        surfaceFilm.evolve();

        if(solvePyrolysisRegion)
        {
            pyrolysis.evolve();
        }

        if (solvePrimaryRegion)
        {
            step(false);
        }
        (void)meanCoNum;
    }

    // one time step: solver/fireFoam.C:84,97-119 with the reference's equation files
    void step(bool increment = true)
    {
        mesh.log.clear();
        if (increment) runTime++;
        // old-time levels (GeometricField::storeOldTimes at the time increment; K's is created on first request)
        rho.storeOldTime(); U.storeOldTime(); thermo.he().storeOldTime(); p_rgh.storeOldTime();
        thermoObj.psi_.storeOldTime(); phi.storeOldTime();
        if (K.old_) K.storeOldTime();
        if (p.old_) p.storeOldTime();
        forAll(Y, i) { Y[i].storeOldTime(); }

        #include "rhoEqn.H"

        // --- PIMPLE loop
        while (pimple.loop())
        {
            #include "UEqn.H"
            #include "YEEqn.H"

            // --- Pressure corrector loop
            while (pimple.correct())
            {
                #include "pEqn.H"
            }

            if (pimple.turbCorr())
            {
                turbulence->correct();
            }
        }

        rho = thermo.rho();
        FFM_FOAM_CHK(ffm_ctx_sync(ctx));
    }

    // solver/fireFoam.C:79: the Courant numbers the log prints in front of a step
    void courant(double* out)
    {
        #include "compressibleCourantNo.H"
        out[0] = meanCoNum; out[1] = CoNum;
    }

    void download(const stecklerCaseData* cs)
    {
        rho.v.toHost(cs->rhoOut); p.v.toHost(cs->pOut); p_rgh.v.toHost(cs->p_rghOut);
        thermo.he().v.toHost(cs->hOut); T.v.toHost(cs->TOut); les->k_.v.toHost(cs->kOut);
        for (int d = 0; d < 3; d++) U.v[d].toHost(cs->UOut + (size_t)d*N);
        forAll(Y, i) { Y[i].v.toHost(cs->YOut[i]); }
        FFM_FOAM_CHK(ffm_faces_from_native(msh, phi.v.data(), cs->phiOutF));
    }
    int logOut(const stecklerCaseData* cs)
    {
        int n = 0;
        for (const solverPerformance& sp : mesh.log) if (n < cs->logCap) {
            cs->resOut[2*n] = sp.initialResidual; cs->resOut[2*n + 1] = sp.finalResidual;
            std::snprintf(cs->namesOut + 16*n, 16, "%s", sp.fieldName.c_str());
            cs->nIterOut[n++] = sp.nIterations;
        }
        return n;
    }
};

// ---- C entry points -------------------------------------------------------------------------------------------------------
// create = createFields.H: the fields, the physics objects, the hydrostatic initialisation (5 DICPCG solves: cs->nIterOut etc.
// hold them afterwards) and turbulence->validate()
extern "C" stecklerSolver* firefoam_steckler_create(ffm_ctx* ctx, ffm_ldu* ldu, ffm_mesh* msh, const stecklerCaseData* cs, int* nSolves)
{
    std::cout.precision(8);              // Time::readDict: writePrecision 8 (cases/steckler/system/controlDict:36-38)
    stecklerSolver* s = new stecklerSolver(ctx, ldu, msh, cs);
    s->startUp(cs->phTopMask, cs->phFluxMask);
    if (nSolves) *nSolves = s->logOut(cs);
    return s;
}
extern "C" void firefoam_steckler_destroy(stecklerSolver* s) { delete s; }
// the Courant numbers {mean, max} of the current flux (compressibleCourantNo.H) and the time step of the next advance(): the
// reference's deltaT sequence (setMultiRegionDeltaT.H + setDeltaT.H + Time::adjustDeltaT towards the write times, asserted on the
// golden log by tests/test_steckler_first_step_cpu.py) is handed over by the caller
extern "C" void firefoam_steckler_courant(stecklerSolver* s, double* out) { s->courant(out); }
extern "C" void firefoam_steckler_set_delta_t(stecklerSolver* s, double deltaT) { s->runTime.setDeltaT(deltaT); }
// the whole loop body with the reference's time-step control: writeControl adjustableRunTime, writeInterval (controlDict:30-32) switch
// Time::adjustDeltaT on; returns like firefoam_steckler_advance, the deltaT used in *deltaTOut
extern "C" int firefoam_steckler_time_step(stecklerSolver* s, const stecklerCaseData* cs, int download, double writeInterval, double* deltaTOut)
{
    s->runTime.setWriteInterval(writeInterval);
    s->timeStep();
    if (deltaTOut) *deltaTOut = s->runTime.deltaTValue();
    if (download) s->download(cs);
    if (cs->contErrOut) cs->contErrOut[0] = s->cumulativeContErr;
    return s->logOut(cs);
}
// one time step; returns the number of linear solves (names, iteration counts and residuals in cs->namesOut / nIterOut / resOut)
extern "C" int firefoam_steckler_advance(stecklerSolver* s, const stecklerCaseData* cs, int download)
{
    s->step();
    if (download) s->download(cs);
    if (cs->contErrOut) cs->contErrOut[0] = s->cumulativeContErr;
    return s->logOut(cs);
}
