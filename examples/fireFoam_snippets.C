/*---------------------------------------------------------------------------*\
  fireFoam_snippets.C -- one time step of the reference's solver loop (solver/fireFoam.C:97-119) in which the four equation
  files are the REFERENCE'S OWN, included unchanged from where they lie under /root/reference/solver:

      #include "rhoEqn.H"   #include "UEqn.H"   #include "YEEqn.H"   #include "pEqn.H"      (and #include "phrghEqn.H", start-up)

  compiled against include/ffmFoam.H (the fvMesh / fvMatrix / fvm:: / fvc:: layer over the C ABI) and include/fireFoamHandles.H
  (the physics handles with the stand-ins of the synthetic plume case).  This file is what solver/createFields.H is to the
  reference: it declares the objects the snippets expect to find in scope.  Nothing of the reference is copied: the build
  (firefoam-dev_amd/csrc/Makefile, target refsnippets; only where /root/reference exists) passes -I/root/reference/solver,
  and the output libffm_refsnippets.so is git-ignored.

  tests/test_reference_snippets_cpu.py builds it (the "compiles unchanged" check of SURVEY 8b B1);
  tests/test_reference_snippets_gpu.py runs the step on the device and compares the fields with the oracle.
\*---------------------------------------------------------------------------*/
#include "fireFoamHandles.H"

using namespace Foam;

struct snippetCase          // host arrays: cell fields [N] in the library's cell order, face fields [F] LDU order, boundary [B]
{
    double deltaT;
    // constants of the synthetic case (oracle/plume.py)
    double RR, Cp, Tref, pRef, mu, Pr, sO2, HC, tau;
    int nSpecies, inertIndex, fuelIndex, o2Index;
    const double* W; const double* nu;
    // state at the start of the step
    const double* rho; const double* U; const double* p; const double* p_rgh; const double* h; const double* const* Y;
    const double* K; const double* dpdt; const double* phiF; const double* phiB;
    const double* gh; const double* ghfF; const double* ghfB;
    // boundary conditions: value fraction f (-1 = inletOutlet: 1 - pos0(phi_b)), refValue; U: 3 components
    const double* fU; const double* refU; const double* fixesU;
    const double* fY; const double* const* refY; const double* fH; const double* refH;
    const double* fluxMaskP; const double* totalMaskP; const double* ph_rgh_b; const double* p_rghB;
    // results
    double* rhoOut; double* UOut; double* pOut; double* p_rghOut; double* hOut; double* const* YOut; double* TOut; double* KOut;
    double* dpdtOut; double* phiOutF; double* phiOutB; double* p_rghBOut; int* nIterOut; int nIterCap;
};

// ---- what solver/createFields.H declares: fvSolution / fvSchemes of the synthetic case (cf. cases/steckler/system/fvSolution:19-101,
// fvSchemes:28-61), thermo / composition / Y, the fields, the physics handles; objects boundary conditions look up by name are
// registered with mesh.store() (inletOutlet: phi; prghTotalHydrostaticPressure: rho, U, phi)
#define FIREFOAM_CREATE_FIELDS(cs)                                                                                            \
    fvMesh mesh(ctx, ldu, msh, cs->deltaT); \
    const label N = mesh.nCells, B = mesh.nBoundary; \
    mesh.solvers["rho"] = mesh.solvers["rhoFinal"] = {FFM_DIAGONAL, FFM_NONE, 1e-6, 0, 0, 1000, 1}; \
    mesh.solvers["U"] = mesh.solvers["UFinal"] = {FFM_PBICGSTAB, FFM_DILU, 1e-6, 0, 0, 1000, 1}; \
    mesh.solvers["Yi"] = mesh.solvers["YiFinal"] = {FFM_PBICGSTAB, FFM_DILU, 1e-8, 0, 0, 1000, 1}; \
    mesh.solvers["h"] = mesh.solvers["hFinal"] = {FFM_PBICGSTAB, FFM_DILU, 1e-8, 0, 0, 1000, 1}; \
    mesh.solvers["p_rgh"] = {FFM_PCG, FFM_DIC, 1e-6, 0.01, 0, 1000, 1}; \
    mesh.solvers["p_rghFinal"] = {FFM_PCG, FFM_DIC, 1e-6, 0, 0, 1000, 1}; \
    if (std::getenv("FFM_PLUME_TIGHT")) for (auto& kv : mesh.solvers) if (kv.second.solver != FFM_DIAGONAL) { kv.second.tolerance = 1e-13; kv.second.relTol = 0; } \
    mesh.divSchemes["div(phi,U)"] = {4, 1, 0, 1}; \
    mesh.divSchemes["div(phi,K)"] = {2, 1, 0, 1}; \
    mesh.divSchemes["div(phiv,p)"] = {2, 1, 0, 1}; \
    mesh.multivariateSelection["div(phi,Yi_h)"]["h"] = {2, 1, 0, 1}; \
    pimpleDict pd; pd.nOuterCorrectors = 1; pd.nCorrectors = 2; pd.nNonOrthogonalCorrectors = 0; pd.hydrostaticInitialization = true; pd.nHydrostaticCorrectors = 5; \
    pimpleControl pimple(mesh, pd); \
 \
    std::vector<double> zeroBh(B, 0.0), oneBh(B, 1.0); \
    auto zeroGradient = [&]() { return std::make_shared<mixedBC>(ctx, B, zeroBh.data(), zeroBh.data(), zeroBh.data()); }; \
    perfectGasConstCpThermo thermoObj(mesh, cs->RR, cs->Cp, cs->Tref); \
    psiReactionThermo& thermo = thermoObj; \
    basicMultiComponentMixture& composition = thermo.composition(); \
    PtrList<volScalarField>& Y = composition.Y(); \
    static const char* specieNames[] = {"O2", "H2O", "C3H8", "CO2", "N2"}; \
    for (label i = 0; i < cs->nSpecies; i++) { \
        composition.species_.push_back(specieNames[i]); composition.active_.push_back(true); \
        thermoObj.W_.push_back(cs->W[i]); \
        Y.append(new volScalarField(specieNames[i], mesh)); \
        Y[i].v.assignHost(cs->Y[i]); \
        Y[i].bc = std::make_shared<mixedBC>(ctx, B, cs->fY, cs->refY[i], zeroBh.data()); \
        mesh.multivariateSelection["div(phi,Yi_h)"][specieNames[i]] = {3, 1, 0, 1}; \
    } \
    const label inertIndex = cs->inertIndex; \
    volScalarField& p = thermo.p(); p.v.assignHost(cs->p); p.b = mesh.patchInternal(p.v); \
    volScalarField& T = thermo.T(); \
    const volScalarField& psi = thermo.psi(); \
    thermo.he().v.assignHost(cs->h); \
    thermo.he().bc = std::make_shared<mixedBC>(ctx, B, cs->fH, cs->refH, zeroBh.data()); \
 \
    volScalarField rho("rho", mesh); rho.v.assignHost(cs->rho); rho.bc = zeroGradient(); rho.correctBoundaryConditions(); \
    volVectorField U("U", mesh); \
    for (int d = 0; d < 3; d++) { U.v[d].assignHost(cs->U + (size_t)d*N); U.bc[d] = std::make_shared<mixedBC>(ctx, B, cs->fU + (size_t)d*B, cs->refU + (size_t)d*B, zeroBh.data()); } \
    U.fixesValue = std::make_shared<dField>(ctx, B); U.fixesValue->assignHost(cs->fixesU); \
    surfaceScalarField phi(mesh); \
    FFM_FOAM_CHK(ffm_faces_to_native(msh, cs->phiF, phi.v.data())); phi.b.assignHost(cs->phiB); \
    volScalarField p_rgh("p_rgh", mesh); p_rgh.v.assignHost(cs->p_rgh); p_rgh.b.assignHost(cs->p_rghB); \
    p_rgh.bc = std::make_shared<mixedBC>(ctx, B, cs->totalMaskP, zeroBh.data(), zeroBh.data()); \
    p_rgh.bc->totalMask = std::make_shared<dField>(ctx, B); p_rgh.bc->totalMask->assignHost(cs->totalMaskP); \
    p_rgh.bc->ph_rgh_b = std::make_shared<dField>(ctx, B); p_rgh.bc->ph_rgh_b->assignHost(cs->ph_rgh_b); \
    p_rgh.fixedFluxMask = std::make_shared<dField>(ctx, B); p_rgh.fixedFluxMask->assignHost(cs->fluxMaskP); \
    volScalarField gh("gh", mesh); gh.v.assignHost(cs->gh); \
    surfaceScalarField ghf(mesh); \
    FFM_FOAM_CHK(ffm_faces_to_native(msh, cs->ghfF, ghf.v.data())); ghf.b.assignHost(cs->ghfB); \
    const dimensionedScalar pRef("pRef", cs->pRef); \
    volScalarField K("K", mesh); K.v.assignHost(cs->K); \
    volScalarField dpdt("dpdt", mesh); dpdt.v.assignHost(cs->dpdt); \
    volScalarField Qdot("Qdot", mesh); \
    multivariateSurfaceInterpolationScheme<scalar>::fieldTable fields; \
    forAll(Y, i) { fields.add(Y[i]); } \
    fields.add(thermo.he()); \
 \
    autoPtr<compressible::turbulenceModel> turbulence(new constantViscosity(mesh, cs->mu, cs->Pr)); \
    std::vector<scalar> nu(cs->nu, cs->nu + cs->nSpecies); \
    autoPtr<combustionModels::psiCombustionModel> combustion(new singleStepEDC(thermo, rho, cs->fuelIndex, cs->o2Index, cs->sO2, cs->tau, cs->HC, nu)); \
    autoPtr<radiation::radiationModel> radiation(new noRadiation()); \
    noParcels parcels(mesh); \
    noSurfaceFilm surfaceFilm(mesh); \
    noFvOptions fvOptions; \
    noMRF MRF; \
    const scalar lewisNo = 1; \
    dimensionedScalar DM("DM", 0.0); \
    const bool constD = false; \
    scalar cumulativeContErr = 0; \
 \
    mesh.store("phi", phi); mesh.store("rho", rho); mesh.store("U", U); \
    (void)0

extern "C" int firefoam_snippets_step(ffm_ctx* ctx, ffm_ldu* ldu, ffm_mesh* msh, const snippetCase* cs)
{
    FIREFOAM_CREATE_FIELDS(cs);
    thermo.correct();                                           // T, psi of the start state
    U.correctBoundaryConditions(); thermo.he().correctBoundaryConditions();
    forAll(Y, i) { Y[i].correctBoundaryConditions(); }
    // ---- runTime++: old-time levels
    rho.storeOldTime(); U.storeOldTime(); thermo.he().storeOldTime(); K.storeOldTime(); p.storeOldTime(); p_rgh.storeOldTime();
    thermoObj.psi_.storeOldTime(); phi.storeOldTime();
    forAll(Y, i) { Y[i].storeOldTime(); }

    // ---- solver/fireFoam.C:97-119 -----------------------------------------------------------------------------------------
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

    // ---- results
    rho.v.toHost(cs->rhoOut); p.v.toHost(cs->pOut); p_rgh.v.toHost(cs->p_rghOut); thermo.he().v.toHost(cs->hOut); T.v.toHost(cs->TOut);
    K.v.toHost(cs->KOut); dpdt.v.toHost(cs->dpdtOut);
    for (int d = 0; d < 3; d++) U.v[d].toHost(cs->UOut + (size_t)d*N);
    forAll(Y, i) { Y[i].v.toHost(cs->YOut[i]); }
    FFM_FOAM_CHK(ffm_faces_from_native(msh, phi.v.data(), cs->phiOutF)); phi.b.toHost(cs->phiOutB); p_rgh.b.toHost(cs->p_rghBOut);
    FFM_FOAM_CHK(ffm_ctx_sync(ctx));
    int n = 0;
    for (const solverPerformance& sp : mesh.log) if (sp.fieldName != "rho" && n < cs->nIterCap) cs->nIterOut[n++] = sp.nIterations;   // (diagonal solves left out)
    (void)psi; (void)inertIndex;
    return n;
}


// solver/createFields.H:100-104 -> solver/phrghEqn.H, the reference's file included unchanged: hydrostatic initialisation
// (nHydrostaticCorrectors solves of fvm::laplacian(rhof, ph_rgh) == fvc::div(phig)).  Inputs: p (uniform pRef), h, Y, U = 0;
// ph_rgh "file": fixedValue 0 where totalMaskP = 1 (the top), fixedFluxPressure where fluxMaskP = 1.
// Results: p_rghOut (= ph_rgh), pOut, rhoOut, p_rghBOut (boundary values of ph_rgh).
extern "C" int firefoam_snippets_hydrostatic(ffm_ctx* ctx, ffm_ldu* ldu, ffm_mesh* msh, const snippetCase* cs)
{
    FIREFOAM_CREATE_FIELDS(cs);
    mesh.solvers["ph_rgh"] = {FFM_PCG, FFM_DIC, 1e-6, 0.01, 0, 1000, 1};           // cases/steckler/system/fvSolution:43-46
    if (std::getenv("FFM_PLUME_TIGHT")) { mesh.solvers["ph_rgh"].tolerance = 1e-13; mesh.solvers["ph_rgh"].relTol = 0; }
    Time runTime(0);
    {
        std::shared_ptr<volScalarField> f(new volScalarField("ph_rgh", mesh));
        f->bc = std::make_shared<mixedBC>(ctx, B, cs->totalMaskP, zeroBh.data(), zeroBh.data());
        f->fixedFluxMask = std::make_shared<dField>(ctx, B); f->fixedFluxMask->assignHost(cs->fluxMaskP);
        mesh.fieldFiles["ph_rgh"] = f;                                              // the case's 0/ph_rgh
    }
    thermo.correct();
    rho = thermo.rho();

    #include "phrghEqn.H"

    rho.v.toHost(cs->rhoOut); p.v.toHost(cs->pOut); p_rgh.v.toHost(cs->p_rghOut); p_rgh.b.toHost(cs->p_rghBOut);
    FFM_FOAM_CHK(ffm_ctx_sync(ctx));
    int n = 0;
    for (const solverPerformance& sp : mesh.log) if (n < cs->nIterCap) cs->nIterOut[n++] = sp.nIterations;
    (void)psi; (void)inertIndex; (void)T; (void)N; (void)lewisNo; (void)constD; (void)cumulativeContErr; (void)fvOptions; (void)MRF;
    return n;
}
