"""ORACLE (test infrastructure): one fireFoam time step (PIMPLE, nOuterCorrectors 1, nCorrectors 2,
nNonOrthogonalCorrectors 0 -- reference cases/steckler/system/fvSolution:84-89) on the synthetic
buoyant-plume box of SURVEY 8(d), restated in numpy with oracle/fv.py operators and the C solvers.

It follows the reference's equation snippets term by term:
  rhoEqn   solver/rhoEqn.H:33-43      fvm::ddt(rho) + fvc::div(phi) == 0                  (diagonalSolver)
  UEqn     solver/UEqn.H:3-33         fvm::ddt(rho,U) + fvm::div(phi,U) + divDevRhoReff(U); solve(UEqn ==
                                      fvc::reconstruct((-ghf*snGrad(rho) - snGrad(p_rgh))*magSf)); K
  YEEqn    solver/YEEqn.H:37-71       per specie: ddt + mvConvection->fvmDiv - laplacian(dEff) == R(Yi); max(0); inert
           solver/YEEqn.H:84-118      he: ddt + fvmDiv + fvc::ddt(rho,K) + fvc::div(phi,K) - dpdt - laplacian(alphaEff)
                                      == Qdot; thermo.correct()
  pEqn     solver/pEqn.H:1-60         rAU, rhorAUf, HbyA, phig, phiHbyA (flux + ddtCorr), constrainPressure, p_rghEqn,
                                      phi, U, p, rhoEqn, K, dpdt                              (x2; second uses p_rghFinal)
  start-up solver/phrghEqn.H:25-56    hydrostatic initialisation, 5 correctors
The physics plug-ins the reference gets from other libraries are replaced by documented stand-ins
(SURVEY 2.2 U18, section 7 step 6): perfect gas with constant Cp and per-cell mixture molecular
weight; constant mu, Pr, Le = 1 (laminar: divDevRhoReff(U) = -fvm::laplacian(mu,U)); an
EDC-shaped single-step source R = rho*min(Yfuel, YO2/s)/tau; no radiation, spray or film; boundary
values of the thermo fields (rho, psi, T-derived) are zero-gradient copies of the cell values.
Schemes: Euler; Gauss limitedLinear 1 for U (through magSqr(U), as LimitedScheme<vector,...,magSqr>),
K and h; limitedLinear01 1 for species; Gauss linear uncorrected laplacians; linear interpolation
(cases/steckler/system/fvSchemes:18-76 with LUST replaced by limitedLinear for div(phi,U)).
"""
import numpy as np

from . import fv, oracle as O

RR = 8314.47
SPECIES = ["O2", "H2O", "C3H8", "CO2", "N2"]          # N2 inert (cases/steckler/constant/thermophysicalProperties:28)
WMOL = np.array([31.9988, 18.0153, 44.0962, 44.01, 28.0134])
Y_AMB = np.array([0.23301, 0.0, 0.0, 0.0, 0.76699])    # cases/steckler/0/{O2,N2}
Y_IN = np.array([0.0, 0.0, 1.0, 0.0, 0.0])            # pure fuel at the inlet patch
INERT = 4
CP, TREF, PREF = 1005.0, 298.15, 101325.0
MU, PR = 1.8e-5, 0.7
S_O2 = 3.6282945                                        # golden log: stoichiometric oxygen-fuel ratio
HC = 46357151.0                                         # golden log: fuel heat of combustion
TAU = 0.05                                              # mixing time of the EDC-shaped stand-in [s]
# products per kg fuel (C3H8 + 5 O2 -> 3 CO2 + 4 H2O)
NU = np.array([-S_O2, 4 * 18.0153 / 44.0962, -1.0, 3 * 44.01 / 44.0962, 0.0])
T_IN, U_IN = 600.0, 0.5
G = np.array([0.0, -9.81, 0.0])


def make_mesh(n, h=0.05, empty=()):
    """empty: boundary groups left out of every patch, i.e. OpenFOAM's `empty` patches of a 2-D case (their faces take part in
    no operator; cases/wallFireSpread2D/system/blockMeshDict:41 is one cell thick in x)"""
    nx, ny, nz = n
    m = fv.HexMesh(n, (0, 0, 0), (nx * h, ny * h, nz * h))
    # split ymin into inlet (central patch, 1 m^2 or the central quarter on small boxes) and floor
    ymin = m._bdefs["ymin"]
    Lx, Lz = nx * h, nz * h
    hw = (min(0.5, Lx / 4), min(0.5, Lz / 4))
    cen = ymin.Cf
    isin = (np.abs(cen[:, 0] - Lx / 2) < hw[0]) & (np.abs(cen[:, 2] - Lz / 2) < hw[1])
    for name, sel in (("inlet_part", isin), ("floor_part", ~isin)):
        m._bdefs[name] = fv.Patch(name, ymin.faceCells[sel], ymin.Sf[sel], ymin.Cf[sel], ymin.deltaCoeffs[sel])
    m.set_patches([("inlet", ["inlet_part"]), ("floor", ["floor_part"]), ("top", ["ymax"]),
                   ("sides", [g for g in ("xmin", "xmax", "zmin", "zmax") if g not in empty])])
    m.solutionD = [-1 if ("xmin" in empty and "xmax" in empty) else 1, 1, -1 if ("zmin" in empty and "zmax" in empty) else 1]
    return m


class Solvers:
    """lduMatrix::solver selection as in fvSolution; `solve(kind, mesh, diag, upper, lower, source, psi0)`."""
    CONTROLS = {
        "p_rgh": dict(solver="PCG", pre="DIC", tolerance=1e-6, relTol=0.01),
        "p_rghFinal": dict(solver="PCG", pre="DIC", tolerance=1e-6, relTol=0.0),
        "ph_rgh": dict(solver="PCG", pre="DIC", tolerance=1e-6, relTol=0.01),
        "U": dict(solver="PBICGSTAB", pre="DILU", tolerance=1e-6, relTol=0.0),
        "Yi": dict(solver="PBICGSTAB", pre="DILU", tolerance=1e-8, relTol=0.0),
        "h": dict(solver="PBICGSTAB", pre="DILU", tolerance=1e-8, relTol=0.0),
        "Ii": dict(solver="PBICGSTAB", pre="DILU", tolerance=1e-4, relTol=0.0),      # reference: GAMG + DILU smoother, 1e-4
    }

    def __init__(self):
        self.log = []

    def solve(self, kind, name, mesh, diag, upper, lower, source, psi0):
        c = self.CONTROLS[kind]
        A = O.Ldu(mesh.nCells, mesh.l, mesh.u).set_coeffs(diag, upper, None if c["solver"] == "PCG" else lower)
        psi, perf = A.solve(getattr(O, c["solver"]), getattr(O, c["pre"]), psi0, source, tolerance=c["tolerance"], relTol=c["relTol"],
                            maxIter=c.get("maxIter", 1000))
        self.log.append((name, perf))
        return psi


class StecklerSolvers(Solvers):
    """the transport-equation selection of cases/steckler/system/fvSolution:49-62: smoothSolver + symGaussSeidel, maxIter 10"""
    CONTROLS = dict(Solvers.CONTROLS,
                    U=dict(solver="SMOOTH", pre="SYMGS", tolerance=1e-6, relTol=0.0, maxIter=10),
                    Yi=dict(solver="SMOOTH", pre="SYMGS", tolerance=1e-8, relTol=0.0, maxIter=10),
                    h=dict(solver="SMOOTH", pre="SYMGS", tolerance=1e-8, relTol=0.0, maxIter=10))


class WallFireSolvers(Solvers):
    """the selection of cases/wallFireSpread2D/system/fvSolution (BASELINE config 5): p_rgh / ph_rgh GAMG + GaussSeidel 1e-5 relTol
    0.01 (:36-47), p_rghFinal 1e-6 relTol 0 (:49-60), U / Yi / h PBiCG + DILU (:67-75,115-152; the momentum solve of a one-outer-
    corrector PIMPLE step reads UFinal: 1e-7).  The multigrid hierarchy depends on the cell numbering, so the GAMG solves run in
    the numbering the device uses (cellOrder / faceOrder = its new -> old maps, (l2, u2, Sf2) the addressing and face areas in
    that numbering): one cached agglomeration per mesh (cacheAgglomeration true)."""
    CONTROLS = dict(Solvers.CONTROLS,
                    p_rgh=dict(solver="GAMG", pre="GaussSeidel", tolerance=1e-5, relTol=0.01),
                    ph_rgh=dict(solver="GAMG", pre="GaussSeidel", tolerance=1e-5, relTol=0.01),
                    p_rghFinal=dict(solver="GAMG", pre="GaussSeidel", tolerance=1e-6, relTol=0.0),
                    U=dict(solver="PBICG", pre="DILU", tolerance=1e-7, relTol=0.0),
                    Yi=dict(solver="PBICG", pre="DILU", tolerance=1e-8, relTol=0.0),
                    h=dict(solver="PBICG", pre="DILU", tolerance=1e-8, relTol=0.0),
                    Ii=dict(solver="GAMG", pre="DILU", tolerance=1e-4, relTol=0.0))          # fvSolution:160-170

    def __init__(self, cellOrder, faceOrder, l2, u2, Sf2):
        Solvers.__init__(self)
        from . import gamg
        self.cOrd, self.fOrd = np.asarray(cellOrder), np.asarray(faceOrder)
        self.agg = gamg.Agglomeration(len(self.cOrd), l2, u2, gamg.face_area_pair_weights(Sf2))
        self.gamg = gamg

    def solve(self, kind, name, mesh, diag, upper, lower, source, psi0):
        c = self.CONTROLS[kind]
        if c["solver"] != "GAMG":
            return Solvers.solve(self, kind, name, mesh, diag, upper, lower, source, psi0)
        asym = lower is not None and kind == "Ii"
        G = self.gamg.GAMGSolver(self.agg, diag[self.cOrd], upper[self.fOrd], lower[self.fOrd] if asym else None, smoother=c["pre"])
        x, perf = G.solve(psi0[self.cOrd], source[self.cOrd], tolerance=c["tolerance"], relTol=c["relTol"])
        psi = np.empty_like(x); psi[self.cOrd] = x
        self.log.append((name, perf))
        return psi


# ---- fvDOM stand-in: the ray set and the per-ray transport equation of the reference's
# packages/thermophysicalModels/radiation/radiationModels/fvDOM (fvDOM/fvDOM.C:55-90, radiativeIntensityRay/
# radiativeIntensityRay.C:126-143,267-322), with a constant absorption coefficient, no scattering, no emission term E
# and the intensity decoupled from the enthalpy equation (radiation->Sh = 0).
SIGMA_SB = 5.670367e-8
K_ABS = 0.1                                                   # 1/m


def ray_set(nPhi=2, nTheta=4):
    """(dAve, omega) of the 4*nPhi*nTheta rays, theta outer / phi inner as fvDOM.C builds them."""
    dPhi, dTheta = np.pi / (2.0 * nPhi), np.pi / nTheta
    rays = []
    for n in range(1, nTheta + 1):
        theta = (2.0 * n - 1.0) * dTheta / 2.0
        for mm in range(1, 4 * nPhi + 1):
            phi = (2.0 * mm - 1.0) * dPhi / 2.0
            omega = 2.0 * np.sin(theta) * np.sin(dTheta / 2.0) * dPhi
            dAve = np.array([np.sin(phi) * np.sin(0.5 * dPhi) * (dTheta - np.cos(2.0 * theta) * np.sin(dTheta)),
                             np.cos(phi) * np.sin(0.5 * dPhi) * (dTheta - np.cos(2.0 * theta) * np.sin(dTheta)),
                             0.5 * dPhi * np.sin(2.0 * theta) * np.sin(dTheta)])
            rays.append((dAve, omega))
    return rays


def conditioned_state(m, Y_amb, h_amb, eps=1e-3):
    """A start state in which no transported field is uniform anywhere: every specie and h carries a smooth, monotone variation of
    relative size eps over the box (its own direction and curvature per field, so that no two fields are multiples of one another),
    the inert specie takes the remainder.  For the multi-step parity tests: the limiter r = 2 (d.gradc)/(phi_N - phi_P) - 1 of
    NVDTVD is then formed from differences ~1e-5 of the field instead of solver noise, i.e. well-conditioned on every face."""
    lo, hi = m.C.min(axis=0), m.C.max(axis=0)
    s = (m.C - lo) / np.where(hi > lo, hi - lo, 1.0)
    x, y, z = s[:, 0], s[:, 1], s[:, 2]
    dirs = np.array([[1.0, 0.6, 0.8], [0.7, 1.0, 0.5], [0.5, 0.8, 1.0], [0.9, 0.5, 0.7], [0.0, 0.0, 0.0], [0.8, 0.9, 0.6]])
    shape = lambda a: (a[0] * x + a[1] * y + a[2] * z) + 0.25 * (a[2] * x * x + a[0] * y * y + a[1] * z * z)
    Y = np.empty((len(SPECIES), m.nCells))
    for i in range(len(SPECIES)):
        if i != INERT:
            Y[i] = Y_amb[i] * (1.0 + eps * shape(dirs[i]))
    Y[INERT] = 1.0 - sum(Y[i] for i in range(len(SPECIES)) if i != INERT)
    return Y, h_amb * (1.0 + eps * shape(dirs[5]))


# composition / enthalpy of the conditioned case (tests): no exact zeros, inflow and ambient values distinct in every field
Y_AMB_COND = np.array([0.22, 0.012, 0.006, 0.015, 0.747])
Y_IN_COND = np.array([0.03, 0.02, 0.88, 0.025, 0.045])
H_AMB_COND = CP * 2.0                                  # ambient at Tref + 2 K


class Plume:
    def __init__(self, n, h=0.05, dt=1e-3, solvers=None, mesh=None, conditioned=False):
        """conditioned: the start state and boundary values of conditioned_state() / Y_AMB_COND / Y_IN_COND / H_AMB_COND instead of
        the quiescent ambient with pure-fuel inflow"""
        self.m = m = mesh if mesh is not None else make_mesh(n, h)
        self.Y_amb, self.Y_in, self.h_amb = (Y_AMB_COND, Y_IN_COND, H_AMB_COND) if conditioned else (Y_AMB, Y_IN, 0.0)
        self.dt, self.rDeltaT = dt, 1.0 / dt
        self.sol = solvers or Solvers()
        N = m.nCells
        self.names = [p.name for p in m.patches]
        self.ghRef = -np.linalg.norm(G) * (n[1] * h)            # hRef = top of the box
        self.gh = m.C @ G - self.ghRef
        self.ghf = m.Cf @ G - self.ghRef
        self.Y = np.tile(Y_AMB[:, None], (1, N))
        self.T = np.full(N, TREF)
        self.h = CP * (self.T - TREF)
        if conditioned:
            self.Y, self.h = conditioned_state(m, self.Y_amb, self.h_amb)
            self.T = TREF + self.h / CP
        self.U = np.zeros((3, N))
        self.p = np.full(N, PREF)
        self.p_rgh = np.zeros(N)
        self.psi = self.calc_psi()
        self.rho = self.psi * self.p
        self.phi = np.zeros(m.nFaces); self.phib = [np.zeros(p.size) for p in m.patches]
        self.K = np.zeros(N); self.dpdt = np.zeros(N)
        self.hydrostatic_init()
        self.time = 0.0
        self.stepNo, self.radFreq = 0, 0                      # set_radiation() switches the fvDOM stand-in on
        self.rays, self.I, self.G = [], [], np.zeros(N)

    # ---- thermo stand-in -------------------------------------------------------------------
    def set_radiation_model(self, a, Ehrr1, Ehrr2):
        """the reference's absorption / emission model and the coupling radiation->Sh(thermo, he) of solver/YEEqn.H:101:
        constant absorption coefficient a [1/m] (constRadFractionEmission: aCont = 0; constantAbsorptionEmission: a, e) and an
        emission E = RadFraction*Qdot with the radScaling of lib/.../constRadFractionEmission/constRadFractionEmission.C:ECont:
        RadFraction = max(min(Ehrr1, Ehrr2), (mlr1*Ehrr1 + mlr2*Ehrr2)/max(SMALL, mlr1 + mlr2)), mlr = -gSum(phi) of the burner
        patch(es) (both lists name the burner in cases/steckler/constant/radiationProperties:44-52).  Ray equation source
        omega/pi*(a sigma T^4 + E/4) (radiativeIntensityRay.C:286-300); Sh = Ru - fvm::Sp(4 Rp T^3/Cp, h) - Rp T^3 (T - 4 h/Cp)
        with Rp = 4 a sigma, Ru = a G - E (fvDOM.C:Rp/Ru, radiationModel.C:229-244)."""
        self.rad_a, self.Ehrr = float(a), (float(Ehrr1), float(Ehrr2))
        self.rad_coupled = True

    rad_coupled, rad_a = False, K_ABS

    rad_patches = (("inlet",), ("inlet",))     # constRadFractionEmissionCoeffs patch1 / patch2 (cases/steckler: both the burner)

    def rad_fraction(self):
        names = [p.name for p in self.m.patches]
        mlr1 = -float(sum(np.sum(self.phib[names.index(n)]) for n in self.rad_patches[0])) if self.rad_patches[0] else 0.0
        mlr2 = -float(sum(np.sum(self.phib[names.index(n)]) for n in self.rad_patches[1])) if self.rad_patches[1] else 0.0
        e1, e2 = self.Ehrr
        return max(min(e1, e2), (mlr1 * e1 + mlr2 * e2) / max(1e-15, mlr1 + mlr2))

    def set_radiation(self, solverFreq=100, nPhi=2, nTheta=4, ordered=True):
        """cases/steckler/constant/radiationProperties:32-40: nPhi 2, nTheta 4 (32 rays), solverFreq 100"""
        self.radFreq = solverFreq
        self.rad_ordered, self._flip = ordered, {}
        self.rays = ray_set(nPhi, nTheta)
        self.I = [np.zeros(self.m.nCells) for _ in self.rays]

    def calc_psi(self):
        return 1.0 / (RR * self.T * (self.Y / WMOL[:, None]).sum(axis=0))

    def thermo_correct(self):
        self.T = TREF + self.h / CP
        self.psi = self.calc_psi()

    def zg(self, vf):            # zero-gradient boundary copy of a cell field
        return [vf[p.faceCells] for p in self.m.patches]

    # the inlet patch as the gas side of a pyrolysing panel (lib/fvPatchFieldsPyrolysis: flowRateInletVelocityPyrolysisCoupled,
    # turbulentTemperatureRadiationQinCoupledMixed fluid branch): per-face velocity [n][3] and enthalpy [n] handed over by
    # oracle/pyrolysis.couple() before every step (tests/test_wallfire_pyrolysis_gpu.py); None = the plume's fixed inlet
    inlet_U, inlet_h = None, None
    mv_selection = True           # the multivariateSelection scheme's common limiter for the species and h (YEEqn in step())

    # ---- boundary conditions (mixed form) --------------------------------------------------
    def bc_U(self):
        m = self.m
        bcs = [fv.MixedBC(m) for _ in range(3)]
        for q, p in enumerate(m.patches):
            for c in range(3):
                if p.name == "inlet":
                    bcs[c].f[q][:] = 1.0
                    bcs[c].ref[q][:] = (U_IN if c == 1 else 0.0) if self.inlet_U is None else self.inlet_U[:, c]
                elif p.name == "floor":
                    bcs[c].f[q][:] = 1.0
                else:   # pressureInletOutletVelocity: tangential components fixed 0 on inflow, normal zeroGradient
                    nrm = np.abs(p.Sf[:, c]) > 0
                    bcs[c].f[q] = np.where(nrm, 0.0, 1.0 - fv.pos0(self.phib[q]))
        return bcs

    def bc_scalar(self, inlet, ambient, floor_fixed=None):
        m = self.m
        bc = fv.MixedBC(m)
        for q, p in enumerate(m.patches):
            if p.name == "inlet":
                bc.f[q][:] = 1.0; bc.ref[q][:] = inlet
            elif p.name == "floor":
                if floor_fixed is not None:
                    bc.f[q][:] = 1.0; bc.ref[q][:] = floor_fixed
            else:       # inletOutlet
                bc.f[q] = 1.0 - fv.pos0(self.phib[q]); bc.ref[q][:] = ambient
        return bc

    def bc_p_rgh(self, gradients):
        """fixedFluxPressure on inlet/floor (gradient from constrainPressure); prghTotalHydrostaticPressure
        on top/sides: p_rgh = ph_rgh - 0.5*rho*(1-pos0(phi))*|U|^2."""
        m = self.m
        bc = fv.MixedBC(m)
        Ub = [np.stack(v) for v in zip(*[b.values(m, self.U[c]) for c, b in enumerate(self.bc_U())])]
        rhob = self.zg(self.rho)
        for q, p in enumerate(m.patches):
            if p.name in ("inlet", "floor"):
                bc.refGrad[q] = gradients[q]
            else:
                bc.f[q][:] = 1.0
                bc.ref[q] = self.ph_rgh_b[q] - 0.5 * rhob[q] * (1.0 - fv.pos0(self.phib[q])) * (Ub[q] ** 2).sum(axis=0)
        return bc

    # ---- solver/phrghEqn.H ----------------------------------------------------------------
    def hydrostatic_init(self, nCorr=5):
        m = self.m
        ph = np.zeros(m.nCells)
        self.p = ph + self.rho * self.gh + PREF
        self.thermo_correct(); self.rho = self.psi * self.p
        for _ in range(nCorr):
            rhof, rhofb = fv.interpolate(m, self.rho, self.zg(self.rho))
            sg, _ = fv.snGrad(m, self.rho, self.zg(self.rho))
            phig = -rhof * self.ghf * sg * m.magSf
            phigb = [np.zeros(p.size) for p in m.patches]
            bc = fv.MixedBC(m)
            for q, p in enumerate(m.patches):
                if p.name == "top":
                    bc.f[q][:] = 1.0
            M = fv.fvm_laplacian(m, rhof, rhofb, [bc])
            M.add_su(fv.surface_integrate(m, phig, phigb))
            d, s = M.solve_system()
            ph = self.sol.solve("ph_rgh", "ph_rgh", m, d, M.upper, M.lower, s, ph)
            self.p = ph + self.rho * self.gh + PREF
            self.thermo_correct(); self.rho = self.psi * self.p
        self.ph_rgh = ph
        self.ph_rgh_b = fv.MixedBC(m, f=bc.f).values(m, ph)
        self.p_rgh = ph.copy()
        # stored_bc: p_rgh keeps the boundary values of its last evaluate() (GeometricField semantics, what the reference's
        # snippets see through include/ffmFoam.H) instead of re-evaluating its conditions with a zero gradient in UEqn
        self.stored_bc = False
        self.p_rgh_b = [b.copy() for b in self.ph_rgh_b]

    # ---- time-step control (solver/fireFoam.C:78-82) ----------------------------------------
    def set_time_controls(self, maxCo, maxDeltaT):
        """controlDict adjustTimeStep yes; maxCo; maxDeltaT (cases/steckler/system/controlDict:44-48)"""
        self.adjust, self.maxCo, self.maxDeltaT = True, maxCo, maxDeltaT

    def set_delta_t(self):
        """compressibleCourantNo.H [upstream], then the reference's solver/setMultiRegionDeltaT.H:34-60 (no pyrolysis region, no
        film: DiNum = -GREAT -> SMALL, maxDi = GREAT, film Courant number 0), then setDeltaT.H [upstream] -- both are included"""
        m = self.m
        GREAT, SMALL = 1.0e15, 1.0e-15
        sumPhi = fv.surface_sum(m, np.abs(self.phi), [np.abs(b) for b in self.phib]) / self.rho
        CoNum = 0.5 * (sumPhi / m.V).max() * self.dt
        self.meanCoNum = 0.5 * (sumPhi.sum() / m.V.sum()) * self.dt
        self.CoNum = CoNum
        # solidRegionDiffusionNo.H: DiNum = pyrolysis.solidRegionDiffNo() (-GREAT without a region -> SMALL); maxDi = pyrolysis.maxDiff()
        DiNum = getattr(self, "solid_DiNum", None)
        DiNum = SMALL if DiNum is None else DiNum
        maxDi = getattr(self, "maxDi", GREAT)
        TFactorFluid = self.maxCo / (CoNum + SMALL)
        TFactorSolid = maxDi / (DiNum + SMALL)
        TFactorFilm = self.maxCo / (0.0 + SMALL)
        dt = min(self.dt * min(min(TFactorFluid, min(TFactorFilm, TFactorSolid)), 1.2), self.maxDeltaT)
        maxDeltaTFact = self.maxCo / (CoNum + SMALL)
        deltaTFact = min(min(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2)
        dt = min(deltaTFact * dt, self.maxDeltaT)
        self.dt, self.rDeltaT = dt, 1.0 / dt

    # ---- one time step (solver/fireFoam.C:76-121) -----------------------------------------
    def rho_eqn(self):
        m = self.m
        d = self.rDeltaT * m.V
        s = self.rDeltaT * self.rho0 * m.V - m.V * fv.surface_integrate(m, self.phi, self.phib)
        self.rho = s / d            # diagonalSolver

    def step(self):
        if getattr(self, "adjust", False):
            self.set_delta_t()
        m, rdt = self.m, self.rDeltaT
        self.sol.log = []
        # oldTime fields
        self.rho0, self.U0, self.h0, self.Y0 = self.rho.copy(), self.U.copy(), self.h.copy(), self.Y.copy()
        self.K0, self.p0, self.psi0, self.p_rgh0 = self.K.copy(), self.p.copy(), self.psi.copy(), self.p_rgh.copy()
        self.phi0, self.phib0 = self.phi.copy(), [b.copy() for b in self.phib]
        self.rho_eqn()
        # ---- UEqn.H
        bcU = self.bc_U()
        muf = np.full(m.nFaces, MU); mub = [np.full(p.size, MU) for p in m.patches]
        Ub = [b.values(m, self.U[c]) for c, b in enumerate(bcU)]
        # div(phi,U) Gauss LUST grad(U) (cases/steckler/system/fvSchemes:32): implicit part with the LUST weights, explicit
        # correction fvc::surfaceIntegrate(phi*correction(U)) added to the matrix (gaussConvectionScheme::fvmDiv)
        UEqn = fv.fvm_ddt(m, rdt, self.rho, self.rho0, self.U0)
        zb = [np.zeros(p.size) for p in m.patches]
        if getattr(self, "divU_scheme", None):
            # div(phi,U) Gauss filteredLinear2V k l (cases/wallFireSpread2D/system/fvSchemes:41): one limiter per face, no correction
            _, kk, ll = self.divU_scheme
            gradU = np.stack([fv.grad(m, self.U[c], Ub[c]) for c in range(3)], axis=2)           # [cell][i][j] = d_i U_j
            divU = fv.fvm_div(m, self.phi, self.phib, fv.filtered_linear2V_weights(m, self.phi, self.U.T.copy(), gradU, kk, ll), bcU)
        else:
            wU = fv.lust_weights(m, self.phi)
            divU = fv.fvm_div(m, self.phi, self.phib, wU, bcU)
            divU.add_vol(np.stack([fv.surface_integrate(m, self.phi * fv.lust_correction(m, self.phi, fv.grad(m, self.U[c], Ub[c])), zb)
                                   for c in range(3)]))
        UEqn += divU
        UEqn -= fv.fvm_laplacian(m, muf, mub, bcU)
        rhob = self.zg(self.rho)
        sgr, _ = fv.snGrad(m, self.rho, rhob)
        bcp = self.bc_p_rgh([np.zeros(p.size) for p in m.patches])
        sgp, sgpb = fv.snGrad(m, self.p_rgh, self.p_rgh_b if self.stored_bc else bcp.values(m, self.p_rgh))
        rec = fv.reconstruct(m, (-self.ghf * sgr - sgp) * m.magSf, [-s * p.magSf for s, p in zip(sgpb, m.patches)])
        for c in range(3):
            if getattr(m, "solutionD", (1, 1, 1))[c] < 0:          # fvMatrix<vector>::solveSegregated skips the empty directions
                continue
            d, s = UEqn.solve_system(c)
            s = s + m.V * rec[:, c]
            self.U[c] = self.sol.solve("U", "U" + "xyz"[c], m, d, UEqn.upper, UEqn.lower, s, self.U[c])
        self.K = 0.5 * (self.U ** 2).sum(axis=0)
        # ---- YEEqn.H
        alphaEff = np.full(m.nCells, MU / PR)
        af, afb = fv.interpolate(m, alphaEff, self.zg(alphaEff))
        fuel, o2 = self.Y[2], self.Y[0]
        wFuel = self.rho * np.minimum(fuel, o2 / S_O2) / TAU            # combustion->correct()
        Qdot = wFuel * HC
        self.Qdot_field = Qdot
        # mvConvection (solver/YEEqn.H:1-10), `Gauss multivariateSelection { Yi limitedLinear01 1; h limitedLinear 1; }`: ONE limiter for
        # the species and h -- the face-wise minimum of the member schemes' limiters over all fields of the table (the five species,
        # the inert one included, and h; solver/createFields.H `fields`), computed when the scheme is constructed, i.e. from the fields
        # at the start of YEEqn.H -- so that all species are interpolated with the same weights.  Pinned by the golden log through
        # oracle/steckler_case.py (tests/test_steckler_whole_log_cpu.py).  mv_selection = False: one limiter per field (round 1).
        bch = self.bc_scalar(CP * (T_IN - TREF) if self.inlet_h is None else self.inlet_h, self.h_amb, floor_fixed=0.0)
        if self.mv_selection:
            lim = fv.limited_limiter(m, "limitedLinear", self.phi, self.h, fv.grad(m, self.h, bch.values(m, self.h)), 1.0)
            Ytb = [np.zeros(p.size) for p in m.patches]
            for i in range(len(SPECIES)):
                if i == INERT:
                    continue
                Yb = self.bc_scalar(self.Y_in[i], self.Y_amb[i]).values(m, self.Y[i])
                Ytb = [a + np.maximum(b, 0.0) for a, b in zip(Ytb, Yb)]
                lim = np.minimum(lim, fv.limited_limiter(m, "limitedLinear01", self.phi, self.Y[i], fv.grad(m, self.Y[i], Yb), 1.0))
            Nb = [np.maximum(1.0 - b, 0.0) for b in Ytb]           # the inert specie's patch values: Y[inertIndex] == 1 - Yt; .max(0)
            lim = np.minimum(lim, fv.limited_limiter(m, "limitedLinear01", self.phi, self.Y[INERT], fv.grad(m, self.Y[INERT], Nb), 1.0))
            w_mv = lim * m.weights + (1.0 - lim) * fv.pos0(self.phi)
            self.w_mv_last = w_mv               # tests hand these to the device (ffm_plume_override_mv_weights): the deciding test
        Yt = np.zeros(m.nCells)
        for i in range(len(SPECIES)):
            if i == INERT:
                continue
            bc = self.bc_scalar(self.Y_in[i], self.Y_amb[i])
            Yb = bc.values(m, self.Y[i])
            w = w_mv if self.mv_selection else fv.limited_weights(m, "limitedLinear01", self.phi, self.Y[i], fv.grad(m, self.Y[i], Yb), 1.0)
            E = fv.fvm_ddt(m, rdt, self.rho, self.rho0, self.Y0[i])
            E += fv.fvm_div(m, self.phi, self.phib, w, [bc])
            E -= fv.fvm_laplacian(m, af, afb, [bc])
            E.add_su(NU[i] * wFuel)                                      # == combustion->R(Yi)
            d, s = E.solve_system()
            self.Y[i] = np.maximum(self.sol.solve("Yi", SPECIES[i], m, d, E.upper, E.lower, s, self.Y[i]), 0.0)
            Yt += self.Y[i]
        self.Y[INERT] = np.maximum(1.0 - Yt, 0.0)
        if self.radFreq > 0 and self.stepNo % self.radFreq == 0:          # radiation->correct(), solver/YEEqn.H:80
            self.radiation_correct()
        hb = bch.values(m, self.h)
        wh = w_mv if self.mv_selection else fv.limited_weights(m, "limitedLinear", self.phi, self.h, fv.grad(m, self.h, hb), 1.0)
        Ub = [b.values(m, self.U[c]) for c, b in enumerate(bcU)]      # U.correctBoundaryConditions() after the solve
        Kb = [0.5 * sum(Ub[c][q] ** 2 for c in range(3)) for q in range(len(m.patches))]
        wK = fv.limited_weights(m, "limitedLinear", self.phi, self.K, fv.grad(m, self.K, Kb), 1.0)
        Kf = wK * self.K[m.l] + (1.0 - wK) * self.K[m.u]
        E = fv.fvm_ddt(m, rdt, self.rho, self.rho0, self.h0)
        E += fv.fvm_div(m, self.phi, self.phib, wh, [bch])
        E -= fv.fvm_laplacian(m, af, afb, [bch])
        # explicit terms on the LHS, one fvMatrix::operator+(tmp<fvMatrix>, tmp<volField>) each (source -= V*term), in the
        # order of solver/YEEqn.H:89-101: fvc::ddt(rho, K), fvc::div(phi, K), -dpdt
        E.add_vol(rdt * (self.rho * self.K - self.rho0 * self.K0))
        E.add_vol(fv.surface_integrate(m, self.phi * Kf, [pb * kb for pb, kb in zip(self.phib, Kb)]))
        E.add_vol(-self.dpdt)
        E.add_su(Qdot)
        if self.rad_coupled and self.rays and hasattr(self, "radE"):
            # + radiation->Sh(thermo, he): Ru - fvm::Sp(4 Rp T^3/Cpv, he) - Rp T^3 (T - 4 he/Cpv)
            Rp = 4.0 * self.rad_a * SIGMA_SB
            T3 = self.T * self.T * self.T
            Ru = self.rad_a * self.G - self.rad_fraction() * Qdot
            self.ShSp = 4.0 * Rp * T3 / CP
            self.ShSu = Ru - Rp * T3 * (self.T - 4.0 * self.h / CP)
            E.diag += m.V * self.ShSp
            E.add_su(self.ShSu)
        d, s = E.solve_system()
        self.h = self.sol.solve("h", "h", m, d, E.upper, E.lower, s, self.h)
        self.thermo_correct()
        # ---- pEqn.H, nCorrectors = 2
        for corr in range(2):
            self.p_corrector(UEqn, final=(corr == 1))
        self.rho = self.psi * self.p
        self.time += self.dt
        self.stepNo += 1

    fvdom = None

    def set_fvdom(self, nPhi, nTheta, solverFreq, maxIter, tolerance, a, Ehrr1, Ehrr2, divScheme="upwind"):
        """the reference's fvDOM itself (oracle/fvdom.py: calculate()'s iteration, greyDiffusiveRadiation walls with their emissivities --
        self.fvdom.emissivity, per patch -- the 2-D ray set where the mesh has an empty z direction) as radiation->correct(), with
        constRadFractionEmission (a, E = RadFraction*Qdot) and radiation->Sh in the enthalpy equation; the rays are solved by the
        solver selection's "Ii" entry.  Wall temperatures: the thermo stand-in's zero-gradient boundary values."""
        from . import fvdom
        m = self.m

        def solve(name, d, upper, lower, s, psi0):
            psi = self.sol.solve("Ii", name, m, d, upper, lower, s, psi0)
            return psi, self.sol.log[-1][1]
        self.fvdom = fvdom.FvDOM(m, nPhi, nTheta, solve, maxIter=maxIter, tolerance=tolerance, divScheme=divScheme,
                                 solutionD=getattr(m, "solutionD", (1, 1, 1)))
        self.radFreq, self.rays = solverFreq, self.fvdom.rays
        self.set_radiation_model(a, Ehrr1, Ehrr2)

    def radiation_correct(self):
        if self.fvdom is not None:
            self.radE = self.rad_fraction() * self.Qdot_field
            self.fvdom.calculate(self.T, self.zg(self.T), self.rad_a, self.radE)
            self.G, self.I = self.fvdom.G, self.fvdom.I
            return
        self._radiation_correct_standin()

    def _radiation_correct_standin(self):
        """One fvDOM sweep: per ray  fvm::div(Ji, Ii) + fvm::Sp(k*omega, Ii) == 1/pi*omega*(k*sigma*T^4), div scheme upwind
        (cases/steckler/system/fvSchemes:60), inflow faces fixed to the ambient black-body intensity, outflow zeroGradient
        (stand-in for greyDiffusiveRadiation); then G = sum_i Ii*omega_i (fvDOM::updateG)."""
        m = self.m
        Ib = SIGMA_SB * ((TREF * TREF) * (TREF * TREF)) / np.pi
        T4 = (self.T * self.T) * (self.T * self.T)
        self.G = np.zeros(m.nCells)
        if self.rad_coupled:
            self.radE = self.rad_fraction() * self.Qdot_field        # absorptionEmission->ECont(): RadFraction*Qdot
        for i, (dAve, omega) in enumerate(self.rays):
            Ji = (dAve[0] * m.Sf[:, 0] + dAve[1] * m.Sf[:, 1]) + dAve[2] * m.Sf[:, 2]
            Jib = [(dAve[0] * p.Sf[:, 0] + dAve[1] * p.Sf[:, 1]) + dAve[2] * p.Sf[:, 2] for p in m.patches]
            bc = fv.MixedBC(m, f=[1.0 - fv.pos0(jb) for jb in Jib], ref=[np.full(p.size, Ib) for p in m.patches])
            E = fv.fvm_div(m, Ji, Jib, fv.pos0(Ji), [bc])
            E.diag += m.V * (self.rad_a * omega)                         # fvm::Sp(k*omega, Ii)
            if self.rad_coupled:
                E.add_su(1.0 / np.pi * omega * (self.rad_a * SIGMA_SB * T4 + self.radE / 4.0))
            else:
                E.add_su(1.0 / np.pi * omega * (self.rad_a * SIGMA_SB * T4))
            d, s = E.solve_system()
            # direction-ordered solve: a ray whose direction has one component of the other sign is solved in the cell order of
            # the box mirrored in that axis, where its upwind matrix is triangular and DILU is exact (1 PBiCGStab iteration)
            neg = int(dAve[0] < 0) + int(dAve[1] < 0) + int(dAve[2] < 0)
            flip = -1
            if self.rad_ordered and neg == 1:
                flip = 0 if dAve[0] < 0 else 1 if dAve[1] < 0 else 2
            elif self.rad_ordered and neg == 2:
                flip = 0 if dAve[0] >= 0 else 1 if dAve[1] >= 0 else 2
            if flip < 0:
                self.I[i] = self.sol.solve("Ii", "I%d" % i, m, d, E.upper, E.lower, s, self.I[i])
            else:
                cm, fm, swap = self._flip[flip] if flip in self._flip else self._flip.setdefault(flip, fv.flip_maps(m, flip))
                uB = np.where(swap, E.lower[fm], E.upper[fm]); lB = np.where(swap, E.upper[fm], E.lower[fm])
                pB = self.sol.solve("Ii", "I%d" % i, m, d[cm], uB, lB, s[cm], self.I[i][cm])
                self.I[i] = np.empty_like(pB); self.I[i][cm] = pB
            self.G = self.G + self.I[i] * omega

    def p_corrector(self, UEqn, final):
        m, rdt = self.m, self.rDeltaT
        self.rho = self.psi * self.p
        rAU = 1.0 / UEqn.A()
        rhorAU = self.rho * rAU
        rhorAUf, rhorAUfb = fv.interpolate(m, rhorAU, self.zg(rhorAU))
        HbyA = rAU * UEqn.H(self.U)
        bcU = self.bc_U()
        Ub = [b.values(m, self.U[c]) for c, b in enumerate(bcU)]
        # constrainHbyA: fixed-value patches take U_b, the others the extrapolated cell value
        HbyAb = [[Ub[c][q] if p.name in ("inlet", "floor") else HbyA[c][p.faceCells] for q, p in enumerate(m.patches)] for c in range(3)]
        rhob = self.zg(self.rho)
        sgr, _ = fv.snGrad(m, self.rho, rhob)
        phig = -rhorAUf * self.ghf * sgr * m.magSf
        rhoH = self.rho * HbyA
        flux = sum(fv.interpolate(m, rhoH[c], [rhob[q] * HbyAb[c][q] for q in range(len(m.patches))])[0] * m.Sf[:, c] for c in range(3))
        fluxb = [sum(rhob[q] * HbyAb[c][q] * p.Sf[:, c] for c in range(3)) for q, p in enumerate(m.patches)]
        # fvc::ddtCorr(rho, U, phi), Euler; zero on boundaries
        rhoU0 = self.rho0 * self.U0
        phiCorr = self.phi0 - sum((m.weights * rhoU0[c][m.l] + (1 - m.weights) * rhoU0[c][m.u]) * m.Sf[:, c] for c in range(3))
        coeff = 1.0 - np.minimum(np.abs(phiCorr) / (np.abs(self.phi0) + 1e-15), 1.0)
        phiHbyA = flux + rhorAUf * (coeff * rdt * phiCorr) + phig
        phiHbyAb = fluxb
        # constrainPressure on the fixedFluxPressure patches
        grads = [(phiHbyAb[q] - rhob[q] * sum(p.Sf[:, c] * Ub[c][q] for c in range(3))) / (p.magSf * rhorAUfb[q])
                 for q, p in enumerate(m.patches)]
        bcp = self.bc_p_rgh(grads)
        E = fv.fvm_ddt(m, rdt, self.psi, self.psi0, self.p_rgh0)
        # fvc::ddt(psi, rho)*gh, fvc::ddt(psi)*pRef, fvc::div(phiHbyA): one source update each (solver/pEqn.H:30-33)
        E.add_vol(rdt * (self.psi * self.rho - self.psi0 * self.rho0) * self.gh)
        E.add_vol(rdt * (self.psi - self.psi0) * PREF)
        E.add_vol(fv.surface_integrate(m, phiHbyA, phiHbyAb))
        E -= fv.fvm_laplacian(m, rhorAUf, rhorAUfb, [bcp])
        d, s = E.solve_system()
        self.p_rgh = self.sol.solve("p_rghFinal" if final else "p_rgh", "p_rgh", m, d, E.upper, E.lower, s, self.p_rgh)
        self.p_rgh_b = bcp.values(m, self.p_rgh)
        fl, flb = E.flux(self.p_rgh)
        self.phi = phiHbyA + fl
        self.phib = [a + b for a, b in zip(phiHbyAb, flb)]
        rec = fv.reconstruct(m, (fl + phig) / rhorAUf, [b / r for b, r in zip(flb, rhorAUfb)])
        self.U = HbyA + rAU * rec.T
        self.p = self.p_rgh + self.rho * self.gh + PREF
        self.rho_eqn()
        self.K = 0.5 * (self.U ** 2).sum(axis=0)
        self.dpdt = rdt * (self.p - self.p0)

    def fields(self):
        out = {"rho": self.rho, "p": self.p, "p_rgh": self.p_rgh, "T": self.T, "h": self.h, "phi": self.phi, "K": self.K}
        for c in range(3):
            out["U" + "xyz"[c]] = self.U[c]
        for i, s in enumerate(SPECIES):
            out[s] = self.Y[i]
        return out
