"""ORACLE (test infrastructure): one time step of the reference's 1-D solid pyrolysis region model, SURVEY 8(f) N3:
reactingOneDim::evolveRegion (packages/regionModels/pyrolysisModels/reactingOneDim/reactingOneDim.C:686-721) =
    solidChemistry->calculate()   one Arrhenius solid reaction  wood^n = char + gas  (cases/pyrolysis1D/constant/panelRegion/reactions)
    solveContinuity()   fvm::ddt(rho) == -RRg                                   (:240-266)   diagonal
    solveSpeciesMass()  fvm::ddt(rho, Yi) == RRs(i); Yi.max(0); last = 1 - Yt    (:269-303)   diagonal
    solveEnergy()       fvm::ddt(rho, h) - fvm::laplacian(alpha, h) + fvc::laplacian(alpha, h) - fvc::laplacian(kappa, T)
                        == chemistryQdot - fvm::Sp(RRg, h)                        (:306-353)   one tridiagonal system per column
    solidThermo.correct()   T from h (hConst thermo: h = Cp (T - Tstd)), Cp mass-fraction weighted, alpha_ = kappa_vol/Cp (class Panel)
and the variant BASELINE config 5 selects, reactingOneDim21 (lib/regionModels/pyrolysisModels/reactingOneDim21/reactingOneDim21.C:
solveEnergy :319-368 with RRs(i) T Cp_i sources instead of -fvm::Sp(RRg, h); evolveRegion :777-815).
on a region mesh extruded from a wall patch (cases/wallFireSpread2D/system/extrudeToRegionMeshDict:17-39: nLayers 8): every
column of cells is coupled only along its own axis (the side faces are `empty`), so the region is nCol independent chains of
nLay cells.  The mesh does not move (cases/pyrolysis1D/constant/pyrolysisZones: moveMesh false, useChemistrySolvers false).
Coupling with the gas region (lib/fvPatchFieldsPyrolysis): in: the heat flux into the exposed face of every column; out: the
temperature of that face and the pyrolysate mass flux through it (phiGas: the column's integrated gas release).

The chemistry follows OpenFOAM-dev's pyrolysisChemistryModel::omega / calculate and solidArrheniusReactionRate as far as this
builder recalls them (not in the reference tree): kf = T < Tcrit ? 0 : A exp(-Ta/T); omega = kf (m_w/m_w0)^n m_w0 with m_w the
cell's virgin-solid mass and m_w0 = rho0 max(Y_w0, 0.001) V its initial value (Ys0_), i.e. per volume kf (rho Y_w/c0)^n c0;
RRs_wood = -omega, RRs_char = (rho_char/rho_wood) omega, RRg = (1 - rho_char/rho_wood) omega; Qdot = -sum Hf_i RRs_i.
The reference holds no output of a pyrolysis run (cases/pyrolysis1D/mlr.plot refers to ./referenceResult, not shipped):
PARITY UNPINNED by reference data -- the tests check conservation laws, a dense solve and the device against this restatement.
Only tests/ may import this module."""
import numpy as np

TSTD = 298.15


class Solid:
    def __init__(self, rho, Cp, kappa, Hf):
        self.rho, self.Cp, self.kappa, self.Hf = rho, Cp, kappa, Hf


WOOD = Solid(114.7, 696.0, 0.135, -1.41e6)      # cases/pyrolysis1D/constant/panelRegion/thermo.solid
CHAR = Solid(11.5, 611.0, 0.4, 0.0)
REACTION = dict(A=7.83e10, Ta=15274.57, Tcrit=400.0, n=4.86)      # .../panelRegion/reactions


class Panel:
    """nCol columns x nLay layers; layer 0 is the exposed (coupled) one.  Fields [nCol][nLay].

    model        "reactingOneDim" (cases/pyrolysis1D; packages/regionModels/pyrolysisModels/reactingOneDim/reactingOneDim.C:306-353:
                 ... == chemistryQdot - fvm::Sp(RRg, h)) or "reactingOneDim21" (cases/wallFireSpread2D/constant/pyrolysisZones:23;
                 lib/regionModels/pyrolysisModels/reactingOneDim21/reactingOneDim21.C:319-368: ... == chemistryQdot + RRs(0) T Cp0 +
                 RRs(1) T Cp1, Cp_i of the pure solids at 300 K)
    alphaScheme  interpolation of laplacian(thermo:alpha,h): "linear" (cases/wallFireSpread2D/system/panelRegion/fvSchemes:45) or
                 "harmonic" (cases/pyrolysis1D/system/panelRegion/fvSchemes:38)
    kappaScheme  interpolation of laplacian(kappa,T): "harmonic" in both cases (fvSchemes:44 / :37); "linear" kept for round 2's tests
    back         None (zero gradient), ("fixed", T) or ("constH", h, Tinf): lib/fvPatchFields/constHTemperatureFvPatchScalarField
                 (.C:156-180: refValue Tinf, refGrad 0, valueFraction 1/(1 + kappa/max(h, SMALL) deltaCoeffs)), the `panel_top`
                 patch of cases/wallFireSpread2D/0/panelRegion/T:25-31 (h 0, Tinf 293)
    radiation    None or dict(v=(absorptivity, emissivity), char=(...)): greyMeanSolidAbsorptionEmission of
                 cases/wallFireSpread2D/constant/panelRegion/radiationProperties:19-31 -- surface emissivity / absorptivity = sum of the
                 solids' values weighted with their VOLUME fractions X_i = (Y_i/rho_i)/sum(Y_j/rho_j) (upstream
                 greyMeanSolidAbsorptionEmission::X / calc), `emissivityMode solidRadiation` on both sides of the coupled patch

    Thermo (heSolidThermo<reactingMixture<constIso, hConst, rhoConst>>, OpenFOAM-dev 2017 as this builder recalls it; not in the
    reference tree): correct() stores alpha_ = kappa_vol/Cp with kappa_vol the VOLUME-fraction-weighted conductivity
    (cellVolMixture) and Cp the mass-fraction-weighted heat capacity (cellMixture); T = Tstd + h/Cp.  solidThermo.kappa() is
    heThermo::kappa() = Cp()*alpha_: the heat capacity of the CURRENT composition (after solveSpeciesMass) times the stored alpha_
    of the last correct() -- that is what fvc::laplacian(kappa(), T()) in solveEnergy and the coupled patch's kappa(*this) see."""

    def __init__(self, nCol, nLay=8, thickness=0.0127, area=1.0, T0=298.15, Yw0=1.0, model="reactingOneDim", alphaScheme="linear",
                 kappaScheme="linear", back=None, radiation=None):
        assert model in ("reactingOneDim", "reactingOneDim21")
        self.model, self.alphaScheme, self.kappaScheme, self.back, self.radiation = model, alphaScheme, kappaScheme, back, radiation
        self.nCol, self.nLay = nCol, nLay
        self.dx = thickness / nLay
        self.A = area
        self.V = self.A * self.dx
        self.Yw0 = Yw0
        self.Yw = np.full((nCol, nLay), Yw0)
        self.rho = np.full((nCol, nLay), 1.0 / (Yw0 / WOOD.rho + (1 - Yw0) / CHAR.rho)) if Yw0 < 1 else np.full((nCol, nLay), WOOD.rho)
        self.T = np.full((nCol, nLay), float(T0))
        self.h = self.Cp() * (self.T - TSTD)
        self.alpha = self.kappa_vol() / self.Cp()             # alpha_ of the constructor's correct()
        self.massGas = np.zeros(nCol)
        self.c0 = self.rho[0, 0] * max(Yw0, 0.001)            # initial partial density of the virgin solid (Ys0_/V)
        self.Twall = np.full(nCol, float(T0))                 # stored value of the coupled patch of T
        self.qSurf = np.zeros(nCol)

    def Cp(self, Yw=None):
        Yw = self.Yw if Yw is None else Yw
        return Yw * WOOD.Cp + (1.0 - Yw) * CHAR.Cp

    def Xw(self, Yw=None):
        """volume fraction of the virgin solid"""
        Yw = self.Yw if Yw is None else Yw
        return (Yw / WOOD.rho) / (Yw / WOOD.rho + (1.0 - Yw) / CHAR.rho)

    def kappa_vol(self, Yw=None):
        X = self.Xw(Yw)
        return X * WOOD.kappa + (1.0 - X) * CHAR.kappa

    def kappa(self):
        """solidThermo.kappa() = Cp()*alpha_ (see the class docstring)"""
        return self.Cp() * self.alpha

    def surface_radiation(self, Yw=None):
        """(absorptivity, emissivity) of the exposed face from layer 0's composition"""
        X = self.Xw(self.Yw[:, 0] if Yw is None else Yw)
        (aV, eV), (aC, eC) = self.radiation["v"], self.radiation["char"]
        return X * aV + (1.0 - X) * aC, X * eV + (1.0 - X) * eC

    @staticmethod
    def _face(scheme, a, b):
        return 0.5 * (a + b) if scheme == "linear" else 1.0 / (0.5 / a + 0.5 / b)

    def _evolve(self, dt, flux, Tback):
        """reactingOneDim(21)::evolveRegion.  flux(kappa_patch) -> heat flux INTO the exposed face [W/m2], called where the
        reference evaluates the coupled patch of T: inside solveEnergy, after solveSpeciesMass, with the old temperatures."""
        R = REACTION
        rdt = 1.0 / dt
        rho0, Yw0f, h0, T0 = self.rho.copy(), self.Yw.copy(), self.h.copy(), self.T.copy()
        alpha = self.alpha
        # ---- solidChemistry->calculate()
        kf = np.where(T0 < R["Tcrit"], 0.0, R["A"] * np.exp(-R["Ta"] / T0))
        omega = kf * np.power(rho0 * Yw0f / self.c0, R["n"]) * self.c0
        sr = CHAR.rho / WOOD.rho
        RRw, RRc, RRg = -omega, sr * omega, (1.0 - sr) * omega
        Qdot = -(WOOD.Hf * RRw + CHAR.Hf * RRc)
        # ---- solveContinuity: fvm::ddt(rho) == -RRg
        self.rho = (rdt * rho0 * self.V - self.V * RRg) / (rdt * self.V)
        # ---- solveSpeciesMass: fvm::ddt(rho, Yw) == RRs(wood); char = 1 - Yt
        self.Yw = np.maximum((rdt * rho0 * Yw0f * self.V + self.V * RRw) / (rdt * self.rho * self.V), 0.0)
        # ---- solveEnergy: tridiagonal in every column
        nL = self.nLay
        kappa = self.Cp() * alpha                                # kappa(): Cp of the new composition, alpha_ of the last correct()
        c_a = self._face(self.alphaScheme, alpha[:, :-1], alpha[:, 1:]) * self.A / self.dx
        c_k = self._face(self.kappaScheme, kappa[:, :-1], kappa[:, 1:]) * self.A / self.dx
        diag = rdt * self.rho * self.V
        if self.model == "reactingOneDim":
            diag = diag + self.V * RRg                           # - fvm::Sp(RRg, h) on the right-hand side
        lower = np.zeros((self.nCol, nL)); upper = np.zeros((self.nCol, nL))
        upper[:, :-1] = -c_a; lower[:, 1:] = -c_a
        diag[:, :-1] += c_a; diag[:, 1:] += c_a
        src = rdt * rho0 * h0 * self.V + self.V * Qdot
        if self.model == "reactingOneDim21":
            src = src + self.V * (RRw * T0 * WOOD.Cp)            # + RRs(0)*T*Cp0
            src = src + self.V * (RRc * T0 * CHAR.Cp)            # + RRs(1)*T*Cp1
        # + fvc::laplacian(alpha, h) - fvc::laplacian(kappa, T) on the LHS: source -= (lapA - lapK)
        fa = c_a * (h0[:, 1:] - h0[:, :-1]); fk = c_k * (T0[:, 1:] - T0[:, :-1])
        lapA = np.zeros((self.nCol, nL)); lapK = np.zeros((self.nCol, nL))
        lapA[:, :-1] += fa; lapA[:, 1:] -= fa
        lapK[:, :-1] += fk; lapK[:, 1:] -= fk
        # exposed face: mixed condition with valueFraction 0, refGrad = q/kappa_patch -> kappa snGrad(T) A = q A; the implicit and
        # explicit alpha-laplacians cancel on a gradient patch
        q = np.asarray(flux(kappa[:, 0]), float)
        self.qSurf = q * np.ones(self.nCol)
        lapK[:, 0] += self.qSurf * self.A
        # back face: mixed condition (f, Tinf): kappa_b A dC f (Tinf - T_c) in fvc::laplacian(kappa,T), and what is left of
        # -fvm::laplacian(alpha,h) + fvc::laplacian(alpha,h) there: alpha_b A dC f (h_new - h_old)
        db = 2.0 / self.dx
        back = ("fixed", Tback) if Tback is not None else self.back
        if back is not None:
            if back[0] == "fixed":
                f, Tinf = 1.0, back[1]
            else:
                f, Tinf = 1.0 / (1.0 + kappa[:, -1] / max(back[1], 1e-15) * db), back[2]
            lapK[:, -1] += kappa[:, -1] * self.A * db * f * (Tinf - T0[:, -1])
            cb = alpha[:, -1] * self.A * db * f
            diag[:, -1] += cb
            src[:, -1] += cb * h0[:, -1]
        src -= (lapA - lapK)
        self.h = thomas(lower, diag, upper, src)
        self.massGas = (RRg * self.V).sum(axis=1)               # phiGas through the exposed face [kg/s]
        # ---- solidThermo.correct()
        Cp = self.Cp()
        self.T = TSTD + self.h / Cp
        self.alpha = self.kappa_vol() / Cp
        return dict(RRg=RRg, Qdot=Qdot, lower=lower, diag=diag, upper=upper, src=src, kappa=kappa)

    def step(self, dt, qSurf, Tback=None):
        """qSurf[nCol]: heat flux into the exposed face [W/m2]; back face as self.back, or held at Tback"""
        return self._evolve(dt, lambda kap: qSurf, Tback)

    def evolve(self, dt, Tgas_cell, kappaDelta_gas, qin, emissivity=None, absorptivity=None):
        """evolveRegion with the exposed face coupled to the gas region by turbulentTemperatureRadiationQinCoupledMixed
        (lib/fvPatchFieldsPyrolysis/.../turbulentTemperatureRadiationQinCoupledMixedFvPatchScalarField.C:176-296, solid branch):
            nbrTotalFlux = nbrKDelta (T_s,cell - T_g,cell) - a qin + e sigma T_w^4;  refGrad = -nbrTotalFlux/kappa(*this), f = 0
        evaluated where OpenFOAM evaluates it -- at the construction of hEqn (mixedEnergy::updateCoeffs -> Tw.evaluate()), i.e. with
        the old cell temperature, the stored wall value T_w and the surface properties of the composition after solveSpeciesMass.
        The new stored wall value follows after the solve: T_w = T_s,cell(new) + refGrad/deltaCoeffs."""
        out = {}

        def flux(kap):
            if self.radiation is not None:
                a, e = self.surface_radiation()
            else:
                a, e = absorptivity, emissivity
            Tw = self.Twall
            total = kappaDelta_gas * (self._T_old0 - Tgas_cell) - a * qin + e * SIGMA_SB * ((Tw * Tw) * (Tw * Tw))
            out["refGrad"] = -total / kap
            return -total
        self._T_old0 = self.T[:, 0].copy()
        res = self._evolve(dt, flux, None)
        self.Twall = self.T[:, 0] + out["refGrad"] / (2.0 / self.dx)
        return res

    def gas_side(self, rho_b, magSf, nf, hocSolid, qFuel):
        """what the gas region's wall patch reads from the panel after its evolve(): refT (fluid branch of the coupled condition:
        the solid's cell temperature, valueFraction 1), U_b (flowRateInletVelocityPyrolysisCoupled, :127-248) and the wall emissivity
        of greyDiffusiveRadiation with `emissivityMode solidRadiation` (radiationCoupledBase.C:150-182)"""
        hocPyr = (hocSolid * WOOD.rho - HOC_CHAR * CHAR.rho) / (WOOD.rho - CHAR.rho)
        phi = self.massGas * hocPyr / qFuel
        U = (-phi / magSf) / rho_b
        emis = self.surface_radiation()[1] if self.radiation is not None else None
        return self.T[:, 0].copy(), nf * U[:, None], emis

    def diff_no(self, dt):
        """reactingOneDim21::solidRegionDiffNo (reactingOneDim21.C:697-714): max over the internal faces of
        sqr(deltaCoeffs)*interpolate(kappa())/interpolate(Cp()*rho)*deltaT (linear interpolation)"""
        kap, cr = self.kappa(), self.Cp() * self.rho
        r = (1.0 / self.dx) * (1.0 / self.dx) * (0.5 * kap[:, :-1] + 0.5 * kap[:, 1:]) / (0.5 * cr[:, :-1] + 0.5 * cr[:, 1:])
        return float(r.max()) * dt

    def surface_T(self):
        """temperature of the exposed layer's cell"""
        return self.T[:, 0].copy()


SIGMA_SB = 5.670367e-08          # constant::physicoChemical::sigma
HOC_CHAR = 32.8e6               # flowRateInletVelocityPyrolysisCoupledFvPatchVectorField.C:204 "hocChar"


def couple(panel, Twall, Tgas_cell, kappaDelta_gas, qin, emissivity, absorptivity, rho_b, magSf, nf, hocSolid, qFuel):
    """The mapped patch conditions between the gas region's wall patch and the panel (lib/fvPatchFieldsPyrolysis), face by face
    (column i <-> gas boundary face i; the caller applies its map):
      solid side, T   turbulentTemperatureRadiationQinCoupledMixedFvPatchScalarField::updateCoeffs (:176-283), radiative branch:
                      nbrConvFlux = nbrKDelta (T_s,cell - T_g,cell); nbrTotalFlux = nbrConvFlux - a qin + e sigma T_w^4;
                      refGrad = -nbrTotalFlux/kappa_s, valueFraction 0  ->  heat flux INTO the solid q = -nbrTotalFlux and the new
                      wall value T_w = T_s,cell + refGrad/deltaCoeffs_s (deltaCoeffs_s = 2/dx), T_w^4 from the previous wall value
      gas side, T     the same class, fluid branch (:285-292): refValue = the solid's cell temperature, valueFraction 1
      gas side, U     flowRateInletVelocityPyrolysisCoupledFvPatchVectorField::updateCoeffs (:127-248): hocPyr = (hocSolid rho_v -
                      hocChar rho_char)/(rho_v - rho_char); phi = phiGas hocPyr/qFuel; U_b = n (-phi/magSf)/rho_b
    Returns (qSurf [nCol], Twall_new [nCol], refT_gas [nCol], U_b [nCol][3])."""
    Ts = panel.T[:, 0]
    kap = panel.kappa()[:, 0]
    panel._T_old0 = Ts
    conv = kappaDelta_gas * (Ts - Tgas_cell)
    total = conv - absorptivity * qin + emissivity * SIGMA_SB * ((Twall * Twall) * (Twall * Twall))
    refGrad = -total / kap
    Tw = Ts + refGrad / (2.0 / panel.dx)
    hocPyr = (hocSolid * WOOD.rho - HOC_CHAR * CHAR.rho) / (WOOD.rho - CHAR.rho)
    phi = panel.massGas * hocPyr / qFuel
    U = (-phi / magSf) / rho_b
    return -total, Tw, Ts.copy(), nf * U[:, None]


def thomas(lower, diag, upper, rhs):
    """batched Thomas algorithm along axis 1 (lower[:, 0] and upper[:, -1] unused)"""
    n = diag.shape[1]
    c = np.zeros_like(diag); d = np.zeros_like(diag)
    c[:, 0] = upper[:, 0] / diag[:, 0]; d[:, 0] = rhs[:, 0] / diag[:, 0]
    for i in range(1, n):
        den = diag[:, i] - lower[:, i] * c[:, i - 1]
        c[:, i] = upper[:, i] / den
        d[:, i] = (rhs[:, i] - lower[:, i] * d[:, i - 1]) / den
    x = np.zeros_like(diag)
    x[:, -1] = d[:, -1]
    for i in range(n - 2, -1, -1):
        x[:, i] = d[:, i] - c[:, i] * x[:, i + 1]
    return x
