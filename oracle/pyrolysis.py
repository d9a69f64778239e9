"""ORACLE (test infrastructure): one time step of the reference's 1-D solid pyrolysis region model, SURVEY 8(f) N3:
reactingOneDim::evolveRegion (packages/regionModels/pyrolysisModels/reactingOneDim/reactingOneDim.C:686-721) =
    solidChemistry->calculate()   one Arrhenius solid reaction  wood^n = char + gas  (cases/pyrolysis1D/constant/panelRegion/reactions)
    solveContinuity()   fvm::ddt(rho) == -RRg                                   (:240-266)   diagonal
    solveSpeciesMass()  fvm::ddt(rho, Yi) == RRs(i); Yi.max(0); last = 1 - Yt    (:269-303)   diagonal
    solveEnergy()       fvm::ddt(rho, h) - fvm::laplacian(alpha, h) + fvc::laplacian(alpha, h) - fvc::laplacian(kappa, T)
                        == chemistryQdot - fvm::Sp(RRg, h)                        (:306-353)   one tridiagonal system per column
    solidThermo.correct()   T from h (hConst thermo: h = Cp (T - Tstd)), kappa and Cp mass-fraction weighted
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
    """nCol columns x nLay layers; layer 0 is the exposed (coupled) one.  Fields [nCol][nLay]."""

    def __init__(self, nCol, nLay=8, thickness=0.0127, area=1.0, T0=298.15, Yw0=1.0):
        self.nCol, self.nLay = nCol, nLay
        self.dx = thickness / nLay
        self.A = area
        self.V = self.A * self.dx
        self.Yw0 = Yw0
        self.Yw = np.full((nCol, nLay), Yw0)
        self.rho = np.full((nCol, nLay), 1.0 / (Yw0 / WOOD.rho + (1 - Yw0) / CHAR.rho)) if Yw0 < 1 else np.full((nCol, nLay), WOOD.rho)
        self.T = np.full((nCol, nLay), float(T0))
        self.h = self.Cp() * (self.T - TSTD)
        self.massGas = np.zeros(nCol)
        self.c0 = self.rho[0, 0] * max(Yw0, 0.001)            # initial partial density of the virgin solid (Ys0_/V)

    def Cp(self):
        return self.Yw * WOOD.Cp + (1.0 - self.Yw) * CHAR.Cp

    def kappa(self):
        return self.Yw * WOOD.kappa + (1.0 - self.Yw) * CHAR.kappa

    def step(self, dt, qSurf, Tback=None):
        """qSurf[nCol]: heat flux into the exposed face [W/m2]; back face adiabatic (Tback None) or held at Tback"""
        R = REACTION
        rdt = 1.0 / dt
        rho0, Yw0f, h0, T0 = self.rho.copy(), self.Yw.copy(), self.h.copy(), self.T.copy()
        kappa, Cp = self.kappa(), self.Cp()
        alpha = kappa / Cp
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
        af = 0.5 * (alpha[:, :-1] + alpha[:, 1:]); kf_ = 0.5 * (kappa[:, :-1] + kappa[:, 1:])     # linear interpolation, uniform layers
        c_a = af * self.A / self.dx; c_k = kf_ * self.A / self.dx
        diag = rdt * self.rho * self.V + self.V * RRg
        lower = np.zeros((self.nCol, nL)); upper = np.zeros((self.nCol, nL))
        upper[:, :-1] = -c_a; lower[:, 1:] = -c_a
        diag[:, :-1] += c_a; diag[:, 1:] += c_a
        src = rdt * rho0 * h0 * self.V + self.V * Qdot
        # + fvc::laplacian(alpha, h) - fvc::laplacian(kappa, T) on the LHS: source -= V*(lapA - lapK)
        fa = c_a * (h0[:, 1:] - h0[:, :-1]); fk = c_k * (T0[:, 1:] - T0[:, :-1])
        lapA = np.zeros((self.nCol, nL)); lapK = np.zeros((self.nCol, nL))
        lapA[:, :-1] += fa; lapA[:, 1:] -= fa
        lapK[:, :-1] += fk; lapK[:, 1:] -= fk
        lapK[:, 0] += np.asarray(qSurf) * self.A               # boundary face of fvc::laplacian(kappa, T): kappa snGrad(T) A = q A
        if Tback is not None:                                    # fixed temperature at the back face
            db = 2.0 / self.dx
            lapK[:, -1] += kappa[:, -1] * self.A * db * (Tback - T0[:, -1])
        src -= (lapA - lapK)
        self.h = thomas(lower, diag, upper, src)
        self.massGas = (RRg * self.V).sum(axis=1)               # phiGas through the exposed face [kg/s]
        # ---- solidThermo.correct()
        self.T = TSTD + self.h / self.Cp()
        return dict(RRg=RRg, Qdot=Qdot, lower=lower, diag=diag, upper=upper, src=src)

    def surface_T(self):
        """temperature of the exposed face: zero-curvature extrapolation is not used upstream; the coupled patch takes the
        fixed-gradient value T_c + q dx/(2 kappa) from the last heat flux (kept by the caller) -- here the cell value"""
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
