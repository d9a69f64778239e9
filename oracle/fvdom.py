"""ORACLE (test infrastructure): the reference's fvDOM radiation model with its iteration and grey-diffusive walls, SURVEY 8(f) N1:
    fvDOM::calculate          packages/thermophysicalModels/radiation/radiationModels/fvDOM/fvDOM/fvDOM.C:547-584
                              do { for every ray not yet converged: maxResidual = max(ray.correct()) } while (maxResidual >
                              tolerance && radIter < maxIter); updateG()
    fvDOM::initialise         :50-200   the ray set: 3-D 4 nPhi nTheta rays; 2-D (x-y plane, z empty) 4 nPhi rays with theta = pi/2,
                              deltaTheta = pi; 1-D 2 rays
    radiativeIntensityRay     radiativeIntensityRay/radiativeIntensityRay.C:126-143 (d, dAve, omega), :267-322 correct():
                              fvm::div(Ji, Ii, "div(Ji,Ii_h)") + fvm::Sp(k omega, Ii) == 1/pi omega (k sigma T^4 + E/4), Ji = dAve & Sf;
                              returns initialResidual*omega/omegaMax
    greyDiffusiveRadiation    derivedFvPatchFields/greyDiffusiveRadiation/greyDiffusiveRadiationMixedFvPatchScalarField.C:150-230
                              updateCoeffs(): qr += Iw nAve; Ir = sum over rays of their qin on the patch (the rays already solved in
                              this iteration carry this iteration's values); faces the ray leaves the wall through ((-n & d) > 0):
                              fixed value (Ir (1 - e) + e sigma T^4)/pi, qem = value nAve; the others zeroGradient, qin = Iw nAve
    fvDOM::updateG            :680-750  G = sum Ii omega; qin, qem, qr summed over the rays (boundary fields)
Selections of BASELINE config 5 (cases/wallFireSpread2D/constant/radiationProperties:28-46: nPhi 2, nTheta 2, convergence 1e-3, maxIter 5,
solverFreq 10; 0/IDefault:21-35: the panel patch `emissivityMode solidRadiation`, every other patch emissivity 1; system/fvSchemes:66
div(Ji,Ii_h) Gauss linearUpwind grad(Ii_h); system/fvSolution:160-170 Ii by GAMG + DILU to 1e-4) and of cases/steckler (maxIter 1,
emissivity 1: pinned by the golden log through oracle/steckler_case.py, which this module reproduces on that case).
When updateCoeffs() runs: at the construction of the ray's fvMatrix (fvMatrix::fvMatrix calls psi.boundaryFieldRef().updateCoeffs()),
i.e. with the patch values Iw the LAST evaluate() of that ray left -- the cell values of its previous solve on zeroGradient faces.
PARITY: the emissivity-1 / maxIter-1 path is pinned by the steckler log; reflecting walls and the iteration are restated from the
files above and unpinned by reference data (the reference ships no output of wallFireSpread2D).  Only tests/ may import this."""
import numpy as np

from . import fv

SIGMA_SB = 5.670367e-8


def ray_set(nPhi, nTheta, solutionD=(1, 1, 1)):
    """[(d, dAve, omega)] as fvDOM::initialise builds them for the mesh's solution directions"""
    nD = sum(1 for s in solutionD if s > 0)
    rays = []

    def ray(phi, theta, dPhi, dTheta):
        st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
        omega = 2.0 * st * np.sin(dTheta / 2.0) * dPhi
        d = np.array([st * sp, st * cp, ct])
        c = np.sin(0.5 * dPhi) * (dTheta - np.cos(2.0 * theta) * np.sin(dTheta))
        dAve = np.array([sp * c, cp * c, 0.5 * dPhi * np.sin(2.0 * theta) * np.sin(dTheta)])
        return d, dAve, omega
    if nD == 3:
        dPhi, dTheta = np.pi / (2.0 * nPhi), np.pi / nTheta
        for n in range(1, nTheta + 1):
            for m in range(1, 4 * nPhi + 1):
                rays.append(ray((2.0 * m - 1.0) * dPhi / 2.0, (2.0 * n - 1.0) * dTheta / 2.0, dPhi, dTheta))
    elif nD == 2:
        if solutionD[2] != -1:
            raise ValueError("Currently 2D solution is limited to the x-y plane")        # fvDOM.C:103-109
        dPhi = np.pi / (2.0 * nPhi)
        for m in range(1, 4 * nPhi + 1):
            rays.append(ray((2.0 * m - 1.0) * dPhi / 2.0, np.pi / 2.0, dPhi, np.pi))
    else:
        if solutionD[0] != 1:
            raise ValueError("Currently 1D solution is limited to the x-direction")
        for m in (1, 2):
            rays.append(ray((2.0 * m - 1.0) * np.pi / 2.0, np.pi / 2.0, np.pi, np.pi))
    return rays


class FvDOM:
    """solve(name, diag, upper, lower, source, psi0) -> (psi, perf) is the caller's linear solver (the case's `Ii` entry)."""

    def __init__(self, m, nPhi, nTheta, solve, maxIter=50, tolerance=0.0, divScheme="upwind", solutionD=(1, 1, 1), emissivity=None):
        self.m, self.solve, self.maxIter, self.tolerance, self.divScheme = m, solve, maxIter, tolerance, divScheme
        self.rays = ray_set(nPhi, nTheta, solutionD)
        self.omegaMax = max(r[2] for r in self.rays)
        nR = len(self.rays)
        zb = lambda: [np.zeros(p.size) for p in m.patches]
        self.I = [np.zeros(m.nCells) for _ in range(nR)]
        self.Ib = [zb() for _ in range(nR)]                          # stored patch values (0/IDefault: value uniform 0)
        self.qin_ray, self.qem_ray, self.qr_ray = [zb() for _ in range(nR)], [zb() for _ in range(nR)], [zb() for _ in range(nR)]
        self.emissivity = emissivity if emissivity is not None else [np.ones(p.size) for p in m.patches]
        self.G = np.zeros(m.nCells)
        self.qin, self.qem, self.qr = zb(), zb(), zb()
        self.log, self.nIterations = [], 0

    def ray_correct(self, i, T, Tb, a, E):
        m = self.m
        d, dAve, omega = self.rays[i]
        nP = len(m.patches)
        self.qr_ray[i] = [np.zeros(p.size) for p in m.patches]
        self.qin_ray[i] = [np.zeros(p.size) for p in m.patches]
        self.qem_ray[i] = [np.zeros(p.size) for p in m.patches]
        Ji = (dAve[0] * m.Sf[:, 0] + dAve[1] * m.Sf[:, 1]) + dAve[2] * m.Sf[:, 2]
        Jib = [(dAve[0] * p.Sf[:, 0] + dAve[1] * p.Sf[:, 1]) + dAve[2] * p.Sf[:, 2] for p in m.patches]
        f, ref = [], []
        for q, p in enumerate(m.patches):
            n = p.Sf / p.magSf[:, None]
            nAve = (n[:, 0] * dAve[0] + n[:, 1] * dAve[1]) + n[:, 2] * dAve[2]
            Iw = self.Ib[i][q]
            self.qr_ray[i][q] = self.qr_ray[i][q] + Iw * nAve
            Ir = self.qin_ray[0][q].copy()
            for j in range(1, len(self.rays)):
                Ir = Ir + self.qin_ray[j][q]
            out = -((n[:, 0] * d[0] + n[:, 1] * d[1]) + n[:, 2] * d[2]) > 0.0
            e = self.emissivity[q]
            tb = Tb[q]
            val = (Ir * (1.0 - e) + e * SIGMA_SB * ((tb * tb) * (tb * tb))) / np.pi
            f.append(np.where(out, 1.0, 0.0)); ref.append(np.where(out, val, 0.0))
            self.qem_ray[i][q] = np.where(out, val * nAve, 0.0)
            self.qin_ray[i][q] = np.where(out, 0.0, Iw * nAve)
        bc = fv.MixedBC(m, f=f, ref=ref)
        M = fv.fvm_div(m, Ji, Jib, fv.pos0(Ji), [bc])
        if self.divScheme == "linearUpwind":
            # gaussConvectionScheme::fvmDiv with a corrected scheme: fvm += fvc::surfaceIntegrate(faceFlux*correction(vf)); the gradient of
            # grad(Ii_h) (Gauss linear) from the stored patch values
            g = fv.grad(m, self.I[i], self.Ib[i])
            M.add_vol(fv.surface_integrate(m, Ji * fv.linear_upwind_correction(m, Ji, g), [np.zeros(p.size) for p in m.patches]))
        elif self.divScheme != "upwind":
            raise ValueError(self.divScheme)
        M.diag += m.V * (a * omega)
        T4 = (T * T) * (T * T)
        M.add_su(1.0 / np.pi * omega * (a * SIGMA_SB * T4 + E / 4.0))
        dg, s = M.solve_system()
        self.I[i], perf = self.solve("ILambda_%d_0" % i, dg, M.upper, M.lower, s, self.I[i])
        self.log.append(("ILambda_%d_0" % i, perf))
        self.Ib[i] = bc.values(m, self.I[i])                          # correctBoundaryConditions()
        return perf["initialResidual"] * omega / self.omegaMax

    def calculate(self, T, Tb, a, E):
        """T [N], Tb per patch; a: absorption coefficient (scalar or [N]); E: emission [N] (absorptionEmission->E())"""
        nR = len(self.rays)
        conv = [False] * nR
        self.log = []
        it = 0
        while True:
            it += 1
            maxRes = 0.0
            for i in range(nR):
                if not conv[i]:
                    r = self.ray_correct(i, T, Tb, a, E)
                    maxRes = max(r, maxRes)
                    if r < self.tolerance:
                        conv[i] = True
            if not (maxRes > self.tolerance and it < self.maxIter):
                break
        self.nIterations = it
        self.update_G()

    def update_G(self):
        m = self.m
        self.G = np.zeros(m.nCells)
        zb = lambda: [np.zeros(p.size) for p in m.patches]
        self.qin, self.qem, self.qr = zb(), zb(), zb()
        for i, (_, _, omega) in enumerate(self.rays):
            self.G = self.G + self.I[i] * omega
            for q in range(len(m.patches)):
                self.qin[q] = self.qin[q] + self.qin_ray[i][q]
                self.qem[q] = self.qem[q] + self.qem_ray[i][q]
                self.qr[q] = self.qr[q] + self.qr_ray[i][q]
