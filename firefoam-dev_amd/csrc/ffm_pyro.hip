// ffm_pyro.hip -- SURVEY 8(f) N3: the reference's 1-D solid pyrolysis region (reactingOneDim::evolveRegion,
// packages/regionModels/pyrolysisModels/reactingOneDim/reactingOneDim.C:686-721: solidChemistry->calculate(), solveContinuity
// :240-266, solveSpeciesMass :269-303, solveEnergy :306-353, solidThermo.correct()) for a panel of independent columns.
//
// The region mesh is extruded from a wall patch of the gas mesh (cases/wallFireSpread2D/system/extrudeToRegionMeshDict:17-39,
// nLayers 8; cases/pyrolysis1D: one column of 8 layers): its LDU graph is nCol disjoint chains, so the reference's three
// fvMatrix solves per step are, per column, two diagonal updates and one tridiagonal system of nLay unknowns.  On the device
// that is ONE kernel: a thread owns a column, keeps its nLay cells in registers, runs the Arrhenius rate, the continuity and
// species updates, assembles the enthalpy equation in the reference's term order and solves it exactly with the Thomas
// algorithm (the reference iterates PCG to 1e-6 on the same matrix).  Fields are stored layer-major ([layer][column]) so that
// the threads of a wave read consecutive addresses; a step streams 4 fields in and out once: 64 B per cell, HBM-bound.
// Coupling with the gas region (lib/fvPatchFieldsPyrolysis): in, the heat flux into every column's exposed face; out, that
// face's cell temperature and the pyrolysate mass flux phiGas of the column.
// oracle/pyrolysis.py is the CPU restatement (same operation order); 'parity unpinned' by reference data (see its header).
#include "ffm_internal.hpp"
#include <algorithm>
#include <cmath>
#include <string>

constexpr int PYRO_MAX_LAYERS = 16;

struct PyroConst { double rhoW, CpW, kW, HfW, rhoC, CpC, kC, HfC, A, Ta, Tcrit, n, c0, dx, area, V; };   // c0: initial partial density of the virgin solid

// selections of the region's dictionaries (ffm_pyro_set_model / _set_back / _set_surface_radiation):
//   model21      0: reactingOneDim (.../reactingOneDim/reactingOneDim.C:306-353: == chemistryQdot - fvm::Sp(RRg, h)); 1: reactingOneDim21
//                (lib/regionModels/pyrolysisModels/reactingOneDim21/reactingOneDim21.C:319-368: == chemistryQdot + RRs(0) T Cp0 + RRs(1) T Cp1)
//   harmA/harmK  laplacian(thermo:alpha,h) / laplacian(kappa,T) with `Gauss harmonic` instead of `Gauss linear` interpolation
//   backMode     0 zero gradient, 1 fixed temperature Tinf, 2 constHTemperature (h = backH, Tinf): valueFraction 1/(1 + kappa/max(h,SMALL) deltaCoeffs)
//   surfRad      greyMeanSolidAbsorptionEmission: surface absorptivity / emissivity from the exposed layer's volume fractions
struct PyroOpts { int model21, harmA, harmK, backMode, surfRad; double backH, Tinf, aW, eW, aC, eC; };

struct ffm_pyro {
    ffm_ctx *ctx = nullptr;
    int nCol = 0, nLay = 0;
    PyroConst k;
    PyroOpts o{0, 0, 0, 0, 0, 0.0, 298.15, 1.0, 1.0, 1.0, 1.0};
    double *rho = nullptr, *Yw = nullptr, *T = nullptr, *h = nullptr, *alpha = nullptr;   // [nLay][nCol]; alpha: thermo's alpha_ of the last correct()
    double *Tsurf = nullptr, *phiGas = nullptr;                               // [nCol]
    double T0 = 298.15, Yw0 = 1.0;                                            // uniform start state
    double *qSurf = nullptr, *Twall = nullptr;                                // [nCol] coupled heat flux / wall temperature (ffm_pyro_couple_d, ffm_pyro_evolve_d)
};

// coupling inputs of ffm_pyro_evolve_d (null Tg: the flux comes from qSurf): gas-side cell temperature, kappaEff*deltaCoeffs and
// incident radiation on the gas boundary faces, column c <-> face map[c]
struct PyroCouple { const int *map; const double *Tg, *kDelta, *qin; double emis, absorp; };

__device__ inline double pyro_face(int harmonic, double a, double b) { return harmonic ? 1.0 / (0.5 / a + 0.5 / b) : 0.5 * (a + b); }

template <int NL>
__global__ __launch_bounds__(256) void k_pyro_step(int nCol, PyroConst k, PyroOpts o, double dt, const double *__restrict__ qSurfIn, int backFixed, double Tback,
                                                   PyroCouple cp_, double *__restrict__ rho_, double *__restrict__ Yw_, double *__restrict__ T_,
                                                   double *__restrict__ h_, double *__restrict__ alpha_, double *__restrict__ Tsurf,
                                                   double *__restrict__ phiGas, double *__restrict__ Twall, double *__restrict__ qSurfOut)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    const double rdt = 1.0 / dt, TSTD = 298.15, V = k.V, A = k.area, dx = k.dx;
    double rho0[NL], h0[NL], T0[NL], kap[NL], alp[NL], RRg[NL], src[NL], rho[NL], Yw[NL];
    const double sr = k.rhoC / k.rhoW;
    double gas = 0.0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const size_t oo = (size_t)i * nCol + c;
        rho0[i] = rho_[oo]; h0[i] = h_[oo]; T0[i] = T_[oo]; alp[i] = alpha_[oo];
        const double Yw0 = Yw_[oo];
        // solidChemistry->calculate(): irreversibleArrheniusSolidReaction wood^n = char + gas
        const double kf = T0[i] < k.Tcrit ? 0.0 : k.A * exp(-k.Ta / T0[i]);
        const double omega = kf * pow(rho0[i] * Yw0 / k.c0, k.n) * k.c0;          // pyrolysisChemistryModel::omega: kf (m/m0)^n m0, per volume
        const double RRw = -omega, RRc = sr * omega;
        RRg[i] = (1.0 - sr) * omega;
        const double Qd = -(k.HfW * RRw + k.HfC * RRc);
        rho[i] = (rdt * rho0[i] * V - V * RRg[i]) / (rdt * V);                               // solveContinuity
        Yw[i] = fmax((rdt * rho0[i] * Yw0 * V + V * RRw) / (rdt * rho[i] * V), 0.0);       // solveSpeciesMass
        gas += RRg[i] * V;
        // solidThermo.kappa() = Cp()*alpha_: the heat capacity of the composition after solveSpeciesMass, alpha_ of the last correct()
        kap[i] = (Yw[i] * k.CpW + (1.0 - Yw[i]) * k.CpC) * alp[i];
        src[i] = rdt * rho0[i] * h0[i] * V + V * Qd;
        if (o.model21) {
            src[i] = src[i] + V * (RRw * T0[i] * k.CpW);                                    // + RRs(0)*T*Cp0
            src[i] = src[i] + V * (RRc * T0[i] * k.CpC);                                    // + RRs(1)*T*Cp1
        }
    }
    // solveEnergy: fvm::ddt(rho,h) - fvm::laplacian(alpha,h) + fvc::laplacian(alpha,h) - fvc::laplacian(kappa,T) == sources
    double dg[NL], lo[NL], up[NL], lapA[NL], lapK[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { dg[i] = rdt * rho[i] * V; if (!o.model21) dg[i] = dg[i] + V * RRg[i]; lo[i] = 0.0; up[i] = 0.0; lapA[i] = 0.0; lapK[i] = 0.0; }
#pragma unroll
    for (int i = 0; i + 1 < NL; i++) {
        const double ca = pyro_face(o.harmA, alp[i], alp[i + 1]) * A / dx, ck = pyro_face(o.harmK, kap[i], kap[i + 1]) * A / dx;
        up[i] = -ca; lo[i + 1] = -ca;
        dg[i] += ca; dg[i + 1] += ca;
        const double fa = ca * (h0[i + 1] - h0[i]), fk = ck * (T0[i + 1] - T0[i]);
        lapA[i] += fa; lapA[i + 1] -= fa;
        lapK[i] += fk; lapK[i + 1] -= fk;
    }
    // exposed face: the coupled patch of T (turbulentTemperatureRadiationQinCoupledMixed, solid branch) evaluated here, where the
    // reference evaluates it (construction of hEqn): old cell temperature, stored wall value, surface properties of the new composition
    double q, refGrad = 0.0;
    if (cp_.Tg) {
        const int b = cp_.map ? cp_.map[c] : c;
        double a = cp_.absorp, e = cp_.emis;
        if (o.surfRad) {
            const double X = (Yw[0] / k.rhoW) / (Yw[0] / k.rhoW + (1.0 - Yw[0]) / k.rhoC);
            a = X * o.aW + (1.0 - X) * o.aC; e = X * o.eW + (1.0 - X) * o.eC;
        }
        const double tw = Twall[c];
        const double total = cp_.kDelta[b] * (T0[0] - cp_.Tg[b]) - a * (cp_.qin ? cp_.qin[b] : 0.0) + e * 5.670367e-08 * ((tw * tw) * (tw * tw));
        refGrad = -total / kap[0];
        q = -total;
    } else q = qSurfIn[c];
    lapK[0] += q * A;
    {   // back face: mixed condition (f, Tinf)
        const double db = 2.0 / dx;
        int mode = backFixed ? 1 : o.backMode; const double Tinf = backFixed ? Tback : o.Tinf;
        if (mode) {
            const double f = mode == 1 ? 1.0 : 1.0 / (1.0 + kap[NL - 1] / fmax(o.backH, 1e-15) * db);
            lapK[NL - 1] += kap[NL - 1] * A * db * f * (Tinf - T0[NL - 1]);
            const double cb = alp[NL - 1] * A * db * f;
            dg[NL - 1] += cb;
            src[NL - 1] += cb * h0[NL - 1];
        }
    }
#pragma unroll
    for (int i = 0; i < NL; i++) src[i] -= (lapA[i] - lapK[i]);
    // Thomas algorithm
    double cp[NL], dp[NL], x[NL];
    cp[0] = up[0] / dg[0]; dp[0] = src[0] / dg[0];
#pragma unroll
    for (int i = 1; i < NL; i++) {
        const double den = dg[i] - lo[i] * cp[i - 1];
        cp[i] = up[i] / den;
        dp[i] = (src[i] - lo[i] * dp[i - 1]) / den;
    }
    x[NL - 1] = dp[NL - 1];
#pragma unroll
    for (int i = NL - 2; i >= 0; i--) x[i] = dp[i] - cp[i] * x[i + 1];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const size_t oo = (size_t)i * nCol + c;
        const double Cp = Yw[i] * k.CpW + (1.0 - Yw[i]) * k.CpC;
        const double X = (Yw[i] / k.rhoW) / (Yw[i] / k.rhoW + (1.0 - Yw[i]) / k.rhoC);
        const double t = TSTD + x[i] / Cp;
        rho_[oo] = rho[i]; Yw_[oo] = Yw[i]; h_[oo] = x[i]; T_[oo] = t;                        // solidThermo.correct()
        alpha_[oo] = (X * k.kW + (1.0 - X) * k.kC) / Cp;
        if (i == 0) {
            Tsurf[c] = t;
            if (cp_.Tg) { Twall[c] = t + refGrad / (2.0 / dx); qSurfOut[c] = q; }
        }
    }
    phiGas[c] = gas;
}

// the region's start state (uniform T0, Yw0) from the solids' properties: rho, h, alpha_ of the constructor's correct() (volume-fraction-
// weighted conductivity over mass-fraction-weighted heat capacity), Ys0_ of the chemistry model
static int pyro_init_state(ffm_pyro *P)
{
    const double Yw0 = P->Yw0, T0 = P->T0;
    const double rho0 = Yw0 < 1 ? 1.0 / (Yw0 / P->k.rhoW + (1 - Yw0) / P->k.rhoC) : P->k.rhoW;
    P->k.c0 = rho0 * std::max(Yw0, 0.001);
    const double Cp = Yw0 * P->k.CpW + (1.0 - Yw0) * P->k.CpC;
    const double X = (Yw0 / P->k.rhoW) / (Yw0 / P->k.rhoW + (1.0 - Yw0) / P->k.rhoC);
    const size_t n = (size_t)P->nCol * P->nLay;
    std::vector<double> v(n);
    auto fill = [&](double *d, double val, size_t m) { std::fill(v.begin(), v.begin() + m, val); return ffm_memcpy_h2d(P->ctx, d, v.data(), sizeof(double) * m); };
    FFM_TRY(fill(P->rho, rho0, n)); FFM_TRY(fill(P->Yw, Yw0, n)); FFM_TRY(fill(P->T, T0, n)); FFM_TRY(fill(P->h, Cp * (T0 - 298.15), n));
    FFM_TRY(fill(P->alpha, (X * P->k.kW + (1.0 - X) * P->k.kC) / Cp, n));
    FFM_TRY(fill(P->phiGas, 0.0, P->nCol)); FFM_TRY(fill(P->qSurf, 0.0, P->nCol));
    FFM_TRY(fill(P->Tsurf, T0, P->nCol)); FFM_TRY(fill(P->Twall, T0, P->nCol));
    return FFM_OK;
}

extern "C" int ffm_pyro_create(ffm_ctx *ctx, int nCol, int nLay, double thickness, double faceArea, double T0, double Yw0, ffm_pyro **out)
{
    if (!ctx || !out || nCol < 1 || nLay < 2 || nLay > PYRO_MAX_LAYERS || thickness <= 0 || faceArea <= 0 || Yw0 <= 0 || Yw0 > 1) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(ctx->device));
    ffm_pyro *P = new ffm_pyro();
    P->ctx = ctx; P->nCol = nCol; P->nLay = nLay;
    // cases/pyrolysis1D/constant/panelRegion/{thermo.solid,reactions}: the defaults; ffm_pyro_set_solids / _set_reaction replace them
    P->k = PyroConst{114.7, 696.0, 0.135, -1.41e6, 11.5, 611.0, 0.4, 0.0, 7.83e10, 15274.57, 400.0, 4.86, 0.0, thickness / nLay, faceArea, faceArea * (thickness / nLay)};
    P->T0 = T0; P->Yw0 = Yw0;
    const size_t n = (size_t)nCol * nLay;
    double **f[5] = {&P->rho, &P->Yw, &P->T, &P->h, &P->alpha};
    for (auto p : f) FFM_HIP(hipMalloc((void **)p, sizeof(double) * n));
    double **g[4] = {&P->Tsurf, &P->phiGas, &P->qSurf, &P->Twall};
    for (auto p : g) FFM_HIP(hipMalloc((void **)p, sizeof(double) * nCol));
    FFM_TRY(pyro_init_state(P));
    *out = P;
    return FFM_OK;
}

extern "C" int ffm_pyro_set_solids(ffm_pyro *P, const double *virgin, const double *charred)
{
    if (!P || !virgin || !charred) return FFM_ERR_ARG;
    P->k.rhoW = virgin[0]; P->k.CpW = virgin[1]; P->k.kW = virgin[2]; P->k.HfW = virgin[3];
    P->k.rhoC = charred[0]; P->k.CpC = charred[1]; P->k.kC = charred[2]; P->k.HfC = charred[3];
    return pyro_init_state(P);          // the dictionaries are read before the first step: the start state follows the solids' properties
}
extern "C" int ffm_pyro_set_reaction(ffm_pyro *P, double A, double Ta, double Tcrit, double n)
{ if (!P) return FFM_ERR_ARG; P->k.A = A; P->k.Ta = Ta; P->k.Tcrit = Tcrit; P->k.n = n; return FFM_OK; }

static int pyro_launch(ffm_pyro *P, double dt, const double *qSurf_d, int backFixed, double Tback, const PyroCouple &cp)
{
    const dim3 grid((P->nCol + 255) / 256), block(256);
    hipStream_t s = P->ctx->stream;
#define PYRO(NL) hipLaunchKernelGGL(k_pyro_step<NL>, grid, block, 0, s, P->nCol, P->k, P->o, dt, qSurf_d, backFixed, Tback, cp, P->rho, P->Yw, P->T, P->h, P->alpha, P->Tsurf, P->phiGas, P->Twall, P->qSurf)
    switch (P->nLay) {
    case 2: PYRO(2); break; case 3: PYRO(3); break; case 4: PYRO(4); break; case 5: PYRO(5); break; case 6: PYRO(6); break;
    case 7: PYRO(7); break; case 8: PYRO(8); break; case 9: PYRO(9); break; case 10: PYRO(10); break; case 11: PYRO(11); break;
    case 12: PYRO(12); break; case 13: PYRO(13); break; case 14: PYRO(14); break; case 15: PYRO(15); break; default: PYRO(16); break;
    }
#undef PYRO
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_pyro_step(ffm_pyro *P, double dt, const double *qSurf_d, int backFixed, double Tback)
{
    if (!P || !qSurf_d || dt <= 0) return FFM_ERR_ARG;
    return pyro_launch(P, dt, qSurf_d, backFixed, Tback, PyroCouple{nullptr, nullptr, nullptr, nullptr, 1.0, 1.0});
}

// evolveRegion with the exposed face coupled to the gas region: see include/ffm.h
extern "C" int ffm_pyro_evolve_d(ffm_pyro *P, double dt, const int *map_d, const double *TgasCell_d, const double *kappaDelta_d, const double *qin_d,
                                 double emissivity, double absorptivity)
{
    if (!P || !TgasCell_d || !kappaDelta_d || dt <= 0) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(P->ctx->device));
    return pyro_launch(P, dt, nullptr, 0, 0.0, PyroCouple{map_d, TgasCell_d, kappaDelta_d, qin_d, emissivity, absorptivity});
}

extern "C" int ffm_pyro_set_model(ffm_pyro *P, int reactingOneDim21, int harmonicAlpha, int harmonicKappa)
{ if (!P) return FFM_ERR_ARG; P->o.model21 = reactingOneDim21 != 0; P->o.harmA = harmonicAlpha != 0; P->o.harmK = harmonicKappa != 0; return FFM_OK; }
extern "C" int ffm_pyro_set_back(ffm_pyro *P, int mode, double h, double Tinf)
{ if (!P || mode < 0 || mode > 2 || h < 0) return FFM_ERR_ARG; P->o.backMode = mode; P->o.backH = h; P->o.Tinf = Tinf; return FFM_OK; }
extern "C" int ffm_pyro_set_surface_radiation(ffm_pyro *P, double absorptivityVirgin, double emissivityVirgin, double absorptivityChar, double emissivityChar)
{
    if (!P || absorptivityVirgin < 0 || emissivityVirgin < 0 || absorptivityChar < 0 || emissivityChar < 0) return FFM_ERR_ARG;
    P->o.surfRad = 1; P->o.aW = absorptivityVirgin; P->o.eW = emissivityVirgin; P->o.aC = absorptivityChar; P->o.eC = emissivityChar;
    return FFM_OK;
}

// what the gas region's wall patch reads from the panel AFTER its evolve (one thread per column, column c <-> gas face map[c]):
// refT = the solid's cell temperature (fluid branch of turbulentTemperatureRadiationQinCoupledMixed, :297-303), U_b of
// flowRateInletVelocityPyrolysisCoupled (:127-248) from the new phiGas, and -- emis_d non-null -- the wall emissivity that
// greyDiffusiveRadiation with `emissivityMode solidRadiation` takes from the solid (radiationCoupledBase.C:150-182)
__global__ void k_pyro_gas_side(int nCol, PyroConst k, PyroOpts o, const int *__restrict__ map, const double *__restrict__ T_, const double *__restrict__ Yw_,
                                const double *__restrict__ phiGas, const double *__restrict__ rhob, const double *__restrict__ magSf,
                                const double *__restrict__ nx, const double *__restrict__ ny, const double *__restrict__ nz, double hocPyr, double qFuel,
                                double *__restrict__ refT, double *__restrict__ Ux, double *__restrict__ Uy, double *__restrict__ Uz, double *__restrict__ emis)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    const int b = map ? map[c] : c;
    refT[b] = T_[c];
    const double phi = phiGas[c] * hocPyr / qFuel;
    const double U = (-phi / magSf[b]) / rhob[b];
    Ux[b] = nx[b] * U; Uy[b] = ny[b] * U; Uz[b] = nz[b] * U;
    if (emis && o.surfRad) {
        const double Yw = Yw_[c];
        const double X = (Yw / k.rhoW) / (Yw / k.rhoW + (1.0 - Yw) / k.rhoC);
        emis[b] = X * o.eW + (1.0 - X) * o.eC;
    }
}
extern "C" int ffm_pyro_gas_side_d(ffm_pyro *P, const int *map_d, const double *rho_b_d, const double *magSf_d, const double *nfx_d, const double *nfy_d,
                                   const double *nfz_d, double hocSolid, double qFuel, double *refT_d, double *Ux_d, double *Uy_d, double *Uz_d, double *emissivity_d)
{
    if (!P || !rho_b_d || !magSf_d || !nfx_d || !nfy_d || !nfz_d || !refT_d || !Ux_d || !Uy_d || !Uz_d || !(qFuel > 0)) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(P->ctx->device));
    const double hocPyr = (hocSolid * P->k.rhoW - 32.8e6 * P->k.rhoC) / (P->k.rhoW - P->k.rhoC);
    hipLaunchKernelGGL(k_pyro_gas_side, dim3((P->nCol + 255) / 256), dim3(256), 0, P->ctx->stream, P->nCol, P->k, P->o, map_d, (const double *)P->T,
                       (const double *)P->Yw, (const double *)P->phiGas, rho_b_d, magSf_d, nfx_d, nfy_d, nfz_d, hocPyr, qFuel, refT_d, Ux_d, Uy_d, Uz_d, emissivity_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_pyro_get(ffm_pyro *P, const char *name, double *out)
{
    if (!P || !name || !out) return FFM_ERR_ARG;
    const std::string n(name);
    const double *src = n == "rho" ? P->rho : n == "Yw" ? P->Yw : n == "T" ? P->T : n == "h" ? P->h : n == "alpha" ? P->alpha : nullptr;
    FFM_HIP(hipStreamSynchronize(P->ctx->stream));
    if (src) {      // [nLay][nCol] on the device -> [nCol][nLay] for the host
        std::vector<double> v((size_t)P->nCol * P->nLay);
        FFM_TRY(ffm_memcpy_d2h(P->ctx, v.data(), src, sizeof(double) * v.size()));
        for (int c = 0; c < P->nCol; c++) for (int i = 0; i < P->nLay; i++) out[(size_t)c * P->nLay + i] = v[(size_t)i * P->nCol + c];
        return FFM_OK;
    }
    const double *col = n == "Tsurf" ? P->Tsurf : n == "phiGas" ? P->phiGas : n == "qSurf" ? P->qSurf : n == "Twall" ? P->Twall : nullptr;
    if (!col) { ffm_set_error("ffm_pyro_get: unknown field %s", name); return FFM_ERR_ARG; }
    FFM_TRY(ffm_memcpy_d2h(P->ctx, out, col, sizeof(double) * P->nCol));
    return FFM_OK;
}
// The mapped patch conditions between the gas region's wall patch and the panel (lib/fvPatchFieldsPyrolysis), one thread per
// column; column i is coupled to gas boundary face map[i] (mappedPatchBase::distribute of an extruded region: a permutation).
//   solid side, T  turbulentTemperatureRadiationQinCoupledMixedFvPatchScalarField::updateCoeffs (:176-283, radiative branch):
//                  nbrTotalFlux = nbrKDelta (T_s,cell - T_g,cell) - a qin + e sigma T_w^4; refGrad = -nbrTotalFlux/kappa_s, f = 0
//                  -> qSurf = -nbrTotalFlux (what ffm_pyro_step takes), T_w = T_s,cell + refGrad dx/2 (T_w^4 from the previous T_w)
//   gas side, T    the same class, fluid branch (:285-292): refValue = T_s,cell, valueFraction 1
//   gas side, U    flowRateInletVelocityPyrolysisCoupledFvPatchVectorField::updateCoeffs (:127-248):
//                  phi = phiGas (hocSolid rho_v - hocChar rho_char)/(rho_v - rho_char)/qFuel; U_b = n (-phi/magSf)/rho_b
__global__ void k_pyro_couple(int nCol, PyroConst k, const int *__restrict__ map, const double *__restrict__ Ts_, const double *__restrict__ Yw_,
                              const double *__restrict__ alpha_, const double *__restrict__ phiGas, double *__restrict__ Twall, double *__restrict__ qSurf,
                              const double *__restrict__ Tg, const double *__restrict__ kDelta, const double *qin, double emis, double absorp,
                              const double *__restrict__ rhob, const double *__restrict__ magSf, const double *__restrict__ nx,
                              const double *__restrict__ ny, const double *__restrict__ nz, double hocPyr, double qFuel,
                              double *__restrict__ refT, double *__restrict__ Ux, double *__restrict__ Uy, double *__restrict__ Uz)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    const int b = map ? map[c] : c;
    const double Ts = Ts_[c], Yw = Yw_[c];                      // layer 0
    const double kap = (Yw * k.CpW + (1.0 - Yw) * k.CpC) * alpha_[c];      // solidThermo.kappa() = Cp()*alpha_
    const double tw = Twall[c];
    const double conv = kDelta[b] * (Ts - Tg[b]);
    const double total = conv - absorp * (qin ? qin[b] : 0.0) + emis * 5.670367e-08 * ((tw * tw) * (tw * tw));
    const double refGrad = -total / kap;
    qSurf[c] = -total;
    Twall[c] = Ts + refGrad / (2.0 / k.dx);
    refT[b] = Ts;
    const double phi = phiGas[c] * hocPyr / qFuel;
    const double U = (-phi / magSf[b]) / rhob[b];
    Ux[b] = nx[b] * U; Uy[b] = ny[b] * U; Uz[b] = nz[b] * U;
}

extern "C" int ffm_pyro_couple_d(ffm_pyro *P, const int *map_d, const double *TgasCell_d, const double *kappaDelta_d, const double *qin_d,
                                 double emissivity, double absorptivity, const double *rho_b_d, const double *magSf_d, const double *nfx_d,
                                 const double *nfy_d, const double *nfz_d, double hocSolid, double qFuel, double *refT_d, double *Ux_d, double *Uy_d,
                                 double *Uz_d)
{
    if (!P || !TgasCell_d || !kappaDelta_d || !rho_b_d || !magSf_d || !nfx_d || !nfy_d || !nfz_d || !refT_d || !Ux_d || !Uy_d || !Uz_d || !(qFuel > 0)) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(P->ctx->device));
    const double hocPyr = (hocSolid * P->k.rhoW - 32.8e6 * P->k.rhoC) / (P->k.rhoW - P->k.rhoC);
    hipLaunchKernelGGL(k_pyro_couple, dim3((P->nCol + 255) / 256), dim3(256), 0, P->ctx->stream, P->nCol, P->k, map_d, (const double *)P->T,
                       (const double *)P->Yw, (const double *)P->alpha, (const double *)P->phiGas, P->Twall, P->qSurf, TgasCell_d, kappaDelta_d, qin_d, emissivity, absorptivity,
                       rho_b_d, magSf_d, nfx_d, nfy_d, nfz_d, hocPyr, qFuel, refT_d, Ux_d, Uy_d, Uz_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
// reactingOneDim21::solidRegionDiffNo (reactingOneDim21.C:697-714): DiNum = max over the internal faces of
// sqr(deltaCoeffs)*interpolate(kappa())/interpolate(Cp()*rho)*deltaT (linear interpolation: interpolationSchemes default linear)
__global__ void k_pyro_diff_no(int nCol, int nLay, PyroConst k, const double *__restrict__ rho_, const double *__restrict__ Yw_, const double *__restrict__ alpha_,
                               double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    double m = 0.0, kapP = 0.0, crP = 0.0;
    for (int i = 0; i < nLay; i++) {
        const size_t o = (size_t)i * nCol + c;
        const double Cp = Yw_[o] * k.CpW + (1.0 - Yw_[o]) * k.CpC, kap = Cp * alpha_[o], cr = Cp * rho_[o];
        if (i > 0) m = fmax(m, (1.0 / k.dx) * (1.0 / k.dx) * (0.5 * kapP + 0.5 * kap) / (0.5 * crP + 0.5 * cr));
        kapP = kap; crP = cr;
    }
    out[c] = m;
}
extern "C" int ffm_reduce_max(ffm_ctx *, const double *, long, double *);
extern "C" int ffm_pyro_diff_no(ffm_pyro *P, double deltaT, double *out)
{
    if (!P || !out) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(P->ctx->device));
    double *tmp = nullptr;
    FFM_TRY(ffm_malloc_uninit(P->ctx, sizeof(double) * P->nCol, (void **)&tmp));
    hipLaunchKernelGGL(k_pyro_diff_no, dim3((P->nCol + 255) / 256), dim3(256), 0, P->ctx->stream, P->nCol, P->nLay, P->k, (const double *)P->rho, (const double *)P->Yw,
                       (const double *)P->alpha, tmp);
    FFM_HIP(hipGetLastError());
    double m = 0.0;
    FFM_TRY(ffm_reduce_max(P->ctx, tmp, P->nCol, &m));
    FFM_TRY(ffm_free(P->ctx, tmp));
    *out = m * deltaT;
    return FFM_OK;
}

extern "C" const double *ffm_pyro_qSurf_d(const ffm_pyro *P) { return P ? P->qSurf : nullptr; }

extern "C" const double *ffm_pyro_surface_T_d(const ffm_pyro *P) { return P ? P->Tsurf : nullptr; }
extern "C" const double *ffm_pyro_phiGas_d(const ffm_pyro *P) { return P ? P->phiGas : nullptr; }
extern "C" int ffm_pyro_destroy(ffm_pyro *P)
{
    if (!P) return FFM_OK;
    hipStreamSynchronize(P->ctx->stream);
    hipFree(P->rho); hipFree(P->Yw); hipFree(P->T); hipFree(P->h); hipFree(P->alpha); hipFree(P->Tsurf); hipFree(P->phiGas); hipFree(P->qSurf); hipFree(P->Twall);
    delete P;
    return FFM_OK;
}
