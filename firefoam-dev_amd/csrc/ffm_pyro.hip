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

struct ffm_pyro {
    ffm_ctx *ctx = nullptr;
    int nCol = 0, nLay = 0;
    PyroConst k;
    double *rho = nullptr, *Yw = nullptr, *T = nullptr, *h = nullptr;      // [nLay][nCol]
    double *Tsurf = nullptr, *phiGas = nullptr;                               // [nCol]
    double *qSurf = nullptr, *Twall = nullptr;                                // [nCol] coupled heat flux / wall temperature (ffm_pyro_couple_d)
};

template <int NL>
__global__ __launch_bounds__(256) void k_pyro_step(int nCol, PyroConst k, double dt, const double *__restrict__ qSurf, int backFixed, double Tback,
                                                   double *__restrict__ rho_, double *__restrict__ Yw_, double *__restrict__ T_, double *__restrict__ h_,
                                                   double *__restrict__ Tsurf, double *__restrict__ phiGas)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    const double rdt = 1.0 / dt, TSTD = 298.15, V = k.V, A = k.area, dx = k.dx;
    double rho0[NL], Yw0[NL], h0[NL], T0[NL], kap[NL], alp[NL], RRg[NL], Qd[NL], rho[NL], Yw[NL];
    const double sr = k.rhoC / k.rhoW;
    double gas = 0.0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const size_t o = (size_t)i * nCol + c;
        rho0[i] = rho_[o]; Yw0[i] = Yw_[o]; h0[i] = h_[o]; T0[i] = T_[o];
        const double Cp = Yw0[i] * k.CpW + (1.0 - Yw0[i]) * k.CpC;
        kap[i] = Yw0[i] * k.kW + (1.0 - Yw0[i]) * k.kC;
        alp[i] = kap[i] / Cp;
        // solidChemistry->calculate(): irreversibleArrheniusSolidReaction wood^n = char + gas
        const double kf = T0[i] < k.Tcrit ? 0.0 : k.A * exp(-k.Ta / T0[i]);
        const double omega = kf * pow(rho0[i] * Yw0[i] / k.c0, k.n) * k.c0;          // pyrolysisChemistryModel::omega: kf (m/m0)^n m0, per volume
        const double RRw = -omega, RRc = sr * omega;
        RRg[i] = (1.0 - sr) * omega;
        Qd[i] = -(k.HfW * RRw + k.HfC * RRc);
        rho[i] = (rdt * rho0[i] * V - V * RRg[i]) / (rdt * V);                               // solveContinuity
        Yw[i] = fmax((rdt * rho0[i] * Yw0[i] * V + V * RRw) / (rdt * rho[i] * V), 0.0);       // solveSpeciesMass
        gas += RRg[i] * V;
    }
    // solveEnergy: fvm::ddt(rho,h) - fvm::laplacian(alpha,h) + fvc::laplacian(alpha,h) - fvc::laplacian(kappa,T) == Qdot - fvm::Sp(RRg,h)
    double dg[NL], lo[NL], up[NL], sc[NL], lapA[NL], lapK[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { dg[i] = rdt * rho[i] * V + V * RRg[i]; lo[i] = 0.0; up[i] = 0.0; lapA[i] = 0.0; lapK[i] = 0.0; }
#pragma unroll
    for (int i = 0; i + 1 < NL; i++) {
        const double ca = 0.5 * (alp[i] + alp[i + 1]) * A / dx, ck = 0.5 * (kap[i] + kap[i + 1]) * A / dx;
        up[i] = -ca; lo[i + 1] = -ca;
        dg[i] += ca; dg[i + 1] += ca;
        const double fa = ca * (h0[i + 1] - h0[i]), fk = ck * (T0[i + 1] - T0[i]);
        lapA[i] += fa; lapA[i + 1] -= fa;
        lapK[i] += fk; lapK[i + 1] -= fk;
    }
    lapK[0] += qSurf[c] * A;
    if (backFixed) lapK[NL - 1] += kap[NL - 1] * A * (2.0 / dx) * (Tback - T0[NL - 1]);
#pragma unroll
    for (int i = 0; i < NL; i++) { sc[i] = rdt * rho0[i] * h0[i] * V + V * Qd[i]; sc[i] -= (lapA[i] - lapK[i]); }
    // Thomas algorithm
    double cp[NL], dp[NL], x[NL];
    cp[0] = up[0] / dg[0]; dp[0] = sc[0] / dg[0];
#pragma unroll
    for (int i = 1; i < NL; i++) {
        const double den = dg[i] - lo[i] * cp[i - 1];
        cp[i] = up[i] / den;
        dp[i] = (sc[i] - lo[i] * dp[i - 1]) / den;
    }
    x[NL - 1] = dp[NL - 1];
#pragma unroll
    for (int i = NL - 2; i >= 0; i--) x[i] = dp[i] - cp[i] * x[i + 1];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const size_t o = (size_t)i * nCol + c;
        const double Cp = Yw[i] * k.CpW + (1.0 - Yw[i]) * k.CpC;
        rho_[o] = rho[i]; Yw_[o] = Yw[i]; h_[o] = x[i]; T_[o] = TSTD + x[i] / Cp;             // solidThermo.correct()
        if (i == 0) Tsurf[c] = TSTD + x[0] / Cp;
    }
    phiGas[c] = gas;
}

extern "C" int ffm_pyro_create(ffm_ctx *ctx, int nCol, int nLay, double thickness, double faceArea, double T0, double Yw0, ffm_pyro **out)
{
    if (!ctx || !out || nCol < 1 || nLay < 2 || nLay > PYRO_MAX_LAYERS || thickness <= 0 || faceArea <= 0 || Yw0 <= 0 || Yw0 > 1) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(ctx->device));
    ffm_pyro *P = new ffm_pyro();
    P->ctx = ctx; P->nCol = nCol; P->nLay = nLay;
    // cases/pyrolysis1D/constant/panelRegion/{thermo.solid,reactions}: the defaults; ffm_pyro_set_solids / _set_reaction replace them
    P->k = PyroConst{114.7, 696.0, 0.135, -1.41e6, 11.5, 611.0, 0.4, 0.0, 7.83e10, 15274.57, 400.0, 4.86, 0.0, thickness / nLay, faceArea, faceArea * (thickness / nLay)};
    P->k.c0 = (Yw0 < 1 ? 1.0 / (Yw0 / P->k.rhoW + (1 - Yw0) / P->k.rhoC) : P->k.rhoW) * std::max(Yw0, 0.001);
    const size_t n = (size_t)nCol * nLay;
    double **f[4] = {&P->rho, &P->Yw, &P->T, &P->h};
    for (auto p : f) FFM_HIP(hipMalloc((void **)p, sizeof(double) * n));
    FFM_HIP(hipMalloc((void **)&P->Tsurf, sizeof(double) * nCol)); FFM_HIP(hipMalloc((void **)&P->phiGas, sizeof(double) * nCol));
    const double rho0 = Yw0 < 1 ? 1.0 / (Yw0 / P->k.rhoW + (1 - Yw0) / P->k.rhoC) : P->k.rhoW;
    const double Cp = Yw0 * P->k.CpW + (1.0 - Yw0) * P->k.CpC;
    std::vector<double> v(n);
    auto fill = [&](double *d, double val) { std::fill(v.begin(), v.end(), val); return hipMemcpy(d, v.data(), sizeof(double) * n, hipMemcpyHostToDevice); };
    FFM_HIP(fill(P->rho, rho0)); FFM_HIP(fill(P->Yw, Yw0)); FFM_HIP(fill(P->T, T0)); FFM_HIP(fill(P->h, Cp * (T0 - 298.15)));
    FFM_HIP(hipMemset(P->phiGas, 0, sizeof(double) * nCol));
    std::vector<double> ts(nCol, T0);
    FFM_HIP(hipMemcpy(P->Tsurf, ts.data(), sizeof(double) * nCol, hipMemcpyHostToDevice));
    FFM_HIP(hipMalloc((void **)&P->qSurf, sizeof(double) * nCol)); FFM_HIP(hipMalloc((void **)&P->Twall, sizeof(double) * nCol));
    FFM_HIP(hipMemset(P->qSurf, 0, sizeof(double) * nCol));
    FFM_HIP(hipMemcpy(P->Twall, ts.data(), sizeof(double) * nCol, hipMemcpyHostToDevice));
    *out = P;
    return FFM_OK;
}

extern "C" int ffm_pyro_set_solids(ffm_pyro *P, const double *virgin, const double *charred)
{
    if (!P || !virgin || !charred) return FFM_ERR_ARG;
    P->k.rhoW = virgin[0]; P->k.CpW = virgin[1]; P->k.kW = virgin[2]; P->k.HfW = virgin[3];
    P->k.rhoC = charred[0]; P->k.CpC = charred[1]; P->k.kC = charred[2]; P->k.HfC = charred[3];
    return FFM_OK;
}
extern "C" int ffm_pyro_set_reaction(ffm_pyro *P, double A, double Ta, double Tcrit, double n)
{ if (!P) return FFM_ERR_ARG; P->k.A = A; P->k.Ta = Ta; P->k.Tcrit = Tcrit; P->k.n = n; return FFM_OK; }

extern "C" int ffm_pyro_step(ffm_pyro *P, double dt, const double *qSurf_d, int backFixed, double Tback)
{
    if (!P || !qSurf_d || dt <= 0) return FFM_ERR_ARG;
    const dim3 grid((P->nCol + 255) / 256), block(256);
    hipStream_t s = P->ctx->stream;
#define PYRO(NL) hipLaunchKernelGGL(k_pyro_step<NL>, grid, block, 0, s, P->nCol, P->k, dt, qSurf_d, backFixed, Tback, P->rho, P->Yw, P->T, P->h, P->Tsurf, P->phiGas)
    switch (P->nLay) {
    case 2: PYRO(2); break; case 3: PYRO(3); break; case 4: PYRO(4); break; case 5: PYRO(5); break; case 6: PYRO(6); break;
    case 7: PYRO(7); break; case 8: PYRO(8); break; case 9: PYRO(9); break; case 10: PYRO(10); break; case 11: PYRO(11); break;
    case 12: PYRO(12); break; case 13: PYRO(13); break; case 14: PYRO(14); break; case 15: PYRO(15); break; default: PYRO(16); break;
    }
#undef PYRO
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_pyro_get(ffm_pyro *P, const char *name, double *out)
{
    if (!P || !name || !out) return FFM_ERR_ARG;
    const std::string n(name);
    const double *src = n == "rho" ? P->rho : n == "Yw" ? P->Yw : n == "T" ? P->T : n == "h" ? P->h : nullptr;
    FFM_HIP(hipStreamSynchronize(P->ctx->stream));
    if (src) {      // [nLay][nCol] on the device -> [nCol][nLay] for the host
        std::vector<double> v((size_t)P->nCol * P->nLay);
        FFM_HIP(hipMemcpy(v.data(), src, sizeof(double) * v.size(), hipMemcpyDeviceToHost));
        for (int c = 0; c < P->nCol; c++) for (int i = 0; i < P->nLay; i++) out[(size_t)c * P->nLay + i] = v[(size_t)i * P->nCol + c];
        return FFM_OK;
    }
    const double *col = n == "Tsurf" ? P->Tsurf : n == "phiGas" ? P->phiGas : n == "qSurf" ? P->qSurf : n == "Twall" ? P->Twall : nullptr;
    if (!col) { ffm_set_error("ffm_pyro_get: unknown field %s", name); return FFM_ERR_ARG; }
    FFM_HIP(hipMemcpy(out, col, sizeof(double) * P->nCol, hipMemcpyDeviceToHost));
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
                              const double *__restrict__ phiGas, double *__restrict__ Twall, double *__restrict__ qSurf,
                              const double *__restrict__ Tg, const double *__restrict__ kDelta, const double *qin, double emis, double absorp,
                              const double *__restrict__ rhob, const double *__restrict__ magSf, const double *__restrict__ nx,
                              const double *__restrict__ ny, const double *__restrict__ nz, double hocPyr, double qFuel,
                              double *__restrict__ refT, double *__restrict__ Ux, double *__restrict__ Uy, double *__restrict__ Uz)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCol) return;
    const int b = map ? map[c] : c;
    const double Ts = Ts_[c], Yw = Yw_[c];                      // layer 0
    const double kap = Yw * k.kW + (1.0 - Yw) * k.kC;
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
                       (const double *)P->Yw, (const double *)P->phiGas, P->Twall, P->qSurf, TgasCell_d, kappaDelta_d, qin_d, emissivity, absorptivity,
                       rho_b_d, magSf_d, nfx_d, nfy_d, nfz_d, hocPyr, qFuel, refT_d, Ux_d, Uy_d, Uz_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" const double *ffm_pyro_qSurf_d(const ffm_pyro *P) { return P ? P->qSurf : nullptr; }

extern "C" const double *ffm_pyro_surface_T_d(const ffm_pyro *P) { return P ? P->Tsurf : nullptr; }
extern "C" const double *ffm_pyro_phiGas_d(const ffm_pyro *P) { return P ? P->phiGas : nullptr; }
extern "C" int ffm_pyro_destroy(ffm_pyro *P)
{
    if (!P) return FFM_OK;
    hipStreamSynchronize(P->ctx->stream);
    hipFree(P->rho); hipFree(P->Yw); hipFree(P->T); hipFree(P->h); hipFree(P->Tsurf); hipFree(P->phiGas); hipFree(P->qSurf); hipFree(P->Twall);
    delete P;
    return FFM_OK;
}
