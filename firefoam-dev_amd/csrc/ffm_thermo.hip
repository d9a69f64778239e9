// ffm_thermo.hip -- SURVEY 8(f) N2: the per-cell physics the reference's time step calls between its equations
//   thermo.correct()        hePsiThermo<reactingMixture<sutherland<janaf<perfectGas<specie>>>>, sensibleEnthalpy>::calculate()
//                           (solver/YEEqn.H:114; cases/steckler/constant/thermophysicalProperties:18-27)
//   combustion->correct()   the reference's eddyDissipationModel::correct
//                           (lib/thermophysicalModels/combustionModels/eddyDissipationModel/eddyDissipationModel.C:93-149)
//   turbulence nut / epsilon   LESModels::kEqn::correctNut, ::epsilon (cases/steckler/constant/turbulenceProperties:18-30)
// as one HBM-streaming kernel each (a thread owns a cell or a patch face; no neighbour access).  The thermo package itself lives
// in OpenFOAM-dev (not in the reference tree); oracle/thermo.py restates it with its sources and is pinned on the numbers the
// reference prints at start-up and, through oracle/steckler_case.py, on the golden log's first time step.
// Mixture of a cell (multiComponentMixture::cellMixture): progressive mass-fraction weighted sum in species order -- specie::
// operator+= (molecular weight), janafThermo::operator+= (Tlow / Thigh, both coefficient sets), sutherlandTransport::operator+=.
// T from he: thermo::T Newton iteration started at the old temperature, |dT| <= T0*1e-4, janaf::limit on every iterate.
#include "ffm_internal.hpp"
#include <algorithm>
#include <cmath>

constexpr int TH_MAXSP = 8;

struct ThermoTable {
    int n;
    double W[TH_MAXSP], Tlow[TH_MAXSP], Thigh[TH_MAXSP], Tcommon[TH_MAXSP], As[TH_MAXSP], Ts[TH_MAXSP];
    double high[TH_MAXSP][7], low[TH_MAXSP][7];          // mass based: the file's coefficients times RR/W (janafThermo constructor)
    double RR;
};
struct ThermoPtrs { const double *Y[TH_MAXSP]; };

struct ffm_thermo { ffm_ctx *ctx; ThermoTable t; };

namespace {
constexpr double TH_SMALL = 1.0e-15, TH_TSTD = 298.15;

struct Mix { double W, Tlow, Thigh, Tcommon, As, Ts, high[7], low[7]; };

__device__ __forceinline__ void cell_mixture(const ThermoTable &t, const ThermoPtrs &y, long i, Mix &m)
{
    double sumY = y.Y[0][i];
    m.W = t.W[0]; m.Tlow = t.Tlow[0]; m.Thigh = t.Thigh[0]; m.Tcommon = t.Tcommon[0]; m.As = t.As[0]; m.Ts = t.Ts[0];
#pragma unroll
    for (int c = 0; c < 7; c++) { m.high[c] = t.high[0][c]; m.low[c] = t.low[0][c]; }
    for (int k = 1; k < t.n; k++) {
        const double Yk = y.Y[k][i];
        const double Y1 = sumY, nw = sumY + Yk;
        if (fabs(nw) > TH_SMALL) {
            m.W = nw / (sumY / m.W + Yk / t.W[k]);                     // specie::operator+=
            const double y1 = Y1 / nw, y2 = Yk / nw;
            m.Tlow = fmax(m.Tlow, t.Tlow[k]); m.Thigh = fmin(m.Thigh, t.Thigh[k]);
#pragma unroll
            for (int c = 0; c < 7; c++) { m.high[c] = y1 * m.high[c] + y2 * t.high[k][c]; m.low[c] = y1 * m.low[c] + y2 * t.low[k][c]; }
            m.As = y1 * m.As + y2 * t.As[k]; m.Ts = y1 * m.Ts + y2 * t.Ts[k];
        }
        sumY = nw;
    }
}
__device__ __forceinline__ const double *coeffs(const Mix &m, double T) { return T < m.Tcommon ? m.low : m.high; }
__device__ __forceinline__ double th_Cp(const Mix &m, double T) { const double *a = coeffs(m, T); return ((((a[4] * T + a[3]) * T + a[2]) * T + a[1]) * T + a[0]); }
__device__ __forceinline__ double th_Ha(const double *a, double T) { return ((((a[4] / 5.0 * T + a[3] / 4.0) * T + a[2] / 3.0) * T + a[1] / 2.0) * T + a[0]) * T + a[5]; }
__device__ __forceinline__ double th_Hs(const Mix &m, double T) { return th_Ha(coeffs(m, T), T) - th_Ha(m.low, TH_TSTD); }
__device__ __forceinline__ double th_limit(const Mix &m, double T) { return fmin(fmax(T, m.Tlow), m.Thigh); }

// hePsiThermo::calculate for n cells (or the faces of a patch: the same arithmetic on the patch-face mixture)
__global__ void k_thermo_correct(long n, ThermoTable t, ThermoPtrs y, const double *__restrict__ he, const double *__restrict__ p,
                                 double *__restrict__ T, double *__restrict__ psi, double *__restrict__ mu, double *__restrict__ alpha, int *fail)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        Mix m; cell_mixture(t, y, i, m);
        const double h = he[i], T0 = T[i], Ttol = T0 * 1.0e-4;
        double Tnew = T0, Test;
        int it = 0;
        do {
            Test = Tnew;
            Tnew = th_limit(m, Test - (th_Hs(m, Test) - h) / th_Cp(m, Test));
            if (it++ > 100) { *fail = 1; break; }
        } while (fabs(Tnew - Test) > Ttol);
        const double R = t.RR / m.W;
        T[i] = Tnew;
        if (psi) psi[i] = 1.0 / (R * Tnew);
        const double muv = m.As * sqrt(Tnew) / (1.0 + m.Ts / Tnew);
        if (mu) mu[i] = muv;
        if (alpha) { const double Cp = th_Cp(m, Tnew), Cv = Cp - R; alpha[i] = muv * Cv * (1.32 + 1.77 * R / Cv) / Cp; }   // modified Eucken kappa / Cp
        (void)p;
    }
}
// psi, mu, alpha at a given temperature (no iteration): the patch faces whose temperature is fixed
__global__ void k_thermo_properties(long n, ThermoTable t, ThermoPtrs y, const double *__restrict__ T, double *__restrict__ psi,
                                    double *__restrict__ mu, double *__restrict__ alpha)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        Mix m; cell_mixture(t, y, i, m);
        const double Tc = T[i], R = t.RR / m.W;
        if (psi) psi[i] = 1.0 / (R * Tc);
        const double muv = m.As * sqrt(Tc) / (1.0 + m.Ts / Tc);
        if (mu) mu[i] = muv;
        if (alpha) { const double Cp = th_Cp(m, Tc), Cv = Cp - R; alpha[i] = muv * Cv * (1.32 + 1.77 * R / Cv) / Cp; }
    }
}
// he = Hs(p, T) of the mixture (patches whose temperature is fixed: hePsiThermo evaluates he there)
__global__ void k_thermo_he(long n, ThermoTable t, ThermoPtrs y, const double *__restrict__ T, double *__restrict__ he)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        Mix m; cell_mixture(t, y, i, m);
        he[i] = th_Hs(m, T[i]);
    }
}
// Cp of the mixture at a given temperature (heThermo::Cp(p, T, patchi): the kappa of compressible::thermalBaffle1D, the gradient
// term of mixedEnergy)
__global__ void k_thermo_cp(long n, ThermoTable t, ThermoPtrs y, const double *__restrict__ T, double *__restrict__ cp)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        Mix m; cell_mixture(t, y, i, m);
        cp[i] = th_Cp(m, T[i]);
    }
}
// eddyDissipationModel::correct: rtTurb = C_EDC*epsilon/max(k, SMALL), rtDiff = C_Diff*alpha/rho/delta^2, rt = max of the two;
// wFuel = rho*min(Y_fuel, Y_O2/s)/deltaT/C_Stiff*(1 - exp(-C_Stiff*deltaT*rt)); Qdot = qFuel*wFuel; epsilon = Ce*k*sqrt(k)/delta
__global__ void k_edc(long n, const double *__restrict__ rho, const double *__restrict__ k, const double *__restrict__ delta,
                      const double *__restrict__ alpha, const double *__restrict__ Yf, const double *__restrict__ Yo, double s, double dt, double Ce,
                      double C_EDC, double C_Diff, double C_Stiff, double qFuel, double *__restrict__ wFuel, double *__restrict__ Qdot)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double kk = k[i], d = delta[i];
        const double eps = Ce * kk * sqrt(kk) / d;
        const double rtTurb = C_EDC * eps / fmax(kk, TH_SMALL);
        const double rtDiff = C_Diff * alpha[i] / rho[i] / (d * d);
        const double rt = fmax(rtTurb, rtDiff);
        const double w = rho[i] * fmin(Yf[i], Yo[i] / s) / dt / C_Stiff * (1.0 - exp(-C_Stiff * dt * rt));
        wFuel[i] = w;
        if (Qdot) Qdot[i] = qFuel * w;
    }
}
// kEqn::correctNut: nut = Ck*sqrt(k)*delta; alphat = rho*nut/Prt
__global__ void k_nut(long n, double Ck, double Prt, const double *__restrict__ k, const double *__restrict__ delta, const double *__restrict__ rho,
                      double *__restrict__ nut, double *__restrict__ alphat)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = Ck * sqrt(k[i]) * delta[i];
        nut[i] = v;
        if (alphat) alphat[i] = rho[i] * v / Prt;
    }
}
inline int tgrid(long n) { return (int)std::max(1L, std::min((n + 255) / 256, (long)RED_BLOCKS)); }
}  // namespace

// species table: per specie molWeight, Tlow, Thigh, Tcommon, highCpCoeffs[7], lowCpCoeffs[7] (as in the thermo file, molar:
// multiplied by RR/W here), sutherland As, Ts.  RR: the universal gas constant the host library uses [J/(kmol K)].
extern "C" int ffm_thermo_create(ffm_ctx *ctx, int nSpecies, const double *W, const double *Tlow, const double *Thigh, const double *Tcommon,
                                 const double *highCpCoeffs, const double *lowCpCoeffs, const double *As, const double *Ts, double RR, ffm_thermo **out)
{
    if (!ctx || !out || nSpecies < 1 || nSpecies > TH_MAXSP || !W || !Tlow || !Thigh || !Tcommon || !highCpCoeffs || !lowCpCoeffs || !As || !Ts || !(RR > 0)) {
        ffm_set_error("ffm_thermo_create: bad argument (at most %d species)", TH_MAXSP); return FFM_ERR_ARG;
    }
    ffm_thermo *th = new ffm_thermo; th->ctx = ctx; th->t.n = nSpecies; th->t.RR = RR;
    for (int i = 0; i < nSpecies; i++) {
        if (!(W[i] > 0)) { delete th; return FFM_ERR_ARG; }
        th->t.W[i] = W[i]; th->t.Tlow[i] = Tlow[i]; th->t.Thigh[i] = Thigh[i]; th->t.Tcommon[i] = Tcommon[i]; th->t.As[i] = As[i]; th->t.Ts[i] = Ts[i];
        const double R = RR / W[i];
        for (int c = 0; c < 7; c++) { th->t.high[i][c] = highCpCoeffs[7 * i + c] * R; th->t.low[i][c] = lowCpCoeffs[7 * i + c] * R; }
    }
    *out = th;
    return FFM_OK;
}
extern "C" int ffm_thermo_destroy(ffm_thermo *th) { delete th; return FFM_OK; }

static int ptrs_of(const ffm_thermo *th, const double *const *Y, ThermoPtrs &y)
{
    for (int i = 0; i < TH_MAXSP; i++) y.Y[i] = nullptr;
    for (int i = 0; i < th->t.n; i++) { if (!Y[i]) return FFM_ERR_ARG; y.Y[i] = Y[i]; }
    return FFM_OK;
}

// hePsiThermo::calculate(): T (in: the old temperature, the Newton iteration's start; out: T(he)), psi, mu, alpha (each nullable)
// for n cells or patch faces; Y_d: nSpecies device pointers
extern "C" int ffm_thermo_correct_d(ffm_thermo *th, long n, const double *const *Y_d, const double *he_d, const double *p_d, double *T_d,
                                    double *psi_d, double *mu_d, double *alpha_d)
{
    if (!th || n < 0 || !Y_d || !he_d || !T_d) return FFM_ERR_ARG;
    ThermoPtrs y; FFM_TRY(ptrs_of(th, Y_d, y));
    FFM_HIP(hipSetDevice(th->ctx->device));
    int *fail = (int *)(th->ctx->scal_d + (NSCAL - 1));                     // last scalar slot as the failure flag
    FFM_HIP(hipMemsetAsync(fail, 0, sizeof(double), th->ctx->stream));
    if (n) hipLaunchKernelGGL(k_thermo_correct, dim3(tgrid(n)), dim3(256), 0, th->ctx->stream, n, th->t, y, he_d, p_d, T_d, psi_d, mu_d, alpha_d, fail);
    FFM_HIP(hipGetLastError());
    int h = 0;
    FFM_HIP(hipMemcpyAsync(&h, fail, sizeof(int), hipMemcpyDeviceToHost, th->ctx->stream));
    FFM_HIP(hipStreamSynchronize(th->ctx->stream));
    if (h) { ffm_set_error("thermo::T: maximum number of iterations exceeded"); return FFM_ERR_ARG; }
    return FFM_OK;
}
extern "C" int ffm_thermo_properties_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *psi_d, double *mu_d, double *alpha_d)
{
    if (!th || n < 0 || !Y_d || !T_d) return FFM_ERR_ARG;
    ThermoPtrs y; FFM_TRY(ptrs_of(th, Y_d, y));
    FFM_HIP(hipSetDevice(th->ctx->device));
    if (n) hipLaunchKernelGGL(k_thermo_properties, dim3(tgrid(n)), dim3(256), 0, th->ctx->stream, n, th->t, y, T_d, psi_d, mu_d, alpha_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_thermo_he_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *he_d)
{
    if (!th || n < 0 || !Y_d || !T_d || !he_d) return FFM_ERR_ARG;
    ThermoPtrs y; FFM_TRY(ptrs_of(th, Y_d, y));
    FFM_HIP(hipSetDevice(th->ctx->device));
    if (n) hipLaunchKernelGGL(k_thermo_he, dim3(tgrid(n)), dim3(256), 0, th->ctx->stream, n, th->t, y, T_d, he_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_thermo_Cp_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *Cp_d)
{
    if (!th || n < 0 || !Y_d || !T_d || !Cp_d) return FFM_ERR_ARG;
    ThermoPtrs y; FFM_TRY(ptrs_of(th, Y_d, y));
    FFM_HIP(hipSetDevice(th->ctx->device));
    if (n) hipLaunchKernelGGL(k_thermo_cp, dim3(tgrid(n)), dim3(256), 0, th->ctx->stream, n, th->t, y, T_d, Cp_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_edc_correct_d(ffm_ctx *ctx, long n, const double *rho_d, const double *k_d, const double *delta_d, const double *alpha_d,
                                 const double *Yfuel_d, const double *YO2_d, double s, double deltaT, double Ce, double C_EDC, double C_Diff,
                                 double C_Stiff, double qFuel, double *wFuel_d, double *Qdot_d)
{
    if (!ctx || n < 0 || !rho_d || !k_d || !delta_d || !alpha_d || !Yfuel_d || !YO2_d || !wFuel_d || !(s > 0) || !(deltaT > 0) || !(C_Stiff > 0)) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k_edc, dim3(tgrid(n)), dim3(256), 0, ctx->stream, n, rho_d, k_d, delta_d, alpha_d, Yfuel_d, YO2_d, s, deltaT, Ce, C_EDC, C_Diff,
                              C_Stiff, qFuel, wFuel_d, Qdot_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_les_keqn_nut_d(ffm_ctx *ctx, long n, double Ck, double Prt, const double *k_d, const double *delta_d, const double *rho_d,
                                  double *nut_d, double *alphat_d)
{
    if (!ctx || n < 0 || !k_d || !delta_d || !nut_d || (alphat_d && (!rho_d || !(Prt > 0)))) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k_nut, dim3(tgrid(n)), dim3(256), 0, ctx->stream, n, Ck, Prt, k_d, delta_d, rho_d, nut_d, alphat_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
