// ffm_solve.hip -- level-scheduled DIC / DILU / Gauss-Seidel sweeps and the
// PCG / PBiCGStab / PBiCG / smoothSolver / diagonalSolver drivers.
//
// Replaces (OpenFOAM-dev @940e28f, not vendored in the reference):
//   .../lduMatrix/preconditioners/{DIC,DILU}Preconditioner/*.C
//   .../lduMatrix/smoothers/{GaussSeidel,symGaussSeidel}/*.C
//   .../lduMatrix/solvers/{PCG,PBiCGStab,PBiCG,smoothSolver,diagonalSolver}/*.C
// selected by the reference in cases/steckler/system/fvSolution:21-61 and
// cases/wallFireSpread2D/system/fvSolution:115-152; entered from
// solver/pEqn.H:39, solver/UEqn.H:19, solver/YEEqn.H:60,111, solver/rhoEqn.H:43.
//
// Exactness: the serial face-order sweeps of the reference carry a dependency
// DAG (owner -> neighbour).  Cells are stored level-major, one kernel launch
// per dependency level, all launches of one sweep pair captured once in a
// hipGraph.  Inside a level every row is accumulated in the reference's face
// order, and the library is compiled with -ffp-contract=off, so the sweeps are
// bitwise equal to the serial loops; only the dot products (two-stage tree
// sums instead of one serial sum) differ, in the last bits.
//
// Scalars (alpha, beta, residual norms) stay on the device; the host reads the
// residual back once per iteration (twice for PBiCGStab) to take the
// convergence decision, as SolverPerformance::checkConvergence does.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

static inline int sgrid(long n) { long g = (n + 255) / 256; return (int)std::max(1L, std::min(g, (long)RED_BLOCKS)); }

// --------------------------------------------------------------- scalar ops ---
enum {
    OP_XREF = 1, OP_NORMF, OP_RES_INIT, OP_RES, OP_PCG_BETA, OP_PCG_ALPHA,
    OP_BS_RHO, OP_BS_ALPHA, OP_BS_OMEGA, OP_BICG_BETA, OP_BICG_ALPHA, OP_RESET
};

#define VSMALL_ 1e-300

__global__ void k_scalar_op(double *__restrict__ s, int op, double arg, int iter)
{
    if (threadIdx.x || blockIdx.x) return;
    switch (op) {
    case OP_RESET:
        s[S_SING] = 0.0; s[S_WARA] = 1e+20; s[S_WARA_OLD] = 1e+20; s[S_RA0RA] = 0.0; s[S_ALPHA] = 0.0; s[S_OMEGA] = 0.0;
        s[S_BETA] = 0.0;
        break;
    case OP_XREF: s[S_XREF] = s[S_TMP0] / arg; break;                    // gAverage(psi)
    case OP_NORMF: s[S_NORMF] = s[S_TMP0] + 1e-20; break;                // + solverPerformance::small_
    case OP_RES_INIT: s[S_RES] = s[S_TMP0] / s[S_NORMF]; s[S_RES0] = s[S_RES]; break;
    case OP_RES: if (s[S_SING] == 0.0) s[S_RES] = s[S_TMP0] / s[S_NORMF]; break;
    case OP_PCG_BETA: {                                                  // wArAold = wArA; wArA = (wA,rA)
        const double old = s[S_WARA], nw = s[S_TMP0];
        s[S_WARA_OLD] = old; s[S_WARA] = nw; s[S_BETA] = nw / old;
        break; }
    case OP_PCG_ALPHA: case OP_BICG_ALPHA: {                             // wApA; checkSingularity(|wApA|/normFactor)
        const double wApA = s[S_TMP0];
        s[S_WAPA] = wApA;
        if (fabs(wApA) / s[S_NORMF] > VSMALL_) { s[S_ALPHA] = s[S_WARA] / wApA; }
        else { s[S_SING] = 1.0; s[S_ALPHA] = 0.0; }
        break; }
    case OP_BS_RHO: {                                                    // rA0rAold = rA0rA; rA0rA = (rA0,rA)
        const double old = s[S_RA0RA], nw = s[S_TMP0];
        s[S_RA0RA_OLD] = old; s[S_RA0RA] = nw;
        if (!(fabs(nw) > VSMALL_)) { s[S_SING] = 1.0; break; }
        if (iter > 0) {
            if (!(fabs(s[S_OMEGA]) > VSMALL_)) { s[S_SING] = 1.0; break; }
            s[S_BETA] = (nw / old) * (s[S_ALPHA] / s[S_OMEGA]);
        }
        break; }
    case OP_BS_ALPHA: if (s[S_SING] == 0.0) { s[S_RA0AYA] = s[S_TMP0]; s[S_ALPHA] = s[S_RA0RA] / s[S_TMP0]; } break;
    case OP_BS_OMEGA: if (s[S_SING] == 0.0) { s[S_TATA] = s[S_TMP0]; s[S_TASA] = s[S_TMP1]; s[S_OMEGA] = s[S_TMP1] / s[S_TMP0]; } break;
    case OP_BICG_BETA: {
        const double old = s[S_WARA], nw = s[S_TMP0];
        s[S_WARA_OLD] = old; s[S_WARA] = nw; s[S_BETA] = nw / old;
        break; }
    }
}

static int scalar_op(ffm_ctx *c, int op, double arg = 0.0, int iter = 0)
{
    hipLaunchKernelGGL(k_scalar_op, dim3(1), dim3(64), 0, c->stream, c->scal_d, op, arg, iter);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
// local sum sits in S_TMP0(..S_TMP0+n-1): all-reduce over ranks, then derive
static int finish_dot(ffm_ctx *c, int op, int nSlots = 1, double arg = 0.0, int iter = 0)
{
    if (c->nRanks > 1 || c->comm) FFM_TRY(ffm_allreduce_slots(c, S_TMP0, nSlots));
    return scalar_op(c, op, arg, iter);
}

// ----------------------------------------------------------- vector kernels ---
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void k_sub(long n, double *__restrict__ r, const double *__restrict__ b, const double *__restrict__ w)
{ GRID_STRIDE(i, n) r[i] = b[i] - w[i]; }

__global__ void k_sub2(long n, double *__restrict__ r, double *__restrict__ rT, const double *__restrict__ b,
                       const double *__restrict__ w, const double *__restrict__ wT)
{ GRID_STRIDE(i, n) { r[i] = b[i] - w[i]; rT[i] = b[i] - wT[i]; } }

__global__ void k_copy(long n, double *__restrict__ d, const double *__restrict__ s)
{ GRID_STRIDE(i, n) d[i] = s[i]; }

// normFactor partial: |Apsi - xRef*sumA| + |source - xRef*sumA|
__global__ __launch_bounds__(256) void k_normf(long n, const double *__restrict__ Ax, const double *__restrict__ b,
                                               const double *__restrict__ sumA, const double *__restrict__ scal,
                                               double *__restrict__ partials)
{
    __shared__ double sm[4];
    const double xRef = scal[S_XREF];
    double acc = 0.0;
    GRID_STRIDE(i, n) { const double t = sumA[i] * xRef; acc += fabs(Ax[i] - t) + fabs(b[i] - t); }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// pA = first ? wA : wA + beta*pA
__global__ void k_p_update(long n, double *__restrict__ p, const double *__restrict__ w, const double *__restrict__ scal, int first)
{
    if (scal[S_SING] != 0.0) return;
    const double beta = scal[S_BETA];
    if (first) { GRID_STRIDE(i, n) p[i] = w[i]; }
    else { GRID_STRIDE(i, n) p[i] = w[i] + beta * p[i]; }
}

// fused PCG: psi += alpha*pA (the update left over from the iteration before), then pA = wA + beta*pA
__global__ void k_p_psi(long n, double *__restrict__ psi, double *__restrict__ p, const double *__restrict__ w, const double *__restrict__ scal)
{
    if (scal[S_SING] != 0.0) return;
    const double alpha = scal[S_ALPHA], beta = scal[S_BETA];
    GRID_STRIDE(i, n) { const double pi = p[i]; psi[i] += alpha * pi; p[i] = w[i] + beta * pi; }
}

// psi += alpha*pA; rA -= alpha*wA; partial sum |rA|      (PCG)
__global__ __launch_bounds__(256) void k_pcg_xr(long n, double *__restrict__ psi, double *__restrict__ r,
                                                const double *__restrict__ p, const double *__restrict__ w,
                                                const double *__restrict__ scal, double *__restrict__ partials)
{
    __shared__ double sm[4];
    const double alpha = scal[S_ALPHA];
    const bool sing = scal[S_SING] != 0.0;
    double acc = 0.0;
    GRID_STRIDE(i, n) {
        double ri = r[i];
        if (!sing) { psi[i] += alpha * p[i]; ri -= alpha * w[i]; r[i] = ri; }
        acc += fabs(ri);
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// PBiCG extra: rT -= alpha*wT
__global__ void k_axmy_alpha(long n, double *__restrict__ r, const double *__restrict__ w, const double *__restrict__ scal)
{
    if (scal[S_SING] != 0.0) return;
    const double alpha = scal[S_ALPHA];
    GRID_STRIDE(i, n) r[i] -= alpha * w[i];
}

// PBiCGStab: pA = first ? rA : rA + beta*(pA - omega*AyA)
__global__ void k_bs_p(long n, double *__restrict__ p, const double *__restrict__ r, const double *__restrict__ AyA,
                       const double *__restrict__ scal, int first)
{
    if (scal[S_SING] != 0.0) return;
    const double beta = scal[S_BETA], omega = scal[S_OMEGA];
    if (first) { GRID_STRIDE(i, n) p[i] = r[i]; }
    else { GRID_STRIDE(i, n) p[i] = r[i] + beta * (p[i] - omega * AyA[i]); }
}

// sA = rA - alpha*AyA; partial |sA|
__global__ __launch_bounds__(256) void k_bs_s(long n, double *__restrict__ sA, const double *__restrict__ r,
                                              const double *__restrict__ AyA, const double *__restrict__ scal,
                                              double *__restrict__ partials)
{
    __shared__ double sm[4];
    const double alpha = scal[S_ALPHA];
    const bool sing = scal[S_SING] != 0.0;
    double acc = 0.0;
    if (!sing) GRID_STRIDE(i, n) { const double v = r[i] - alpha * AyA[i]; sA[i] = v; acc += fabs(v); }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// psi += alpha*yA
__global__ void k_axpy_alpha(long n, double *__restrict__ psi, const double *__restrict__ y, const double *__restrict__ scal)
{
    const double alpha = scal[S_ALPHA];
    GRID_STRIDE(i, n) psi[i] += alpha * y[i];
}

// two dots in one pass: partials[0..] = tA.tA, partials[RED_BLOCKS..] = tA.sA
__global__ __launch_bounds__(256) void k_dot2(long n, const double *__restrict__ t, const double *__restrict__ s,
                                              double *__restrict__ partials)
{
    __shared__ double sm[4];
    double a = 0.0, b = 0.0;
    GRID_STRIDE(i, n) { const double ti = t[i]; a += ti * ti; b += ti * s[i]; }
    const double ra = block_sum(a, sm);
    const double rb = block_sum(b, sm);
    if (threadIdx.x == 0) { partials[blockIdx.x] = ra; partials[RED_BLOCKS + blockIdx.x] = rb; }
}

__global__ __launch_bounds__(1024) void k_sum_partials2(int n, const double *__restrict__ partials, double *__restrict__ scal,
                                                        int slot, int nSums)
{
    __shared__ double sm[16];
    for (int k = 0; k < nSums; k++) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[k * RED_BLOCKS + i];
        const double r = block_sum(acc, sm);
        if (threadIdx.x == 0) scal[slot + k] = r;
    }
}

// psi += alpha*yA + omega*zA; rA = sA - omega*tA; partial |rA|
__global__ __launch_bounds__(256) void k_bs_xr(long n, double *__restrict__ psi, double *__restrict__ r,
                                               const double *__restrict__ yA, const double *__restrict__ zA,
                                               const double *__restrict__ sA, const double *__restrict__ tA,
                                               const double *__restrict__ scal, double *__restrict__ partials)
{
    __shared__ double sm[4];
    const double alpha = scal[S_ALPHA], omega = scal[S_OMEGA];
    const bool sing = scal[S_SING] != 0.0;
    double acc = 0.0;
    if (!sing) GRID_STRIDE(i, n) {
        psi[i] += alpha * yA[i] + omega * zA[i];
        const double v = sA[i] - omega * tA[i];
        r[i] = v; acc += fabs(v);
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// psi = source/diag (diagonalSolver)
__global__ void k_diag_solve(long n, double *__restrict__ psi, const double *__restrict__ b, const double *__restrict__ d)
{ GRID_STRIDE(i, n) psi[i] = b[i] / d[i]; }

__global__ void k_mul(long n, double *__restrict__ w, const double *__restrict__ a, const double *__restrict__ b)
{ GRID_STRIDE(i, n) w[i] = a[i] * b[i]; }

__global__ void k_recip(long n, double *__restrict__ d, const double *__restrict__ s)
{ GRID_STRIDE(i, n) d[i] = 1.0 / s[i]; }

static int partial_sum_to(ffm_ctx *c, int nb, int slot, int nSums = 1)
{
    hipLaunchKernelGGL(k_sum_partials2, dim3(1), dim3(1024), 0, c->stream, nb, c->partials_d, c->scal_d, slot, nSums);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ------------------------------------------------------------ level kernels ---
// A launch covers the cells [c0,c1) of one dependency level (contiguous in the level-major
// numbering).  c0 need not be slice-aligned: thread t handles cell (c0 & ~63) + t so that a
// wave still maps onto one slice and every index load stays unit-stride.

// calcReciprocalD, forward part:  rD[c] = diag[c] - sum_k upper*lower/rD[l]   (not yet inverted)
template <int W>
__global__ void k_rD_level(int c0, int c1, LduView v, const double *__restrict__ upper,
                           const double *__restrict__ lower, const double *__restrict__ diag, double *__restrict__ rD)
{
    const int c = (c0 & ~63) + blockIdx.x * blockDim.x + threadIdx.x;
    if (c < c0 || c >= c1) return;
    RowEnt<W> L; load_lower<W>(v, c, L);
    double au[W], al[W], rn[W];
#pragma unroll
    for (int s = 0; s < W; s++) { au[s] = upper[L.f[s]]; al[s] = lower[L.f[s]]; rn[s] = L.on[s] ? rD[L.nb[s]] : 1.0; }
    double d = diag[c];
#pragma unroll
    for (int s = 0; s < W; s++) if (L.on[s]) d -= au[s] * al[s] / rn[s];
    rD[c] = d;
}

// forward sweep of one level:  w[c] = rD[c]*r[c] - sum_k rD[c]*coef[f]*w[l]
template <int W>
__global__ void k_fwd_level(int c0, int c1, LduView v, const double *__restrict__ coef,
                            const double *__restrict__ rD, const double *__restrict__ r, double *__restrict__ w)
{
    const int c = (c0 & ~63) + blockIdx.x * blockDim.x + threadIdx.x;
    if (c < c0 || c >= c1) return;
    RowEnt<W> L; load_lower<W>(v, c, L);
    const double rd = rD[c], rc = r[c];
    double a[W], wn[W];
#pragma unroll
    for (int s = 0; s < W; s++) { a[s] = coef[L.f[s]]; wn[s] = L.on[s] ? w[L.nb[s]] : 0.0; }
    double wc = rd * rc;
#pragma unroll
    for (int s = 0; s < W; s++) if (L.on[s]) wc -= rd * a[s] * wn[s];
    w[c] = wc;
}

// backward sweep of one level: w[c] -= rD[c]*coef[f]*w[u], faces of c in descending order
template <int W>
__global__ void k_bwd_level(int p0, int p1, const int *__restrict__ order, LduView v, const double *__restrict__ coef,
                            const double *__restrict__ rD, double *__restrict__ w)
{
    int c;
    if (order) { const int p = p0 + blockIdx.x * blockDim.x + threadIdx.x; if (p >= p1) return; c = order[p]; }
    else { c = (p0 & ~63) + blockIdx.x * blockDim.x + threadIdx.x; if (c < p0 || c >= p1) return; }
    RowEnt<W> U; load_upper<W, true>(v, c, U);          // block-Jacobi: ghost neighbours are ignored
    const double rd = rD[c];
    double wc = w[c];
    double a[W], wn[W];
#pragma unroll
    for (int s = 0; s < W; s++) { a[s] = coef[U.f[s]]; wn[s] = U.on[s] ? w[U.nb[s]] : 0.0; }
#pragma unroll
    for (int s = W - 1; s >= 0; s--) if (U.on[s]) wc -= rd * a[s] * wn[s];
    w[c] = wc;
}

// Gauss-Seidel forward level: psi_c = (bP_c - sum_lower lower*psi_l - sum_upper upper*psi_u)/diag_c;
// the value after the lower sum is kept in bSave for the reverse sweep of symGaussSeidel
template <int W>
__global__ void k_gs_fwd_level(int c0, int c1, LduView v, const double *__restrict__ upper,
                               const double *__restrict__ lower, const double *__restrict__ diag,
                               const double *__restrict__ bP, double *__restrict__ bSave, double *__restrict__ psi)
{
    const int c = (c0 & ~63) + blockIdx.x * blockDim.x + threadIdx.x;
    if (c < c0 || c >= c1) return;
    RowEnt<W> L, U; load_lower<W>(v, c, L); load_upper<W>(v, c, U);
    double al[W], au[W], pl[W], pu[W];
#pragma unroll
    for (int s = 0; s < W; s++) {
        al[s] = lower[L.f[s]]; au[s] = upper[U.f[s]];
        pl[s] = L.on[s] ? psi[L.nb[s]] : 0.0; pu[s] = U.on[s] ? psi[U.nb[s]] : 0.0;
    }
    double val = bP[c];
#pragma unroll
    for (int s = 0; s < W; s++) if (L.on[s]) val -= al[s] * pl[s];
    bSave[c] = val;
#pragma unroll
    for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
    psi[c] = val / diag[c];
}

template <int W>
__global__ void k_gs_bwd_level(int p0, int p1, const int *__restrict__ order, LduView v, const double *__restrict__ upper,
                               const double *__restrict__ diag, const double *__restrict__ bSave, double *__restrict__ psi)
{
    int c;
    if (order) { const int p = p0 + blockIdx.x * blockDim.x + threadIdx.x; if (p >= p1) return; c = order[p]; }
    else { c = (p0 & ~63) + blockIdx.x * blockDim.x + threadIdx.x; if (c < p0 || c >= p1) return; }
    RowEnt<W> U; load_upper<W>(v, c, U);
    double au[W], pu[W];
#pragma unroll
    for (int s = 0; s < W; s++) { au[s] = upper[U.f[s]]; pu[s] = U.on[s] ? psi[U.nb[s]] : 0.0; }
    double val = bSave[c];
#pragma unroll
    for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
    psi[c] = val / diag[c];
}

// ------------------------------------------------- single-workgroup sweeps ---
// Small matrices (the coarse levels of a GAMG hierarchy, small unstructured meshes): one launch per dependency level costs ~3.6 us
// each and these have hundreds of levels with a handful of cells.  Here ONE workgroup of 1024 threads walks all levels of a sweep
// (and of the sweep pair of a preconditioner application / symGaussSeidel), one workgroup barrier per level.  Values written in an
// earlier level are read with agent-scope loads (they bypass the CU's L1, where a line fetched for a neighbouring cell may hold the
// old value); the barrier's release waits for the stores.  Per-cell arithmetic is that of the level kernels above, bit for bit.
constexpr int SMALL_T = 1024;
constexpr long FFM_SMALL_SWEEP_DEFAULT = 131072L;
__device__ __forceinline__ double s_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
enum { SM_RD = 0, SM_PRECOND = 1, SM_GS = 2, SM_SYMGS = 3 };
struct SmallArgs {
    LduView v;
    int nLevels, nBwd, N;
    const int *fwdStart, *bwdRange, *order;
    const double *upper, *lower, *diag, *cf, *cb, *r, *bP;
    double *rD, *w, *bSave, *psi;
};
template <int MODE, int W>
__global__ __launch_bounds__(SMALL_T) void k_small_sweep(SmallArgs a)
{
    const int tid = threadIdx.x;
    // ---- forward levels
    for (int Lv = 0; Lv < a.nLevels; Lv++) {
        const int c0 = a.fwdStart[Lv], c1 = a.fwdStart[Lv + 1];
        for (int c = c0 + tid; c < c1; c += SMALL_T) {
            if (MODE == SM_RD) {
                RowEnt<W> L; load_lower<W>(a.v, c, L);
                double au[W], al[W], rn[W];
#pragma unroll
                for (int s = 0; s < W; s++) { au[s] = a.upper[L.f[s]]; al[s] = a.lower[L.f[s]]; rn[s] = L.on[s] ? s_ld(&a.rD[L.nb[s]]) : 1.0; }
                double d = a.diag[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) d -= au[s] * al[s] / rn[s];
                a.rD[c] = d;
            } else if (MODE == SM_PRECOND) {
                RowEnt<W> L; load_lower<W>(a.v, c, L);
                const double rd = a.rD[c], rc = a.r[c];
                double q[W], wn[W];
#pragma unroll
                for (int s = 0; s < W; s++) { q[s] = a.cf[L.f[s]]; wn[s] = L.on[s] ? s_ld(&a.w[L.nb[s]]) : 0.0; }
                double wc = rd * rc;
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) wc -= rd * q[s] * wn[s];
                a.w[c] = wc;
            } else {
                RowEnt<W> L, U; load_lower<W>(a.v, c, L); load_upper<W>(a.v, c, U);
                double al[W], au[W], pl[W], pu[W];
#pragma unroll
                for (int s = 0; s < W; s++) {
                    al[s] = a.lower[L.f[s]]; au[s] = a.upper[U.f[s]];
                    pl[s] = L.on[s] ? s_ld(&a.psi[L.nb[s]]) : 0.0; pu[s] = U.on[s] ? s_ld(&a.psi[U.nb[s]]) : 0.0;
                }
                double val = a.bP[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) val -= al[s] * pl[s];
                a.bSave[c] = val;
#pragma unroll
                for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
                a.psi[c] = val / a.diag[c];
            }
        }
        __syncthreads();
    }
    if (MODE == SM_RD) {            // rD = 1/rD (k_recip)
        for (int c = tid; c < a.N; c += SMALL_T) a.rD[c] = 1.0 / s_ld(&a.rD[c]);
        return;
    }
    if (MODE == SM_GS) return;
    // ---- backward levels (preconditioner: level 0 = cells without owned faces, nothing to do there)
    for (int b = (MODE == SM_PRECOND ? 1 : 0); b < a.nBwd; b++) {
        const int p0 = a.bwdRange[2 * b], p1 = a.bwdRange[2 * b + 1];
        for (int p = p0 + tid; p < p1; p += SMALL_T) {
            const int c = a.order ? a.order[p] : p;
            if (MODE == SM_PRECOND) {
                RowEnt<W> U; load_upper<W, true>(a.v, c, U);          // block-Jacobi: ghost neighbours are ignored
                const double rd = a.rD[c];
                double wc = s_ld(&a.w[c]);
                double q[W], wn[W];
#pragma unroll
                for (int s = 0; s < W; s++) { q[s] = a.cb[U.f[s]]; wn[s] = U.on[s] ? s_ld(&a.w[U.nb[s]]) : 0.0; }
#pragma unroll
                for (int s = W - 1; s >= 0; s--) if (U.on[s]) wc -= rd * q[s] * wn[s];
                a.w[c] = wc;
            } else {
                RowEnt<W> U; load_upper<W>(a.v, c, U);
                double au[W], pu[W];
#pragma unroll
                for (int s = 0; s < W; s++) { au[s] = a.upper[U.f[s]]; pu[s] = U.on[s] ? s_ld(&a.psi[U.nb[s]]) : 0.0; }
                double val = s_ld(&a.bSave[c]);
#pragma unroll
                for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
                a.psi[c] = val / a.diag[c];
            }
        }
        __syncthreads();
    }
}

// usable for this matrix?  (level-scheduled mode, at most FFM_SMALL_SWEEP_CELLS owned cells -- 0 switches it off; default 0, or
// 131072 when the dataflow sweeps are switched off)
static bool small_usable(ffm_ldu *A)
{
    if (A->smallState) return A->smallState > 0;
    const char *e = getenv("FFM_SMALL_SWEEP_CELLS");            // read per matrix: tests switch between the paths
    const char *f = getenv("FFM_FLOW_SWEEP");
    // measured (GAMG V-cycle at 96^3 and 200^3): the dataflow sweeps below beat this form at every size, so it is only the default
    // where they are switched off
    const long limit = e ? atol(e) : ((f && !strcmp(f, "0")) ? FFM_SMALL_SWEEP_DEFAULT : 0L);
    A->smallState = -1;
    if (A->sweepMode == 2 || A->nOwned > limit || A->nLevels < 1) return false;
    std::vector<int> br(2 * (size_t)std::max(A->nBwdLevels, 1), 0);
    for (int b = 0; b < A->nBwdLevels; b++) {
        const int s = A->h_bwdLevelStart[b], e = A->h_bwdLevelStart[b + 1];
        if (A->bwdContig) { br[2 * b] = A->h_bwdFirstCell[b]; br[2 * b + 1] = A->h_bwdFirstCell[b] + (e - s); }
        else { br[2 * b] = s; br[2 * b + 1] = e; }
    }
    if (hipMalloc((void **)&A->smallFwdStart, sizeof(int) * (A->nLevels + 1)) != hipSuccess) return false;
    if (hipMalloc((void **)&A->smallBwdRange, sizeof(int) * br.size()) != hipSuccess) return false;
    if (ffm_h2d(A->ctx, A->smallFwdStart, A->h_fwdLevelStart.data(), sizeof(int) * (A->nLevels + 1)) != FFM_OK) return false;
    if (ffm_h2d(A->ctx, A->smallBwdRange, br.data(), sizeof(int) * br.size()) != FFM_OK) return false;
    A->smallState = 1;
    return true;
}
static SmallArgs small_args(ffm_ldu *A)
{
    SmallArgs a{};
    a.v = ffm_view(A); a.nLevels = A->nLevels; a.nBwd = A->nBwdLevels; a.N = A->nOwned;
    a.fwdStart = A->smallFwdStart; a.bwdRange = A->smallBwdRange; a.order = A->bwdContig ? nullptr : A->bwdOrder;
    a.upper = A->upper; a.lower = A->lower; a.diag = A->diag; a.rD = A->rD;
    return a;
}
#define SMALL_LAUNCH(MODE, a) FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_small_sweep<MODE, W>), dim3(1), dim3(SMALL_T), 0, A->ctx->stream, a))

// --------------------------------------------------------- dataflow sweeps ---
// Level-scheduled matrices too large for one workgroup (unstructured meshes, the upper coarse levels of a GAMG hierarchy): instead of
// one launch -- or one device-wide barrier -- per dependency level, ONE launch in which every cell waits for the values it needs.
// Workgroups take chunks of 256 cells by an atomic ticket, forward chunks in the level-major cell order, then backward chunks in
// backward-level order: every value a cell waits for belongs to a chunk with a lower ticket, i.e. to a workgroup that is running or
// has finished, so the grid drains whatever the dispatch order or the residency (the scheme of the tiled sweeps, ffm_tile.hip).
// Values are published in sentinel-filled arrays with agent-scope stores and polled with agent-scope loads: the value is its own
// flag.  A lane never blocks in front of its store: the wait is a retry loop with the store inside and a wave-uniform exit (a ballot
// over the lanes not yet done -- with a per-lane exit the compiler may sink the store behind the loop, where a finished lane would
// wait for the lanes that wait for it), so a chain of dependent cells in one wavefront advances one link per trip.  Every wait is bounded and raises the abort word (reported when the solve ends).
// Per-cell arithmetic is that of the level kernels above, term for term in the same order: results are bitwise equal.
constexpr int FLOW_T = 256;            // measured: 64 is slower at 200^3 (3.6 against 2.36 ms per DIC application), 1024 equal there and slower on the GAMG levels
constexpr unsigned FLOW_SPIN_LIMIT = 1u << 21;
constexpr unsigned long long FLOW_SENT = 0xFFF8C0DEFEEDF10Full;       // a NaN no computation produces
__device__ __forceinline__ bool f_pending(double v) { return (unsigned long long)__double_as_longlong(v) == FLOW_SENT; }
__device__ __forceinline__ void f_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct FlowArgs {
    LduView v;
    int N, nChunkF, nChunkB, nap;
    const int *order;                       // [N] cells by backward level
    const double *upper, *lower, *diag, *cf, *cb, *r, *bP;
    double *rD, *w, *psi;
    double *mf, *mb, *sv;                   // published values: forward, backward, bSave (sentinel-filled before the launch)
    unsigned int *ticket;                   // [0] ticket counter, [1] abort word
};
__global__ void k_flow_fill(long n, double *a, double *b, double *c, unsigned int *ticket)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ticket[0] = 0u;
    const double sent = __longlong_as_double((long long)FLOW_SENT);
    GRID_STRIDE(i, n) { if (a) a[i] = sent; if (b) b[i] = sent; if (c) c[i] = sent; }
}
// one trip of a wait: false = keep waiting; raises / follows the abort word
__device__ __forceinline__ bool f_give_up(unsigned &spins, unsigned int *ticket, int nap)
{
    if (nap) __builtin_amdgcn_s_sleep(2);
    if ((++spins & 1023u) == 0u) {
        if (__hip_atomic_load(&ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (spins > FLOW_SPIN_LIMIT) { __hip_atomic_store(&ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    }
    return false;
}
template <int MODE, int W>
__global__ __launch_bounds__(FLOW_T) void k_flow_sweep(FlowArgs a)
{
    __shared__ unsigned shTicket;
    if (threadIdx.x == 0) shTicket = atomicAdd(&a.ticket[0], 1u);
    __syncthreads();
    const unsigned tk = shTicket;
    unsigned spins = 0;
    if (tk < (unsigned)a.nChunkF) {
        // ---- forward
        const int c = (int)tk * FLOW_T + threadIdx.x;
        if (c >= a.N) return;
        if (MODE == SM_RD) {
            RowEnt<W> L; load_lower<W>(a.v, c, L);
            double au[W], al[W], rn[W];
#pragma unroll
            for (int s = 0; s < W; s++) { au[s] = a.upper[L.f[s]]; al[s] = a.lower[L.f[s]]; rn[s] = 1.0; }
            const double dg = a.diag[c];
            bool done = false; do { if (!done) {
                bool ready = true;
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) { rn[s] = s_ld(&a.mf[L.nb[s]]); ready = ready && !f_pending(rn[s]); }
                if (ready) {
                    double d = dg;
#pragma unroll
                    for (int s = 0; s < W; s++) if (L.on[s]) d -= au[s] * al[s] / rn[s];
                    f_st(&a.mf[c], d);
                    a.rD[c] = 1.0 / d;              // k_recip
                    done = true;
                } else done = f_give_up(spins, a.ticket, a.nap);
            } } while (__ballot(!done) != 0ull);
        } else if (MODE == SM_PRECOND) {
            RowEnt<W> L; load_lower<W>(a.v, c, L);
            const double rd = a.rD[c], rc = a.r[c];
            double q[W], wn[W];
#pragma unroll
            for (int s = 0; s < W; s++) { q[s] = a.cf[L.f[s]]; wn[s] = 0.0; }
            bool done = false; do { if (!done) {
                bool ready = true;
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) { wn[s] = s_ld(&a.mf[L.nb[s]]); ready = ready && !f_pending(wn[s]); }
                if (ready) {
                    double wc = rd * rc;
#pragma unroll
                    for (int s = 0; s < W; s++) if (L.on[s]) wc -= rd * q[s] * wn[s];
                    f_st(&a.mf[c], wc);
                    done = true;
                } else done = f_give_up(spins, a.ticket, a.nap);
            } } while (__ballot(!done) != 0ull);
        } else {
            RowEnt<W> L, U; load_lower<W>(a.v, c, L); load_upper<W>(a.v, c, U);
            double al[W], au[W], pl[W], pu[W];
#pragma unroll
            for (int s = 0; s < W; s++) {
                al[s] = a.lower[L.f[s]]; au[s] = a.upper[U.f[s]];
                pl[s] = 0.0; pu[s] = U.on[s] ? a.psi[U.nb[s]] : 0.0;          // upper neighbours: the values before this sweep
            }
            const double bc = a.bP[c], dg = a.diag[c];
            bool done = false; do { if (!done) {
                bool ready = true;
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) { pl[s] = s_ld(&a.mf[L.nb[s]]); ready = ready && !f_pending(pl[s]); }
                if (ready) {
                    double val = bc;
#pragma unroll
                    for (int s = 0; s < W; s++) if (L.on[s]) val -= al[s] * pl[s];
                    const double save = val;
#pragma unroll
                    for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
                    val /= dg;
                    if (MODE == SM_SYMGS) f_st(&a.sv[c], save); else a.psi[c] = val;
                    f_st(&a.mf[c], val);
                    done = true;
                } else done = f_give_up(spins, a.ticket, a.nap);
            } } while (__ballot(!done) != 0ull);
        }
        return;
    }
    // ---- backward (preconditioner application, symGaussSeidel)
    if (MODE == SM_RD || MODE == SM_GS) return;
    const int p = (int)(tk - (unsigned)a.nChunkF) * FLOW_T + threadIdx.x;
    if (p >= a.N) return;
    const int c = a.order[p];
    if (MODE == SM_PRECOND) {
        RowEnt<W> U; load_upper<W, true>(a.v, c, U);          // block-Jacobi: ghost neighbours are ignored
        const double rd = a.rD[c];
        double q[W], wn[W];
#pragma unroll
        for (int s = 0; s < W; s++) { q[s] = a.cb[U.f[s]]; wn[s] = 0.0; }
        bool done = false; do { if (!done) {
            double wc = s_ld(&a.mf[c]);
            bool ready = !f_pending(wc);
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) { wn[s] = s_ld(&a.mb[U.nb[s]]); ready = ready && !f_pending(wn[s]); }
            if (ready) {
#pragma unroll
                for (int s = W - 1; s >= 0; s--) if (U.on[s]) wc -= rd * q[s] * wn[s];
                f_st(&a.mb[c], wc);
                a.w[c] = wc;
                done = true;
            } else done = f_give_up(spins, a.ticket, a.nap);
        } } while (__ballot(!done) != 0ull);
    } else {
        RowEnt<W> U; load_upper<W>(a.v, c, U);
        double au[W], pu[W];
        bool own[W];
#pragma unroll
        for (int s = 0; s < W; s++) {
            au[s] = a.upper[U.f[s]]; own[s] = U.on[s] && U.nb[s] < a.N;
            pu[s] = (U.on[s] && !own[s]) ? a.psi[U.nb[s]] : 0.0;              // ghost neighbours: unchanged by the sweep
        }
        const double dg = a.diag[c];
        bool done = false; do { if (!done) {
            double val = s_ld(&a.sv[c]);
            bool ready = !f_pending(val);
#pragma unroll
            for (int s = 0; s < W; s++) if (own[s]) { pu[s] = s_ld(&a.mb[U.nb[s]]); ready = ready && !f_pending(pu[s]); }
            if (ready) {
#pragma unroll
                for (int s = 0; s < W; s++) if (U.on[s]) val -= au[s] * pu[s];
                val /= dg;
                f_st(&a.mb[c], val);
                a.psi[c] = val;
                done = true;
            } else done = f_give_up(spins, a.ticket, a.nap);
        } } while (__ballot(!done) != 0ull);
    }
}

// usable for this matrix?  level-scheduled mode, not taken by the single-workgroup sweeps; FFM_FLOW_SWEEP=0 switches it off (one
// launch per level, the round-1 path), FFM_FLOW_SWEEP=all takes every matrix whatever FFM_SMALL_SWEEP_CELLS says.
// Measured on one MI355X: a DIC application (two sweeps) in level-major numbering 4.30 -> 2.36 ms at 200^3 (598 levels: 2.0 us per
// level and sweep, the store -> poll round trip between workgroups on different XCDs; the poll's s_sleep makes no difference),
// 1.66 -> 1.05 ms at 100^3; a GAMG V-cycle 29.0 -> 15.5 ms at 200^3 and 8.6 -> 5.8 ms at 96^3 (profiles/README.md)
static bool flow_usable(ffm_ldu *A)
{
    if (A->flowState) return A->flowState > 0;
    const char *e = getenv("FFM_FLOW_SWEEP");                   // read per matrix: tests switch between the paths
    A->flowState = -1;
    if (A->sweepMode == 2 || A->nLevels < 1 || A->nOwned < 1 || (e && !strcmp(e, "0"))) return false;
    std::vector<int> ord((size_t)A->nOwned);
    if (A->bwdContig) {
        size_t k = 0;
        for (int b = 0; b < A->nBwdLevels; b++)
            for (int i = 0, n = A->h_bwdLevelStart[b + 1] - A->h_bwdLevelStart[b]; i < n; i++) ord[k++] = A->h_bwdFirstCell[b] + i;
        if (k != ord.size()) return false;
    } else {
        if (hipStreamSynchronize(A->ctx->stream) != hipSuccess) return false;        // (uploaded on the context's non-blocking stream)
        if (ffm_d2h(A->ctx, ord.data(), A->bwdOrder, sizeof(int) * ord.size()) != FFM_OK) return false;
    }
    if (hipMalloc((void **)&A->flowOrder, sizeof(int) * ord.size()) != hipSuccess) return false;
    if (ffm_h2d(A->ctx, A->flowOrder, ord.data(), sizeof(int) * ord.size()) != FFM_OK) return false;
    if (!A->sweepTicket) {
        if (hipMalloc((void **)&A->sweepTicket, 2 * sizeof(unsigned int)) != hipSuccess) return false;
        // on the context's stream: a null-stream hipMemset returns before it has run and is not ordered against that (non-blocking)
        // stream -- it could zero the ticket counter in the middle of the first sweep
        if (hipMemsetAsync(A->sweepTicket, 0, 2 * sizeof(unsigned int), A->ctx->stream) != hipSuccess) return false;
    }
    A->flowState = 1;
    return true;
}
static bool flow_preferred(ffm_ldu *A)
{
    const char *e = getenv("FFM_FLOW_SWEEP");
    return e && !strcmp(e, "all") && flow_usable(A);
}
static int flow_args(ffm_ldu *A, int mode, FlowArgs &a)
{
    a = FlowArgs{};
    a.v = ffm_view(A); a.N = A->nOwned; a.nChunkF = (A->nOwned + FLOW_T - 1) / FLOW_T;
    a.nChunkB = (mode == SM_PRECOND || mode == SM_SYMGS) ? a.nChunkF : 0;
    a.nap = 1;
    a.order = A->flowOrder; a.upper = A->upper; a.lower = A->lower; a.diag = A->diag; a.rD = A->rD; a.ticket = A->sweepTicket;
    FFM_TRY(ffm_ldu_work(A, 20, &a.mf));
    if (a.nChunkB) FFM_TRY(ffm_ldu_work(A, 21, &a.mb));
    if (mode == SM_SYMGS) FFM_TRY(ffm_ldu_work(A, 22, &a.sv));
    hipLaunchKernelGGL(k_flow_fill, dim3(sgrid(A->nOwned)), dim3(256), 0, A->ctx->stream, (long)A->nOwned, a.mf, a.mb, a.sv, a.ticket);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
#define FLOW_LAUNCH(MODE, a) FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_flow_sweep<MODE, W>), dim3((a).nChunkF + (a).nChunkB), dim3(FLOW_T), 0, A->ctx->stream, a))

// the abort word of the dataflow sweeps (a bounded wait ran out): reported when a solve ends
int ffm_flow_check_abort(ffm_ldu *A)
{
    if (A->flowState <= 0 || !A->sweepTicket) return FFM_OK;
    unsigned int h[2] = {0, 0};
    FFM_HIP(hipMemcpyAsync(h, A->sweepTicket, sizeof(h), hipMemcpyDeviceToHost, A->ctx->stream));
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (h[1]) {
        ffm_set_error("dataflow sweep timed out waiting for the value of a predecessor cell (abort word set)");
        unsigned int z = 0;
        ffm_h2d(A->ctx, A->sweepTicket + 1, &z, sizeof(z));
        return FFM_ERR_HIP;
    }
    return FFM_OK;
}

// blocks needed to cover cells [c0,c1) when thread 0 of block 0 sits on the slice start of c0
static inline int level_grid(int c0, int c1) { return ffm_grid(c1 - (c0 & ~63), 256); }

// ------------------------------------------------------ level-sweep drivers ---
enum { SW_RD = 1, SW_PRECOND = 2, SW_GS = 3, SW_SYMGS = 4 };

template <class Body>
static int run_graphed(ffm_ldu *A, const SweepGraphKey &key, Body body)
{
    static const bool noGraph = getenv("FFM_NO_GRAPH") != nullptr;
    hipStream_t s = A->ctx->stream;
    if (noGraph) return body();
    auto it = A->graphs.find(key);
    if (it == A->graphs.end()) {
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        FFM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = body();
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) { ffm_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return FFM_ERR_HIP; }
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) { ffm_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return FFM_ERR_HIP; }
        if (A->graphs.size() >= FFM_MAX_SWEEP_GRAPHS) {        // bounded cache: drop the oldest graph
            auto old = A->graphs.find(A->graphOrder.front());
            if (old != A->graphs.end()) { hipGraphExecDestroy(old->second); A->graphs.erase(old); }
            A->graphOrder.erase(A->graphOrder.begin());
        }
        it = A->graphs.emplace(key, ge).first;
        A->graphOrder.push_back(key);
    }
    FFM_HIP(hipGraphLaunch(it->second, s));
    return FFM_OK;
}

static inline void bwd_range(const ffm_ldu *A, int b, int &p0, int &p1, const int *&order)
{
    const int s = A->h_bwdLevelStart[b], e = A->h_bwdLevelStart[b + 1];
    if (A->bwdContig) { p0 = A->h_bwdFirstCell[b]; p1 = p0 + (e - s); order = nullptr; }
    else { p0 = s; p1 = e; order = A->bwdOrder; }
}

// rD = 1/(diag - sum upper*lower/rD[l]) in face order (DIC: lower == upper)
static int calc_rD(ffm_ldu *A)
{
    hipStream_t s = A->ctx->stream;
    if (A->sweepMode == 2) {
        FFM_TRY(ffm_tile_calc_rD(A));
        hipLaunchKernelGGL(k_recip, dim3(sgrid(A->nOwned)), dim3(256), 0, s, (long)A->nOwned, A->rD, A->rD);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    if (!flow_preferred(A) && small_usable(A)) {
        SmallArgs a = small_args(A);
        SMALL_LAUNCH(SM_RD, a);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    if (flow_usable(A)) {
        FlowArgs a; FFM_TRY(flow_args(A, SM_RD, a));
        FLOW_LAUNCH(SM_RD, a);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    SweepGraphKey key{SW_RD, {A->lower, A->upper, A->diag, A->rD, nullptr, nullptr}};
    FFM_TRY(run_graphed(A, key, [&]() -> int {
        for (int L = 0; L < A->nLevels; L++) {
            const int c0 = A->h_fwdLevelStart[L], c1 = A->h_fwdLevelStart[L + 1];
            if (c1 == c0) continue;
            FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL(k_rD_level<W>, dim3(level_grid(c0, c1)), dim3(256), 0, s, c0, c1, ffm_view(A),
                               A->upper, A->lower, A->diag, A->rD));
        }
        hipLaunchKernelGGL(k_recip, dim3(sgrid(A->nOwned)), dim3(256), 0, s, (long)A->nOwned, A->rD, A->rD);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }));
    return FFM_OK;
}

int ffm_precond_setup_i(ffm_ldu *A, int precond)
{
    if (A->rDKind == precond && A->rDEpoch == A->coeffEpoch) return FFM_OK;
    hipStream_t s = A->ctx->stream;
    switch (precond) {
    case FFM_NONE: break;
    case FFM_DIC:
        if (!A->symmetric) { ffm_set_error("DIC needs a symmetric matrix"); return FFM_ERR_UNSUPPORTED; }
        FFM_TRY(calc_rD(A)); break;
    case FFM_DILU: FFM_TRY(calc_rD(A)); break;
    case FFM_DIAGONALP:
        hipLaunchKernelGGL(k_recip, dim3(sgrid(A->nOwned)), dim3(256), 0, s, (long)A->nOwned, A->rD, A->diag);
        FFM_HIP(hipGetLastError());
        break;
    default: ffm_set_error("unknown preconditioner %d", precond); return FFM_ERR_UNSUPPORTED;
    }
    A->rDKind = precond; A->rDEpoch = A->coeffEpoch;
    return FFM_OK;
}

int ffm_precond_apply_i(ffm_ldu *A, int precond, bool transpose, const double *r, double *w)
{
    hipStream_t s = A->ctx->stream;
    const long N = A->nOwned;
    if (precond == FFM_NONE) { hipLaunchKernelGGL(k_copy, dim3(sgrid(N)), dim3(256), 0, s, N, w, r); FFM_HIP(hipGetLastError()); return FFM_OK; }
    if (precond == FFM_DIAGONALP) { hipLaunchKernelGGL(k_mul, dim3(sgrid(N)), dim3(256), 0, s, N, w, A->rD, r); FFM_HIP(hipGetLastError()); return FFM_OK; }
    // DIC: fwd upper / bwd upper.  DILU: fwd lower / bwd upper.  DILU^T: fwd upper / bwd lower.
    const double *cf = (precond == FFM_DIC) ? A->upper : (transpose ? A->upper : A->lower);
    const double *cb = (precond == FFM_DIC) ? A->upper : (transpose ? A->lower : A->upper);
    if (A->sweepMode == 2) return ffm_tile_precond(A, precond, transpose, r, w);
    if (!flow_preferred(A) && small_usable(A)) {
        SmallArgs a = small_args(A);
        a.cf = cf; a.cb = cb; a.r = r; a.w = w;
        SMALL_LAUNCH(SM_PRECOND, a);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    if (flow_usable(A)) {
        FlowArgs a; FFM_TRY(flow_args(A, SM_PRECOND, a));
        a.cf = cf; a.cb = cb; a.r = r; a.w = w;
        FLOW_LAUNCH(SM_PRECOND, a);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    SweepGraphKey key{SW_PRECOND + 16 * (transpose ? 1 : 0) + 32 * precond, {r, w, cf, cb, A->rD, nullptr}};
    return run_graphed(A, key, [&]() -> int {
        for (int L = 0; L < A->nLevels; L++) {
            const int c0 = A->h_fwdLevelStart[L], c1 = A->h_fwdLevelStart[L + 1];
            if (c1 == c0) continue;
            FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL(k_fwd_level<W>, dim3(level_grid(c0, c1)), dim3(256), 0, s, c0, c1, ffm_view(A), cf, A->rD, r, w));
        }
        // backward level 0 = cells without owned faces: nothing to do
        for (int b = 1; b < A->nBwdLevels; b++) {
            int p0, p1; const int *order; bwd_range(A, b, p0, p1, order);
            if (p1 == p0) continue;
            FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL(k_bwd_level<W>, dim3(order ? ffm_grid(p1 - p0, 256) : level_grid(p0, p1)), dim3(256), 0, s,
                                                       p0, p1, order, ffm_view(A), cb, A->rD, w));
        }
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    });
}

// GaussSeidelSmoother::smooth / symGaussSeidelSmoother::smooth
int ffm_gs_smooth_i(ffm_ldu *A, bool sym, int nSweeps, double *psi, const double *b)
{
    hipStream_t s = A->ctx->stream;
    const long N = A->nOwned;
    double *bP, *bSave;
    FFM_TRY(ffm_ldu_work(A, 10, &bP)); FFM_TRY(ffm_ldu_work(A, 11, &bSave));
    for (int sw = 0; sw < nSweeps; sw++) {
        const double *bUse = b;
        if (!A->ifaces.empty()) {   // bPrime = source + bou*pnf (negated interface coefficients)
            hipLaunchKernelGGL(k_copy, dim3(sgrid(N)), dim3(256), 0, s, N, bP, b);
            FFM_TRY(ffm_halo_update(A, psi, bP, A->ifBou, +1.0));
            bUse = bP;
        }
        // decomposed block: every sweep starts from the neighbour ranks' current values, whatever form the sweep takes (the
        // reference's smoothers update the interfaces before every sweep: GaussSeidelSmoother.C, initMatrixInterfaces /
        // updateMatrixInterfaces inside the sweep loop) -- the single-workgroup, dataflow and per-level forms read psi of the ghost cells
        if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange(A, psi));
        if (A->sweepMode == 2 && ffm_tile_gs_usable(A)) {
            if (!A->ghNbrRank.empty()) {        // move the ghost cells' terms into bPrime
                if (bUse == b) { hipLaunchKernelGGL(k_copy, dim3(sgrid(N)), dim3(256), 0, s, N, bP, b); bUse = bP; }
                FFM_TRY(ffm_tile_gs_ghost_terms(A, psi, bP));
            }
            if (!A->gsProd) FFM_HIP(hipMalloc((void **)&A->gsProd, sizeof(double) * 3 * (size_t)std::max(A->nCells, 1)));
            FFM_TRY(ffm_tile_gs(A, sym, psi, bUse, bSave, A->gsProd));
            continue;
        }
        if (!flow_preferred(A) && small_usable(A)) {
            SmallArgs a = small_args(A);
            a.bP = bUse; a.bSave = bSave; a.psi = psi;
            if (sym) SMALL_LAUNCH(SM_SYMGS, a); else SMALL_LAUNCH(SM_GS, a);
            FFM_HIP(hipGetLastError());
            continue;
        }
        if (flow_usable(A)) {
            FlowArgs a; FFM_TRY(flow_args(A, sym ? SM_SYMGS : SM_GS, a));
            a.bP = bUse; a.psi = psi;
            if (sym) FLOW_LAUNCH(SM_SYMGS, a); else FLOW_LAUNCH(SM_GS, a);
            FFM_HIP(hipGetLastError());
            continue;
        }
        SweepGraphKey key{sym ? SW_SYMGS : SW_GS, {psi, bUse, A->lower, A->upper, A->diag, bSave}};
        FFM_TRY(run_graphed(A, key, [&]() -> int {
            for (int L = 0; L < A->nLevels; L++) {
                const int c0 = A->h_fwdLevelStart[L], c1 = A->h_fwdLevelStart[L + 1];
                if (c1 == c0) continue;
                FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL(k_gs_fwd_level<W>, dim3(level_grid(c0, c1)), dim3(256), 0, s, c0, c1, ffm_view(A),
                                                           A->upper, A->lower, A->diag, bUse, bSave, psi));
            }
            if (sym) for (int bl = 0; bl < A->nBwdLevels; bl++) {
                int p0, p1; const int *order; bwd_range(A, bl, p0, p1, order);
                if (p1 == p0) continue;
                FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL(k_gs_bwd_level<W>, dim3(order ? ffm_grid(p1 - p0, 256) : level_grid(p0, p1)), dim3(256),
                                                           0, s, p0, p1, order, ffm_view(A), A->upper, A->diag, bSave, psi));
            }
            FFM_HIP(hipGetLastError());
            return FFM_OK;
        }));
    }
    return FFM_OK;
}

// ------------------------------------------------------------------ solvers ---
struct Controls { double tol, relTol; int minIter, maxIter, nSweeps; };

static inline bool check_convergence(ffm_perf *p, const Controls &k)
{
    p->converged = (p->finalResidual < k.tol || (k.relTol > 1e-20 && p->finalResidual < k.relTol * p->initialResidual)) ? 1 : 0;
    return p->converged;
}

// normFactor + initial residual: wA = A psi already computed, rA = source - wA already formed
// tmp holds sumA on entry (ffm_k_spmv_sumA at the top of every solver)
static int norm_and_initial(ffm_ldu *A, const double *psi, const double *source, const double *Apsi, double *tmp,
                            const double *rA, ffm_perf *perf)
{
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned;
    FFM_TRY(ffm_k_sum(c, psi, N, S_TMP0));
    FFM_TRY(finish_dot(c, OP_XREF, 1, (double)A->globalCells));
    const int g = sgrid(N);
    hipLaunchKernelGGL(k_normf, dim3(g), dim3(256), 0, s, N, Apsi, source, tmp, c->scal_d, c->partials_d);
    FFM_TRY(partial_sum_to(c, g, S_TMP0));
    FFM_TRY(finish_dot(c, OP_NORMF));
    FFM_TRY(ffm_k_summag(c, rA, N, S_TMP0));
    FFM_TRY(finish_dot(c, OP_RES_INIT));
    FFM_TRY(ffm_read_scalars(c));
    perf->initialResidual = c->scal_h[S_RES0];
    perf->finalResidual = perf->initialResidual;
    perf->nIterations = 0; perf->singular = 0; perf->converged = 0;
    return FFM_OK;
}

// PCG::solve
static int pcg(ffm_ldu *A, int precond, const Controls &k, double *psi, const double *source, ffm_perf *perf)
{
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned; const int g = sgrid(N);
    double *pA, *wA, *rA;
    FFM_TRY(ffm_ldu_work(A, 1, &pA)); FFM_TRY(ffm_ldu_work(A, 2, &wA)); FFM_TRY(ffm_ldu_work(A, 3, &rA));
    FFM_TRY(scalar_op(c, OP_RESET));
    FFM_TRY(ffm_k_spmv_sumA(A, psi, wA, pA));
    hipLaunchKernelGGL(k_sub, dim3(g), dim3(256), 0, s, N, rA, source, wA);
    FFM_TRY(norm_and_initial(A, psi, source, wA, pA, rA, perf));
    if (k.minIter > 0 || !check_convergence(perf, k)) {
        FFM_TRY(ffm_precond_setup_i(A, precond));
        if (precond == FFM_DIC && A->sweepMode == 2 && ffm_tile_pcg_fusable(A)) {
            // The same iteration with the vector updates carried by the sweeps (ffm_tile.hip, k_tile<.., FUSE>): the residual
            // update and its norm ride on the forward sweep of the NEXT preconditioner application (which is therefore started
            // before the convergence check: one unused forward sweep per solve), wA.rA on the backward sweep, and psi += alpha pA
            // is deferred into the next search-direction update.  Per-cell arithmetic unchanged.
            // one more fusion where the tiled Amul runs without ghost faces: the direction update and the deferred solution update
            // ride on the Amul (k_tile_amul<true, true>: pNew = wA + beta*pOld, psi += alpha*pOld, qA = A pNew, pNew.qA), with
            // two direction buffers and the Amul result in a vector of its own (FFM_PCG_AMUL_UNFUSED=1: separate k_p_psi)
            const char *ea = getenv("FFM_PCG_AMUL_UNFUSED");
            const bool fuseP = ffm_tile_amul_pcg_usable(A) && !(ea && atoi(ea) != 0);
            double *pB = nullptr, *qA = wA;
            if (fuseP) { FFM_TRY(ffm_ldu_work(A, 4, &pB)); FFM_TRY(ffm_ldu_work(A, 5, &qA)); }
            do {
                if (perf->nIterations == 0) {
                    FFM_TRY(ffm_precond_apply_i(A, precond, false, rA, wA));
                    FFM_TRY(ffm_k_dot(c, wA, rA, N, S_TMP0));
                } else FFM_TRY(ffm_tile_pcg_bwd(A, rA, wA, S_TMP0));
                FFM_TRY(finish_dot(c, OP_PCG_BETA));
                if (perf->nIterations == 0) {
                    hipLaunchKernelGGL(k_p_update, dim3(g), dim3(256), 0, s, N, pA, wA, c->scal_d, 1);
                    FFM_TRY(ffm_k_spmv_dot(A, pA, qA, S_TMP0));
                } else if (fuseP) {
                    FFM_TRY(ffm_tile_amul_pcg(A, wA, pA, pB, psi, qA, S_TMP0));
                    std::swap(pA, pB);
                } else {
                    hipLaunchKernelGGL(k_p_psi, dim3(g), dim3(256), 0, s, N, psi, pA, wA, c->scal_d);
                    FFM_TRY(ffm_k_spmv_dot(A, pA, qA, S_TMP0));
                }
                FFM_TRY(finish_dot(c, OP_PCG_ALPHA));
                FFM_TRY(ffm_tile_pcg_fwd(A, rA, wA, S_TMP0, qA == wA ? nullptr : qA));
                FFM_TRY(finish_dot(c, OP_RES));
                FFM_TRY(ffm_read_scalars(c));
                if (c->scal_h[S_SING] != 0.0) { perf->singular = 1; break; }
                perf->finalResidual = c->scal_h[S_RES];
            } while ((++perf->nIterations < k.maxIter && !check_convergence(perf, k)) || perf->nIterations < k.minIter);
            if (!perf->singular) hipLaunchKernelGGL(k_axpy_alpha, dim3(g), dim3(256), 0, s, N, psi, pA, c->scal_d);
            FFM_HIP(hipGetLastError());
            return FFM_OK;
        }
        do {
            FFM_TRY(ffm_precond_apply_i(A, precond, false, rA, wA));
            FFM_TRY(ffm_k_dot(c, wA, rA, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_PCG_BETA));
            hipLaunchKernelGGL(k_p_update, dim3(g), dim3(256), 0, s, N, pA, wA, c->scal_d, perf->nIterations == 0 ? 1 : 0);
            FFM_TRY(ffm_k_spmv_dot(A, pA, wA, S_TMP0));
            FFM_TRY(finish_dot(c, OP_PCG_ALPHA));
            hipLaunchKernelGGL(k_pcg_xr, dim3(g), dim3(256), 0, s, N, psi, rA, pA, wA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
            FFM_TRY(ffm_read_scalars(c));
            if (c->scal_h[S_SING] != 0.0) { perf->singular = 1; break; }
            perf->finalResidual = c->scal_h[S_RES];
        } while ((++perf->nIterations < k.maxIter && !check_convergence(perf, k)) || perf->nIterations < k.minIter);
    }
    return FFM_OK;
}

// PBiCGStab::solve
static int pbicgstab(ffm_ldu *A, int precond, const Controls &k, double *psi, const double *source, ffm_perf *perf)
{
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned; const int g = sgrid(N);
    double *yA, *rA, *pA, *AyA, *sA, *zA, *tA, *rA0;
    FFM_TRY(ffm_ldu_work(A, 1, &pA)); FFM_TRY(ffm_ldu_work(A, 2, &yA)); FFM_TRY(ffm_ldu_work(A, 3, &rA));
    FFM_TRY(scalar_op(c, OP_RESET));
    FFM_TRY(ffm_k_spmv_sumA(A, psi, yA, pA));
    hipLaunchKernelGGL(k_sub, dim3(g), dim3(256), 0, s, N, rA, source, yA);
    FFM_TRY(norm_and_initial(A, psi, source, yA, pA, rA, perf));
    if (k.minIter > 0 || !check_convergence(perf, k)) {
        FFM_TRY(ffm_ldu_work(A, 4, &AyA)); FFM_TRY(ffm_ldu_work(A, 5, &sA)); FFM_TRY(ffm_ldu_work(A, 6, &zA));
        FFM_TRY(ffm_ldu_work(A, 7, &tA)); FFM_TRY(ffm_ldu_work(A, 8, &rA0));
        hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, N, rA0, rA);
        FFM_TRY(ffm_precond_setup_i(A, precond));
        do {
            FFM_TRY(ffm_k_dot(c, rA0, rA, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BS_RHO, 1, 0.0, perf->nIterations));
            hipLaunchKernelGGL(k_bs_p, dim3(g), dim3(256), 0, s, N, pA, rA, AyA, c->scal_d, perf->nIterations == 0 ? 1 : 0);
            FFM_TRY(ffm_precond_apply_i(A, precond, false, pA, yA));
            FFM_TRY(ffm_k_spmv(A, yA, AyA, false));
            FFM_TRY(ffm_k_dot(c, rA0, AyA, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BS_ALPHA));
            hipLaunchKernelGGL(k_bs_s, dim3(g), dim3(256), 0, s, N, sA, rA, AyA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
            FFM_TRY(ffm_read_scalars(c));
            if (c->scal_h[S_SING] != 0.0) { perf->singular = 1; break; }
            perf->finalResidual = c->scal_h[S_RES];
            if (check_convergence(perf, k)) {
                hipLaunchKernelGGL(k_axpy_alpha, dim3(g), dim3(256), 0, s, N, psi, yA, c->scal_d);
                perf->nIterations++;
                return FFM_OK;
            }
            FFM_TRY(ffm_precond_apply_i(A, precond, false, sA, zA));
            FFM_TRY(ffm_k_spmv(A, zA, tA, false));
            hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, s, N, tA, sA, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0, 2));
            FFM_TRY(finish_dot(c, OP_BS_OMEGA, 2));
            hipLaunchKernelGGL(k_bs_xr, dim3(g), dim3(256), 0, s, N, psi, rA, yA, zA, sA, tA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
            FFM_TRY(ffm_read_scalars(c));
            perf->finalResidual = c->scal_h[S_RES];
        } while ((++perf->nIterations < k.maxIter && !check_convergence(perf, k)) || perf->nIterations < k.minIter);
    }
    return FFM_OK;
}

// PBiCG::solve
static int pbicg(ffm_ldu *A, int precond, const Controls &k, double *psi, const double *source, ffm_perf *perf)
{
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned; const int g = sgrid(N);
    double *pA, *pT, *wA, *wT, *rA, *rT;
    FFM_TRY(ffm_ldu_work(A, 1, &pA)); FFM_TRY(ffm_ldu_work(A, 2, &wA)); FFM_TRY(ffm_ldu_work(A, 3, &rA));
    FFM_TRY(ffm_ldu_work(A, 4, &pT)); FFM_TRY(ffm_ldu_work(A, 5, &wT)); FFM_TRY(ffm_ldu_work(A, 6, &rT));
    FFM_TRY(scalar_op(c, OP_RESET));
    FFM_TRY(ffm_k_spmv_sumA(A, psi, wA, pA));
    FFM_TRY(ffm_k_spmv(A, psi, wT, true));
    hipLaunchKernelGGL(k_sub2, dim3(g), dim3(256), 0, s, N, rA, rT, source, wA, wT);
    FFM_TRY(norm_and_initial(A, psi, source, wA, pA, rA, perf));
    if (k.minIter > 0 || !check_convergence(perf, k)) {
        FFM_TRY(ffm_precond_setup_i(A, precond));
        do {
            FFM_TRY(ffm_precond_apply_i(A, precond, false, rA, wA));
            FFM_TRY(ffm_precond_apply_i(A, precond, true, rT, wT));
            FFM_TRY(ffm_k_dot(c, wA, rT, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BICG_BETA));
            const int first = perf->nIterations == 0 ? 1 : 0;
            hipLaunchKernelGGL(k_p_update, dim3(g), dim3(256), 0, s, N, pA, wA, c->scal_d, first);
            hipLaunchKernelGGL(k_p_update, dim3(g), dim3(256), 0, s, N, pT, wT, c->scal_d, first);
            FFM_TRY(ffm_k_spmv(A, pA, wA, false));
            FFM_TRY(ffm_k_spmv(A, pT, wT, true));
            FFM_TRY(ffm_k_dot(c, wA, pT, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BICG_ALPHA));
            hipLaunchKernelGGL(k_axmy_alpha, dim3(g), dim3(256), 0, s, N, rT, wT, c->scal_d);
            hipLaunchKernelGGL(k_pcg_xr, dim3(g), dim3(256), 0, s, N, psi, rA, pA, wA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
            FFM_TRY(ffm_read_scalars(c));
            if (c->scal_h[S_SING] != 0.0) { perf->singular = 1; break; }
            perf->finalResidual = c->scal_h[S_RES];
        } while ((++perf->nIterations < k.maxIter && !check_convergence(perf, k)) || perf->nIterations < k.minIter);
    }
    return FFM_OK;
}

// smoothSolver::solve
static int smooth(ffm_ldu *A, int smoother, const Controls &k, double *psi, const double *source, ffm_perf *perf)
{
    if (smoother != FFM_GS && smoother != FFM_SYMGS) { ffm_set_error("smoothSolver: smoother must be GaussSeidel or symGaussSeidel"); return FFM_ERR_UNSUPPORTED; }
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned; const int g = sgrid(N);
    const int nSweeps = k.nSweeps > 0 ? k.nSweeps : 1;
    double *Apsi, *tmp, *res;
    FFM_TRY(ffm_ldu_work(A, 1, &Apsi)); FFM_TRY(ffm_ldu_work(A, 2, &tmp)); FFM_TRY(ffm_ldu_work(A, 3, &res));
    FFM_TRY(scalar_op(c, OP_RESET));
    FFM_TRY(ffm_k_spmv_sumA(A, psi, Apsi, tmp));
    hipLaunchKernelGGL(k_sub, dim3(g), dim3(256), 0, s, N, res, source, Apsi);
    FFM_TRY(norm_and_initial(A, psi, source, Apsi, tmp, res, perf));
    if (k.minIter > 0 || !check_convergence(perf, k)) {
        do {
            FFM_TRY(ffm_gs_smooth_i(A, smoother == FFM_SYMGS, nSweeps, psi, source));
            FFM_TRY(ffm_k_residual(A, psi, source, res));
            FFM_TRY(ffm_k_summag(c, res, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
            FFM_TRY(ffm_read_scalars(c));
            perf->finalResidual = c->scal_h[S_RES];
        } while (((perf->nIterations += nSweeps) < k.maxIter && !check_convergence(perf, k)) || perf->nIterations < k.minIter);
    }
    return FFM_OK;
}

static int diagonal(ffm_ldu *A, double *psi, const double *source, ffm_perf *perf)
{
    hipLaunchKernelGGL(k_diag_solve, dim3(sgrid(A->nOwned)), dim3(256), 0, A->ctx->stream, (long)A->nOwned, psi, source, A->diag);
    FFM_HIP(hipGetLastError());
    perf->initialResidual = perf->finalResidual = 0.0; perf->nIterations = 0; perf->converged = 1; perf->singular = 0;
    return FFM_OK;
}

static int solve_internal(ffm_ldu *A, int solver, int precond, const Controls &k, double *psi, const double *source, ffm_perf *perf)
{
    switch (solver) {
    case FFM_PCG:
        if (!A->symmetric) { ffm_set_error("PCG needs a symmetric matrix"); return FFM_ERR_UNSUPPORTED; }
        return pcg(A, precond, k, psi, source, perf);
    case FFM_PBICGSTAB: return pbicgstab(A, precond, k, psi, source, perf);
    case FFM_PBICG: return pbicg(A, precond, k, psi, source, perf);
    case FFM_DIAGONAL: return diagonal(A, psi, source, perf);
    case FFM_SMOOTH: return smooth(A, precond, k, psi, source, perf);
    }
    ffm_set_error("unknown solver %d", solver);
    return FFM_ERR_UNSUPPORTED;
}

// library-internal: vectors in the matrix's internal cell order, no final synchronisation (ffm_gamg.hip: coarsest level)
extern "C" int ffm_solve_internal_i(ffm_ldu *A, int solver, int precond, double tol, double relTol, int minIter, int maxIter, int nSweeps,
                                    double *psi, const double *source, ffm_perf *perf)
{
    memset(perf, 0, sizeof(*perf));
    Controls k{tol, relTol, minIter, maxIter, nSweeps};
    return solve_internal(A, solver, precond, k, psi, source, perf);
}

extern "C" int ffm_solve_d(ffm_ldu *A, int solver, int precond, double tol, double relTol, int minIter, int maxIter,
                           int nSweeps, double *psi_d, const double *source_d, ffm_perf *out)
{
    if (!A || !psi_d || !source_d || !out) { ffm_set_error("ffm_solve_d: null argument"); return FFM_ERR_ARG; }
    if ((solver == FFM_PCG || solver == FFM_PBICGSTAB || solver == FFM_PBICG) &&
        !(precond == FFM_NONE || precond == FFM_DIC || precond == FFM_DILU || precond == FFM_DIAGONALP)) {
        ffm_set_error("solver %d: preconditioner %d not offered", solver, precond); return FFM_ERR_UNSUPPORTED;
    }
    memset(out, 0, sizeof(*out));
    Controls k{tol, relTol, minIter, maxIter, nSweeps};
    FFM_HIP(hipSetDevice(A->ctx->device));
    if (A->identity) {
        FFM_TRY(solve_internal(A, solver, precond, k, psi_d, source_d, out));
    } else {
        const double *si; double *pi;
        FFM_TRY(ffm_to_internal(A, source_d, 1, &si));
        const double *pin; FFM_TRY(ffm_to_internal(A, psi_d, 2, &pin)); pi = A->permIn[2];
        FFM_TRY(solve_internal(A, solver, precond, k, pi, si, out));
        FFM_TRY(ffm_from_internal(A, pi, psi_d));
    }
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (A->sweepMode == 2) FFM_TRY(ffm_tile_check_abort(A)); else FFM_TRY(ffm_flow_check_abort(A));
    return FFM_OK;
}

// ------------------------------------------------------------------ several systems with common off-diagonals ---
// fvMatrix::solveSegregated of a vector equation and the species loop under a multivariateSelection scheme solve systems that differ in
// the diagonal and the right-hand side only.  PBiCGStab + DILU (or DIC) of nSys <= FFM_TILE_MAXSYS such systems run in lock step: every
// system keeps its own solver state (scalars, work vectors, reciprocal diagonal) and goes through exactly the operations of pbicgstab()
// above in the same order -- its numbers and its iteration count are those of a solve of its own -- but the preconditioner sweeps and
// calcReciprocalD of the systems still iterating are ONE tiled sweep each (ffm_tile.hip: k_tile_m), and the host reads all residuals
// back with one synchronisation per half iteration.
namespace {
struct Lane {
    const double *diag, *source; double *psi; ffm_perf *perf;
    double *scal_d, *scal_h;
    double *rD, *pA, *yA, *rA, *AyA, *sA, *zA, *tA, *rA0;
    bool active;
};
struct LaneGuard {          // the context's scalar block and the matrix's diagonal / reciprocal diagonal are switched per lane
    ffm_ldu *A; double *scal_d, *scal_h, *diag, *rD; int rDKind; unsigned long rDEpoch;
    explicit LaneGuard(ffm_ldu *a) : A(a), scal_d(a->ctx->scal_d), scal_h(a->ctx->scal_h), diag(a->diag), rD(a->rD), rDKind(a->rDKind), rDEpoch(a->rDEpoch) {}
    void use(const Lane &l) { A->ctx->scal_d = l.scal_d; A->ctx->scal_h = l.scal_h; A->diag = const_cast<double *>(l.diag); A->rD = l.rD; }
    ~LaneGuard() { A->ctx->scal_d = scal_d; A->ctx->scal_h = scal_h; A->diag = diag; A->rD = rD; A->rDKind = -1; A->rDEpoch = ~0ul; }
};
}
static int read_lane_scalars(ffm_ctx *c, Lane *L, int n)
{
    for (int i = 0; i < n; i++) if (L[i].active)
        FFM_HIP(hipMemcpyAsync(L[i].scal_h, L[i].scal_d, sizeof(double) * NSCAL, hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}
static int precond_lanes(ffm_ldu *A, int precond, Lane *L, int n, double *Lane::*in, double *Lane::*out, LaneGuard &G)
{
    const double *rD[FFM_TILE_MAXSYS], *r[FFM_TILE_MAXSYS]; double *w[FFM_TILE_MAXSYS]; int m = 0, only = -1;
    for (int i = 0; i < n; i++) if (L[i].active) { rD[m] = L[i].rD; r[m] = L[i].*in; w[m] = L[i].*out; m++; only = i; }
    if (m >= 2) return ffm_tile_precond_multi(A, precond, m, rD, r, w);
    if (m == 1) { G.use(L[only]); return ffm_precond_apply_i(A, precond, false, L[only].*in, L[only].*out); }
    return FFM_OK;
}
// out = A in for the lanes still iterating: one pass over the coefficients for all of them (ffm_ldu.hip: k_rows_m)
static int amul_lanes(ffm_ldu *A, Lane *L, int n, double *Lane::*in, double *Lane::*out, LaneGuard &G)
{
    const double *dg[FFM_TILE_MAXSYS], *x[FFM_TILE_MAXSYS]; double *y[FFM_TILE_MAXSYS]; int m = 0, only = -1;
    for (int i = 0; i < n; i++) if (L[i].active) { dg[m] = L[i].diag; x[m] = L[i].*in; y[m] = L[i].*out; m++; only = i; }
    if (m >= 2) return ffm_k_spmv_multi(A, m, dg, x, y, nullptr);
    if (m == 1) { G.use(L[only]); return ffm_k_spmv(A, L[only].*in, L[only].*out, false); }
    return FFM_OK;
}
static int pbicgstab_multi(ffm_ldu *A, int precond, const Controls &k, int n, Lane *L)
{
    ffm_ctx *c = A->ctx; hipStream_t s = c->stream; const long N = A->nOwned; const int g = sgrid(N);
    LaneGuard G(A);
    {   // wA = A psi and sumA of every system in one pass
        const double *dg[FFM_TILE_MAXSYS], *x[FFM_TILE_MAXSYS]; double *y[FFM_TILE_MAXSYS], *sm[FFM_TILE_MAXSYS];
        for (int i = 0; i < n; i++) { dg[i] = L[i].diag; x[i] = L[i].psi; y[i] = L[i].yA; sm[i] = L[i].pA; }
        FFM_TRY(ffm_k_spmv_multi(A, n, dg, x, y, sm));
    }
    for (int i = 0; i < n; i++) {
        Lane &l = L[i]; G.use(l);
        FFM_TRY(scalar_op(c, OP_RESET));
        hipLaunchKernelGGL(k_sub, dim3(g), dim3(256), 0, s, N, l.rA, l.source, l.yA);
        FFM_TRY(norm_and_initial(A, l.psi, l.source, l.yA, l.pA, l.rA, l.perf));          // (reads this lane's scalars back)
        l.active = k.minIter > 0 || !check_convergence(l.perf, k);
        if (l.active) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, N, l.rA0, l.rA);
    }
    {   // calcReciprocalD of the systems that iterate
        const double *dg[FFM_TILE_MAXSYS]; double *D[FFM_TILE_MAXSYS]; int m = 0, only = -1;
        for (int i = 0; i < n; i++) if (L[i].active) { dg[m] = L[i].diag; D[m] = L[i].rD; m++; only = i; }
        if (m >= 2) {
            FFM_TRY(ffm_tile_calc_rD_multi(A, m, dg, D));
            for (int j = 0; j < m; j++) hipLaunchKernelGGL(k_recip, dim3(g), dim3(256), 0, s, N, D[j], (const double *)D[j]);
        } else if (m == 1) { G.use(L[only]); A->rDKind = -1; FFM_TRY(ffm_precond_setup_i(A, precond)); }
        A->rDKind = precond; A->rDEpoch = A->coeffEpoch;         // every lane's rD is current: the single-lane calls below must not redo it
    }
    for (;;) {
        bool any = false;
        for (int i = 0; i < n; i++) if (L[i].active) {
            Lane &l = L[i]; G.use(l); any = true;
            FFM_TRY(ffm_k_dot(c, l.rA0, l.rA, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BS_RHO, 1, 0.0, l.perf->nIterations));
            hipLaunchKernelGGL(k_bs_p, dim3(g), dim3(256), 0, s, N, l.pA, l.rA, l.AyA, c->scal_d, l.perf->nIterations == 0 ? 1 : 0);
        }
        if (!any) break;
        FFM_TRY(precond_lanes(A, precond, L, n, &Lane::pA, &Lane::yA, G));
        FFM_TRY(amul_lanes(A, L, n, &Lane::yA, &Lane::AyA, G));
        for (int i = 0; i < n; i++) if (L[i].active) {
            Lane &l = L[i]; G.use(l);
            FFM_TRY(ffm_k_dot(c, l.rA0, l.AyA, N, S_TMP0));
            FFM_TRY(finish_dot(c, OP_BS_ALPHA));
            hipLaunchKernelGGL(k_bs_s, dim3(g), dim3(256), 0, s, N, l.sA, l.rA, l.AyA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
        }
        FFM_TRY(read_lane_scalars(c, L, n));
        for (int i = 0; i < n; i++) if (L[i].active) {
            Lane &l = L[i]; G.use(l);
            if (l.scal_h[S_SING] != 0.0) { l.perf->singular = 1; l.active = false; continue; }
            l.perf->finalResidual = l.scal_h[S_RES];
            if (check_convergence(l.perf, k)) {
                hipLaunchKernelGGL(k_axpy_alpha, dim3(g), dim3(256), 0, s, N, l.psi, l.yA, c->scal_d);
                l.perf->nIterations++;
                l.active = false;
            }
        }
        FFM_TRY(precond_lanes(A, precond, L, n, &Lane::sA, &Lane::zA, G));
        FFM_TRY(amul_lanes(A, L, n, &Lane::zA, &Lane::tA, G));
        for (int i = 0; i < n; i++) if (L[i].active) {
            Lane &l = L[i]; G.use(l);
            hipLaunchKernelGGL(k_dot2, dim3(g), dim3(256), 0, s, N, l.tA, l.sA, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0, 2));
            FFM_TRY(finish_dot(c, OP_BS_OMEGA, 2));
            hipLaunchKernelGGL(k_bs_xr, dim3(g), dim3(256), 0, s, N, l.psi, l.rA, l.yA, l.zA, l.sA, l.tA, c->scal_d, c->partials_d);
            FFM_TRY(partial_sum_to(c, g, S_TMP0));
            FFM_TRY(finish_dot(c, OP_RES));
        }
        FFM_TRY(read_lane_scalars(c, L, n));
        for (int i = 0; i < n; i++) if (L[i].active) {
            Lane &l = L[i];
            l.perf->finalResidual = l.scal_h[S_RES];
            l.active = (++l.perf->nIterations < k.maxIter && !check_convergence(l.perf, k)) || l.perf->nIterations < k.minIter;
        }
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_solve_multi_d(ffm_ldu *A, int nSys, int solver, int precond, double tol, double relTol, int minIter, int maxIter,
                                 const double *const *diag_d, const double *upper_d, const double *lower_d, double *const *psi_d,
                                 const double *const *source_d, ffm_perf *out)
{
    if (!A || nSys < 1 || !diag_d || !psi_d || !source_d || !out) { ffm_set_error("ffm_solve_multi_d: null argument"); return FFM_ERR_ARG; }
    for (int i = 0; i < nSys; i++) if (!diag_d[i] || !psi_d[i] || !source_d[i]) { ffm_set_error("ffm_solve_multi_d: null array"); return FFM_ERR_ARG; }
    FFM_HIP(hipSetDevice(A->ctx->device));
    Controls k{tol, relTol, minIter, maxIter, 1};
    const bool batched = nSys >= 2 && nSys <= FFM_TILE_MAXSYS && solver == FFM_PBICGSTAB && (precond == FFM_DILU || precond == FFM_DIC) &&
                         A->identity && A->sweepMode == 2 && ffm_tile_multi_usable(A) && A->ifaces.empty();
    if (!batched) {          // one after the other (any solver, any matrix)
        for (int i = 0; i < nSys; i++) {
            FFM_TRY(ffm_ldu_bind_coeffs_native_d(A, diag_d[i], upper_d, lower_d, i > 0));
            FFM_TRY(ffm_solve_d(A, solver, precond, tol, relTol, minIter, maxIter, 1, psi_d[i], source_d[i], &out[i]));
        }
        return FFM_OK;
    }
    FFM_TRY(ffm_ldu_bind_coeffs_native_d(A, diag_d[0], upper_d, lower_d, 0));
    if (precond == FFM_DIC && !A->symmetric) { ffm_set_error("DIC needs a symmetric matrix"); return FFM_ERR_UNSUPPORTED; }
    ffm_ctx *c = A->ctx;
    if (!c->multiScal_d) {
        FFM_HIP(hipMalloc((void **)&c->multiScal_d, sizeof(double) * NSCAL * FFM_TILE_MAXSYS));
        FFM_HIP(hipHostMalloc((void **)&c->multiScal_h, sizeof(double) * NSCAL * FFM_TILE_MAXSYS, hipHostMallocDefault));
        FFM_HIP(hipMemsetAsync(c->multiScal_d, 0, sizeof(double) * NSCAL * FFM_TILE_MAXSYS, c->stream));
    }
    Lane L[FFM_TILE_MAXSYS];
    for (int i = 0; i < nSys; i++) {
        Lane &l = L[i];
        memset(&out[i], 0, sizeof(ffm_perf));
        l.diag = diag_d[i]; l.source = source_d[i]; l.psi = psi_d[i]; l.perf = &out[i]; l.active = false;
        l.scal_d = c->multiScal_d + (size_t)i * NSCAL; l.scal_h = c->multiScal_h + (size_t)i * NSCAL;
        double **w[9] = {&l.rD, &l.pA, &l.yA, &l.rA, &l.AyA, &l.sA, &l.zA, &l.tA, &l.rA0};
        if (i == 0) { l.rD = A->rD; for (int j = 1; j <= 8; j++) FFM_TRY(ffm_ldu_work(A, j, w[j])); }
        else for (int j = 0; j < 9; j++) FFM_TRY(ffm_ldu_work(A, 32 + 9 * (i - 1) + j, w[j]));
    }
    FFM_TRY(pbicgstab_multi(A, precond, k, nSys, L));
    FFM_HIP(hipStreamSynchronize(c->stream));
    FFM_TRY(ffm_tile_check_abort(A));
    return FFM_OK;
}

extern "C" int ffm_solve(ffm_ldu *A, int solver, int precond, double tol, double relTol, int minIter, int maxIter,
                         int nSweeps, double *psi, const double *source, ffm_perf *out)
{
    if (!A || !psi || !source || !out) return FFM_ERR_ARG;
    double *p = nullptr, *b = nullptr;
    const size_t nb = sizeof(double) * (size_t)std::max(A->nCells, 1);
    FFM_HIP(hipMalloc((void **)&p, nb)); FFM_HIP(hipMalloc((void **)&b, nb));
    FFM_TRY(ffm_h2d(A->ctx, p, psi, sizeof(double) * A->nOwned));
    FFM_TRY(ffm_h2d(A->ctx, b, source, sizeof(double) * A->nOwned));
    int rc = ffm_solve_d(A, solver, precond, tol, relTol, minIter, maxIter, nSweeps, p, b, out);
    if (!rc) rc = ffm_d2h(A->ctx, psi, p, sizeof(double) * A->nOwned);
    hipFree(p); hipFree(b);
    return rc;
}

// ------------------------------------------- preconditioner entry (tests) ---
extern "C" int ffm_precond_setup(ffm_ldu *A, int precond, double *rD_out_d)
{
    if (!A) return FFM_ERR_ARG;
    FFM_TRY(ffm_precond_setup_i(A, precond));
    if (rD_out_d) FFM_TRY(ffm_from_internal(A, A->rD, rD_out_d));
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (A->sweepMode == 2) FFM_TRY(ffm_tile_check_abort(A)); else FFM_TRY(ffm_flow_check_abort(A));
    return FFM_OK;
}

// timing helper for bench.py: `reps` preconditioner applications (DIC: a forward and a backward sweep each) on the matrix bound
// to the LDU, bracketed by HIP events on the context stream; average time of ONE application in milliseconds
extern "C" int ffm_bench_precond(ffm_ldu *A, int precond, const double *r_d, double *w_d, int reps, double *avg_ms)
{
    if (!A || !r_d || !w_d || reps < 1 || !avg_ms) return FFM_ERR_ARG;
    if (!A->identity) { ffm_set_error("ffm_bench_precond needs an LDU in the library's cell order"); return FFM_ERR_UNSUPPORTED; }
    FFM_TRY(ffm_precond_setup_i(A, precond));
    FFM_TRY(ffm_precond_apply_i(A, precond, false, r_d, w_d));      // warm-up
    hipEvent_t e0, e1;
    FFM_HIP(hipEventCreate(&e0)); FFM_HIP(hipEventCreate(&e1));
    FFM_HIP(hipEventRecord(e0, A->ctx->stream));
    for (int i = 0; i < reps; i++) FFM_TRY(ffm_precond_apply_i(A, precond, false, r_d, w_d));
    FFM_HIP(hipEventRecord(e1, A->ctx->stream));
    FFM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FFM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return FFM_OK;
}

extern "C" int ffm_precond_apply(ffm_ldu *A, int precond, int transpose, const double *r_d, double *w_d)
{
    if (!A || !r_d || !w_d) return FFM_ERR_ARG;
    FFM_TRY(ffm_precond_setup_i(A, precond));
    const double *ri; FFM_TRY(ffm_to_internal(A, r_d, 0, &ri));
    if (A->identity) { FFM_TRY(ffm_precond_apply_i(A, precond, transpose != 0, ri, w_d)); }
    else {
        double *wi; FFM_TRY(ffm_ldu_work(A, 0, &wi));
        FFM_TRY(ffm_precond_apply_i(A, precond, transpose != 0, ri, wi));
        FFM_TRY(ffm_from_internal(A, wi, w_d));
    }
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (A->sweepMode == 2) FFM_TRY(ffm_tile_check_abort(A)); else FFM_TRY(ffm_flow_check_abort(A));
    return FFM_OK;
}

extern "C" int ffm_gs_smooth(ffm_ldu *A, int symmetric_sweep, int nSweeps, double *psi_d, const double *b_d)
{
    if (!A || !psi_d || !b_d || nSweeps < 1) return FFM_ERR_ARG;
    if (A->identity) { FFM_TRY(ffm_gs_smooth_i(A, symmetric_sweep != 0, nSweeps, psi_d, b_d)); }
    else {
        const double *bi, *pin;
        FFM_TRY(ffm_to_internal(A, b_d, 1, &bi)); FFM_TRY(ffm_to_internal(A, psi_d, 2, &pin));
        FFM_TRY(ffm_gs_smooth_i(A, symmetric_sweep != 0, nSweeps, A->permIn[2], bi));
        FFM_TRY(ffm_from_internal(A, A->permIn[2], psi_d));
    }
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (A->sweepMode == 2) FFM_TRY(ffm_tile_check_abort(A)); else FFM_TRY(ffm_flow_check_abort(A));
    return FFM_OK;
}
