// ffm_pipe.hip -- pipelined, persistent DIC / DILU / Gauss-Seidel sweeps.
//
// Same operators as the level-scheduled sweeps in ffm_solve.hip (exact face-order semantics of
// OpenFOAM-dev's DICPreconditioner / DILUPreconditioner / (sym)GaussSeidelSmoother, reference
// selection cases/steckler/system/fvSolution:21-61), but without one kernel launch per dependency
// level: a hex box of n^3 cells has 3n-2 levels, and at ~3.6 us per dependent launch the sweeps of
// one preconditioner application cost 8.6 ms at 400^3 while their HBM traffic is worth ~2 ms.
//
// Scheme.  The owned cells are split into G groups = contiguous chunks of the caller's (topological)
// cell order, and numbered group-major / level-major inside a group (ffm_ldu.hip:analyse).  ONE
// workgroup sweeps ONE group level by level.  Cross-group dependencies always point from a lower to
// a higher group (forward sweep; reversed for the backward sweep), so a workgroup only ever waits
// for groups that took their ticket before it did: groups are handed out by an atomic ticket, hence
// every group a workgroup waits for is already running or finished and the pipeline cannot deadlock,
// whatever the dispatch order or residency (MI355X_MICROARCH "Workgroup dispatch").
//   * inside a group the values of the previous levels are exchanged through an LDS ring;
//   * between groups through global memory: every w store and every cross-group w load is an
//     agent-scope (sc1) access, a workgroup publishes "all my levels < L are complete" in a 64-bit
//     progress word after every wave has drained its stores (s_waitcnt vmcnt(0)) and the workgroup
//     has met at a barrier, and a consumer polls that word with sc1 loads from a few lanes, then a
//     barrier, then loads (the hand-off form measured valid on gfx950: one flag lane per storing
//     workgroup, sc1 data both sides);
//   * progress words carry the launch epoch (ticket / G), so nothing is reset between launches;
//   * every spin is bounded; on a time-out the kernel raises an abort word that all pollers watch,
//     and the host turns it into FFM_ERR_HIP (no silent fallback).
// Rows are accumulated in the reference's face order with FMA contraction off, so results are
// bitwise those of the level-scheduled kernels and of the serial CPU loops.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>

constexpr int RING = 4096;           // doubles per workgroup ring (32 KiB of LDS)
constexpr int PIPE_THREADS = 256;
constexpr unsigned SPIN_LIMIT = 1u << 23;

struct PipeView {
    int G, nOwn;
    const int *grpCell, *entStart, *entLevel, *entCell, *predStart, *preds, *bwdCells;
    unsigned long long *progress;
    unsigned int *ticket;            // [0] ticket counter, [1] abort word
    int useRing;
};

enum { PF_PRECOND = 0, PF_RD = 1, PF_GS = 2 };

__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wait until every predecessor group has published (epoch, >= level).  Lane t polls predecessor t with agent-scope
// loads and remembers the last value it saw, so a predecessor that is ahead costs no memory access.  Bounded: after
// SPIN_LIMIT polls the abort word is raised; every poller watches it.
__device__ __forceinline__ void wait_preds(const PipeView &pl, int p0, int p1, unsigned long long need, unsigned long long &seen,
                                           int *shAbort)
{
    for (int q = p0 + (int)threadIdx.x; q < p1; q += blockDim.x) {
        const bool cached = (p1 - p0) <= (int)blockDim.x;       // one predecessor per lane: `seen` is that predecessor's value
        unsigned long long val = cached ? seen : 0ull;
        if (val < need) {
            const unsigned long long *addr = &pl.progress[pl.preds[q]];
            unsigned spins = 0;
            val = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (val < need) {
                __builtin_amdgcn_s_sleep(16);
                if ((++spins & 255u) == 0u) {
                    if (__hip_atomic_load(&pl.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *shAbort = 1; break; }
                    if (spins > SPIN_LIMIT) {
                        __hip_atomic_store(&pl.ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *shAbort = 1; break;
                    }
                }
                val = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (cached) seen = val;
        }
    }
}

// ------------------------------------------------------------------ forward ---
// PF_PRECOND: w[c] = rD[c]*r[c] - sum_k rD[c]*cA[f]*w[l]                      (DIC/DILU forward)
// PF_RD     : w[c] = dg[c] - sum_k cA[f]*cB[f]/w[l]                           (calcReciprocalD, not yet inverted)
// PF_GS     : v = r[c] - sum_k cA[f]*w[l]; w2[c] = v; v -= sum_upper cB[e]*w_old[u]; w[c] = v/dg[c]
template <int W, int OP>
__global__ __launch_bounds__(PIPE_THREADS) void k_pipe_fwd(LduView v, PipeView pl, const double *__restrict__ cA,
                                                          const double *__restrict__ cB, const double *__restrict__ dg,
                                                          const double *__restrict__ rD, const double *__restrict__ r,
                                                          double *w, double *__restrict__ w2)
{
    __shared__ double ring[RING];
    __shared__ int shG, shAbort;
    __shared__ unsigned long long shEpoch;
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&pl.ticket[0], 1u);
        shG = (int)(t % (unsigned)pl.G); shEpoch = (unsigned long long)(t / (unsigned)pl.G + 1u) << 32; shAbort = 0;
    }
    __syncthreads();
    const int g = shG;
    const unsigned long long epoch = shEpoch;
    const int gs = pl.grpCell[g];
    const int e0 = pl.entStart[g], e1 = pl.entStart[g + 1];
    const int p0 = pl.predStart[g], p1 = pl.predStart[g + 1];
    unsigned long long seen = 0;
    for (int e = e0; e < e1; e++) {
        const int L = pl.entLevel[e], cs = pl.entCell[e], ce = pl.entCell[e + 1];
        // levels < L of every predecessor must be complete and visible
        wait_preds(pl, p0, p1, epoch | (unsigned long long)(unsigned)L, seen, &shAbort);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores of the previous entry have been performed
        __syncthreads();                                      // ... every wave's; ring of the previous entry is visible
        if (shAbort) return;
        if (threadIdx.x == 0)
            __hip_atomic_store(&pl.progress[g], epoch | (unsigned long long)(unsigned)L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ringLo = ce - RING;                         // ring slots of cells >= ringLo survive this entry's writes
        for (int c = cs + (int)threadIdx.x; c < ce; c += blockDim.x) {
            RowEnt<W> Lw; load_lower<W>(v, c, Lw);
            double a[W], b[W], x[W];
#pragma unroll
            for (int s = 0; s < W; s++) {
                a[s] = cA[Lw.f[s]];
                b[s] = (OP == PF_RD) ? cB[Lw.f[s]] : 0.0;
                const int nb = Lw.nb[s];
                if (!Lw.on[s]) x[s] = 1.0;
                else if (pl.useRing && nb >= gs && nb >= ringLo) x[s] = ring[(nb - gs) & (RING - 1)];
                else x[s] = ld_agent(&w[nb]);
            }
            double val;
            if (OP == PF_PRECOND) {
                const double rd = rD[c];
                val = rd * r[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (Lw.on[s]) val -= rd * a[s] * x[s];
            } else if (OP == PF_RD) {
                val = dg[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (Lw.on[s]) val -= a[s] * b[s] / x[s];
            } else {
                val = r[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (Lw.on[s]) val -= a[s] * x[s];
                w2[c] = val;
                RowEnt<W> Uw; load_upper<W>(v, c, Uw);
                double au[W], xu[W];
#pragma unroll
                for (int s = 0; s < W; s++) { au[s] = cB[Uw.f[s]]; xu[s] = Uw.on[s] ? w[Uw.nb[s]] : 0.0; }   // old values
#pragma unroll
                for (int s = 0; s < W; s++) if (Uw.on[s]) val -= au[s] * xu[s];
                val = val / dg[c];
            }
            st_agent(&w[c], val);
            if (pl.useRing) ring[(c - gs) & (RING - 1)] = val;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(&pl.progress[g], epoch | 0x7fffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ backward ---
// PF_PRECOND: w[c] -= sum rD[c]*cA[e]*w[u], faces of c in descending order, ghost neighbours ignored
// PF_GS     : v = w2[c] - sum cA[e]*w[u] (ascending, ghost neighbours with their lagged value); w[c] = v/dg[c]
template <int W, int OP>
__global__ __launch_bounds__(PIPE_THREADS) void k_pipe_bwd(LduView v, PipeView pl, const double *__restrict__ cA,
                                                          const double *__restrict__ dg, const double *__restrict__ rD,
                                                          double *w, const double *__restrict__ w2)
{
    __shared__ double ring[RING];
    __shared__ int shG, shAbort;
    __shared__ unsigned long long shEpoch;
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&pl.ticket[0], 1u);
        shG = pl.G - 1 - (int)(t % (unsigned)pl.G); shEpoch = (unsigned long long)(t / (unsigned)pl.G + 1u) << 32; shAbort = 0;
    }
    __syncthreads();
    const int g = shG;
    const unsigned long long epoch = shEpoch;
    const int gs = pl.grpCell[g], ge = pl.grpCell[g + 1];
    const int e0 = pl.entStart[g], e1 = pl.entStart[g + 1];
    const int p0 = pl.predStart[g], p1 = pl.predStart[g + 1];
    unsigned long long seen = 0;
    for (int e = e0; e < e1; e++) {
        const int L = pl.entLevel[e], q0 = pl.entCell[e], q1 = pl.entCell[e + 1];   // positions into bwdCells
        wait_preds(pl, p0, p1, epoch | (unsigned long long)(unsigned)L, seen, &shAbort);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (shAbort) return;
        if (threadIdx.x == 0)
            __hip_atomic_store(&pl.progress[g], epoch | (unsigned long long)(unsigned)L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // with the reverse order, position q holds cell ge-1-(q-gs): this entry writes cells [cLo, cHi]
        const int cLo = ge - 1 - (q1 - 1 - gs);
        const int ringHi = cLo + RING;                        // ring slots of cells < ringHi survive this entry's writes
        for (int q = q0 + (int)threadIdx.x; q < q1; q += blockDim.x) {
            const int c = pl.bwdCells[q];
            RowEnt<W> Uw;
            if (OP == PF_PRECOND) load_upper<W, true>(v, c, Uw); else load_upper<W, false>(v, c, Uw);
            double a[W], x[W];
#pragma unroll
            for (int s = 0; s < W; s++) {
                a[s] = cA[Uw.f[s]];
                const int nb = Uw.nb[s];
                if (!Uw.on[s]) x[s] = 0.0;
                else if (pl.useRing && nb < ge && nb < ringHi) x[s] = ring[(nb - gs) & (RING - 1)];
                else x[s] = ld_agent(&w[nb]);
            }
            double val;
            if (OP == PF_PRECOND) {
                const double rd = rD[c];
                val = w[c];
#pragma unroll
                for (int s = W - 1; s >= 0; s--) if (Uw.on[s]) val -= rd * a[s] * x[s];
            } else {
                val = w2[c];
#pragma unroll
                for (int s = 0; s < W; s++) if (Uw.on[s]) val -= a[s] * x[s];
                val = val / dg[c];
            }
            st_agent(&w[c], val);
            if (pl.useRing) ring[(c - gs) & (RING - 1)] = val;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(&pl.progress[g], epoch | 0x7fffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ host side ---
static PipeView fwd_view(const ffm_ldu *A)
{
    PipeView p; p.G = A->nGroups; p.nOwn = A->nOwned; p.grpCell = A->grpCell; p.entStart = A->fEntStart; p.entLevel = A->fEntLevel;
    p.entCell = A->fEntCell; p.predStart = A->fPredStart; p.preds = A->fPreds; p.bwdCells = A->bwdCells;
    p.progress = A->pipeProgress; p.ticket = A->pipeTicket; p.useRing = 1;
    return p;
}
static PipeView bwd_view(const ffm_ldu *A)
{
    PipeView p = fwd_view(A);
    p.entStart = A->bEntStart; p.entLevel = A->bEntLevel; p.entCell = A->bEntPos; p.predStart = A->bPredStart; p.preds = A->bPreds;
    p.useRing = A->bwdIsReverse ? 1 : 0;
    return p;
}

int ffm_pipe_check_abort(ffm_ldu *A)
{
    unsigned int h[2] = {0, 0};
    FFM_HIP(hipMemcpyAsync(h, A->pipeTicket, sizeof(h), hipMemcpyDeviceToHost, A->ctx->stream));
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (h[1]) {
        ffm_set_error("pipelined sweep timed out waiting for a predecessor group (abort word set)");
        unsigned int z = 0;
        hipMemcpy(A->pipeTicket + 1, &z, sizeof(z), hipMemcpyHostToDevice);
        return FFM_ERR_HIP;
    }
    return FFM_OK;
}

#define PIPE_LAUNCH(kern, ...)                                                                                          \
    do {                                                                                                                \
        if (A->nGroups > 0) hipLaunchKernelGGL(kern, dim3(A->nGroups), dim3(PIPE_THREADS), 0, A->ctx->stream, __VA_ARGS__); \
    } while (0)

// rD = diag - sum upper*lower/rD[l] (un-inverted; the caller inverts)
int ffm_pipe_calc_rD(ffm_ldu *A)
{
    FFM_DISPATCH_W(A->maxW, PIPE_LAUNCH((k_pipe_fwd<W, PF_RD>), ffm_view(A), fwd_view(A), A->upper, A->lower, A->diag,
                                        (const double *)nullptr, (const double *)nullptr, A->rD, (double *)nullptr));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_pipe_precond(ffm_ldu *A, const double *cf, const double *cb, const double *r, double *w)
{
    FFM_DISPATCH_W(A->maxW, PIPE_LAUNCH((k_pipe_fwd<W, PF_PRECOND>), ffm_view(A), fwd_view(A), cf, (const double *)nullptr,
                                        (const double *)nullptr, A->rD, r, w, (double *)nullptr));
    FFM_DISPATCH_W(A->maxW, PIPE_LAUNCH((k_pipe_bwd<W, PF_PRECOND>), ffm_view(A), bwd_view(A), cb, (const double *)nullptr, A->rD, w,
                                        (const double *)nullptr));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_pipe_gs(ffm_ldu *A, bool sym, double *psi, const double *bP, double *bSave)
{
    FFM_DISPATCH_W(A->maxW, PIPE_LAUNCH((k_pipe_fwd<W, PF_GS>), ffm_view(A), fwd_view(A), A->lower, A->upper, A->diag,
                                        (const double *)nullptr, bP, psi, bSave));
    if (sym)
        FFM_DISPATCH_W(A->maxW, PIPE_LAUNCH((k_pipe_bwd<W, PF_GS>), ffm_view(A), bwd_view(A), A->upper, A->diag, (const double *)nullptr,
                                            psi, (const double *)bSave));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
