// ffm_tile.hip -- tiled wavefront sweeps for DIC / DILU (forward + backward substitution).
//
// Same operators, bitwise, as the level-scheduled kernels of ffm_solve.hip (OpenFOAM-dev DICPreconditioner /
// DILUPreconditioner::precondition, reference selection cases/steckler/system/fvSolution:21-46).  Motivation and
// measurements: a hand-off between workgroups costs about as much as a dependent kernel launch on MI355X (3-5 us under
// load), so neither one launch per dependency level (3n-2 levels for an n^3 box) nor a persistent kernel that hands off
// once per level (ffm_pipe.hip) can beat ~3.6 us per level.  This kernel needs a hand-off only once per BATCH of levels:
//   * cells are split into groups (2-D tiles of cell columns when the host gives a hint, chunks of the cell order
//     otherwise) whose dependency graph is acyclic; one workgroup sweeps one group, level by level;
//   * inside a group, the values of earlier levels are exchanged through an LDS ring: one barrier per level;
//   * values owned by other groups (tile edges) are fetched once per batch of up to 16 levels into an LDS halo buffer,
//     after the producing groups have published that those levels are complete; a consumer therefore runs a batch behind
//     its producers and polls / synchronises once per batch instead of once per level;
//   * groups are handed out by an atomic ticket in topological order, so a workgroup only waits for groups that are
//     already running (no residency assumption, no deadlock); every spin is bounded and raises the abort word.
// Data hand-off uses agent-scope (sc1) stores and loads for w, a drained wave (s_waitcnt vmcnt(0)) + workgroup barrier
// before the single-lane progress store, and a barrier between the poll and the halo loads (MI355X_MICROARCH,
// "Valid forms").
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <climits>

constexpr int T_RING = 4096;        // doubles
constexpr int T_HALO = 1024;        // doubles
constexpr int T_ENT = 256;          // max cells per entry = workgroup size
constexpr int T_KB = 16;            // max entries (levels) per batch
constexpr int T_THREADS = 256;
constexpr unsigned T_SPIN_LIMIT = 1u << 22;

struct TileDir {            // one sweep direction (device arrays)
    int nEnt = 0, nBat = 0;
    int *grpBat = nullptr;      // [G+1] batches of each group
    int *batEnt = nullptr;      // [nBat+1] entries of each batch
    int *batNeed = nullptr;     // [nBat] progress every predecessor must have published
    int *batPub = nullptr;      // [nBat] progress to publish after the batch
    int *batHalo = nullptr;     // [nBat+1] range into haloCells
    int *entCell = nullptr;     // [nEnt+1] first cell of each entry (forward numbering); backward: entCell[e+1] .. entCell[e] descending
    int *haloCells = nullptr;
    int *ref = nullptr;         // per lower entry / upper slot: >= 0 cell id (ring), <= -2 halo slot, -1 none
};

struct ffm_tile_plan {
    bool usable = false;
    int G = 0;
    TileDir f, b;
    double *loCoefU = nullptr, *loCoefL = nullptr;     // coefficients in lower-entry layout
    unsigned long epochU = ~0ul, epochL = ~0ul;
};

static void free_dir(TileDir &d)
{
    hipFree(d.grpBat); hipFree(d.batEnt); hipFree(d.batNeed); hipFree(d.batPub); hipFree(d.batHalo); hipFree(d.entCell);
    hipFree(d.haloCells); hipFree(d.ref);
    d = TileDir();
}
void ffm_tile_free(ffm_ldu *A)
{
    if (!A->tile) return;
    free_dir(A->tile->f); free_dir(A->tile->b);
    hipFree(A->tile->loCoefU); hipFree(A->tile->loCoefL);
    delete A->tile; A->tile = nullptr;
}
bool ffm_tile_usable(const ffm_ldu *A) { return A->tile && A->tile->usable; }

template <class T> static int upv(T **d, const std::vector<T> &v)
{
    FFM_HIP(hipMalloc((void **)d, sizeof(T) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) FFM_HIP(hipMemcpy(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return FFM_OK;
}

// Build one direction.  fwd: neighbours = lower entries (cells with smaller index); bwd: upper slots.
static int build_dir(ffm_ldu *A, bool fwd, const std::vector<int> &lvl, const std::vector<int> &grpCell,
                     const std::vector<int> &grpOfCell, TileDir &D, bool &ok)
{
    const int G = (int)grpCell.size() - 1, nOwn = A->nOwned;
    const std::vector<int> &off = fwd ? A->h_loOff : A->h_upOff;
    const std::vector<int> &ent = fwd ? A->h_loEnt : A->h_upNbr;
    std::vector<int> ref(ent.size(), -1), grpBat(G + 1, 0), batEnt(1, 0), batNeed, batPub, batHalo(1, 0), entCell, haloCells;
    std::vector<int> entLevel;
    std::vector<int> stamp(nOwn, -1), slotOf(nOwn, 0);
    int batchId = 0;
    for (int g = 0; g < G; g++) {
        const int gs = grpCell[g], ge = grpCell[g + 1];
        // entries: runs of equal level, at most T_ENT cells; forward ascending cells, backward descending
        std::vector<std::pair<int, int>> ents;      // [first, last) in processing order (backward: first > last side handled below)
        if (fwd) {
            for (int c = gs; c < ge;) { int e = c; while (e < ge && lvl[e] == lvl[c] && e - c < T_ENT) e++; ents.emplace_back(c, e); c = e; }
        } else {
            for (int c = ge; c > gs;) { int e = c; while (e > gs && lvl[e - 1] == lvl[c - 1] && c - e < T_ENT) e--; ents.emplace_back(e, c); c = e; }
        }
        // levels must be non-decreasing along the processing order (true for the forward order by construction; for the
        // backward order only when the backward levels decrease with the forward order inside the group)
        for (size_t i = 1; i < ents.size(); i++) if (lvl[ents[i].first] < lvl[ents[i - 1].first]) { ok = false; return FFM_OK; }
        size_t i = 0;
        while (i < ents.size()) {
            // open a batch
            int nE = 0, nCells = 0;
            std::vector<int> halo;
            const size_t iStart = i;
            while (i < ents.size() && nE < T_KB) {
                const int c0 = ents[i].first, c1 = ents[i].second;
                if (nE > 0 && nCells + (c1 - c0) > T_RING / 2 - T_ENT) break;
                // externals of this entry not yet in the batch halo
                std::vector<int> add;
                for (int c = c0; c < c1; c++) {
                    const int sl = c >> 6, lane = c & 63, wdt = (off[sl + 1] - off[sl]) / 64;
                    for (int s = 0; s < wdt; s++) {
                        const int q = off[sl] + s * 64 + lane, e = ent[q];
                        if (e < 0) continue;
                        const int nb = fwd ? (e >> 4) : e;
                        if (nb >= nOwn) continue;                                   // ghost: ignored by the block-Jacobi sweeps
                        const bool inRing = grpOfCell[nb] == g && std::abs(c - nb) <= T_RING - 2 * T_ENT;
                        if (!inRing && stamp[nb] != batchId) { stamp[nb] = batchId; slotOf[nb] = -1; add.push_back(nb); }
                    }
                }
                if (nE > 0 && halo.size() + add.size() > (size_t)T_HALO) { for (int nb : add) stamp[nb] = -1; break; }
                if (add.size() > (size_t)T_HALO) { ok = false; return FFM_OK; }
                for (int nb : add) { slotOf[nb] = (int)halo.size(); halo.push_back(nb); }
                nE++; nCells += c1 - c0; i++;
            }
            // assign references of the batch's cells
            int maxNbLevelPlus1 = 0;
            for (size_t k = iStart; k < i; k++) {
                for (int c = ents[k].first; c < ents[k].second; c++) {
                    const int sl = c >> 6, lane = c & 63, wdt = (off[sl + 1] - off[sl]) / 64;
                    for (int s = 0; s < wdt; s++) {
                        const int q = off[sl] + s * 64 + lane, e = ent[q];
                        if (e < 0) continue;
                        const int nb = fwd ? (e >> 4) : e;
                        if (nb >= nOwn) continue;
                        if (stamp[nb] == batchId && slotOf[nb] >= 0) {
                            ref[q] = -2 - slotOf[nb];
                            if (grpOfCell[nb] != g) maxNbLevelPlus1 = std::max(maxNbLevelPlus1, lvl[nb] + 1);
                            else if (nb >= ents[iStart].first && nb < ents[i - 1].second && fwd) { ok = false; return FFM_OK; }   // far ref inside the batch
                        } else ref[q] = nb;
                    }
                }
                entCell.push_back(fwd ? ents[k].first : ents[k].second);   // backward: entry covers [entCell[e+1]', ...) see kernel
                entLevel.push_back(lvl[ents[k].first]);
            }
            batEnt.push_back((int)entCell.size());
            batNeed.push_back(maxNbLevelPlus1);          // predecessors must have completed every level < this
            batPub.push_back(i < ents.size() ? lvl[ents[i].first] : INT_MAX);
            haloCells.insert(haloCells.end(), halo.begin(), halo.end());
            batHalo.push_back((int)haloCells.size());
            batchId++;
        }
        grpBat[g + 1] = (int)batNeed.size();
        // sentinel for the last entry of the group
        if (fwd) { /* entCell[e+1] of the last entry must be ge */ }
    }
    // entry bounds: store both ends explicitly (2 ints per entry) to keep forward/backward uniform
    // rebuild entCell as [first,last) pairs
    {
        std::vector<int> pairs;
        pairs.reserve(entCell.size() * 2);
        size_t eidx = 0;
        for (int g = 0; g < G; g++) {
            const int gs = grpCell[g], ge = grpCell[g + 1];
            if (fwd) {
                for (int c = gs; c < ge;) { int e = c; while (e < ge && lvl[e] == lvl[c] && e - c < T_ENT) e++; pairs.push_back(c); pairs.push_back(e); c = e; eidx++; }
            } else {
                for (int c = ge; c > gs;) { int e = c; while (e > gs && lvl[e - 1] == lvl[c - 1] && c - e < T_ENT) e--; pairs.push_back(e); pairs.push_back(c); c = e; eidx++; }
            }
        }
        if (eidx != entCell.size()) { ffm_set_error("internal: tile plan entry mismatch"); return FFM_ERR_ADDR; }
        entCell.swap(pairs);
    }
    D.nEnt = (int)entLevel.size(); D.nBat = (int)batNeed.size();
    FFM_TRY(upv(&D.grpBat, grpBat)); FFM_TRY(upv(&D.batEnt, batEnt)); FFM_TRY(upv(&D.batNeed, batNeed)); FFM_TRY(upv(&D.batPub, batPub));
    FFM_TRY(upv(&D.batHalo, batHalo)); FFM_TRY(upv(&D.entCell, entCell)); FFM_TRY(upv(&D.haloCells, haloCells)); FFM_TRY(upv(&D.ref, ref));
    return FFM_OK;
}

int ffm_tile_build(ffm_ldu *A, const std::vector<int> &l, const std::vector<int> &u, const std::vector<int> &lev,
                   const std::vector<int> &bl, const std::vector<int> &grpCell)
{
    (void)l; (void)u;
    A->tile = new ffm_tile_plan();
    ffm_tile_plan *T = A->tile;
    T->G = (int)grpCell.size() - 1;
    if (A->maxW > 4 || T->G <= 0) return FFM_OK;       // not usable: the caller falls back to ffm_pipe
    std::vector<int> grpOfCell(A->nOwned);
    for (int g = 0; g < T->G; g++) for (int c = grpCell[g]; c < grpCell[g + 1]; c++) grpOfCell[c] = g;
    bool ok = true;
    FFM_TRY(build_dir(A, true, lev, grpCell, grpOfCell, T->f, ok));
    if (ok) FFM_TRY(build_dir(A, false, bl, grpCell, grpOfCell, T->b, ok));
    T->usable = ok;
    if (ok) {
        FFM_HIP(hipMalloc((void **)&T->loCoefU, sizeof(double) * std::max(A->loTotal, 1)));
        FFM_HIP(hipMalloc((void **)&T->loCoefL, sizeof(double) * std::max(A->loTotal, 1)));
    }
    return FFM_OK;
}

// ------------------------------------------------------------------ device ---
struct TileView {
    int G;
    const int *grpCell, *grpBat, *batEnt, *batNeed, *batPub, *batHalo, *entCell, *haloCells, *ref, *predStart, *preds;
    unsigned long long *progress;
    unsigned int *ticket;
};

__device__ __forceinline__ double t_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void t_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void t_wait_preds(const TileView &t, int p0, int p1, unsigned long long need, unsigned long long &seen, int *shAbort)
{
    for (int q = p0 + (int)threadIdx.x; q < p1; q += blockDim.x) {
        const bool cached = (p1 - p0) <= (int)blockDim.x;
        unsigned long long val = cached ? seen : 0ull;
        if (val < need) {
            const unsigned long long *addr = &t.progress[t.preds[q]];
            unsigned spins = 0;
            val = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (val < need) {
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 255u) == 0u) {
                    if (__hip_atomic_load(&t.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *shAbort = 1; break; }
                    if (spins > T_SPIN_LIMIT) { __hip_atomic_store(&t.ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *shAbort = 1; break; }
                }
                val = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (cached) seen = val;
        }
    }
}

// FWD: w[c] = rD[c]*r[c] - sum_k rD[c]*coef[q]*w[l]   (coef in lower-entry layout)
// BWD: w[c] -= sum_s rD[c]*coef[e]*w[u], slots in descending order (coef in upper-slot layout)
template <int W, bool FWD>
__global__ __launch_bounds__(T_THREADS) void k_tile(LduView v, TileView t, const double *__restrict__ coef, const double *__restrict__ rD,
                                                    const double *__restrict__ r, double *w)
{
    __shared__ double ring[T_RING];
    __shared__ double halo[T_HALO];
    __shared__ int shG, shAbort;
    __shared__ unsigned long long shEpoch;
    if (threadIdx.x == 0) {
        const unsigned tk = atomicAdd(&t.ticket[0], 1u);
        const int k = (int)(tk % (unsigned)t.G);
        shG = FWD ? k : t.G - 1 - k; shEpoch = (unsigned long long)(tk / (unsigned)t.G + 1u) << 32; shAbort = 0;
    }
    __syncthreads();
    const int g = shG;
    const unsigned long long epoch = shEpoch;
    const int gs = t.grpCell[g];
    const int b0 = t.grpBat[g], b1 = t.grpBat[g + 1];
    const int p0 = t.predStart[g], p1 = t.predStart[g + 1];
    unsigned long long seen = 0;
    for (int b = b0; b < b1; b++) {
        const int need = t.batNeed[b];
        if (need > 0) t_wait_preds(t, p0, p1, epoch | (unsigned long long)(unsigned)need, seen, &shAbort);
        __syncthreads();                                    // poll done for everyone; previous batch finished with the halo buffer
        if (shAbort) return;
        const int h0 = t.batHalo[b], h1 = t.batHalo[b + 1];
        for (int i = h0 + (int)threadIdx.x; i < h1; i += blockDim.x) halo[i - h0] = t_ld(&w[t.haloCells[i]]);
        __syncthreads();
        const int e0 = t.batEnt[b], e1 = t.batEnt[b + 1];
        for (int e = e0; e < e1; e++) {
            const int c0 = t.entCell[2 * e], c1 = t.entCell[2 * e + 1];
            const int c = FWD ? c0 + (int)threadIdx.x : c1 - 1 - (int)threadIdx.x;
            if (c >= c0 && c < c1) {
                const int sl = c >> 6, lane = c & 63;
                const int base = FWD ? lo_base(v, sl) : up_base(v, sl);
                const int wdt = FWD ? lo_width(v, sl) : up_width(v, sl);
                int rf[W]; double a[W], x[W];
#pragma unroll
                for (int s = 0; s < W; s++) {
                    const int q = base + s * 64 + lane;
                    rf[s] = (s < wdt) ? t.ref[q] : -1;
                    a[s] = (s < wdt) ? coef[q] : 0.0;
                }
                const double rd = rD[c];
                double val = FWD ? rd * r[c] : w[c];
#pragma unroll
                for (int s = 0; s < W; s++) x[s] = (rf[s] >= 0) ? ring[(rf[s] - gs) & (T_RING - 1)] : (rf[s] <= -2 ? halo[-2 - rf[s]] : 0.0);
                if (FWD) {
#pragma unroll
                    for (int s = 0; s < W; s++) if (rf[s] != -1) val -= rd * a[s] * x[s];
                } else {
#pragma unroll
                    for (int s = W - 1; s >= 0; s--) if (rf[s] != -1) val -= rd * a[s] * x[s];
                }
                t_st(&w[c], val);
                ring[(c - gs) & (T_RING - 1)] = val;
            }
            __syncthreads();                                // one barrier per level: the ring is visible to the next entry
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's stores of the batch have been performed
        __syncthreads();                                    // ... every wave's
        if (threadIdx.x == 0) {
            const int pub = t.batPub[b];
            __hip_atomic_store(&t.progress[g], epoch | (unsigned long long)(unsigned)pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// coefficients in lower-entry layout: out[q] = src[face_of(owner, slot)] for every lower entry q
__global__ void k_gather_lo(LduView v, int loTotal, const double *__restrict__ src, double *__restrict__ out)
{
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < loTotal; q += (long)gridDim.x * blockDim.x) {
        const int e = v.loEnt[q];
        out[q] = (e >= 0) ? src[face_of(v, e >> 4, e & 15)] : 0.0;
    }
}

static TileView tview(const ffm_ldu *A, const TileDir &d, bool fwd)
{
    TileView t; t.G = A->tile->G; t.grpCell = A->grpCell; t.grpBat = d.grpBat; t.batEnt = d.batEnt; t.batNeed = d.batNeed; t.batPub = d.batPub;
    t.batHalo = d.batHalo; t.entCell = d.entCell; t.haloCells = d.haloCells; t.ref = d.ref;
    t.predStart = fwd ? A->fPredStart : A->bPredStart; t.preds = fwd ? A->fPreds : A->bPreds;
    t.progress = A->pipeProgress; t.ticket = A->pipeTicket;
    return t;
}

int ffm_tile_precond(ffm_ldu *A, int precond, bool transpose, const double *r, double *w)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    // forward coefficient: DIC upper; DILU lower; DILU^T upper.  backward: DIC upper; DILU upper; DILU^T lower.
    const bool fwdUpper = (precond == FFM_DIC) || transpose;
    const double *cb = (precond == FFM_DIC) ? A->upper : (transpose ? A->lower : A->upper);
    double *cf = fwdUpper ? T->loCoefU : T->loCoefL;
    unsigned long &ep = fwdUpper ? T->epochU : T->epochL;
    if (ep != A->coeffEpoch) {
        const int g = std::max(1, std::min(ffm_grid(A->loTotal, 256), RED_BLOCKS));
        hipLaunchKernelGGL(k_gather_lo, dim3(g), dim3(256), 0, s, ffm_view(A), A->loTotal, fwdUpper ? A->upper : A->lower, cf);
        ep = A->coeffEpoch;
    }
    hipLaunchKernelGGL((k_tile<4, true>), dim3(T->G), dim3(T_THREADS), 0, s, ffm_view(A), tview(A, T->f, true), (const double *)cf, A->rD, r, w);
    hipLaunchKernelGGL((k_tile<4, false>), dim3(T->G), dim3(T_THREADS), 0, s, ffm_view(A), tview(A, T->b, false), cb, A->rD, r, w);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
