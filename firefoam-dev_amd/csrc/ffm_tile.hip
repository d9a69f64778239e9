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
//     already running (no residency assumption, no deadlock); every spin is bounded and raises the abort word;
//   * everything an entry (= one level of one group, <= 256 cells) needs from memory except neighbour values lives in
//     cell-major streams private to this kernel (16-bit neighbour codes, coefficients gathered once per coefficient
//     update), whose addresses depend on the cell index only, and is fetched T_PF entries ahead into registers: the
//     per-level critical path is LDS reads, a few FMAs, one LDS write and one barrier.
// Data hand-off uses agent-scope (sc1) stores and loads for w, a drained wave (s_waitcnt vmcnt(0)) + workgroup barrier
// before the single-lane progress store, and a barrier between the poll and the halo loads (MI355X_MICROARCH,
// "Valid forms").
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <climits>

constexpr int T_RING = 4096;        // doubles
constexpr int T_HALO = 1024;        // doubles (power of two)
constexpr int T_ENT = 256;          // max cells per entry = workgroup size
constexpr int T_KB = 16;            // max entries (levels) per batch
constexpr int T_THREADS = 256;
constexpr int T_PF = 8;             // entries fetched ahead of the one being computed
constexpr unsigned T_SPIN_LIMIT = 1u << 22;
constexpr unsigned short T_NONE = 0xFFFFu, T_HALOBIT = 0x8000u;
static_assert(T_HALO <= 4 * T_THREADS, "halo prefetch holds 4 cells per thread");
static_assert(T_RING - 2 * T_ENT < T_HALOBIT && T_HALO < T_HALOBIT, "neighbour codes are 15 bit");

struct TileDir {            // one sweep direction (device arrays)
    int nEnt = 0, nBat = 0;
    int *grpBat = nullptr;      // [G+1] batches of each group
    int *batEnt = nullptr;      // [nBat+1] entries of each batch
    int *batNeed = nullptr;     // [nBat] progress every predecessor must have published
    int *batPub = nullptr;      // [nBat] progress to publish after the batch
    int *batHalo = nullptr;     // [nBat+4] range into haloCells
    int *entCell = nullptr;     // [2*nEnt + pad] cell range [first,last) of each entry
    int *haloCells = nullptr;
    unsigned short *code = nullptr;   // [4*nOwn] per cell and slot: ring distance | T_HALOBIT+halo slot | T_NONE
    int *src = nullptr;         // [W*nOwn] native coefficient index of each slot (-1: none)
    double *coefU = nullptr, *coefL = nullptr;        // [W*nOwn] gathered upper / lower coefficients (lazily allocated)
    unsigned long epochU = ~0ul, epochL = ~0ul;
};

struct ffm_tile_plan {
    bool usable = false;
    int G = 0, W = 4;
    TileDir f, b;
};

static void free_dir(TileDir &d)
{
    hipFree(d.grpBat); hipFree(d.batEnt); hipFree(d.batNeed); hipFree(d.batPub); hipFree(d.batHalo); hipFree(d.entCell);
    hipFree(d.haloCells); hipFree(d.code); hipFree(d.src); hipFree(d.coefU); hipFree(d.coefL);
    d = TileDir();
}
void ffm_tile_free(ffm_ldu *A)
{
    if (!A->tile) return;
    free_dir(A->tile->f); free_dir(A->tile->b);
    delete A->tile; A->tile = nullptr;
}
bool ffm_tile_usable(const ffm_ldu *A) { return A->tile && A->tile->usable; }

template <class T> static int upv(T **d, const std::vector<T> &v)
{
    FFM_HIP(hipMalloc((void **)d, sizeof(T) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) FFM_HIP(hipMemcpy(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return FFM_OK;
}

// Build one direction.  fwd: neighbours = lower entries (cells with smaller index); bwd: upper slots.  Neighbours that are
// ghost cells are dropped (block-Jacobi sweeps ignore them); the remaining ones keep their order, which is the order of
// the reference's face loop.
static int build_dir(ffm_ldu *A, bool fwd, int W, const std::vector<int> &lvl, const std::vector<int> &grpCell,
                     const std::vector<int> &grpOfCell, TileDir &D, bool &ok)
{
    const int G = (int)grpCell.size() - 1, nOwn = A->nOwned;
    const std::vector<int> &off = fwd ? A->h_loOff : A->h_upOff;
    const std::vector<int> &ent = fwd ? A->h_loEnt : A->h_upNbr;
    // compact per-cell neighbour lists: nbr[W*c + k], src[W*c + k]
    std::vector<int> nbr((size_t)W * nOwn, -1), src((size_t)W * nOwn, -1);
    for (int c = 0; c < nOwn; c++) {
        const int sl = c >> 6, lane = c & 63, wdt = (off[sl + 1] - off[sl]) / 64;
        int k = 0;
        for (int s = 0; s < wdt; s++) {
            const int q = off[sl] + s * 64 + lane, e = ent[q];
            if (e < 0) continue;
            const int nb = fwd ? (e >> 4) : e;
            if (nb >= nOwn) continue;
            if (k >= W) { ok = false; return FFM_OK; }
            nbr[(size_t)W * c + k] = nb;
            src[(size_t)W * c + k] = fwd ? (A->h_upOff[(e >> 4) >> 6] + (e & 15) * 64 + ((e >> 4) & 63)) : q;
            k++;
        }
    }
    std::vector<unsigned short> code((size_t)4 * nOwn, T_NONE);
    std::vector<int> grpBat(G + 1, 0), batEnt(1, 0), batNeed, batPub, batHalo(1, 0), entCell, haloCells;
    std::vector<int> stamp(nOwn, -1), slotOf(nOwn, 0);
    int batchId = 0, nEnt = 0;
    int maxKB = T_KB;
    if (const char *e = getenv("FFM_TILE_KB")) maxKB = std::max(1, atoi(e));
    for (int g = 0; g < G; g++) {
        const int gs = grpCell[g], ge = grpCell[g + 1];
        // entries: runs of equal level, at most T_ENT cells; forward ascending cells, backward descending
        std::vector<std::pair<int, int>> ents;      // cell range [first, last) of each entry, in processing order
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
            int nE = 0;
            std::vector<int> halo;
            const size_t iStart = i;
            while (i < ents.size() && nE < maxKB) {
                const int c0 = ents[i].first, c1 = ents[i].second;
                // cells the batch has already swept: forward [first cell of the batch, c0), backward [c1, end of the batch)
                const int doneLo = fwd ? ents[iStart].first : c1, doneHi = fwd ? c0 : ents[iStart].second;
                bool farInBatch = false;
                std::vector<int> add;               // externals of this entry not yet in the batch halo
                for (int c = c0; c < c1; c++) for (int k = 0; k < W; k++) {
                    const int nb = nbr[(size_t)W * c + k];
                    if (nb < 0) continue;
                    const bool inRing = grpOfCell[nb] == g && std::abs(c - nb) <= T_RING - 2 * T_ENT;
                    // a value of this group that has left the ring is re-read from memory, which is only valid once a batch
                    // boundary has been passed since it was written
                    if (!inRing && grpOfCell[nb] == g && nb >= doneLo && nb < doneHi) farInBatch = true;
                    if (!inRing && stamp[nb] != batchId) { stamp[nb] = batchId; slotOf[nb] = -1; add.push_back(nb); }
                }
                if (nE > 0 && (farInBatch || halo.size() + add.size() > (size_t)T_HALO)) { for (int nb : add) stamp[nb] = -1; break; }
                if (farInBatch || add.size() > (size_t)T_HALO) { ok = false; return FFM_OK; }
                for (int nb : add) { slotOf[nb] = (int)halo.size(); halo.push_back(nb); }
                nE++; i++;
            }
            int maxNbLevelPlus1 = 0;
            for (size_t k2 = iStart; k2 < i; k2++) {
                for (int c = ents[k2].first; c < ents[k2].second; c++) for (int k = 0; k < W; k++) {
                    const int nb = nbr[(size_t)W * c + k];
                    if (nb < 0) continue;
                    if (stamp[nb] == batchId && slotOf[nb] >= 0) {
                        code[(size_t)4 * c + k] = (unsigned short)(T_HALOBIT | slotOf[nb]);
                        if (grpOfCell[nb] != g) maxNbLevelPlus1 = std::max(maxNbLevelPlus1, lvl[nb] + 1);
                    } else code[(size_t)4 * c + k] = (unsigned short)std::abs(c - nb);
                }
                entCell.push_back(ents[k2].first); entCell.push_back(ents[k2].second);
                nEnt++;
            }
            batEnt.push_back(nEnt);
            batNeed.push_back(maxNbLevelPlus1);          // predecessors must have completed every level < this
            batPub.push_back(i < ents.size() ? lvl[ents[i].first] : INT_MAX);
            haloCells.insert(haloCells.end(), halo.begin(), halo.end());
            batHalo.push_back((int)haloCells.size());
            batchId++;
        }
        grpBat[g + 1] = (int)batNeed.size();
    }
    // read-ahead padding
    for (int k = 0; k < 2 * (2 * T_PF + 2); k++) entCell.push_back(0);
    for (int k = 0; k < 3; k++) batHalo.push_back(batHalo.back());
    batEnt.push_back(batEnt.back()); batEnt.push_back(batEnt.back());
    batNeed.push_back(0); batPub.push_back(0);
    for (int k = 0; k < T_THREADS * 4; k++) haloCells.push_back(0);
    D.nEnt = nEnt; D.nBat = (int)batNeed.size() - 1;
    FFM_TRY(upv(&D.grpBat, grpBat)); FFM_TRY(upv(&D.batEnt, batEnt)); FFM_TRY(upv(&D.batNeed, batNeed)); FFM_TRY(upv(&D.batPub, batPub));
    FFM_TRY(upv(&D.batHalo, batHalo)); FFM_TRY(upv(&D.entCell, entCell)); FFM_TRY(upv(&D.haloCells, haloCells));
    FFM_TRY(upv(&D.code, code)); FFM_TRY(upv(&D.src, src));
    return FFM_OK;
}

int ffm_tile_build(ffm_ldu *A, const std::vector<int> &l, const std::vector<int> &u, const std::vector<int> &lev,
                   const std::vector<int> &bl, const std::vector<int> &grpCell)
{
    (void)l; (void)u;
    A->tile = new ffm_tile_plan();
    ffm_tile_plan *T = A->tile;
    T->G = (int)grpCell.size() - 1;
    if (T->G <= 0) return FFM_OK;                       // not usable: the caller falls back to ffm_pipe
    const int nOwn = A->nOwned;
    // slots per cell (ghost neighbours dropped): 3 for a hexahedral block, at most 4 supported
    int W = 0;
    for (int dir = 0; dir < 2; dir++) {
        const std::vector<int> &off = dir ? A->h_upOff : A->h_loOff;
        const std::vector<int> &ent = dir ? A->h_upNbr : A->h_loEnt;
        for (int c = 0; c < nOwn; c++) {
            const int sl = c >> 6, lane = c & 63, wdt = (off[sl + 1] - off[sl]) / 64;
            int k = 0;
            for (int s = 0; s < wdt; s++) { const int e = ent[off[sl] + s * 64 + lane]; if (e >= 0 && (dir ? e : (e >> 4)) < nOwn) k++; }
            W = std::max(W, k);
        }
    }
    if (W > 4) return FFM_OK;
    T->W = W <= 3 ? 3 : 4;
    std::vector<int> grpOfCell(nOwn);
    for (int g = 0; g < T->G; g++) for (int c = grpCell[g]; c < grpCell[g + 1]; c++) grpOfCell[c] = g;
    bool ok = true;
    FFM_TRY(build_dir(A, true, T->W, lev, grpCell, grpOfCell, T->f, ok));
    if (ok) FFM_TRY(build_dir(A, false, T->W, bl, grpCell, grpOfCell, T->b, ok));
    T->usable = ok;
    return FFM_OK;
}

// ------------------------------------------------------------------ device ---
struct TileView {
    int G;
    const int *grpCell, *grpBat, *batEnt, *batNeed, *batPub, *batHalo, *entCell, *haloCells, *predStart, *preds;
    const ushort4 *code;
    unsigned long long *progress;
    unsigned int *ticket;
};

__device__ __forceinline__ double t_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void t_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// workgroup barrier that only waits for this wave's LDS traffic: outstanding global loads (the read-ahead) stay in flight
__device__ __forceinline__ void t_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void t_wait_preds(const TileView &t, int p0, int p1, unsigned long long need, unsigned long long &seen, int *shAbort)
{
    // after a time-out anywhere the sweep runs on without waiting (no early exit: keeps the main loop free of exits); the
    // host finds the abort word set when the solve ends and reports the failure
    if (*(volatile int *)shAbort) return;
    for (int q = p0 + (int)threadIdx.x; q < p1; q += blockDim.x) {
        const bool cached = (p1 - p0) <= (int)blockDim.x;
        unsigned long long val = cached ? seen : 0ull;
        if (val < need) {
            const unsigned long long *addr = &t.progress[t.preds[q]];
            unsigned spins = 0;
            val = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (val < need) {
                __builtin_amdgcn_s_sleep(4);
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

__device__ __forceinline__ unsigned code_of(const ushort4 &q, int s) { return s == 0 ? q.x : s == 1 ? q.y : s == 2 ? q.z : q.w; }

// FWD: w[c] = rD[c]*r[c] - sum_k rD[c]*coef[c][k]*w[l_k]          (k ascending: the reference's face order)
// BWD: w[c] -= sum_k rD[c]*coef[c][k]*w[u_k]                      (k descending)
template <int W, bool FWD>
__global__ __launch_bounds__(T_THREADS) void k_tile(TileView t, const double *__restrict__ coef, const double *__restrict__ rD,
                                                    const double *__restrict__ r, double *w)
{
    __shared__ double ring[T_RING];
    __shared__ double halo[T_HALO];
    __shared__ int shG, shAbort;
    __shared__ unsigned shEpoch;
    const int tid = (int)threadIdx.x;
    if (tid == 0) {
        const unsigned tk = atomicAdd(&t.ticket[0], 1u);
        const int k = (int)(tk % (unsigned)t.G);
        shG = FWD ? k : t.G - 1 - k; shEpoch = tk / (unsigned)t.G + 1u; shAbort = 0;
    }
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(shG);
    const unsigned long long epoch = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)shEpoch) << 32;
    const int gs = __builtin_amdgcn_readfirstlane(t.grpCell[g]);
    const int b0 = __builtin_amdgcn_readfirstlane(t.grpBat[g]), b1 = __builtin_amdgcn_readfirstlane(t.grpBat[g + 1]);
    const int p0 = __builtin_amdgcn_readfirstlane(t.predStart[g]), p1 = __builtin_amdgcn_readfirstlane(t.predStart[g + 1]);
    if (b0 >= b1) return;
    const int eBeg = __builtin_amdgcn_readfirstlane(t.batEnt[b0]), eEnd = __builtin_amdgcn_readfirstlane(t.batEnt[b1]);

    // read-ahead registers: slot k holds the entries e with (e - eBeg) % T_PF == k
    int pc[T_PF];
    ushort4 pq[T_PF];
    double pa[T_PF][W], prd[T_PF], pv[T_PF];
    // cell ranges are looked up a further T_PF entries ahead: slot k holds the range of the entry that slot k fetches next
    int pb0[T_PF], pb1[T_PF];
#define T_BOUNDS(k, e) { const int ee_ = min((e), eEnd); pb0[k] = t.entCell[2 * ee_]; pb1[k] = t.entCell[2 * ee_ + 1]; }
#define T_FETCH(k, e) {                                                                                \
        const int c_ = FWD ? pb0[k] + tid : pb1[k] - 1 - tid;                                           \
        const bool ok_ = (e) < eEnd && c_ >= pb0[k] && c_ < pb1[k];                                     \
        const int cc_ = ok_ ? c_ : gs;                                                                  \
        pq[k] = t.code[cc_];                                                                            \
        _Pragma("unroll") for (int s = 0; s < W; s++) pa[k][s] = coef[(size_t)cc_ * W + s];             \
        prd[k] = rD[cc_];                                                                               \
        pv[k] = FWD ? r[cc_] : w[cc_];                                                                  \
        pc[k] = ok_ ? c_ : -1;                                                                          \
    }
#pragma unroll
    for (int k = 0; k < T_PF; k++) T_BOUNDS(k, eBeg + k);
#pragma unroll
    for (int k = 0; k < T_PF; k++) { T_FETCH(k, eBeg + k); T_BOUNDS(k, eBeg + T_PF + k); }

    int hc[4];                                              // halo cells of the next batch to open (4 per thread)
    {
        const int h0 = __builtin_amdgcn_readfirstlane(t.batHalo[b0]), h1 = __builtin_amdgcn_readfirstlane(t.batHalo[b0 + 1]);
#pragma unroll
        for (int j = 0; j < 4; j++) { const int i = h0 + tid + j * T_THREADS; hc[j] = (i < h1) ? t.haloCells[i] : -1; }
    }
    int bcur = b0, bStartE = eBeg, bEndE = eBeg, bPub = 0;
    // metadata of the next batch to open, loaded one batch ahead (uniform values held in vector registers until used)
    int mNeed = t.batNeed[b0], mH1 = t.batHalo[b0 + 1], mH2 = t.batHalo[b0 + 2], mEnd = t.batEnt[b0 + 1], mPub = t.batPub[b0];
    unsigned long long seen = 0;
    for (int e = eBeg; e < eEnd; e += T_PF) {
#pragma unroll
        for (int k = 0; k < T_PF; k++) {
            const int ee = e + k;
            if (ee == bStartE && ee < eEnd) {      // (entries past the end are padding of the unrolled loop: barriers only)
                // ---- open batch bcur: wait for the producers, fetch the halo values, read ahead the next batch's halo list
                const int need = __builtin_amdgcn_readfirstlane(mNeed);
                const int h1 = __builtin_amdgcn_readfirstlane(mH1), h2 = __builtin_amdgcn_readfirstlane(mH2);
                bEndE = __builtin_amdgcn_readfirstlane(mEnd);
                bPub = __builtin_amdgcn_readfirstlane(mPub);
                mNeed = t.batNeed[bcur + 1]; mH1 = t.batHalo[bcur + 2]; mH2 = t.batHalo[bcur + 3]; mEnd = t.batEnt[bcur + 2]; mPub = t.batPub[bcur + 1];
                if (need > 0) t_wait_preds(t, p0, p1, epoch | (unsigned long long)(unsigned)need, seen, &shAbort);
                __syncthreads();                        // poll done for everyone before anyone reads a published value
                double hv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) hv[j] = t_ld(&w[max(hc[j], 0)]);
#pragma unroll
                for (int j = 0; j < 4; j++) if (hc[j] >= 0) halo[tid + j * T_THREADS] = hv[j];
#pragma unroll
                for (int j = 0; j < 4; j++) { const int i = h1 + tid + j * T_THREADS; hc[j] = (i < h2) ? t.haloCells[i] : -1; }
                t_barrier();
            }
            // ---- entry ee from slot k
            {
                const int c = pc[k];
                if (c >= 0) {
                    double x[W];
#pragma unroll
                    for (int s = 0; s < W; s++) {
                        const unsigned cd = code_of(pq[k], s);
                        const int ri = ((FWD ? c - (int)cd : c + (int)cd) - gs) & (T_RING - 1);
                        const double xr = ring[ri], xh = halo[cd & (T_HALO - 1)];
                        x[s] = (cd & T_HALOBIT) ? xh : xr;
                    }
                    const double rd = prd[k];
                    double val = FWD ? rd * pv[k] : pv[k];
                    if (FWD) {
#pragma unroll
                        for (int s = 0; s < W; s++) if (code_of(pq[k], s) != T_NONE) val -= rd * pa[k][s] * x[s];
                    } else {
#pragma unroll
                        for (int s = W - 1; s >= 0; s--) if (code_of(pq[k], s) != T_NONE) val -= rd * pa[k][s] * x[s];
                    }
                    t_st(&w[c], val);
                    ring[(c - gs) & (T_RING - 1)] = val;
                }
            }
            // ---- refill slot k with entry ee + T_PF and look up the range of the one after
            T_FETCH(k, ee + T_PF);
            T_BOUNDS(k, ee + 2 * T_PF);
            t_barrier();                                // one barrier per level: the ring is visible to the next entry
            if (ee + 1 == bEndE) {
                // ---- close the batch: every wave's stores have been performed before lane 0 publishes
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t_barrier();
                if (tid == 0) __hip_atomic_store(&t.progress[g], epoch | (unsigned long long)(unsigned)bPub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bcur++; bStartE = bEndE;
            }
        }
    }
#undef T_FETCH
#undef T_BOUNDS
}

// out[i] = native[src[i]] (0 where a cell has fewer than W neighbours): coefficients in the kernel's cell-major layout
__global__ void k_tile_gather(long n, const int *__restrict__ src, const double *__restrict__ native, double *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int q = src[i];
        out[i] = (q >= 0) ? native[q] : 0.0;
    }
}

static TileView tview(const ffm_ldu *A, const TileDir &d, bool fwd)
{
    TileView t; t.G = A->tile->G; t.grpCell = A->grpCell; t.grpBat = d.grpBat; t.batEnt = d.batEnt; t.batNeed = d.batNeed; t.batPub = d.batPub;
    t.batHalo = d.batHalo; t.entCell = d.entCell; t.haloCells = d.haloCells; t.code = (const ushort4 *)d.code;
    t.predStart = fwd ? A->fPredStart : A->bPredStart; t.preds = fwd ? A->fPreds : A->bPreds;
    t.progress = A->pipeProgress; t.ticket = A->pipeTicket;
    return t;
}

// gathered coefficients of one direction, refreshed when the matrix coefficients changed
static int tile_coef(ffm_ldu *A, TileDir &d, bool upper, const double **out)
{
    const long n = (long)A->tile->W * A->nOwned;
    double *&buf = upper ? d.coefU : d.coefL;
    unsigned long &ep = upper ? d.epochU : d.epochL;
    if (!buf) { FFM_HIP(hipMalloc((void **)&buf, sizeof(double) * std::max<long>(n, 1))); ep = ~0ul; }
    if (ep != A->coeffEpoch) {
        const int g = std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS));
        hipLaunchKernelGGL(k_tile_gather, dim3(g), dim3(256), 0, A->ctx->stream, n, d.src, upper ? A->upper : A->lower, buf);
        ep = A->coeffEpoch;
    }
    *out = buf;
    return FFM_OK;
}

int ffm_tile_precond(ffm_ldu *A, int precond, bool transpose, const double *r, double *w)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    // forward coefficient: DIC upper; DILU lower; DILU^T upper.  backward: DIC upper; DILU upper; DILU^T lower.
    const bool fwdUpper = (precond == FFM_DIC) || transpose;
    const bool bwdUpper = (precond == FFM_DIC) || !transpose;
    const double *cf, *cb;
    FFM_TRY(tile_coef(A, T->f, fwdUpper, &cf));
    FFM_TRY(tile_coef(A, T->b, bwdUpper, &cb));
    if (T->W == 3) {
        hipLaunchKernelGGL((k_tile<3, true>), dim3(T->G), dim3(T_THREADS), 0, s, tview(A, T->f, true), cf, A->rD, r, w);
        hipLaunchKernelGGL((k_tile<3, false>), dim3(T->G), dim3(T_THREADS), 0, s, tview(A, T->b, false), cb, A->rD, r, w);
    } else {
        hipLaunchKernelGGL((k_tile<4, true>), dim3(T->G), dim3(T_THREADS), 0, s, tview(A, T->f, true), cf, A->rD, r, w);
        hipLaunchKernelGGL((k_tile<4, false>), dim3(T->G), dim3(T_THREADS), 0, s, tview(A, T->b, false), cb, A->rD, r, w);
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
