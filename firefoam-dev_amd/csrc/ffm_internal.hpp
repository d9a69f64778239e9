// ffm_internal.hpp -- private structures of libffm.so (not part of the C ABI).
#pragma once
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include <thread>
#include <algorithm>
#include <unordered_map>

#include "../../include/ffm.h"

struct ncclComm;
struct ffm_tile_plan;

void ffm_set_error(const char *fmt, ...);

#define FFM_HIP(call)                                                            \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            ffm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call,           \
                          hipGetErrorString(e_));                                \
            return FFM_ERR_HIP;                                                  \
        }                                                                        \
    } while (0)

#define FFM_TRY(call)                                                            \
    do {                                                                         \
        int r_ = (call);                                                         \
        if (r_ != FFM_OK) return r_;                                             \
    } while (0)

// Fixed reduction geometry: every dot product is a two-stage, atomics-free,
// order-fixed sum (RED_BLOCKS partials, then one block), so results are
// bitwise reproducible from run to run.
constexpr int RED_THREADS = 256;
constexpr int RED_BLOCKS = 2048;
constexpr int NSCAL = 64;  // device scalar slots per context

// device scalar slots
enum {
    S_TMP0 = 0, S_TMP1, S_TMP2, S_TMP3,
    S_XREF, S_NORMF, S_RES, S_RES0,
    S_WARA, S_WARA_OLD, S_WAPA, S_ALPHA, S_BETA, S_OMEGA,
    S_RA0RA, S_RA0RA_OLD, S_RA0AYA, S_TATA, S_TASA,
    S_SING,      // != 0 => singular flag
    S_NEG_ALPHA, S_ONE
};

struct ffm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    double *scal_d = nullptr;      // [NSCAL]
    double *partials_d = nullptr;  // [4][RED_BLOCKS]
    double *scal_h = nullptr;      // pinned [NSCAL]
    double *multiScal_d = nullptr, *multiScal_h = nullptr;      // [FFM_TILE_MAXSYS][NSCAL]: the lanes of ffm_solve_multi_d (allocated on first use)
    // communicator (RCCL); nRanks == 1 => serial
    int rank = 0, nRanks = 1;
    ncclComm *comm = nullptr;
    // a communicator of its own for the ghost exchange that overlaps the interior rows on commStream: RCCL serialises the operations of
    // ONE communicator, so collectives of one communicator issued from two streams (halo group on commStream, dot-product all-reduces on
    // stream) may be launched in different orders on different ranks and deadlock; two communicators have independent queues
    ncclComm *haloComm = nullptr;
    void *hostUser = nullptr;
    ffm_host_allreduce_fn hostAllreduce = nullptr;
    ffm_host_exchange_fn hostExchange = nullptr;
    ffm_host_exchange2_fn hostExchange2 = nullptr;
    int cuCount = 256;
    // pairGAMGAgglomeration::forward_: a STATIC upstream -- the direction in which the next pair agglomeration visits the cells, toggled by
    // every agglomeration of the run, so a second GAMG mesh / region starts in the direction the first one ended with (ffm_ctx_set_gamg_forward)
    bool gamgForward = true;
    // second stream for the RCCL ghost exchange of an Amul whose interior rows do not need the ghost values (tiled Amul):
    // pack (main stream) -> evPack -> send / receive (commStream) -> evRecv -> ghost-face tail (main stream)
    hipStream_t commStream = nullptr;
    hipEvent_t evPack = nullptr, evRecv = nullptr;
    // stream-ordered caching allocator behind ffm_malloc / ffm_free (the Foam layer makes one temporary per operator):
    // a freed block goes to the free list of its size class without synchronising -- every consumer runs on `stream`, so a
    // later owner's kernels are ordered after the previous owner's -- and is handed out again by the next ffm_malloc of that
    // class; ffm_ctx_trim / ffm_ctx_destroy return the cached blocks to the runtime
    std::unordered_map<size_t, std::vector<void *>> poolFree;
    std::unordered_map<void *, size_t> poolSize;
    size_t poolCachedBytes = 0, poolCapBytes = (size_t)96 << 30;
};

// Host <-> device copies, correct by construction: the context's stream is created non-blocking, so a null-stream hipMemcpy neither
// waits for work queued on it (a pooled block's zero-fill, a kernel that writes the table being read back) nor is waited for by it.
// Every copy of the library therefore goes through these two -- queued on the context's stream and waited for -- and the bare
// runtime calls are poisoned for the rest of every translation unit that includes this header.
inline int ffm_h2d(ffm_ctx *c, void *dst_d, const void *src, size_t bytes)
{
    if (!bytes) return FFM_OK;
    FFM_HIP(hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, c->stream)); FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}
inline int ffm_d2h(ffm_ctx *c, void *dst, const void *src_d, size_t bytes)
{
    if (!bytes) return FFM_OK;
    FFM_HIP(hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, c->stream)); FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}
inline int ffm_dzero(ffm_ctx *c, void *dst_d, size_t bytes)
{
    if (!bytes) return FFM_OK;
    FFM_HIP(hipMemsetAsync(dst_d, 0, bytes, c->stream));
    return FFM_OK;
}
template <class T> inline int ffm_upload_vec(ffm_ctx *c, T **d, const std::vector<T> &v, size_t minCount = 1)
{
    FFM_HIP(hipMalloc((void **)d, sizeof(T) * std::max<size_t>(v.size(), minCount)));
    return ffm_h2d(c, *d, v.data(), sizeof(T) * v.size());
}
#pragma GCC poison hipMemcpy hipMemset

// Host set-up loops over cells / faces (renumbered addressing, geometry in the native layout, the reconstruction tensors): independent
// iterations split over the host's cores (at most 16 threads; FFM_HOST_THREADS overrides; below 1M iterations: the caller's thread)
template <class Fn> inline void ffm_parallel_for(long n, Fn fn, long serialBelow = 1L << 20)
{
    static const int nT = [] { const char *e = getenv("FFM_HOST_THREADS"); int t = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return std::max(1, std::min(t, 16)); }();
    if (n < serialBelow || nT == 1) { fn(0L, n); return; }
    std::vector<std::thread> th;
    const long chunk = (n + nT - 1) / nT;
    for (int t = 0; t < nT; t++) { const long lo = t * chunk, hi = std::min(n, lo + chunk); if (lo < hi) th.emplace_back([=] { fn(lo, hi); }); }
    for (auto &x : th) x.join();
}

// key of a cached hipGraph of level-scheduled sweeps: EVERY device pointer the captured kernels bake in
struct SweepGraphKey {
    int kind; const void *p[6];
    bool operator<(const SweepGraphKey &o) const {
        if (kind != o.kind) return kind < o.kind;
        for (int i = 0; i < 6; i++) if (p[i] != o.p[i]) return p[i] < o.p[i];
        return false;
    }
};
constexpr size_t FFM_MAX_SWEEP_GRAPHS = 24;     // cap of the per-matrix graph cache (oldest entries are dropped first)

struct ffm_iface {
    int size = 0, nbrRank = -1;
    int offset = 0;                 // offset into the packed halo buffers
};

struct ffm_ldu {
    ffm_ctx *ctx = nullptr;
    int nCells = 0, nFaces = 0;        // nCells = owned + ghost cells (array length of every cell field)
    int nOwned = 0;                    // rows of this rank; ghost cells [nOwned, nCells) are copies of neighbour-rank cells
    long globalCells = 0;
    std::vector<int> h_ghSendCaller;        // the send cells of the ghost exchange in the caller's cell labels (GAMG: agglomerated per level)
    bool identity = true;           // caller numbering == internal numbering
    bool symmetric = true;
    bool bwdContig = true;          // backward levels are contiguous cell ranges

    // host copies of the analysis (internal numbering)
    std::vector<int> h_newToOldCell, h_newToOldFace;
    std::vector<int> h_fwdLevelStart;   // [nLevels+1] cell ranges (level-major)
    std::vector<int> h_bwdLevelStart;   // [nBwdLevels+1] ranges into bwdOrder
    std::vector<int> h_bwdFirstCell;    // [nBwdLevels] first cell of each level (bwdContig)
    int nLevels = 0, nBwdLevels = 0;

    // device addressing (internal cell numbering): sliced owner-ELL.
    // Cells are grouped in slices of 64 (one wavefront).  Slice sl stores its faces slot-major:
    //   upper part: entry e = upOff[sl] + s*64 + lane, s < upper width of the slice
    //       upNbr[e]  = neighbour cell of the s-th face OWNED by cell sl*64+lane (-1 = padding)
    //       upper[e] / lower[e] = the face's upper / lower coefficient (0 on padding)
    //     e is the library's NATIVE FACE INDEX; the owner of a native face is implicit.
    //   lower part: entry q = loOff[sl] + s*64 + lane
    //       loEnt[q] = (owner cell << 4) | slot of that face in the owner's row  (-1 = padding)
    //     listing the faces whose NEIGHBOUR is the cell, in the caller's face order.
    // Every load of an index or coefficient array is unit-stride across the wave; the symmetric
    // coefficient is stored once and re-read through the (nbr, slot) pair by the neighbour row.
    int nSlices = 0, upTotal = 0, loTotal = 0;
    int upWidthUniform = -1;                     // >= 0: every slice has this upper width
    int loWidthUniform = -1;
    int maxW = 0;                                // max slots (lower or upper) of any row
    int *upOff = nullptr, *loOff = nullptr;      // [nSlices+1]
    int *upNbr = nullptr;                        // [upTotal]
    int *loEnt = nullptr;                        // [loTotal]
    int *bwdOrder = nullptr;                     // [N] (only when !bwdContig)
    int *smallFwdStart = nullptr;                // [nLevels+1] device copy of h_fwdLevelStart (single-workgroup sweeps, ffm_solve.hip)
    int *smallBwdRange = nullptr;                // [2*nBwdLevels] {p0, p1} of every backward level
    int smallState = 0;                          // 0 not decided, 1 usable, -1 not (too large / switched off)
    int *flowOrder = nullptr;                    // [nOwned] cells by backward level (dataflow sweeps, ffm_solve.hip)
    int flowState = 0;                           // 0 not decided, 1 usable, -1 not
    int *cellPerm = nullptr;                     // [N] new->old (only when !identity)
    int *faceSrc = nullptr;                      // [upTotal] native face -> caller face id (-1 padding)
    std::vector<int> h_callerToNative;           // [F] caller face id -> native face index
    std::vector<int> h_upOff, h_loOff, h_loEnt, h_upNbr;   // host copies for the sweep planners (entries freed after use)
    int *callerToNative = nullptr;               // device copy (on demand)

    // coefficients (internal numbering / native face index)
    double *diag = nullptr, *upper = nullptr, *lower = nullptr;  // lower==upper when symmetric
    double *lowerBuf = nullptr;                                  // storage for asymmetric lower

    // XCD-aware row schedule of the row kernels: list of 256-row chunks; entry i is processed by a workgroup with
    // blockIdx % 8 == i % 8, i.e. (observed round-robin dispatch) always on the same XCD, and chunks with the same i % 8
    // cover the same eighth of every dependency level, so a row's neighbours in the adjacent levels were touched by the
    // same XCD and are found in its L2.  Speed only; correctness does not depend on the placement.
    int *rowSched = nullptr; int nSched = 0;

    // preconditioner state
    double *rD = nullptr;          // reciprocal diagonal (DIC/DILU/diagonal)
    int rDKind = -1;               // which preconditioner rD currently holds
    unsigned long coeffEpoch = 0, rDEpoch = ~0ul;
    unsigned long offDiagEpoch = 0;       // bumped when upper / lower change (coeffEpoch: any coefficient, diag included)
    double *diagBuf = nullptr, *upperBuf = nullptr;     // the library's own coefficient storage; diag/upper/lower may instead
                                                        // point at caller arrays (ffm_ldu_bind_coeffs_native_d)

    // work vectors (internal numbering), allocated on demand
    std::vector<double *> work;    // [nWork] each N doubles
    double *gsProd = nullptr;      // [3N] scratch of the tiled Gauss-Seidel sweep
    double *permIn[3] = {nullptr, nullptr, nullptr};  // staging for caller-order vectors

    // interfaces (processor patches), packed patch after patch
    std::vector<ffm_iface> ifaces;
    std::vector<int> ifaceTags, ghTags;   // pair tags: posting order of the point-to-point messages (ffm_ldu_set_exchange_tags)
    int haloTotal = 0;
    int *ifFaceCells = nullptr;    // [haloTotal] internal numbering
    double *ifBou = nullptr, *ifInt = nullptr;   // [haloTotal]
    double *haloSend = nullptr, *haloRecv = nullptr;
    double *haloSend_h = nullptr, *haloRecv_h = nullptr;   // pinned, host transport only
    // boundary cells grouped: cell ifCell[j] owns packed items ifItem[ifCellStart[j]..ifCellStart[j+1])
    // listed in (patch, face) order, so a cell's interface terms are added in the reference's order
    int nIfCells = 0;
    int *ifCell = nullptr, *ifCellStart = nullptr, *ifItem = nullptr;

    // ghost-cell halo plan (native decomposed path): for neighbour q, send x[ghSendCells[ghSendOff[q]..]] and receive
    // straight into x[nOwned + ghRecvOff[q] ..]
    std::vector<int> ghNbrRank, ghSendOff, ghRecvOff;   // offsets have nNbr+1 entries
    int *ghSendCells = nullptr;
    bool ghPending = false;        // an overlapped exchange is in flight on ctx->commStream (ffm_ghost_exchange_begin / _end)
    double *ghSendBuf = nullptr, *ghSendBuf_h = nullptr, *ghRecvBuf_h = nullptr;

    // ---- tiled sweep plan (sweepMode == 2): cells are numbered group-major, level-major inside a group;
    // group g = cells [grpCell[g], grpCell[g+1]); one workgroup sweeps one group level by level (ffm_tile.hip)
    int sweepMode = 0;              // 0: one launch per level (level-major numbering); 2: tiled wavefront sweeps
    int nGroups = 0;
    bool bwdIsReverse = false;      // inside every group the backward order is the exact reverse of the forward order
    int *grpCell = nullptr;         // [G+1]
    int *bwdCells = nullptr;        // [nOwned] cells in (group, backward level) order
    unsigned int *sweepTicket = nullptr;          // [2]: ticket counter (zeroed on the stream before every sweep), abort flag
    ffm_tile_plan *tile = nullptr;                // tiled wavefront plan (sweepMode == 2)

    // cached hipGraphs of level-scheduled sweeps
    std::map<SweepGraphKey, hipGraphExec_t> graphs;
    std::vector<SweepGraphKey> graphOrder;         // insertion order, for the cap
};

// FFM_TIMING=1: wall time of the host-side set-up stages to stderr
struct FfmStageTimer {
    const char *what; double t0; bool on;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    explicit FfmStageTimer(const char *w) : what(w), t0(now()), on(getenv("FFM_TIMING") != nullptr) {}
    ~FfmStageTimer() { if (on) fprintf(stderr, "ffm timing: %-28s %.2f s\n", what, now() - t0); }
};
// the same in laps: lap("x") prints the time since the previous lap (FFM_TIMING=2)
struct FfmLapTimer {
    const char *what; double t0; bool on;
    explicit FfmLapTimer(const char *w) : what(w), t0(FfmStageTimer::now()), on(getenv("FFM_TIMING") != nullptr && atoi(getenv("FFM_TIMING")) >= 2) {}
    void lap(const char *stage) { if (!on) return; const double t = FfmStageTimer::now(); fprintf(stderr, "ffm timing:   %s: %-30s %.2f s\n", what, stage, t - t0); t0 = t; }
};
// label of the 2-D tile (a, b) of cell columns.  Ties between tiles that are ready at the same time are broken by label
// (ffm_ldu.hip: topological ranking), so the label orders the tickets: anti-diagonal major = the order of the sweep's wavefront
// (FFM_TILE_ROW_ORDER=1: row-major, the round-1 order)
static inline int ffm_tile_label(int a, int b)
{
    static const bool row = getenv("FFM_TILE_ROW_ORDER") && atoi(getenv("FFM_TILE_ROW_ORDER")) != 0;
    a = a < 2047 ? a : 2047; b = b < 2047 ? b : 2047;
    return row ? a + 32768 * b : (a + b) * 4096 + b;
}
struct LduView;
LduView ffm_view(const ffm_ldu *A);
// ---- internal helpers shared between translation units -------------------
int ffm_ldu_work(ffm_ldu *A, int idx, double **out);           // lazily allocated N-vectors
int ffm_to_internal(ffm_ldu *A, const double *x_d, int slot, const double **out);
int ffm_from_internal(ffm_ldu *A, const double *xin, double *x_d);
int ffm_k_spmv(ffm_ldu *A, const double *x, double *y, bool transpose);
int ffm_k_spmv_dot(ffm_ldu *A, const double *x, double *y, int slot);  // y=Ax, scal[slot]=x.y (local)
int ffm_k_residual(ffm_ldu *A, const double *x, const double *b, double *r);
int ffm_k_sumA(ffm_ldu *A, double *s);
int ffm_k_spmv_sumA(ffm_ldu *A, const double *x, double *y, double *s);
int ffm_halo_update(ffm_ldu *A, const double *x, double *y, const double *coeffs, double sign);  // exchange + apply
int ffm_halo_apply(ffm_ldu *A, double *y, const double *coeffs, const double *vals /*null => 1*/, double sign);
int ffm_allreduce_slots(ffm_ctx *ctx, int firstSlot, int n);   // sum over ranks, in stream
int ffm_allreduce_minmax(ffm_ctx *ctx, int slot, int isMax);
void ffm_comm_finalize_i(ffm_ctx *ctx);
int ffm_precond_setup_i(ffm_ldu *A, int precond);
int ffm_precond_apply_i(ffm_ldu *A, int precond, bool transpose, const double *r, double *w);
int ffm_tile_build(ffm_ldu *A, const std::vector<int> &lev, const std::vector<int> &bl, const std::vector<int> &grpCell,
                   const std::vector<int> *bwdCells /* null: the backward order mirrors the forward order */);
bool ffm_tile_feasible(int nOwn, int F, const int *l, const int *u);
int ffm_tile_calc_rD(ffm_ldu *A);
int ffm_flow_check_abort(ffm_ldu *A);                    // dataflow sweeps of level-scheduled matrices (ffm_solve.hip)
bool ffm_tile_gs_usable(const ffm_ldu *A);
int ffm_tile_gs_ghost_terms(ffm_ldu *A, const double *psi, double *bP);
int ffm_ghost_exchange(ffm_ldu *A, double *x);
int ffm_ghost_exchange_begin(ffm_ldu *A, double *x);     // starts the refresh of x[nOwned..nCells); may return with it in flight
int ffm_ghost_exchange_end(ffm_ldu *A);                  // the main stream waits for it
int ffm_tile_gs(ffm_ldu *A, bool sym, double *psi, const double *bP, double *bSave, double *prod3);
bool ffm_tile_amul_usable(const ffm_ldu *A);
int ffm_tile_amul(ffm_ldu *A, const double *x, double *y, int dotSlot);     // returns 1 if the dot product is left to the caller
void ffm_tile_free(ffm_ldu *A);
bool ffm_tile_usable(const ffm_ldu *A);
struct FfmFvSegs { int nSeg; const int4 *seg; const int4 *rec; };
bool ffm_tile_fv_segments(ffm_ldu *A, int runLength, FfmFvSegs *out);   // single block, tile plan present: FV face passes through LDS (ffm_fused.hip)
bool ffm_tile_pcg_fusable(const ffm_ldu *A);
int ffm_tile_pcg_fwd(ffm_ldu *A, double *rA, double *wA, int slot, const double *qA = nullptr);
bool ffm_tile_amul_pcg_usable(const ffm_ldu *A);
bool ffm_tile_amul_asym_usable(const ffm_ldu *A);
int ffm_tile_amul_asym(ffm_ldu *A, const double *x, double *y);
int ffm_tile_amul_pcg(ffm_ldu *A, const double *w, const double *pin, double *pout, double *psi, double *y, int dotSlot);
int ffm_tile_pcg_bwd(ffm_ldu *A, const double *rA, double *wA, int slot);
int ffm_tile_precond(ffm_ldu *A, int precond, bool transpose, const double *r, double *w);
// several systems with the matrix's off-diagonal coefficients in one sweep (ffm_tile.hip: k_tile_m)
#define FFM_TILE_MAXSYS 4
bool ffm_tile_multi_usable(const ffm_ldu *A);
int ffm_tile_precond_multi(ffm_ldu *A, int precond, int n, const double *const *rD, const double *const *r, double *const *w);
int ffm_tile_calc_rD_multi(ffm_ldu *A, int n, const double *const *diag, double *const *D);
int ffm_k_spmv_multi(ffm_ldu *A, int n, const double *const *diag, const double *const *x, double *const *y, double *const *sumA);
int ffm_tile_check_abort(ffm_ldu *A);
int ffm_gs_smooth_i(ffm_ldu *A, bool sym, int nSweeps, double *psi, const double *b);
int ffm_halo_exchange(ffm_ldu *A, const double *x);
int ffm_ghost_exchange(ffm_ldu *A, double *x);                 // refresh x[nOwned..nCells) from the neighbour ranks          // pack x[faceCells], exchange into haloRecv
int ffm_read_scalars(ffm_ctx *ctx);  // scal_d -> scal_h, synchronises the stream

// reductions into device scalar slots (local partial sums; caller all-reduces)
int ffm_k_dot(ffm_ctx *ctx, const double *x, const double *y, long n, int slot);
int ffm_k_summag(ffm_ctx *ctx, const double *x, long n, int slot);
int ffm_k_sum(ffm_ctx *ctx, const double *x, long n, int slot);
int ffm_k_sumsqr(ffm_ctx *ctx, const double *x, long n, int slot);

static inline int ffm_grid(long n, int threads) { return (int)((n + threads - 1) / threads); }
