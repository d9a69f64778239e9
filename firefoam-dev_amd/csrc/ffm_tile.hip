// ffm_tile.hip -- tiled wavefront sweeps for DIC / DILU: calcReciprocalD, forward and backward substitution.
//
// Same operators, bitwise, as the level-scheduled kernels of ffm_solve.hip (OpenFOAM-dev DICPreconditioner /
// DILUPreconditioner::calcReciprocalD and ::precondition, reference selection cases/steckler/system/fvSolution:21-46).
// Motivation and measurements (MI355X, 400^3 box = 1198 dependency levels): a hand-off between workgroups costs about as
// much as a dependent kernel launch (3-5 us under load), so neither one launch per level nor a persistent kernel that
// publishes a progress word per level (tried in round 1, removed) gets under ~3.6 us per level.  Here
//   * cells are split into groups (2-D tiles of cell columns: a host hint or a detected blockMesh box) whose dependency
//     graph is acyclic; cells are numbered group-major and level-major inside a group; one workgroup sweeps one group,
//     one "entry" (<= 256 cells of one level) per step;
//   * inside a group the values of earlier levels are exchanged through an LDS ring: one workgroup barrier per level;
//   * a value another group needs (tile faces, ~12 % of the cells) is also stored to a "mailbox" slot with one 8-byte
//     agent-scope store.  Mailboxes are filled with a signalling-NaN sentinel before every sweep, and a sweep never
//     produces that bit pattern, so the value is its own ready flag: the consumer loads the slot with an agent-scope
//     load and re-loads it while it still reads the sentinel (cdna_hip_programming.md G16 form R2, "the data IS the
//     flag": one aligned 8-byte granule, no fence, no flag, no drain).  No progress words, no per-batch synchronisation;
//   * everything an entry needs from memory lives in streams private to this kernel whose addresses depend on the cell
//     index only (16-bit neighbour codes, coefficients gathered once per coefficient update) and is fetched T_PF entries
//     ahead into registers; entry records 2*T_PF ahead; mailbox values T_PM ahead.  The consumer therefore trails its
//     producers by a few levels and normally finds its values ready; the per-level critical path is LDS reads, a few
//     FMAs, one LDS write and one barrier;
//   * groups are handed out by an atomic ticket in topological order, so a workgroup only waits for groups that are
//     already running or finished (no residency assumption, no deadlock); every spin is bounded and raises the abort word
//     that ffm_tile_check_abort() reports when the solve ends.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <atomic>
#include <climits>

constexpr int T_RING = 4096;        // doubles (power of two)
constexpr int T_ENT = 256;          // max cells per entry = workgroup size
constexpr int T_THREADS = 256;
constexpr int T_W = 3;              // neighbour slots per cell and direction (hexahedra: 3)
#ifndef FFM_T_PF
#define FFM_T_PF 8
#endif
constexpr int T_PF = FFM_T_PF;      // entries fetched ahead of the one being computed
constexpr int T_PM = 3;             // mailbox values are loaded this many entries ahead (2 <= T_PM < T_PF)
constexpr int T_XMAX = 64;          // max external references per entry (one lane of the mail wave each; power of two)
constexpr int T_RINGD = T_RING - 2 * T_ENT;     // largest cell distance served by the ring
constexpr unsigned T_SPIN_LIMIT = 1u << 22;
constexpr unsigned T_NONE = 0xFFFFu;               // absent neighbour / cell that publishes nothing
constexpr int T_LDS = T_RING + 2 * T_XMAX + 1;      // ring, two external buffers (entry parity), one dummy slot read by absent neighbours
constexpr unsigned long long T_SENT = 0x7FF4DEADBEEFCAFEull;    // signalling NaN: no arithmetic result has these bits
static_assert(T_LDS < (int)T_NONE && T_ENT <= 256, "neighbour codes are LDS slots");
static_assert(T_PM >= 2 && T_PM < T_PF, "read-ahead distances");

struct TileDir {            // one sweep direction (device arrays)
    int nEnt = 0, nPub = 0;
    int *grpEnt = nullptr;      // [G+1] entries of each group
    int4 *rec = nullptr;        // [nEnt + pad] {first cell, cells | externals << 16, first mailbox slot, first extSrc index}
    int *extSrc = nullptr;      // mailbox slot of every external reference, entry by entry
    unsigned short *code = nullptr;   // [4*nOwn] per cell: 3 neighbour codes = the LDS slot holding the neighbour's value when the cell is computed
                                      // (ring slot (nb - group start) mod T_RING, or T_RING + parity(entry)*T_XMAX + external index; T_NONE: no neighbour), publish slot | T_NONE
    int *src = nullptr;         // [3*nOwn] native coefficient index of each neighbour slot (-1: none)
    int *nbrCell = nullptr;     // [3*nOwn] the neighbour cell of each slot (-1: none)
    double *coefU = nullptr, *coefL = nullptr;        // [3*nOwn] gathered upper / lower coefficients (lazily allocated)
    unsigned long epochU = ~0ul, epochL = ~0ul;
    double *mail = nullptr;     // [nPub + 1] (inside ffm_tile_plan::mailAll)
};

struct ffm_tile_plan {
    bool usable = false;
    int G = 0;
    TileDir f, b;
    double *mailAll = nullptr;
    long nMail = 0;
    double *mailMulti = nullptr;    // [FFM_TILE_MAXSYS][nMail]: the mailboxes of the multi-system sweeps (k_tile_m), allocated on first use
    unsigned long long *trace = nullptr;    // diagnostics (ffm_debug_tile_trace): per group {start, first entry done, end, re-loads} of the last launch
    // ---- general backward order: when the backward dependency order inside a group is not the mirror image of the forward
    // one (baffles, unstructured meshes), the backward sweep runs in "position space": position q holds cell cellOf[q]
    // (cells of a group sorted by backward level), all backward streams are indexed by position, and the vectors of a
    // sweep are permuted into / out of position space by a gather / scatter pass around the kernel
    bool mirror = true;
    const int *cellOf = nullptr;    // [nOwn] (= ffm_ldu::bwdCells, not owned)
    double *wp = nullptr, *rDp = nullptr;
    unsigned long rDpEpoch = ~0ul; int rDpKind = -1;
    // ---- tiled Amul (symmetric matrices): same groups and entries as the forward sweep
    bool amulUsable = false;
    bool gsTables = false;          // the cell-space upper-neighbour tables below exist (Gauss-Seidel sweeps, Amul tail)
    std::vector<int> grpEntHost;    // forward entries of each group (host copy)
    int4 *arec = nullptr;           // [nEnt + pad] {first cell, cells | externals << 16, first index into aext, 0}
    uint4 *acode = nullptr;         // [nOwn] 8 x 16 bit: 3 lower codes, 3 upper codes (A_* encoding below), 2 spare
    int2 *aext = nullptr;           // per external reference {cell whose x is needed, native index of the coefficient or -1}
    // owned upper neighbours of every cell in CELL space (the backward direction's tables are in position space when
    // !mirror): native coefficient index and neighbour cell of slot k, and the upper coefficients gathered through them
    int *upSrcCell = nullptr, *upNbrCell = nullptr;
    double *upCoefCell = nullptr, *diagp = nullptr;
    unsigned long upCoefEpoch = ~0ul, diagpEpoch = ~0ul;
    int nSeg = 0;                   // workgroups of the tiled Amul: a segment = a run of entries of one group
    int4 *aseg = nullptr;           // [nSeg] {group, first entry, end entry, end entry of the group}
    double *amulPartials = nullptr; // [nSeg]
    int4 *fvSeg = nullptr; int nFvSeg = 0, fvRun = 0;      // segments of the FV face passes (ffm_tile_fv_segments)
    int nTail = 0;                  // cells that own faces towards ghost cells (processed after the tiled kernel, in face order)
    int *tailCell = nullptr, *tailStart = nullptr, *tailFace = nullptr, *tailNbr = nullptr;
};

// ---- FV face passes on the tile numbering (ffm_fused.hip: k_mv_tile): runs of forward entries of one group, one workgroup each
// {first entry of the group, first entry of the run, end entry of the run, end entry of the group}; entry records = the forward plan's
bool ffm_tile_fv_segments(ffm_ldu *A, int runLength, FfmFvSegs *out)
{
    ffm_tile_plan *T = A->tile;
    if (!T || !T->usable || T->grpEntHost.empty() || A->nCells != A->nOwned) return false;
    if (!T->fvSeg || T->fvRun != runLength) {
        std::vector<int4> seg;
        const std::vector<int> &ge = T->grpEntHost;
        for (int g = 0; g < T->G; g++) for (int a = ge[g]; a < ge[g + 1]; a += runLength) seg.push_back(make_int4(ge[g], a, std::min(a + runLength, ge[g + 1]), ge[g + 1]));
        if (T->fvSeg) { hipFree(T->fvSeg); T->fvSeg = nullptr; }
        if (ffm_upload_vec(A->ctx, &T->fvSeg, seg) != FFM_OK) return false;
        T->nFvSeg = (int)seg.size(); T->fvRun = runLength;
    }
    out->nSeg = T->nFvSeg; out->seg = T->fvSeg; out->rec = T->f.rec;
    return true;
}

static void free_dir(TileDir &d)
{
    hipFree(d.grpEnt); hipFree(d.rec); hipFree(d.extSrc); hipFree(d.code); hipFree(d.src); hipFree(d.nbrCell); hipFree(d.coefU); hipFree(d.coefL);
    d = TileDir();
}
void ffm_tile_free(ffm_ldu *A)
{
    if (!A->tile) return;
    free_dir(A->tile->f); free_dir(A->tile->b);
    hipFree(A->tile->mailAll); hipFree(A->tile->mailMulti); hipFree(A->tile->trace); hipFree(A->tile->wp); hipFree(A->tile->rDp);
    hipFree(A->tile->arec); hipFree(A->tile->acode); hipFree(A->tile->aext); hipFree(A->tile->aseg); hipFree(A->tile->amulPartials);
    hipFree(A->tile->fvSeg); hipFree(A->tile->upSrcCell); hipFree(A->tile->upNbrCell); hipFree(A->tile->upCoefCell); hipFree(A->tile->diagp);
    hipFree(A->tile->tailCell); hipFree(A->tile->tailStart); hipFree(A->tile->tailFace); hipFree(A->tile->tailNbr);
    delete A->tile; A->tile = nullptr;
}
bool ffm_tile_usable(const ffm_ldu *A) { return A->tile && A->tile->usable; }

template <class T> static int upv(ffm_ctx *c, T **d, const std::vector<T> &v) { return ffm_upload_vec(c, d, v); }

// The tiled sweeps need at most T_W lower and T_W upper neighbours per owned cell (ghost neighbours not counted) and,
// inside every group, a backward order that is the reverse of the forward order (LduAnalysis::bwdIsReverse).
bool ffm_tile_feasible(int nOwn, int F, const int *l, const int *u)
{
    std::vector<unsigned char> nl(nOwn, 0), nu(nOwn, 0);
    for (int f = 0; f < F; f++) {
        if (u[f] >= nOwn) continue;
        if (++nu[l[f]] > T_W || ++nl[u[f]] > T_W) return false;
    }
    return true;
}

// Build one direction.  fwd: neighbours = lower entries (cells with smaller index); bwd: upper slots.  Neighbours that are
// ghost cells are dropped (block-Jacobi sweeps ignore them); the remaining ones keep their order, which is the order of
// the reference's face loop.
static int build_dir(ffm_ldu *A, bool fwd, const std::vector<int> &lvl, const std::vector<int> &grpCell,
                     const std::vector<int> &grpOfCell, TileDir &D, bool &ok, std::vector<int4> *recOut = nullptr,
                     std::vector<int> *grpEntOut = nullptr)
{
    const int G = (int)grpCell.size() - 1, nOwn = A->nOwned, W = T_W;
    const std::vector<int> &off = fwd ? A->h_loOff : A->h_upOff;
    const std::vector<int> &ent = fwd ? A->h_loEnt : A->h_upNbr;
    std::vector<int> nbr((size_t)W * nOwn, -1), src((size_t)W * nOwn, -1);
    std::vector<unsigned char> exposed(nOwn, 0);
    auto isExt = [&](int c, int nb) { return grpOfCell[nb] != grpOfCell[c] || std::abs(c - nb) > T_RINGD; };
    std::atomic<int> tooMany(0);
    ffm_parallel_for(nOwn, [&](long c0_, long c1_) {
        for (long c = c0_; c < c1_; c++) {
            const int sl = (int)(c >> 6), lane = (int)(c & 63), wdt = (off[sl + 1] - off[sl]) / 64;
            int k = 0;
            for (int s = 0; s < wdt; s++) {
                const int q = off[sl] + s * 64 + lane, e = ent[q];
                if (e < 0) continue;
                const int nb = fwd ? (e >> 4) : e;
                if (nb >= nOwn) continue;
                if (k >= W) { tooMany = 1; break; }
                nbr[(size_t)W * c + k] = nb;
                src[(size_t)W * c + k] = fwd ? (A->h_upOff[(e >> 4) >> 6] + (e & 15) * 64 + ((e >> 4) & 63)) : q;
                if (isExt((int)c, nb)) exposed[nb] = 1;          // (several threads may store the same 1)
                k++;
            }
        }
    });
    if (tooMany) { ok = false; return FFM_OK; }
    // entries: runs of one level, at most T_ENT cells and T_XMAX external references; forward ascending, backward descending
    std::vector<unsigned short> code((size_t)4 * nOwn, (unsigned short)T_NONE);
    std::vector<int> grpEnt(G + 1, 0), mailIdx(nOwn, -1);
    std::vector<int4> rec;
    int nPub = 0;
    auto extRefs = [&](int c) { int n = 0; for (int k = 0; k < W; k++) { const int nb = nbr[(size_t)W * c + k]; if (nb >= 0 && isExt(c, nb)) n++; } return n; };
    for (int g = 0; g < G; g++) {
        const int gs = grpCell[g], ge = grpCell[g + 1];
        int prevLevel = -1;
        auto close = [&](int c0, int c1, int nExt) {          // cells [c0,c1)
            if (lvl[c0] < prevLevel) ok = false;               // processing order must not go back in level
            prevLevel = lvl[c0];
            int slot = 0;
            for (int c = c0; c < c1; c++) if (exposed[c]) { code[(size_t)4 * c + 3] = (unsigned short)slot; mailIdx[c] = nPub + slot; slot++; }
            rec.push_back(make_int4(c0, (c1 - c0) | (nExt << 16), nPub, 0));
            nPub += slot;
        };
        if (fwd) {
            for (int c = gs; c < ge;) {
                int e = c, nx = 0;
                while (e < ge && lvl[e] == lvl[c] && e - c < T_ENT) { const int x = extRefs(e); if (e > c && nx + x > T_XMAX) break; nx += x; e++; }
                close(c, e, nx); c = e;
            }
        } else {
            for (int c = ge; c > gs;) {
                int e = c, nx = 0;
                while (e > gs && lvl[e - 1] == lvl[c - 1] && c - e < T_ENT) { const int x = extRefs(e - 1); if (e < c && nx + x > T_XMAX) break; nx += x; e--; }
                close(e, c, nx); c = e;
            }
        }
        if (!ok) return FFM_OK;
        grpEnt[g + 1] = (int)rec.size();
    }
    // neighbour codes and the external lists (mailbox slots are known for every group now)
    // (the entries' external lists start at the running sum of their external counts: the entries are independent of each other)
    size_t nExtAll = 0;
    for (int4 &R : rec) { R.w = (int)nExtAll; nExtAll += (size_t)(R.y >> 16); }
    std::vector<int> extSrc(nExtAll);
    std::atomic<int> bad(0);
    ffm_parallel_for((long)rec.size(), [&](long e0_, long e1_) {
        for (long ei = e0_; ei < e1_; ei++) {
            const int4 &R = rec[ei];
            const int c0 = R.x, cnt = R.y & 0xFFFF;
            const int gs = cnt ? grpCell[grpOfCell[c0]] : 0;
            int t = 0;
            for (int c = c0; c < c0 + cnt; c++) for (int k = 0; k < W; k++) {
                const int nb = nbr[(size_t)W * c + k];
                if (nb < 0) continue;
                if (isExt(c, nb)) {
                    if (mailIdx[nb] < 0 || t >= (R.y >> 16)) { bad = mailIdx[nb] < 0 ? 1 : 2; continue; }
                    code[(size_t)4 * c + k] = (unsigned short)(T_RING + (int)(ei & 1) * T_XMAX + t); extSrc[(size_t)R.w + t] = mailIdx[nb]; t++;
                } else code[(size_t)4 * c + k] = (unsigned short)((nb - gs) & (T_RING - 1));
            }
            if (t != (R.y >> 16)) bad = 2;
        }
    }, nOwn >= (1 << 20) ? 1024 : (1L << 40));
    if (bad == 1) { ffm_set_error("internal: tile plan references an unpublished cell"); return FFM_ERR_ADDR; }
    if (bad) { ffm_set_error("internal: tile plan external count mismatch"); return FFM_ERR_ADDR; }
    D.nEnt = (int)rec.size(); D.nPub = nPub;
    if (recOut) *recOut = rec;
    if (grpEntOut) *grpEntOut = grpEnt;
    for (int k = 0; k < 2 * T_PF + 2; k++) rec.push_back(make_int4(0, 0, 0, 0));      // read-ahead padding
    for (int k = 0; k < T_THREADS; k++) extSrc.push_back(0);
    FFM_TRY(upv(A->ctx, &D.grpEnt, grpEnt)); FFM_TRY(upv(A->ctx, &D.rec, rec)); FFM_TRY(upv(A->ctx, &D.extSrc, extSrc)); FFM_TRY(upv(A->ctx, &D.code, code)); FFM_TRY(upv(A->ctx, &D.src, src));
    if (!fwd) FFM_TRY(upv(A->ctx, &D.nbrCell, nbr));
    return FFM_OK;
}


// Backward direction in position space (see ffm_tile_plan::mirror): position q of the group-major array bwdCells holds the
// cell processed q-th; neighbours = the owned upper neighbours of that cell, referred to by their positions (always smaller
// inside the group).  Same outputs as build_dir, every per-cell array indexed by position.
static int build_dir_pos(ffm_ldu *A, const std::vector<int> &bl, const std::vector<int> &grpCell, const std::vector<int> &grpOfCell,
                         const std::vector<int> &bwdCells, TileDir &D, bool &ok)
{
    const int G = (int)grpCell.size() - 1, nOwn = A->nOwned, W = T_W;
    std::vector<int> posOf(nOwn);
    for (int q = 0; q < nOwn; q++) posOf[bwdCells[q]] = q;
    std::vector<int> nbr((size_t)W * nOwn, -1), src((size_t)W * nOwn, -1);        // neighbour POSITIONS
    std::vector<unsigned char> exposed(nOwn, 0);                                   // by position
    auto isExt = [&](int q, int nq) { return grpOfCell[bwdCells[nq]] != grpOfCell[bwdCells[q]] || std::abs(q - nq) > T_RINGD; };
    for (int q = 0; q < nOwn; q++) {
        const int c = bwdCells[q], sl = c >> 6, lane = c & 63, wdt = (A->h_upOff[sl + 1] - A->h_upOff[sl]) / 64;
        int k = 0;
        for (int s = 0; s < wdt; s++) {
            const int e = A->h_upOff[sl] + s * 64 + lane, nb = A->h_upNbr[e];
            if (nb < 0 || nb >= nOwn) continue;
            if (k >= W) { ok = false; return FFM_OK; }
            const int nq = posOf[nb];
            if (grpOfCell[nb] == grpOfCell[c] && nq >= q) { ok = false; return FFM_OK; }      // must have been processed earlier
            nbr[(size_t)W * q + k] = nq; src[(size_t)W * q + k] = e;
            if (isExt(q, nq)) exposed[nq] = 1;
            k++;
        }
    }
    std::vector<unsigned short> code((size_t)4 * nOwn, (unsigned short)T_NONE);
    std::vector<int> grpEnt(G + 1, 0), mailIdx(nOwn, -1);                          // mailIdx by position
    std::vector<int4> rec;
    int nPub = 0;
    auto extRefs = [&](int q) { int n = 0; for (int k = 0; k < W; k++) { const int nq = nbr[(size_t)W * q + k]; if (nq >= 0 && isExt(q, nq)) n++; } return n; };
    for (int g = 0; g < G; g++) {
        const int gs = grpCell[g], ge = grpCell[g + 1];
        int prevLevel = -1;
        for (int q = gs; q < ge;) {
            int e = q, nx = 0;
            while (e < ge && bl[bwdCells[e]] == bl[bwdCells[q]] && e - q < T_ENT) { const int x = extRefs(e); if (e > q && nx + x > T_XMAX) break; nx += x; e++; }
            if (bl[bwdCells[q]] < prevLevel) { ok = false; return FFM_OK; }
            prevLevel = bl[bwdCells[q]];
            int slot = 0;
            for (int p = q; p < e; p++) if (exposed[p]) { code[(size_t)4 * p + 3] = (unsigned short)slot; mailIdx[p] = nPub + slot; slot++; }
            rec.push_back(make_int4(q, (e - q) | (nx << 16), nPub, 0));
            nPub += slot;
            q = e;
        }
        grpEnt[g + 1] = (int)rec.size();
    }
    std::vector<int> extSrc;
    for (size_t ei = 0; ei < rec.size(); ei++) {
        int4 &R = rec[ei];
        const int q0 = R.x, cnt = R.y & 0xFFFF;
        const int gs = cnt ? grpCell[grpOfCell[bwdCells[q0]]] : 0;
        R.w = (int)extSrc.size();
        int t = 0;
        for (int q = q0; q < q0 + cnt; q++) for (int k = 0; k < W; k++) {
            const int nq = nbr[(size_t)W * q + k];
            if (nq < 0) continue;
            if (isExt(q, nq)) {
                if (mailIdx[nq] < 0) { ffm_set_error("internal: tile plan references an unpublished cell"); return FFM_ERR_ADDR; }
                code[(size_t)4 * q + k] = (unsigned short)(T_RING + (int)(ei & 1) * T_XMAX + t); extSrc.push_back(mailIdx[nq]); t++;
            } else code[(size_t)4 * q + k] = (unsigned short)((nq - gs) & (T_RING - 1));
        }
        if (t != (R.y >> 16)) { ffm_set_error("internal: tile plan external count mismatch"); return FFM_ERR_ADDR; }
    }
    D.nEnt = (int)rec.size(); D.nPub = nPub;
    for (int k = 0; k < 2 * T_PF + 2; k++) rec.push_back(make_int4(0, 0, 0, 0));
    for (int k = 0; k < T_THREADS; k++) extSrc.push_back(0);
    FFM_TRY(upv(A->ctx, &D.grpEnt, grpEnt)); FFM_TRY(upv(A->ctx, &D.rec, rec)); FFM_TRY(upv(A->ctx, &D.extSrc, extSrc)); FFM_TRY(upv(A->ctx, &D.code, code)); FFM_TRY(upv(A->ctx, &D.src, src));
    return FFM_OK;
}

// ------------------------------------------------------------------ tiled Amul: plan ---
// y = A x for a symmetric matrix on the tile numbering.  One workgroup streams through one group entry by entry (the
// forward sweep's entries); x and the upper coefficients of the entries [e-3, e+3] live in LDS rings, so a row finds its
// neighbours' x and, for its lower faces, the owner's coefficient there: every coefficient, x, diag is read from memory
// once (56 B per cell + 16 B of codes).  Neighbours outside the window or in another group ("externals", tile faces) are
// fetched by the mail wave from x / the native coefficient array.  Faces towards ghost cells come last in a row's face
// order and are added afterwards by k_amul_tail, so the sum of every row runs in the reference's face order.
constexpr int A_RING = 1024;            // cells in the x / coefficient rings (power of two; >= 8 entries of 256)
constexpr int A_WIN = 1;                // entries on either side served by the rings
constexpr int A_AHEAD = A_WIN + 1;      // the rings are filled this many entries ahead of the entry being computed
constexpr int A_XMAX = 128;             // externals per entry (two per lane of the mail wave)
constexpr int A_PF = 6;                 // read-ahead (entries), > A_AHEAD
constexpr unsigned A_NONE = 0xFFFFu, A_EXT = 0x8000u;      // in-ring code: bits 0-10 cell distance, bits 12-13 owner's slot
static_assert((A_WIN + A_AHEAD + 1) * T_ENT <= A_RING && A_PF > A_AHEAD, "ring window");

static int build_amul(ffm_ldu *A, const std::vector<int> &grpOfCell, const std::vector<int4> &recF)
{
    ffm_tile_plan *T = A->tile;
    const int nOwn = A->nOwned, nEnt = (int)recF.size();
    // ---- cell-space tables of the owned upper neighbours + the faces towards ghost cells (Gauss-Seidel sweeps, Amul tail):
    // structural requirements only (<= 3 owned upper neighbours, ghost neighbours after the owned ones in slot order)
    std::vector<int> tailCell, tailStart(1, 0), tailFace, tailNbr;
    std::vector<int> upSrc((size_t)3 * nOwn, -1), upNb((size_t)3 * nOwn, -1);
    for (int c = 0; c < nOwn; c++) {
        const int sl = c >> 6, lane = c & 63;
        const int uw = (A->h_upOff[sl + 1] - A->h_upOff[sl]) / 64;
        int k = 0;
        bool ghostSeen = false;
        for (int s = 0; s < uw; s++) {
            const int idx = A->h_upOff[sl] + s * 64 + lane, nb = A->h_upNbr[idx];
            if (nb < 0) continue;
            if (nb >= nOwn) {
                if (!ghostSeen) { tailCell.push_back(c); ghostSeen = true; }
                tailFace.push_back(idx); tailNbr.push_back(nb);
                continue;
            }
            if (ghostSeen || k >= 3) return FFM_OK;                      // owned after ghost, or too many: no cell-space tables
            upSrc[(size_t)3 * c + k] = idx; upNb[(size_t)3 * c + k] = nb;
            k++;
        }
        if (ghostSeen) tailStart.push_back((int)tailFace.size());
    }
    FFM_TRY(upv(A->ctx, &T->upSrcCell, upSrc)); FFM_TRY(upv(A->ctx, &T->upNbrCell, upNb));
    T->nTail = (int)tailCell.size();
    FFM_TRY(upv(A->ctx, &T->tailCell, tailCell)); FFM_TRY(upv(A->ctx, &T->tailStart, tailStart)); FFM_TRY(upv(A->ctx, &T->tailFace, tailFace)); FFM_TRY(upv(A->ctx, &T->tailNbr, tailNbr));
    T->gsTables = true;
#define AMUL_GIVE_UP(why) do { if (getenv("FFM_VERBOSE")) fprintf(stderr, "ffm: no ring plan for the tiled Amul (%s, entry %d of %d): row kernel\n", why, e, nEnt); return FFM_OK; } while (0)
    // ---- ring plan of the tiled Amul: may give up (too many out-of-window neighbours in one entry), Amul then runs on the row kernel
    std::vector<int> entOf(nOwn, -1);
    for (int e = 0; e < nEnt; e++) for (int c = recF[e].x; c < recF[e].x + (recF[e].y & 0xFFFF); c++) entOf[c] = e;
    std::vector<unsigned short> code((size_t)8 * nOwn, (unsigned short)A_NONE);
    std::vector<int4> arec(nEnt);
    std::vector<int2> aext;
    auto inRing = [&](int c, int nb) { return grpOfCell[nb] == grpOfCell[c] && std::abs(entOf[nb] - entOf[c]) <= A_WIN; };
    for (int e = 0; e < nEnt; e++) {
        const int c0 = recF[e].x, cnt = recF[e].y & 0xFFFF;
        const int extOff = (int)aext.size();
        int t = 0;
        for (int c = c0; c < c0 + cnt; c++) {
            const int sl = c >> 6, lane = c & 63;
            // lower entries in face order
            const int lw = (A->h_loOff[sl + 1] - A->h_loOff[sl]) / 64;
            int k = 0;
            for (int s = 0; s < lw; s++) {
                const int q = A->h_loEnt[A->h_loOff[sl] + s * 64 + lane];
                if (q < 0) continue;
                const int o = q >> 4, os = q & 15;
                if (k >= 3) { AMUL_GIVE_UP("more than 3 lower neighbours"); }
                if (inRing(c, o) && os < 3 && c - o < 2048) code[(size_t)8 * c + k] = (unsigned short)((c - o) | (os << 12));
                else {
                    if (t >= A_XMAX) { AMUL_GIVE_UP("more than A_XMAX externals in one entry"); }
                    code[(size_t)8 * c + k] = (unsigned short)(A_EXT | t);
                    aext.push_back(make_int2(o, A->h_upOff[o >> 6] + os * 64 + (o & 63))); t++;
                }
                k++;
            }
            // owned upper neighbours in slot order (cell-space tables above)
            for (k = 0; k < 3; k++) {
                const int nb = upNb[(size_t)3 * c + k];
                if (nb < 0) break;
                if (inRing(c, nb) && nb - c < 2048) code[(size_t)8 * c + 3 + k] = (unsigned short)(nb - c);
                else {
                    if (t >= A_XMAX) { AMUL_GIVE_UP("more than A_XMAX externals in one entry"); }
                    code[(size_t)8 * c + 3 + k] = (unsigned short)(A_EXT | t);
                    aext.push_back(make_int2(nb, -1)); t++;
                }
            }
        }
        arec[e] = make_int4(c0, cnt | (t << 16), extOff, 0);
    }
    for (int k = 0; k < 2 * A_PF + 2; k++) arec.push_back(make_int4(0, 0, 0, 0));
    for (int k = 0; k < 2 * A_XMAX; k++) aext.push_back(make_int2(0, -1));
    FFM_TRY(upv(A->ctx, &T->arec, arec)); FFM_TRY(upv(A->ctx, &T->aext, aext));
    FFM_HIP(hipMalloc((void **)&T->acode, sizeof(unsigned short) * 8 * std::max<size_t>(nOwn, 1)));
    FFM_TRY(ffm_h2d(A->ctx, T->acode, code.data(), sizeof(unsigned short) * code.size()));
    // segments: enough workgroups to fill the chip, each long enough to amortise the A_WIN entries read twice at either end
    {
        const std::vector<int> &grpEntH = T->grpEntHost;
        // measured at 400^3 (r02, scripts/tile_probe.py): 16 entries 0.82 ms, 24 0.83, 32 0.84, 64 0.87, a whole tile (430) 0.93, 8 0.94 --
        // short segments balance the 256 CUs better than their two extra ring entries cost
        int L = std::min(20, std::max(16, nEnt / 2048));
        if (const char *e = getenv("FFM_AMUL_SEG")) L = std::max(1, atoi(e));
        std::vector<int4> seg;
        for (int g = 0; g < T->G; g++) for (int a = grpEntH[g]; a < grpEntH[g + 1]; a += L) seg.push_back(make_int4(g, a, std::min(a + L, grpEntH[g + 1]), grpEntH[g + 1]));
        T->nSeg = (int)seg.size();
        FFM_TRY(upv(A->ctx, &T->aseg, seg));
        FFM_HIP(hipMalloc((void **)&T->amulPartials, sizeof(double) * std::max(T->nSeg, 1)));      // x.y of every segment (fused dot)
    }
    T->amulUsable = true;
    return FFM_OK;
}

int ffm_tile_build(ffm_ldu *A, const std::vector<int> &lev, const std::vector<int> &bl, const std::vector<int> &grpCell,
                   const std::vector<int> *bwdCells)
{
    A->tile = new ffm_tile_plan();
    ffm_tile_plan *T = A->tile;
    T->G = (int)grpCell.size() - 1;
    if (T->G <= 0) return FFM_OK;
    const int nOwn = A->nOwned;
    if ((long)A->nCells * 24 >= (1L << 32)) return FFM_OK;         // the kernel addresses its streams with 32-bit byte offsets
    std::vector<int> grpOfCell(nOwn);
    for (int g = 0; g < T->G; g++) for (int c = grpCell[g]; c < grpCell[g + 1]; c++) grpOfCell[c] = g;
    bool ok = true;
    std::vector<int4> recF;
    { FfmStageTimer tm_("tile: forward plan"); FFM_TRY(build_dir(A, true, lev, grpCell, grpOfCell, T->f, ok, &recF, &T->grpEntHost)); }
    T->mirror = bwdCells == nullptr;
    if (ok) {
        FfmStageTimer tm_("tile: backward plan");
        if (T->mirror) FFM_TRY(build_dir(A, false, bl, grpCell, grpOfCell, T->b, ok));
        else { FFM_TRY(build_dir_pos(A, bl, grpCell, grpOfCell, *bwdCells, T->b, ok)); T->cellOf = A->bwdCells; }
    }
    T->usable = ok;
    if (ok) {
        T->nMail = (long)T->f.nPub + T->b.nPub + 2;
        FFM_HIP(hipMalloc((void **)&T->mailAll, sizeof(double) * T->nMail));
        T->f.mail = T->mailAll; T->b.mail = T->mailAll + T->f.nPub + 1;
        FfmStageTimer tm_("tile: amul plan");
        FFM_TRY(build_amul(A, grpOfCell, recF));
        // the Gauss-Seidel sweeps need the cell-space tables: without them the matrix gets level-scheduled sweeps
        if (!T->gsTables) T->usable = false;
    }
    return FFM_OK;
}

// ------------------------------------------------------------------ device ---
struct TileView {
    int G;
    const int *grpCell, *grpEnt, *extSrc;
    const int4 *rec;
    const uint2 *code;
    double *mail;
    unsigned int *ticket;       // [0] ticket counter, [1] abort word
    unsigned long long *trace;  // diagnostics or nullptr
};

__device__ __forceinline__ double t_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void t_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// workgroup barrier that only waits for this wave's LDS traffic: outstanding global loads (the read-ahead) stay in flight
__device__ __forceinline__ void t_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ bool t_pending(double v) { return (unsigned long long)__double_as_longlong(v) == T_SENT; }

enum { TM_FWD = 0, TM_BWD = 1, TM_RD = 2, TM_GSF = 3, TM_GSB = 4 };
struct __attribute__((aligned(8))) T3 { double a, b, c; };       // the three coefficients of a cell
static_assert(T_W == 3, "T3");

// slow path of a mailbox read: re-load until the producer's value has replaced the sentinel; bounded, watches the abort word
template <bool TRACE>
__device__ __noinline__ double t_wait_value(const double *addr, unsigned int *ticket, int *shAbort, unsigned long long *traceWord)
{
    double v = t_ld(addr);
    if (*(volatile int *)shAbort) return v;
    unsigned spins = 0;
    while (t_pending(v)) {
        __builtin_amdgcn_s_sleep(1);
        v = t_ld(addr);
        if ((++spins & 1023u) == 0u) {
            if (__hip_atomic_load(&ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { *shAbort = 1; break; }
            if (spins > T_SPIN_LIMIT) { __hip_atomic_store(&ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *shAbort = 1; break; }
        }
    }
    if (TRACE) atomicAdd(traceWord, (unsigned long long)spins);
    return v;
}

// TM_FWD: w[c] = rD[c]*r[c] - sum_k rD[c]*a[c][k]*w[l_k]           (k ascending: the reference's face order)
// TM_BWD: w[c] -= sum_k rD[c]*a[c][k]*w[u_k]                       (k descending)
// TM_RD : D[c] = diag[c] - sum_k a[c][k]*b[c][k]/D[l_k]            (k ascending; the caller inverts D afterwards)
// TM_GSF: Gauss-Seidel forward row sweep: v = bPrime[c] - sum_k lower[c][k]*psi[l_k] (new values); aux[c] = v (kept for the
//         reverse sweep of symGaussSeidel); v -= b[c][k] for k = 0..2 (b = upper*psi_old of the upper neighbours, computed
//         by k_tile_gs_products; +0.0 where there is none); psi[c] = v/diag[c]
// TM_GSB: reverse row sweep: v = aux[c] - sum_k upper[c][k]*psi[u_k] (new values, k ascending); psi[c] = v/diag[c]
// The per-level path is issue-bound (one wave per SIMD runs the whole entry), so
//   * waves 0-3 ("compute") are kept free of divergent branches: idle lanes work on a dummy cell, only stores are predicated;
//   * wave 4 ("mail") does nothing but bring the external values of the next entry into LDS: entry record -> mailbox slot
//     list -> mailbox values, each stage read ahead, re-loading a value while it still reads the sentinel.  Its few loads
//     have a vmcnt stream of their own, so waiting for a mailbox value never waits for the compute waves' read-ahead.
// Every wave executes the same number of workgroup barriers: one after the prologue, one per entry of the padded loop.
// FUSE (PCG, DIC): the vector updates on either side of the preconditioner ride on the sweeps' own streams --
//   TM_FWD: aux = the residual, updated in place first: rA -= alpha*q (q = A pA: r when given, else w on entry; PCG.C "rA[cell] -= alpha*wA[cell]"),
//           partials[group] = sum |rA| (the residual norm of the iteration that just ended);
//   TM_BWD: partials[group] = sum wA*rA of the finished preconditioned residual (PCG.C wArA = gSumProd(wA, rA)).
// The per-cell arithmetic is that of k_pcg_xr / the unfused sweep; only the order of the two sums differs (as in any reduction).
template <int MODE, bool TRACE, bool POSB = false, bool FUSE = false>
__global__ __launch_bounds__(T_THREADS + 64) void k_tile(TileView t, const double *__restrict__ ca, const double *__restrict__ cb,
                                                         const double *__restrict__ dg, const double *__restrict__ r, double *w, double *aux,
                                                         const double *__restrict__ scal = nullptr, double *__restrict__ partials = nullptr)
{
    static_assert(!FUSE || MODE == TM_FWD || MODE == TM_BWD, "fused vector updates: DIC/DILU application only");
    __shared__ double shSum[4];
    constexpr bool ASC = MODE != TM_BWD && MODE != TM_GSB;
    constexpr bool TWO = MODE == TM_RD || MODE == TM_GSF;       // a second triple of per-cell doubles (cb)
    constexpr int W = T_W;
    constexpr int T_PF = TWO ? 6 : ::T_PF;                      // (register budget) read-ahead of this mode
    __shared__ double ring[T_LDS];                    // ring of this group's values, the two halo buffers, the dummy slot
    double *const halo = ring + T_RING;
    __shared__ int4 shRec[4];           // entry records handed from the mail wave to the compute waves
    __shared__ int shG;
    __shared__ int shAbort;
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        const unsigned tk = atomicAdd(&t.ticket[0], 1u);
        const int k = (int)(tk % (unsigned)t.G);
        shG = ASC ? k : t.G - 1 - k; shAbort = 0;
    }
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(shG);
    const unsigned gs = (unsigned)__builtin_amdgcn_readfirstlane(t.grpCell[g]);
    const int e0 = __builtin_amdgcn_readfirstlane(t.grpEnt[g]), e1 = __builtin_amdgcn_readfirstlane(t.grpEnt[g + 1]);
    if (e0 >= e1) { if (FUSE && tid == 0) partials[g] = 0.0; return; }
    if (TRACE && tid == 0) { t.trace[4 * g] = wall_clock64(); t.trace[4 * g + 3] = 0; }

    if (tid >= (unsigned)T_THREADS) {
        // ------------------------------------------------------------------ mail wave
        const unsigned lane = tid - (unsigned)T_THREADS;
        int4 qrec[T_PF];                // record of entry e + 2*T_PF .. (slot = (e - e0) % T_PF)
        unsigned qne[T_PF], qxi[T_PF];  // externals and this lane's mailbox slot of entry e + T_PF ..
        double qxv[T_PF];               // mailbox value of entry e + T_PM ..
#define Q_REC(k, e) { qrec[k] = t.rec[min((e), e1)]; }
#define Q_IDX(k, e) { const int4 R_ = qrec[k]; qne[k] = ((e) < e1) ? ((unsigned)R_.y >> 16) : 0u; qxi[k] = (unsigned)t.extSrc[(unsigned)R_.w + (lane < qne[k] ? lane : 0u)]; }
#define Q_MAIL(k) { qxv[k] = t_ld(&t.mail[lane < qne[k] ? qxi[k] : 0u]); }
#define Q_PUT(k, e) {                                                                                    \
        if (lane < qne[k]) {                                                                              \
            double v_ = qxv[k];                                                                           \
            if (__builtin_expect(t_pending(v_), 0)) v_ = t_wait_value<TRACE>(&t.mail[qxi[k]], t.ticket, &shAbort, TRACE ? &t.trace[4 * g + 3] : nullptr); \
            halo[(((e) & 1) * T_XMAX) + lane] = v_;                                                       \
        }                                                                                                 \
    }
#pragma unroll
        for (int k = 0; k < T_PF; k++) Q_REC(k, e0 + k);
#pragma unroll
        for (int k = 0; k < T_PF; k++) { Q_IDX(k, e0 + k); Q_REC(k, e0 + T_PF + k); }
#pragma unroll
        for (int k = 0; k < T_PM; k++) Q_MAIL(k);
        Q_PUT(0, e0);
        if (lane == 0) shRec[(e0 + T_PF) & 3] = qrec[0];
        t_barrier();
        for (int e = e0; e < e1; e += T_PF) {
#pragma unroll
            for (int k = 0; k < T_PF; k++) {
                const int ee = e + k;
                Q_PUT((k + 1) % T_PF, ee + 1);          // externals of the next entry
                if (lane == 0) shRec[(ee + 1 + T_PF) & 3] = qrec[(k + 1) % T_PF];     // record the compute waves fetch from next
                Q_IDX(k, ee + T_PF);
                Q_REC(k, ee + 2 * T_PF);
                Q_MAIL((k + T_PM) % T_PF);
                t_barrier();
            }
        }
#undef Q_REC
#undef Q_IDX
#undef Q_MAIL
#undef Q_PUT
        return;             // (the fused sums are reduced by the compute waves alone)
    }

    // ---------------------------------------------------------------------- compute waves
    // read-ahead registers: slot k serves the entries e with (e - e0) % T_PF == k
    unsigned pc[T_PF], ppb[T_PF];
    bool pok[T_PF];
    uint2 pq[T_PF];
    double pa[T_PF][W], pb[T_PF][TWO ? W : 1], pd[T_PF], pv[T_PF], pv2[FUSE ? T_PF : 1];
    double fsum = 0.0;
    const double fAlpha = (FUSE && MODE == TM_FWD) ? scal[S_ALPHA] : 0.0;
    const bool fSing = (FUSE && MODE == TM_FWD) ? (scal[S_SING] != 0.0) : false;
#define T_FETCH(k, e, R_) {                                                                             \
        const unsigned cnt_ = ((e) < e1) ? ((unsigned)R_.y & 0xFFFFu) : 0u;                              \
        const bool ok_ = tid < cnt_;                                                                     \
        const unsigned cc_ = ok_ ? ((ASC || POSB) ? (unsigned)R_.x + tid : (unsigned)R_.x + cnt_ - 1u - tid) : gs; \
        const unsigned o8_ = cc_ * 8u, o24_ = cc_ * 24u;        /* 32-bit byte offsets: arrays < 4 GiB (host check) */  \
        pq[k] = *(const uint2 *)((const char *)t.code + o8_);                                            \
        { const T3 v_ = *(const T3 *)((const char *)ca + o24_); pa[k][0] = v_.a; pa[k][1] = v_.b; pa[k][2] = v_.c; }     \
        if (TWO) { const T3 v_ = *(const T3 *)((const char *)cb + o24_); pb[k][0] = v_.a; pb[k][TWO ? 1 : 0] = v_.b; pb[k][TWO ? 2 : 0] = v_.c; } \
        pd[k] = *(const double *)((const char *)dg + o8_);                                               \
        pv[k] = (FUSE && MODE == TM_FWD) ? *(const double *)((const char *)aux + o8_) : (MODE == TM_FWD || MODE == TM_GSF || MODE == TM_GSB) ? *(const double *)((const char *)r + o8_) : (MODE == TM_BWD ? *(const double *)((const char *)w + o8_) : 0.0); \
        if (FUSE) pv2[FUSE ? k : 0] = (MODE == TM_FWD) ? *(const double *)((const char *)(r ? r : (const double *)w) + o8_) : *(const double *)((const char *)r + o8_); \
        pc[k] = cc_; pok[k] = ok_; ppb[k] = (unsigned)R_.z;                                              \
    }
#pragma unroll
    for (int k = 0; k < T_PF; k++) { const int4 R0 = t.rec[min(e0 + k, e1)]; T_FETCH(k, e0 + k, R0); }
    t_barrier();
    if (TRACE && tid == 0) t.trace[4 * g + 1] = wall_clock64();
    for (int e = e0; e < e1; e += T_PF) {
#pragma unroll
        for (int k = 0; k < T_PF; k++) {
            const int ee = e + k;                           // entries past e1 are padding of the unrolled loop: barriers only
            const int4 Rn = shRec[(ee + T_PF) & 3];         // record of the entry to fetch below (from the mail wave)
            // ---- entry ee from slot k (idle lanes work on the dummy cell gs; nothing of theirs is stored)
            {
                const unsigned c = pc[k];
                const unsigned cd[4] = {pq[k].x & 0xFFFFu, pq[k].x >> 16, pq[k].y & 0xFFFFu, pq[k].y >> 16};
                double x[W];
#pragma unroll
                for (int s = 0; s < W; s++) x[s] = ring[min(cd[s], (unsigned)(T_LDS - 1))];     // the plan stores the LDS slot itself
                const double d = pd[k];
                double val;
                if (MODE == TM_FWD) {
                    double ri = pv[k];
                    if (FUSE) {
                        if (!fSing) { ri -= fAlpha * pv2[FUSE ? k : 0]; if (pok[k]) *(double *)((char *)aux + c * 8u) = ri; }
                        fsum += pok[k] ? fabs(ri) : 0.0;
                    }
                    val = d * ri;
#pragma unroll
                    for (int s = 0; s < W; s++) { const double nv = val - d * pa[k][s] * x[s]; val = (cd[s] != T_NONE) ? nv : val; }
                } else if (MODE == TM_BWD) {
                    val = pv[k];
#pragma unroll
                    for (int s = W - 1; s >= 0; s--) { const double nv = val - d * pa[k][s] * x[s]; val = (cd[s] != T_NONE) ? nv : val; }
                } else if (MODE == TM_RD) {
                    val = d;
#pragma unroll
                    for (int s = 0; s < W; s++) { const double nv = val - pa[k][s] * pb[k][TWO ? s : 0] / x[s]; val = (cd[s] != T_NONE) ? nv : val; }
                } else if (MODE == TM_GSF) {
                    val = pv[k];
#pragma unroll
                    for (int s = 0; s < W; s++) { const double nv = val - pa[k][s] * x[s]; val = (cd[s] != T_NONE) ? nv : val; }
                    if (pok[k]) *(double *)((char *)aux + c * 8u) = val;
#pragma unroll
                    for (int s = 0; s < W; s++) val -= pb[k][TWO ? s : 0];
                    val = val / d;
                } else {
                    val = pv[k];
#pragma unroll
                    for (int s = 0; s < W; s++) { const double nv = val - pa[k][s] * x[s]; val = (cd[s] != T_NONE) ? nv : val; }
                    val = val / d;
                }
                if (pok[k]) { *(double *)((char *)w + c * 8u) = val; ring[(c - gs) & (unsigned)(T_RING - 1)] = val; }
                if (FUSE && MODE == TM_BWD) fsum += pok[k] ? val * pv2[FUSE ? k : 0] : 0.0;
                if (pok[k] && cd[3] != T_NONE) t_st(&t.mail[ppb[k] + cd[3]], val);
            }
            // ---- refill slot k with entry ee + T_PF
            T_FETCH(k, ee + T_PF, Rn);
            t_barrier();                                    // one barrier per level: ring and halo are visible to the next entry
        }
    }
    if (TRACE && tid == 0) t.trace[4 * g + 2] = wall_clock64();
#undef T_FETCH
    if (FUSE) {             // the four compute waves (the mail wave has left): wave sums through LDS, lane 0 adds them in wave order
        double v = fsum;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63u) == 0u) shSum[tid >> 6] = v;
        t_barrier();
        if (tid == 0) partials[g] = ((shSum[0] + shSum[1]) + shSum[2]) + shSum[3];
    }
}

// out[i] = native[src[i]] (0 where a cell has fewer than T_W neighbours): coefficients in the kernel's cell-major layout
__global__ void k_tile_gather(long n, const int *__restrict__ src, const double *__restrict__ native, double *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int q = src[i];
        out[i] = (q >= 0) ? native[q] : 0.0;
    }
}
// mailbox fill before a sweep (pair); it also zeroes the group ticket counter, so that the 32-bit counter never wraps and every
// launch hands its groups out as ticket % G with ticket < 2 G (forward sweep 0..G-1, backward sweep G..2G-1)
__global__ void k_tile_fill(long n, unsigned long long *p, unsigned long long v, unsigned int *ticket)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ticket[0] = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// GATHER: out[q] = in[cellOf[q]]; else scatter: out[cellOf[q]] = in[q]
template <bool GATHER>
__global__ void k_tile_permute(long n, const int *__restrict__ cellOf, const double *__restrict__ in, double *__restrict__ out)
{
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        if (GATHER) out[q] = in[cellOf[q]]; else out[cellOf[q]] = in[q];
    }
}

static TileView tview(const ffm_ldu *A, const TileDir &d)
{
    TileView t; t.G = A->tile->G; t.grpCell = A->grpCell; t.grpEnt = d.grpEnt; t.extSrc = d.extSrc; t.rec = d.rec;
    t.code = (const uint2 *)d.code; t.mail = d.mail; t.ticket = A->sweepTicket; t.trace = A->tile->trace;
    return t;
}

// gathered coefficients of one direction, refreshed when the matrix coefficients changed
static int tile_coef(ffm_ldu *A, TileDir &d, bool upper, const double **out)
{
    const long n = (long)T_W * A->nOwned;
    double *&buf = upper ? d.coefU : d.coefL;
    unsigned long &ep = upper ? d.epochU : d.epochL;
    if (!buf) { FFM_HIP(hipMalloc((void **)&buf, sizeof(double) * std::max<long>(n, 1))); ep = ~0ul; }
    if (ep != A->offDiagEpoch) {
        const int g = std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS));
        hipLaunchKernelGGL(k_tile_gather, dim3(g), dim3(256), 0, A->ctx->stream, n, d.src, upper ? A->upper : A->lower, buf);
        ep = A->offDiagEpoch;
    }
    *out = buf;
    return FFM_OK;
}

static void tile_fill(ffm_ldu *A, double *p, long n)
{
    const int g = std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS));
    hipLaunchKernelGGL(k_tile_fill, dim3(g), dim3(256), 0, A->ctx->stream, n, (unsigned long long *)p, T_SENT, A->sweepTicket);
}

// upper coefficients of the owned upper neighbours in cell space: the backward direction's gathered array on mirror-ordered
// meshes, a gather of its own otherwise
static int tile_up_coef_cell(ffm_ldu *A, const double **out)
{
    ffm_tile_plan *T = A->tile;
    if (T->mirror) return tile_coef(A, T->b, true, out);
    const long n = (long)T_W * A->nOwned;
    if (!T->upCoefCell) { FFM_HIP(hipMalloc((void **)&T->upCoefCell, sizeof(double) * std::max<long>(n, 1))); T->upCoefEpoch = ~0ul; }
    if (T->upCoefEpoch != A->offDiagEpoch) {
        hipLaunchKernelGGL(k_tile_gather, dim3(std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS))), dim3(256), 0, A->ctx->stream, n,
                           (const int *)T->upSrcCell, (const double *)A->upper, T->upCoefCell);
        T->upCoefEpoch = A->offDiagEpoch;
    }
    *out = T->upCoefCell;
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
    tile_fill(A, T->mailAll, T->nMail);
    if (!T->mirror) {
        // backward sweep in position space: rD and the forward result are gathered to positions, the result scattered back
        const long n = A->nOwned;
        const int g = std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS));
        if (!T->wp) { FFM_HIP(hipMalloc((void **)&T->wp, sizeof(double) * std::max<long>(n, 1))); FFM_HIP(hipMalloc((void **)&T->rDp, sizeof(double) * std::max<long>(n, 1))); }
        if (T->rDpEpoch != A->rDEpoch || T->rDpKind != A->rDKind) {
            hipLaunchKernelGGL(k_tile_permute<true>, dim3(g), dim3(256), 0, s, n, T->cellOf, (const double *)A->rD, T->rDp);
            T->rDpEpoch = A->rDEpoch; T->rDpKind = A->rDKind;
        }
        hipLaunchKernelGGL((k_tile<TM_FWD, false>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->f), cf, (const double *)nullptr, (const double *)A->rD, r, w, (double *)nullptr);
        hipLaunchKernelGGL(k_tile_permute<true>, dim3(g), dim3(256), 0, s, n, T->cellOf, (const double *)w, T->wp);
        hipLaunchKernelGGL((k_tile<TM_BWD, false, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cb, (const double *)nullptr, (const double *)T->rDp, r, T->wp, (double *)nullptr);
        hipLaunchKernelGGL(k_tile_permute<false>, dim3(g), dim3(256), 0, s, n, T->cellOf, (const double *)T->wp, w);
        FFM_HIP(hipGetLastError());
        return FFM_OK;
    }
    if (T->trace) {
        hipLaunchKernelGGL((k_tile<TM_FWD, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->f), cf, (const double *)nullptr, (const double *)A->rD, r, w, (double *)nullptr);
        hipLaunchKernelGGL((k_tile<TM_BWD, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cb, (const double *)nullptr, (const double *)A->rD, r, w, (double *)nullptr);
    } else {
        hipLaunchKernelGGL((k_tile<TM_FWD, false>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->f), cf, (const double *)nullptr, (const double *)A->rD, r, w, (double *)nullptr);
        hipLaunchKernelGGL((k_tile<TM_BWD, false>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cb, (const double *)nullptr, (const double *)A->rD, r, w, (double *)nullptr);
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

__global__ void k_tile_sum_partials(int n, const double *__restrict__ partials, double *__restrict__ scal, int slot);
// PCG with DIC on a mirror-ordered tile plan: the two halves of the preconditioner application with the neighbouring vector
// updates fused in (k_tile<.., FUSE>).  Forward: rA -= scal[S_ALPHA]*wA, scal[slot] = sum |rA| (local), wA = forward sweep of rA.
// Backward: finishes wA, scal[slot] = sum wA*rA (local).
bool ffm_tile_pcg_fusable(const ffm_ldu *A)
{
    const char *e = getenv("FFM_PCG_UNFUSED");
    const bool off = e && atoi(e) != 0;
    return !off && ffm_tile_usable(A) && A->tile->mirror && !A->tile->trace && A->tile->G <= 4 * RED_BLOCKS;
}
int ffm_tile_pcg_fwd(ffm_ldu *A, double *rA, double *wA, int slot, const double *qA)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *cf, *cb;
    FFM_TRY(tile_coef(A, T->f, true, &cf));
    FFM_TRY(tile_coef(A, T->b, true, &cb));
    tile_fill(A, T->mailAll, T->nMail);
    hipLaunchKernelGGL((k_tile<TM_FWD, false, false, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->f), cf, (const double *)nullptr,
                       (const double *)A->rD, qA, wA, rA, (const double *)A->ctx->scal_d, A->ctx->partials_d);
    hipLaunchKernelGGL(k_tile_sum_partials, dim3(1), dim3(1024), 0, s, T->G, (const double *)A->ctx->partials_d, A->ctx->scal_d, slot);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
int ffm_tile_pcg_bwd(ffm_ldu *A, const double *rA, double *wA, int slot)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *cb;
    FFM_TRY(tile_coef(A, T->b, true, &cb));
    hipLaunchKernelGGL((k_tile<TM_BWD, false, false, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cb, (const double *)nullptr,
                       (const double *)A->rD, rA, wA, (double *)nullptr, (const double *)A->ctx->scal_d, A->ctx->partials_d);
    hipLaunchKernelGGL(k_tile_sum_partials, dim3(1), dim3(1024), 0, s, T->G, (const double *)A->ctx->partials_d, A->ctx->scal_d, slot);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// upper*psi_old of the (owned) upper neighbours of every cell, slot by slot; +0.0 where a cell has fewer than three
__global__ void k_tile_gs_products(long n3, const int *__restrict__ nbrCell, const double *__restrict__ coef, const double *__restrict__ psi,
                                   double *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (long)gridDim.x * blockDim.x) {
        const int nb = nbrCell[i];
        out[i] = (nb >= 0) ? coef[i] * psi[nb] : 0.0;
    }
}

bool ffm_tile_gs_usable(const ffm_ldu *A) { return ffm_tile_usable(A) && A->tile->gsTables; }

// the abort word of the sweep kernels (a bounded mailbox wait ran out): reported when a solve ends
int ffm_tile_check_abort(ffm_ldu *A)
{
    unsigned int h[2] = {0, 0};
    FFM_HIP(hipMemcpyAsync(h, A->sweepTicket, sizeof(h), hipMemcpyDeviceToHost, A->ctx->stream));
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (h[1]) {
        ffm_set_error("tiled sweep timed out waiting for a value of a predecessor group (abort word set)");
        unsigned int z = 0;
        ffm_h2d(A->ctx, A->sweepTicket + 1, &z, sizeof(z));
        return FFM_ERR_HIP;
    }
    return FFM_OK;
}

// One GaussSeidelSmoother / symGaussSeidelSmoother sweep (forward rows, then reverse rows when sym): psi in place, bP = bPrime
// (source with the lagged interface terms), bSave = scratch [nCells].  prod = scratch [3*nCells].
int ffm_tile_gs(ffm_ldu *A, bool sym, double *psi, const double *bP, double *bSave, double *prod)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *cl, *cuCell, *cuB;
    FFM_TRY(tile_coef(A, T->f, A->lower == A->upper, &cl));        // lower coefficients, lower-neighbour layout
    FFM_TRY(tile_up_coef_cell(A, &cuCell));                         // upper coefficients, cell space
    FFM_TRY(tile_coef(A, T->b, true, &cuB));                        // upper coefficients in the backward direction's layout
    const long n = A->nOwned, n3 = (long)T_W * n;
    const int g1 = std::max(1, std::min(ffm_grid(n, 256), 8 * RED_BLOCKS));
    hipLaunchKernelGGL(k_tile_gs_products, dim3(std::max(1, std::min(ffm_grid(n3, 256), 8 * RED_BLOCKS))), dim3(256), 0, s, n3,
                       (const int *)T->upNbrCell, cuCell, (const double *)psi, prod);
    tile_fill(A, T->mailAll, T->nMail);
    hipLaunchKernelGGL((k_tile<TM_GSF, false>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->f), cl, (const double *)prod, (const double *)A->diag, bP, psi, bSave);
    if (sym && T->mirror)
        hipLaunchKernelGGL((k_tile<TM_GSB, false>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cuB, (const double *)nullptr, (const double *)A->diag,
                           (const double *)bSave, psi, (double *)nullptr);
    else if (sym) {
        // reverse sweep in position space: bSave and diag gathered to positions, psi scattered back
        if (!T->wp) { FFM_HIP(hipMalloc((void **)&T->wp, sizeof(double) * std::max<long>(n, 1))); FFM_HIP(hipMalloc((void **)&T->rDp, sizeof(double) * std::max<long>(n, 1))); T->rDpEpoch = ~0ul; }
        if (!T->diagp) { FFM_HIP(hipMalloc((void **)&T->diagp, sizeof(double) * std::max<long>(n, 1))); T->diagpEpoch = ~0ul; }
        if (T->diagpEpoch != A->coeffEpoch) {
            hipLaunchKernelGGL(k_tile_permute<true>, dim3(g1), dim3(256), 0, s, n, T->cellOf, (const double *)A->diag, T->diagp);
            T->diagpEpoch = A->coeffEpoch;
        }
        double *bSp = prod;                                         // the products are consumed: reuse their first n doubles
        hipLaunchKernelGGL(k_tile_permute<true>, dim3(g1), dim3(256), 0, s, n, T->cellOf, (const double *)bSave, bSp);
        hipLaunchKernelGGL((k_tile<TM_GSB, false, true>), dim3(T->G), dim3(T_THREADS + 64), 0, s, tview(A, T->b), cuB, (const double *)nullptr, (const double *)T->diagp,
                           (const double *)bSp, T->wp, (double *)nullptr);
        hipLaunchKernelGGL(k_tile_permute<false>, dim3(g1), dim3(256), 0, s, n, T->cellOf, (const double *)T->wp, psi);
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// D = diag - sum upper*lower/D[l] in face order (un-inverted; the caller inverts)
int ffm_tile_calc_rD(ffm_ldu *A)
{
    ffm_tile_plan *T = A->tile;
    const double *cu, *cl;
    FFM_TRY(tile_coef(A, T->f, true, &cu));
    if (A->lower == A->upper) cl = cu; else FFM_TRY(tile_coef(A, T->f, false, &cl));
    tile_fill(A, T->f.mail, (long)T->f.nPub + 1);
    hipLaunchKernelGGL((k_tile<TM_RD, false>), dim3(T->G), dim3(T_THREADS + 64), 0, A->ctx->stream, tview(A, T->f), cu, cl, (const double *)A->diag,
                       (const double *)nullptr, A->rD, (double *)nullptr);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ------------------------------------------------------------------ several systems in one sweep ---
// The segregated solves of a vector equation (fvMatrix::solveSegregated: the components of U differ in the boundary diagonal and the
// source only) and the species equations under a multivariateSelection scheme (solver/YEEqn.H:43-60: one set of face weights, one
// diffusivity) have the SAME off-diagonal coefficients.  k_tile_m sweeps NF such systems at once: neighbour codes and coefficients are
// read once, every system has its own ring in LDS, its own mailboxes and its own rD / right-hand side / result; a level costs one
// barrier for all of them and their dependency chains interleave.  Per system the arithmetic is k_tile's, operation for operation.
constexpr int TM_PF = 4;            // read-ahead of the multi-system sweeps (NF systems' operands per slot)
static_assert(T_PM < TM_PF, "read-ahead distances");
template <int NF> struct TileMulti { const double *dg[NF]; const double *r[NF]; double *w[NF]; double *mail[NF]; };

template <int MODE, int NF>
__global__ __launch_bounds__(T_THREADS + 64) void k_tile_m(TileView t, const double *__restrict__ ca, const double *__restrict__ cb, TileMulti<NF> m)
{
    static_assert(MODE == TM_FWD || MODE == TM_BWD || MODE == TM_RD, "DILU / DIC application and calcReciprocalD");
    constexpr bool ASC = MODE != TM_BWD;
    constexpr bool TWO = MODE == TM_RD;
    constexpr int W = T_W, PF = TM_PF;
    __shared__ double ring[NF][T_LDS];
    __shared__ int4 shRec[4];
    __shared__ int shG;
    __shared__ int shAbort;
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        const unsigned tk = atomicAdd(&t.ticket[0], 1u);
        const int k = (int)(tk % (unsigned)t.G);
        shG = ASC ? k : t.G - 1 - k; shAbort = 0;
    }
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(shG);
    const unsigned gs = (unsigned)__builtin_amdgcn_readfirstlane(t.grpCell[g]);
    const int e0 = __builtin_amdgcn_readfirstlane(t.grpEnt[g]), e1 = __builtin_amdgcn_readfirstlane(t.grpEnt[g + 1]);
    if (e0 >= e1) return;

    if (tid >= (unsigned)T_THREADS) {
        // ------------------------------------------------------------------ mail wave (as in k_tile, NF values per external)
        const unsigned lane = tid - (unsigned)T_THREADS;
        int4 qrec[PF];
        unsigned qne[PF], qxi[PF];
        double qxv[PF][NF];
#define Q_REC(k, e) { qrec[k] = t.rec[min((e), e1)]; }
#define Q_IDX(k, e) { const int4 R_ = qrec[k]; qne[k] = ((e) < e1) ? ((unsigned)R_.y >> 16) : 0u; qxi[k] = (unsigned)t.extSrc[(unsigned)R_.w + (lane < qne[k] ? lane : 0u)]; }
#define Q_MAIL(k) { _Pragma("unroll") for (int i_ = 0; i_ < NF; i_++) qxv[k][i_] = t_ld(&m.mail[i_][lane < qne[k] ? qxi[k] : 0u]); }
#define Q_PUT(k, e) {                                                                                    \
        if (lane < qne[k]) {                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NF; i_++) {                                           \
                double v_ = qxv[k][i_];                                                                   \
                if (__builtin_expect(t_pending(v_), 0)) v_ = t_wait_value<false>(&m.mail[i_][qxi[k]], t.ticket, &shAbort, nullptr); \
                ring[i_][T_RING + (((e) & 1) * T_XMAX) + lane] = v_;                                      \
            }                                                                                             \
        }                                                                                                 \
    }
#pragma unroll
        for (int k = 0; k < PF; k++) Q_REC(k, e0 + k);
#pragma unroll
        for (int k = 0; k < PF; k++) { Q_IDX(k, e0 + k); Q_REC(k, e0 + PF + k); }
#pragma unroll
        for (int k = 0; k < T_PM; k++) Q_MAIL(k);
        Q_PUT(0, e0);
        if (lane == 0) shRec[(e0 + PF) & 3] = qrec[0];
        t_barrier();
        for (int e = e0; e < e1; e += PF) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int ee = e + k;
                Q_PUT((k + 1) % PF, ee + 1);
                if (lane == 0) shRec[(ee + 1 + PF) & 3] = qrec[(k + 1) % PF];
                Q_IDX(k, ee + PF);
                Q_REC(k, ee + 2 * PF);
                Q_MAIL((k + T_PM) % PF);
                t_barrier();
            }
        }
#undef Q_REC
#undef Q_IDX
#undef Q_MAIL
#undef Q_PUT
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    unsigned pc[PF], ppb[PF];
    bool pok[PF];
    uint2 pq[PF];
    double pa[PF][W], pb[PF][TWO ? W : 1], pd[PF][NF], pv[PF][TWO ? 1 : NF];
#define TM_FETCH(k, e, R_) {                                                                            \
        const unsigned cnt_ = ((e) < e1) ? ((unsigned)R_.y & 0xFFFFu) : 0u;                              \
        const bool ok_ = tid < cnt_;                                                                     \
        const unsigned cc_ = ok_ ? (ASC ? (unsigned)R_.x + tid : (unsigned)R_.x + cnt_ - 1u - tid) : gs; \
        const unsigned o8_ = cc_ * 8u, o24_ = cc_ * 24u;                                                 \
        pq[k] = *(const uint2 *)((const char *)t.code + o8_);                                            \
        { const T3 v_ = *(const T3 *)((const char *)ca + o24_); pa[k][0] = v_.a; pa[k][1] = v_.b; pa[k][2] = v_.c; }     \
        if (TWO) { const T3 v_ = *(const T3 *)((const char *)cb + o24_); pb[k][0] = v_.a; pb[k][TWO ? 1 : 0] = v_.b; pb[k][TWO ? 2 : 0] = v_.c; } \
        _Pragma("unroll") for (int i_ = 0; i_ < NF; i_++) {                                              \
            pd[k][i_] = *(const double *)((const char *)m.dg[i_] + o8_);                                 \
            if (!TWO) pv[k][TWO ? 0 : i_] = *(const double *)((const char *)(MODE == TM_FWD ? m.r[i_] : (const double *)m.w[i_]) + o8_); \
        }                                                                                                \
        pc[k] = cc_; pok[k] = ok_; ppb[k] = (unsigned)R_.z;                                              \
    }
#pragma unroll
    for (int k = 0; k < PF; k++) { const int4 R0 = t.rec[min(e0 + k, e1)]; TM_FETCH(k, e0 + k, R0); }
    t_barrier();
    for (int e = e0; e < e1; e += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int ee = e + k;
            const int4 Rn = shRec[(ee + PF) & 3];
            {
                const unsigned c = pc[k];
                const unsigned cd[4] = {pq[k].x & 0xFFFFu, pq[k].x >> 16, pq[k].y & 0xFFFFu, pq[k].y >> 16};
                unsigned sl[W];
#pragma unroll
                for (int s = 0; s < W; s++) sl[s] = min(cd[s], (unsigned)(T_LDS - 1));
                double val[NF];
#pragma unroll
                for (int i = 0; i < NF; i++) {
                    const double d = pd[k][i];
                    double v;
                    if (MODE == TM_FWD) {
                        v = d * pv[k][TWO ? 0 : i];
#pragma unroll
                        for (int s = 0; s < W; s++) { const double nv = v - d * pa[k][s] * ring[i][sl[s]]; v = (cd[s] != T_NONE) ? nv : v; }
                    } else if (MODE == TM_BWD) {
                        v = pv[k][TWO ? 0 : i];
#pragma unroll
                        for (int s = W - 1; s >= 0; s--) { const double nv = v - d * pa[k][s] * ring[i][sl[s]]; v = (cd[s] != T_NONE) ? nv : v; }
                    } else {
                        v = d;
#pragma unroll
                        for (int s = 0; s < W; s++) { const double nv = v - pa[k][s] * pb[k][TWO ? s : 0] / ring[i][sl[s]]; v = (cd[s] != T_NONE) ? nv : v; }
                    }
                    val[i] = v;
                }
                if (pok[k]) {
#pragma unroll
                    for (int i = 0; i < NF; i++) {
                        *(double *)((char *)m.w[i] + c * 8u) = val[i];
                        ring[i][(c - gs) & (unsigned)(T_RING - 1)] = val[i];
                        if (cd[3] != T_NONE) t_st(&m.mail[i][ppb[k] + cd[3]], val[i]);
                    }
                }
            }
            TM_FETCH(k, ee + PF, Rn);
            t_barrier();
        }
    }
#undef TM_FETCH
}


bool ffm_tile_multi_usable(const ffm_ldu *A)
{
    const char *e = getenv("FFM_NO_MULTI_SWEEP");
    return !(e && atoi(e) != 0) && ffm_tile_usable(A) && A->tile->mirror && !A->tile->trace;
}
static int tile_multi_mail(ffm_ldu *A)
{
    ffm_tile_plan *T = A->tile;
    if (!T->mailMulti) FFM_HIP(hipMalloc((void **)&T->mailMulti, sizeof(double) * (size_t)FFM_TILE_MAXSYS * (size_t)T->nMail));
    return FFM_OK;
}
template <int MODE, int NF>
static void tile_multi_launch(ffm_ldu *A, const TileDir &d, const double *ca, const double *cb, const double *const *dg, const double *const *r, double *const *w)
{
    ffm_tile_plan *T = A->tile;
    TileMulti<NF> m;
    for (int i = 0; i < NF; i++) { m.dg[i] = dg[i]; m.r[i] = r ? r[i] : nullptr; m.w[i] = w[i]; m.mail[i] = T->mailMulti + (size_t)i * T->nMail + (d.mail - T->mailAll); }
    hipLaunchKernelGGL((k_tile_m<MODE, NF>), dim3(T->G), dim3(T_THREADS + 64), 0, A->ctx->stream, tview(A, d), ca, cb, m);
}
#define TILE_MULTI(MODE, n, ...) switch (n) { case 2: tile_multi_launch<MODE, 2>(__VA_ARGS__); break; case 3: tile_multi_launch<MODE, 3>(__VA_ARGS__); break; \
                                             default: tile_multi_launch<MODE, 4>(__VA_ARGS__); break; }
// DILU / DIC application (not transposed) of n = 2 .. FFM_TILE_MAXSYS systems with the matrix's off-diagonal coefficients and their
// own reciprocal diagonals: w[i] = precondition(r[i]).  Mirror-ordered tile plans (ffm_tile_multi_usable).
int ffm_tile_precond_multi(ffm_ldu *A, int precond, int n, const double *const *rD, const double *const *r, double *const *w)
{
    ffm_tile_plan *T = A->tile;
    if (n < 2 || n > FFM_TILE_MAXSYS || !ffm_tile_multi_usable(A)) return FFM_ERR_ARG;
    const double *cf, *cb;
    FFM_TRY(tile_coef(A, T->f, precond == FFM_DIC, &cf));
    FFM_TRY(tile_coef(A, T->b, true, &cb));
    FFM_TRY(tile_multi_mail(A));
    tile_fill(A, T->mailMulti, (long)n * T->nMail);
    TILE_MULTI(TM_FWD, n, A, T->f, cf, nullptr, rD, r, w);
    TILE_MULTI(TM_BWD, n, A, T->b, cb, nullptr, rD, nullptr, w);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
// D[i] = diag[i] - sum upper*lower/D[i][l] in face order (un-inverted) for n systems
int ffm_tile_calc_rD_multi(ffm_ldu *A, int n, const double *const *diag, double *const *D)
{
    ffm_tile_plan *T = A->tile;
    if (n < 2 || n > FFM_TILE_MAXSYS || !ffm_tile_multi_usable(A)) return FFM_ERR_ARG;
    const double *cu, *cl;
    FFM_TRY(tile_coef(A, T->f, true, &cu));
    if (A->lower == A->upper) cl = cu; else FFM_TRY(tile_coef(A, T->f, false, &cl));
    FFM_TRY(tile_multi_mail(A));
    tile_fill(A, T->mailMulti, (long)n * T->nMail);
    TILE_MULTI(TM_RD, n, A, T->f, cu, cl, diag, nullptr, D);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}


// ------------------------------------------------------------------ tiled Amul: kernels ---
struct AmulView {
    int G;
    const int *grpCell, *grpEnt;
    const int4 *rec, *seg;
    const uint4 *code;
    const int2 *ext;
};

// y[c] = diag[c]*x[c] + sum_lower coef*x[l] + sum_upper(owned) coef*x[u]   (lduMatrix::Amul row order); DOT: partial of x.y
// FUSEP (PCG): x is not read but formed on the way into the rings, x = w + beta*pin (PCG.C "pA = wA + beta*pA"), and stored to pout
// by the segment that owns the row, together with psi += alpha*pin (the solution update left over from the iteration before);
// pin / pout are two buffers (a neighbouring workgroup may still need the old direction), y must not alias w.
// ASYM (asymmetric matrices: the transport equations): the coefficient of a lower face is not the owner's upper coefficient from
// the ring but the cell's own triple lc[3c + k] of lower coefficients (the array the forward DILU sweep streams, gathered once
// per coefficient update); x still comes through the rings / external buffers.
template <bool DOT, bool FUSEP = false, bool ASYM = false>
__global__ __launch_bounds__(T_THREADS + 64) void k_tile_amul(AmulView t, const double *__restrict__ bc, const double *__restrict__ upper,
                                                              const double *__restrict__ diag, const double *__restrict__ x,
                                                              double *__restrict__ y, double *__restrict__ partials,
                                                              const double *__restrict__ pin = nullptr, double *__restrict__ pout = nullptr,
                                                              double *__restrict__ psi = nullptr, const double *__restrict__ scal = nullptr,
                                                              const double *__restrict__ lc = nullptr)
{
    // FUSEP: x = w on entry.  A singular flag leaves direction and solution as they are (k_p_psi's guard)
    const bool fSing = FUSEP ? (scal[S_SING] != 0.0) : false;
    const double fBeta = FUSEP ? scal[S_BETA] : 0.0, fAlpha = FUSEP ? scal[S_ALPHA] : 0.0;
    __shared__ double xring[A_RING + 2 * A_XMAX];           // x of the window, then the two external-x buffers
    __shared__ double cring[3 * A_RING + 2 * A_XMAX];       // upper coefficients of the window, then the external coefficients
    __shared__ int4 shRec[4];
    __shared__ double sm[8];
    const unsigned tid = threadIdx.x;
    // segment [sa, sb) of group g: rows of these entries are computed; the rings also need the A_WIN entries on either side
    const int4 S = t.seg[blockIdx.x];
    const int g = S.x, sa = S.y, sb = S.z;
    const unsigned gs = (unsigned)t.grpCell[g];
    const int e0 = max(t.grpEnt[g], sa - A_WIN), e1 = min(S.w, sb + A_WIN);     // entries loaded
    double dot = 0.0;
    if (sa < sb) {
        if (tid >= (unsigned)T_THREADS) {
            // ---------------------------------------------------------------- mail wave: externals of the next entry
            const unsigned lane = tid - (unsigned)T_THREADS;
            int4 qrec[A_PF];
            unsigned qne[A_PF];
            int2 qi0[A_PF], qi1[A_PF];
            double qx0[A_PF], qx1[A_PF], qc0[A_PF], qc1[A_PF], qp0[FUSEP ? A_PF : 1], qp1[FUSEP ? A_PF : 1];
#define Q_REC(k, e) { qrec[k] = t.rec[min((e), e1)]; }
#define Q_IDX(k, e) { const int4 R_ = qrec[k]; qne[k] = ((e) < e1) ? ((unsigned)R_.y >> 16) : 0u;                       \
                      qi0[k] = t.ext[(unsigned)R_.z + (lane < qne[k] ? lane : 0u)];                                     \
                      qi1[k] = t.ext[(unsigned)R_.z + (lane + 64u < qne[k] ? lane + 64u : 0u)]; }
#define Q_VAL(k) { qx0[k] = x[qi0[k].x]; qx1[k] = x[qi1[k].x]; qc0[k] = upper[max(qi0[k].y, 0)]; qc1[k] = upper[max(qi1[k].y, 0)];     \
                   if (FUSEP) { qp0[FUSEP ? k : 0] = pin[qi0[k].x]; qp1[FUSEP ? k : 0] = pin[qi1[k].x]; } }
#define Q_X0(k) (FUSEP ? (fSing ? qp0[FUSEP ? k : 0] : qx0[k] + fBeta * qp0[FUSEP ? k : 0]) : qx0[k])
#define Q_X1(k) (FUSEP ? (fSing ? qp1[FUSEP ? k : 0] : qx1[k] + fBeta * qp1[FUSEP ? k : 0]) : qx1[k])
#define Q_PUT(k, e) { const unsigned hb_ = (unsigned)((e) & 1) * (unsigned)A_XMAX;                                      \
                      if (lane < qne[k]) { xring[A_RING + hb_ + lane] = Q_X0(k); cring[3 * A_RING + hb_ + lane] = qc0[k]; } \
                      if (lane + 64u < qne[k]) { xring[A_RING + hb_ + lane + 64u] = Q_X1(k); cring[3 * A_RING + hb_ + lane + 64u] = qc1[k]; } }
#pragma unroll
            for (int k = 0; k < A_PF; k++) Q_REC(k, e0 + k);
#pragma unroll
            for (int k = 0; k < A_PF; k++) { Q_IDX(k, e0 + k); Q_REC(k, e0 + A_PF + k); }
#pragma unroll
            for (int k = 0; k < T_PM; k++) Q_VAL(k);
            Q_PUT(0, e0);
            if (lane == 0) shRec[(e0 + A_PF) & 3] = qrec[0];
            t_barrier();
            for (int e = e0; e < sb; e += A_PF) {
#pragma unroll
                for (int k = 0; k < A_PF; k++) {
                    const int ee = e + k;
                    Q_PUT((k + 1) % A_PF, ee + 1);
                    if (lane == 0) shRec[(ee + 1 + A_PF) & 3] = qrec[(k + 1) % A_PF];
                    Q_IDX(k, ee + A_PF);
                    Q_REC(k, ee + 2 * A_PF);
                    Q_VAL((k + T_PM) % A_PF);
                    t_barrier();
                }
            }
#undef Q_REC
#undef Q_IDX
#undef Q_VAL
#undef Q_PUT
#undef Q_X0
#undef Q_X1
        } else {
            // ---------------------------------------------------------------- compute waves
            unsigned pc[A_PF];
            bool pok[A_PF], pst[A_PF];
            uint4 pq[A_PF];
            double pb[A_PF][3], pd[A_PF], px[A_PF], pp[FUSEP ? A_PF : 1], ps[FUSEP ? A_PF : 1], pl[ASYM ? A_PF : 1][3];
#define A_FETCH(k, e, R_) {                                                                              \
        const unsigned cnt_ = ((e) < e1) ? ((unsigned)(R_).y & 0xFFFFu) : 0u;                            \
        const bool ok_ = tid < cnt_;                                                                     \
        const unsigned cc_ = ok_ ? (unsigned)(R_).x + tid : gs;                                          \
        const unsigned o8_ = cc_ * 8u;                                                                   \
        pq[k] = *(const uint4 *)((const char *)t.code + cc_ * 16u);                                      \
        { const T3 v_ = *(const T3 *)((const char *)bc + cc_ * 24u); pb[k][0] = v_.a; pb[k][1] = v_.b; pb[k][2] = v_.c; } \
        pd[k] = *(const double *)((const char *)diag + o8_);                                             \
        px[k] = *(const double *)((const char *)x + o8_);                                                \
        if (FUSEP) { pp[FUSEP ? k : 0] = *(const double *)((const char *)pin + o8_); ps[FUSEP ? k : 0] = *(const double *)((const char *)psi + o8_); } \
        if (ASYM) { const T3 v_ = *(const T3 *)((const char *)lc + cc_ * 24u); pl[ASYM ? k : 0][0] = v_.a; pl[ASYM ? k : 0][1] = v_.b; pl[ASYM ? k : 0][2] = v_.c; } \
        pc[k] = cc_; pok[k] = ok_; pst[k] = ok_ && (e) >= sa && (e) < sb;                                \
    }
#define A_FILL(k) { if (FUSEP) px[k] = fSing ? pp[FUSEP ? k : 0] : px[k] + fBeta * pp[FUSEP ? k : 0];                      \
                    if (pok[k]) { const unsigned i_ = (pc[k] - gs) & (unsigned)(A_RING - 1); xring[i_] = px[k];    \
                                   cring[3u * i_] = pb[k][0]; cring[3u * i_ + 1u] = pb[k][1]; cring[3u * i_ + 2u] = pb[k][2]; } }
#pragma unroll
            for (int k = 0; k < A_PF; k++) { const int4 R0 = t.rec[min(e0 + k, e1)]; A_FETCH(k, e0 + k, R0); }
#pragma unroll
            for (int k = 0; k < A_AHEAD; k++) A_FILL(k);
            t_barrier();
            for (int e = e0; e < sb; e += A_PF) {
#pragma unroll
                for (int k = 0; k < A_PF; k++) {
                    const int ee = e + k;
                    const int4 Rn = shRec[(ee + A_PF) & 3];
                    {
                        const unsigned c = pc[k];
                        const unsigned hb = (unsigned)(ee & 1) * (unsigned)A_XMAX;
                        const unsigned cd[6] = {pq[k].x & 0xFFFFu, pq[k].x >> 16, pq[k].y & 0xFFFFu, pq[k].y >> 16, pq[k].z & 0xFFFFu, pq[k].z >> 16};
                        const double xc = px[k];
                        double acc = pd[k] * xc;
#pragma unroll
                        for (int s = 0; s < 3; s++) {       // lower faces: x and the coefficient from the owner's ring slot
                            const unsigned m = (unsigned)((int)(cd[s] << 16) >> 31);
                            const unsigned ri = (c - (cd[s] & 0x7FFu) - gs) & (unsigned)(A_RING - 1);
                            const unsigned hi = hb + (cd[s] & (unsigned)(A_XMAX - 1));
                            const unsigned xi = ri ^ ((ri ^ ((unsigned)A_RING + hi)) & m);
                            const unsigned ci = (3u * ri + ((cd[s] >> 12) & 3u)) ^ (((3u * ri + ((cd[s] >> 12) & 3u)) ^ (3u * (unsigned)A_RING + hi)) & m);
                            const double nv = acc + (ASYM ? pl[ASYM ? k : 0][s] : cring[ci]) * xring[xi];
                            acc = (cd[s] != A_NONE) ? nv : acc;
                        }
#pragma unroll
                        for (int s = 0; s < 3; s++) {       // upper faces: own coefficient
                            const unsigned u = cd[3 + s];
                            const unsigned m = (unsigned)((int)(u << 16) >> 31);
                            const unsigned ri = (c + (u & 0x7FFu) - gs) & (unsigned)(A_RING - 1);
                            const unsigned xi = ri ^ ((ri ^ ((unsigned)A_RING + hb + (u & (unsigned)(A_XMAX - 1)))) & m);
                            const double nv = acc + pb[k][s] * xring[xi];
                            acc = (u != A_NONE) ? nv : acc;
                        }
                        if (pst[k]) __builtin_nontemporal_store(acc, (double *)((char *)y + c * 8u));
                        if (FUSEP && pst[k]) {
                            *(double *)((char *)pout + c * 8u) = xc;
                            if (!fSing) *(double *)((char *)psi + c * 8u) = ps[FUSEP ? k : 0] + fAlpha * pp[FUSEP ? k : 0];
                        }
                        if (DOT) dot += pst[k] ? acc * xc : 0.0;
                    }
                    A_FILL((k + A_AHEAD) % A_PF);           // rings: entry ee + A_AHEAD
                    A_FETCH(k, ee + A_PF, Rn);
                    t_barrier();
                }
            }
#undef A_FETCH
#undef A_FILL
        }
    }
    if (DOT) {
        const double rsum = block_sum(dot, sm);
        if (tid == 0) partials[blockIdx.x] = rsum;
    }
}

// faces towards ghost cells, in slot order, added to the finished owned part of the row
template <bool SUB>
__global__ void k_amul_tail(int n, const int *__restrict__ cell, const int *__restrict__ start, const int *__restrict__ face,
                            const int *__restrict__ nbr, const double *__restrict__ upper, const double *__restrict__ x, double *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell[i];
    double acc = y[c];
    for (int q = start[i]; q < start[i + 1]; q++) { const double t = upper[face[q]] * x[nbr[q]]; acc = SUB ? acc - t : acc + t; }
    y[c] = acc;
}

// Gauss-Seidel on a block with ghost cells: bPrime = source - (faces towards ghost cells)*psi_ghost, the lagged explicit
// treatment OpenFOAM gives coupled patches (GaussSeidelSmoother: bPrime = source; updateMatrixInterfaces with negated coeffs)
int ffm_tile_gs_ghost_terms(ffm_ldu *A, const double *psi, double *bP)
{
    ffm_tile_plan *T = A->tile;
    if (T->nTail > 0)
        hipLaunchKernelGGL(k_amul_tail<true>, dim3((T->nTail + 255) / 256), dim3(256), 0, A->ctx->stream, T->nTail, (const int *)T->tailCell,
                           (const int *)T->tailStart, (const int *)T->tailFace, (const int *)T->tailNbr, (const double *)A->upper, psi, bP);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

__global__ __launch_bounds__(1024) void k_tile_sum_partials(int n, const double *__restrict__ partials, double *__restrict__ scal, int slot)
{
    __shared__ double sm[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) scal[slot] = r;
}

bool ffm_tile_amul_usable(const ffm_ldu *A) { return ffm_tile_usable(A) && A->tile->amulUsable && A->symmetric && A->ifaces.empty(); }

// y = A x for an asymmetric matrix on the tile plan (transport equations): upper coefficients from the cell-space upper table,
// lower coefficients from the forward sweep's gathered array; ghost faces by the tail kernel as in the symmetric case
bool ffm_tile_amul_asym_usable(const ffm_ldu *A)
{
    return ffm_tile_usable(A) && A->tile->amulUsable && !A->symmetric && A->ifaces.empty() && !getenv("FFM_NO_TILE_AMUL_ASYM");
}
int ffm_tile_amul_asym(ffm_ldu *A, const double *x, double *y)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *bcoef, *lcoef;
    FFM_TRY(tile_up_coef_cell(A, &bcoef));
    FFM_TRY(tile_coef(A, T->f, false, &lcoef));
    AmulView v; v.G = T->G; v.grpCell = A->grpCell; v.grpEnt = T->f.grpEnt; v.rec = T->arec; v.seg = T->aseg; v.code = T->acode; v.ext = T->aext;
    hipLaunchKernelGGL((k_tile_amul<false, false, true>), dim3(T->nSeg), dim3(T_THREADS + 64), 0, s, v, bcoef, (const double *)A->upper, (const double *)A->diag, x, y,
                       (double *)nullptr, (const double *)nullptr, (double *)nullptr, (double *)nullptr, (const double *)nullptr, lcoef);
    FFM_TRY(ffm_ghost_exchange_end(A));
    if (T->nTail > 0)
        hipLaunchKernelGGL(k_amul_tail<false>, dim3((T->nTail + 255) / 256), dim3(256), 0, s, T->nTail, (const int *)T->tailCell, (const int *)T->tailStart,
                           (const int *)T->tailFace, (const int *)T->tailNbr, (const double *)A->upper, x, y);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// PCG: pout = w + beta*pin, psi += alpha*pin, y = A pout, scal[dotSlot] = pout.y (local), in one pass (k_tile_amul<true, true>)
bool ffm_tile_amul_pcg_usable(const ffm_ldu *A)
{
    return ffm_tile_amul_usable(A) && A->tile->nTail == 0 && A->ghNbrRank.empty();
}
int ffm_tile_amul_pcg(ffm_ldu *A, const double *w, const double *pin, double *pout, double *psi, double *y, int dotSlot)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *bcoef;
    FFM_TRY(tile_up_coef_cell(A, &bcoef));
    AmulView v; v.G = T->G; v.grpCell = A->grpCell; v.grpEnt = T->f.grpEnt; v.rec = T->arec; v.seg = T->aseg; v.code = T->acode; v.ext = T->aext;
    hipLaunchKernelGGL((k_tile_amul<true, true>), dim3(T->nSeg), dim3(T_THREADS + 64), 0, s, v, bcoef, (const double *)A->upper, (const double *)A->diag, w, y,
                       T->amulPartials, pin, pout, psi, (const double *)A->ctx->scal_d);
    hipLaunchKernelGGL(k_tile_sum_partials, dim3(1), dim3(1024), 0, s, T->nSeg, (const double *)T->amulPartials, A->ctx->scal_d, dotSlot);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// y = A x (symmetric A); dotSlot >= 0: also scal[dotSlot] = x.y over the owned rows (local part)
int ffm_tile_amul(ffm_ldu *A, const double *x, double *y, int dotSlot)
{
    ffm_tile_plan *T = A->tile;
    hipStream_t s = A->ctx->stream;
    const double *bcoef;
    FFM_TRY(tile_up_coef_cell(A, &bcoef));
    AmulView v; v.G = T->G; v.grpCell = A->grpCell; v.grpEnt = T->f.grpEnt; v.rec = T->arec; v.seg = T->aseg; v.code = T->acode; v.ext = T->aext;
    const bool fusedDot = dotSlot >= 0 && T->nTail == 0;
    if (fusedDot) {
        FFM_TRY(ffm_ghost_exchange_end(A));          // (no ghost faces on this rank: nothing in flight that the kernel would need)
        hipLaunchKernelGGL((k_tile_amul<true>), dim3(T->nSeg), dim3(T_THREADS + 64), 0, s, v, bcoef, (const double *)A->upper, (const double *)A->diag, x, y, T->amulPartials);
        hipLaunchKernelGGL(k_tile_sum_partials, dim3(1), dim3(1024), 0, s, T->nSeg, (const double *)T->amulPartials, A->ctx->scal_d, dotSlot);
    } else {
        hipLaunchKernelGGL((k_tile_amul<false>), dim3(T->nSeg), dim3(T_THREADS + 64), 0, s, v, bcoef, (const double *)A->upper, (const double *)A->diag, x, y, (double *)nullptr);
        FFM_TRY(ffm_ghost_exchange_end(A));          // an overlapped ghost refresh (ffm_ghost_exchange_begin) must have landed before the tail
        if (T->nTail > 0)
            hipLaunchKernelGGL(k_amul_tail<false>, dim3((T->nTail + 255) / 256), dim3(256), 0, s, T->nTail, (const int *)T->tailCell, (const int *)T->tailStart,
                               (const int *)T->tailFace, (const int *)T->tailNbr, (const double *)A->upper, x, y);
    }
    FFM_HIP(hipGetLastError());
    return fusedDot || dotSlot < 0 ? FFM_OK : 1;       // 1: the caller still has to take the dot product
}

// Diagnostics: the first call switches tracing on; later calls copy out, for the LAST tiled launch, 4 words per group:
// wall_clock64 (100 MHz) at start, after the first entry's externals, at the end, and the number of mailbox re-loads.
extern "C" int ffm_debug_tile_trace(ffm_ldu *A, unsigned long long *out, int nWords)
{
    if (!A || !ffm_tile_usable(A)) return FFM_ERR_ARG;
    ffm_tile_plan *T = A->tile;
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    if (!T->trace) {
        FFM_HIP(hipMalloc((void **)&T->trace, sizeof(unsigned long long) * 4 * T->G));
        FFM_TRY(ffm_dzero(A->ctx, T->trace, sizeof(unsigned long long) * 4 * T->G)); FFM_HIP(hipStreamSynchronize(A->ctx->stream));
        return T->G;
    }
    if (out && nWords > 0) FFM_TRY(ffm_d2h(A->ctx, out, T->trace, sizeof(unsigned long long) * std::min(nWords, 4 * T->G)));
    return T->G;
}

extern "C" int ffm_debug_set_sweep_ticket(ffm_ldu *A, unsigned int value)
{
    if (!A || !A->sweepTicket) return FFM_ERR_ARG;
    FFM_HIP(hipStreamSynchronize(A->ctx->stream));
    FFM_TRY(ffm_h2d(A->ctx, A->sweepTicket, &value, sizeof(value)));
    return FFM_OK;
}

// Diagnostics: resident workgroups per CU the runtime predicts for the forward / backward sweep kernels and the tiled Amul
extern "C" int ffm_debug_tile_occupancy(int *out3)
{
    if (!out3) return FFM_ERR_ARG;
    FFM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[0], k_tile<TM_FWD, false, false>, T_THREADS + 64, 0));
    FFM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[1], k_tile<TM_BWD, false, false>, T_THREADS + 64, 0));
    FFM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], k_tile_amul<true>, T_THREADS + 64, 0));
    return FFM_OK;
}
