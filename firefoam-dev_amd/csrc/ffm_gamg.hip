// ffm_gamg.hip -- GAMG (geometric-algebraic multigrid) as the reference's dictionaries select it:
//   p_rgh / ph_rgh: solver GAMG, smoother GaussSeidel, agglomerator faceAreaPair, mergeLevels 1, nCellsInCoarsestLevel 10,
//                   cacheAgglomeration true          (cases/wallFireSpread2D/system/fvSolution:36-60)
//   Ii:             solver GAMG, smoother DILU        (cases/steckler/system/fvSolution:63-73)
// The algorithm is OpenFOAM-dev's (GAMGSolver, pairGAMGAgglomeration, GAMGAgglomeration; not part of the reference tree); the
// oracle restatement with its sources is oracle/gamg.py, which this file is tested against.
//
// Layout.  The agglomeration (host, once per mesh: cacheAgglomeration) yields, per level, a restriction map and a coarse LDU
// addressing; every coarse level is an ffm_ldu of its own, so smoothing, Amul, residuals and the coarsest-level Krylov solve
// are the kernels of ffm_ldu / ffm_solve / ffm_tile on that level's matrix.  All level vectors live in the INTERNAL cell order
// of their ffm_ldu (the dependency-level numbering the sweeps need); the maps between levels are stored in those orders, while
// their entry lists keep the reference's summation order (fine cells / fine faces ascending in the level's own numbering):
//   restrictField    cf[c] = sum of ff over the fine cells of c           one thread per coarse cell, sequential
//   prolongField     ff[i] = cf[map[i]]
//   agglomerateMatrix coarse diag / upper / lower from the fine coefficients in caller (agglomeration) order, then handed to
//                    ffm_ldu_set_coeffs_d of the coarse level
// The V-cycle is launched level by level from the host; nothing is read back inside it (the correction scale factors stay on
// the device) except by the coarsest-level solver, which checks its own convergence.
//
// Decomposed meshes (OpenFOAM's default, no processorAgglomerator: cases/wallFireSpread2D/system/fvSolution:36-60 on 4 ranks): every rank
// agglomerates ITS OWN cells over its internal faces (pairGAMGAgglomeration on the rank's lduAddressing); the processor interfaces are
// agglomerated with them (GAMGInterface / processorGAMGInterface: a coarse interface face per pair of coarse cells either side, its
// coefficient the sum of its fine faces').  In this library's ghost-cell form: a level has owned coarse cells + coarse GHOST cells -- one
// per distinct coarse cell of a neighbour rank that touches this rank, numbered rank by rank in the order of first appearance along the
// fine receive list -- the fine cut faces agglomerate into coarse cut faces, and the level's ghost exchange sends, per neighbour, the
// coarse cells of the fine send list in the order of first appearance (the same order on both sides, because a receive list mirrors
// the neighbour's send list).  The neighbour's restriction map reaches this rank through the fine level's ghost exchange.  Smoothers,
// Amul and the coarsest-level Krylov solve then are the decomposed forms of ffm_ldu / ffm_solve (ghost refresh before every sweep /
// product, block-Jacobi DIC / DILU, all-reduced dots); continueAgglomerating, normFactor, residual norms and the correction scale
// factors are global (all-reduced).
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <unordered_map>

extern "C" int ffm_solve_internal_i(ffm_ldu *A, int solver, int precond, double tol, double relTol, int minIter, int maxIter, int nSweeps,
                                    double *psi, const double *source, ffm_perf *perf);

namespace {

constexpr double GREAT_ = 1e15;

struct Level {              // matrix k (0 = the caller's matrix); maps lead to matrix k + 1
    ffm_ldu *A = nullptr;
    bool owned = false;
    int nCells = 0, nFaces = 0;             // OWNED cells; faces including the cut faces towards ghost cells
    int nGhost = 0;                         // ghost cells behind the owned ones (decomposed meshes)
    std::vector<int> l, u;                  // caller (agglomeration) order
    // ---- device, towards the coarser level
    int nCoarse = 0, nCoarseFaces = 0;
    int *rStart = nullptr, *rItem = nullptr;        // [nCoarse+1], [nCells]: fine INTERNAL cells of every coarse INTERNAL cell, ascending caller index
    int *toCoarse = nullptr;                        // [nCells] fine internal -> coarse internal
    // coefficient agglomeration, caller orders
    int *cStart = nullptr, *cItem = nullptr;        // fine caller cells of every coarse caller cell
    int *iStart = nullptr, *iItem = nullptr;        // interior fine faces of every coarse caller cell
    int *fStart = nullptr, *fItem = nullptr;        // fine faces (| flip << 31) of every coarse face
    // ---- coefficients of this matrix in caller order (level 0: the caller's arrays), its diagonal in internal order
    double *diag = nullptr, *upper = nullptr, *lower = nullptr;
    double *dInt = nullptr;
    // ---- vectors (internal order)
    double *src = nullptr, *corr = nullptr, *acf = nullptr, *pre = nullptr, *tmp = nullptr;
};

template <class T> int up(ffm_ctx *c, T **d, const std::vector<T> &v) { return ffm_upload_vec(c, d, v); }
int dalloc(double **d, size_t n) { FFM_HIP(hipMalloc((void **)d, sizeof(double) * std::max<size_t>(n, 1))); return FFM_OK; }

// ------------------------------------------------------------------ host: agglomeration
// pairGAMGAgglomeration::agglomerate(nCoarseCells, addressing, faceWeights)
void pair_agglomerate(int nFine, const std::vector<int> &l, const std::vector<int> &u, const std::vector<double> &w, bool forward,
                      std::vector<int> &cmap, int &nCoarse)
{
    const int nF = (int)l.size();
    std::vector<int> off(nFine + 1, 0), cnt(nFine, 0), cellFaces(2 * (size_t)nF);
    for (int f = 0; f < nF; f++) { off[u[f] + 1]++; off[l[f] + 1]++; }
    for (int c = 0; c < nFine; c++) off[c + 1] += off[c];
    for (int f = 0; f < nF; f++) cellFaces[off[u[f]] + cnt[u[f]]++] = f;
    for (int f = 0; f < nF; f++) cellFaces[off[l[f]] + cnt[l[f]]++] = f;
    cmap.assign(nFine, -1);
    nCoarse = 0;
    for (int ci = 0; ci < nFine; ci++) {
        const int c = forward ? ci : nFine - ci - 1;
        if (cmap[c] >= 0) continue;
        int match = -1; double best = -GREAT_;
        for (int q = off[c]; q < off[c + 1]; q++) {
            const int f = cellFaces[q];
            if (cmap[u[f]] < 0 && cmap[l[f]] < 0 && w[f] > best) { match = f; best = w[f]; }
        }
        if (match >= 0) { cmap[u[match]] = nCoarse; cmap[l[match]] = nCoarse; nCoarse++; }
        else {
            match = -1; best = -GREAT_;
            for (int q = off[c]; q < off[c + 1]; q++) { const int f = cellFaces[q]; if (w[f] > best) { match = f; best = w[f]; } }
            if (match >= 0) cmap[c] = std::max(cmap[u[match]], cmap[l[match]]);
        }
    }
    for (int ci = 0; ci < nFine; ci++) { const int c = forward ? ci : nFine - ci - 1; if (cmap[c] < 0) cmap[c] = nCoarse++; }
    if (!forward) for (int c = 0; c < nFine; c++) cmap[c] = (nCoarse - 1) - cmap[c];
}

// GAMGAgglomeration::agglomerateLduAddressing: coarse faces in order of discovery per coarse owner
void agglomerate_addressing(const std::vector<int> &l, const std::vector<int> &u, const std::vector<int> &rmap, int nCoarse,
                            std::vector<int> &cl, std::vector<int> &cu, std::vector<int> &fra, std::vector<char> &flip, int nCoarseOwned = -1)
{
    const int nF = (int)l.size();
    int maxN = 10;
    std::vector<int> cCellnFaces(nCoarse, 0), cCellFaces((size_t)maxN * nCoarse), initNbr;
    fra.assign(nF, 0); initNbr.reserve(nF / 2 + 1);
    for (int f = 0; f < nF; f++) {
        const int ru = rmap[u[f]], rl = rmap[l[f]];
        if (ru == rl) { fra[f] = -(ru + 1); continue; }
        const int own = std::min(ru, rl), nei = std::max(ru, rl);
        int *cc = &cCellFaces[(size_t)maxN * own];
        bool found = false;
        for (int i = 0; i < cCellnFaces[own]; i++) if (initNbr[cc[i]] == nei) { found = true; fra[f] = cc[i]; break; }
        if (found) continue;
        if (cCellnFaces[own] >= maxN) {
            const int oldN = maxN; maxN *= 2;
            std::vector<int> grown((size_t)maxN * nCoarse);
            for (int c = 0; c < nCoarse; c++) std::copy(&cCellFaces[(size_t)oldN * c], &cCellFaces[(size_t)oldN * c] + cCellnFaces[c], &grown[(size_t)maxN * c]);
            cCellFaces.swap(grown);
            cc = &cCellFaces[(size_t)maxN * own];
        }
        cc[cCellnFaces[own]++] = (int)initNbr.size();
        fra[f] = (int)initNbr.size();
        initNbr.push_back(nei);
    }
    const int nCF = (int)initNbr.size();
    cl.resize(nCF); cu.resize(nCF);
    std::vector<int> fmap(nCF);
    int k = 0;
    // decomposed levels (nCoarseOwned given): the faces of an owner towards OWNED cells first, then its cut faces towards ghost cells, each
    // group in order of discovery -- the layout of every decomposed ffm_ldu (the packed lower entries need the owned faces in the first 16 slots)
    for (int c = 0; c < nCoarse; c++) for (int pass = 0; pass < (nCoarseOwned >= 0 ? 2 : 1); pass++) for (int i = 0; i < cCellnFaces[c]; i++) {
        const int cf = cCellFaces[(size_t)maxN * c + i];
        if (nCoarseOwned >= 0 && (initNbr[cf] >= nCoarseOwned) != (pass == 1)) continue;
        cl[k] = c; cu[k] = initNbr[cf]; fmap[cf] = k++;
    }
    flip.assign(nF, 0);
    for (int f = 0; f < nF; f++) if (fra[f] >= 0) { fra[f] = fmap[fra[f]]; flip[f] = rmap[u[f]] < rmap[l[f]]; }
}

// CSR of `item` grouped by key[item] (items ascending inside a group)
void group_by(int nKeys, int nItems, const std::vector<int> &key, std::vector<int> &start, std::vector<int> &items,
              const std::vector<int> *value = nullptr)
{
    start.assign(nKeys + 1, 0);
    for (int i = 0; i < nItems; i++) if (key[i] >= 0) start[key[i] + 1]++;
    for (int k = 0; k < nKeys; k++) start[k + 1] += start[k];
    items.assign(start[nKeys], 0);
    std::vector<int> pos(start.begin(), start.end() - 1);
    for (int i = 0; i < nItems; i++) if (key[i] >= 0) items[pos[key[i]]++] = value ? (*value)[i] : i;
}

// ------------------------------------------------------------------ device kernels
#define GS_LOOP(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void k_restrict(long nCoarse, const int *__restrict__ start, const int *__restrict__ item, const double *__restrict__ ff, double *__restrict__ cf)
{
    GS_LOOP(c, nCoarse) {
        double s = 0.0;
        for (int q = start[c]; q < start[c + 1]; q++) s += ff[item[q]];
        cf[c] = s;
    }
}
__global__ void k_prolong(long nFine, const int *__restrict__ map, const double *__restrict__ cf, double *__restrict__ ff)
{ GS_LOOP(i, nFine) ff[i] = cf[map[i]]; }

// coarse diagonal: the fine diagonals of the cluster (restrictField), then its interior faces in fine-face order
__global__ void k_agg_diag(long nCoarse, const int *__restrict__ cStart, const int *__restrict__ cItem, const int *__restrict__ iStart,
                           const int *__restrict__ iItem, const double *__restrict__ fd, const double *__restrict__ fu, const double *fl, double *__restrict__ cd)
{
    GS_LOOP(c, nCoarse) {
        double s = 0.0;
        for (int q = cStart[c]; q < cStart[c + 1]; q++) s += fd[cItem[q]];
        for (int q = iStart[c]; q < iStart[c + 1]; q++) { const int f = iItem[q]; s += fl ? fu[f] + fl[f] : 2 * fu[f]; }
        cd[c] = s;
    }
}
__global__ void k_agg_faces(long nCF, const int *__restrict__ fStart, const int *__restrict__ fItem, const double *__restrict__ fu, const double *fl,
                            double *__restrict__ cu, double *cl)
{
    GS_LOOP(k, nCF) {
        double su = 0.0, sl = 0.0;
        for (int q = fStart[k]; q < fStart[k + 1]; q++) {
            const int e = fItem[q], f = e & 0x7FFFFFFF; const bool flip = e < 0;
            if (fl) { su += flip ? fl[f] : fu[f]; sl += flip ? fu[f] : fl[f]; }
            else su += fu[f];
        }
        cu[k] = su; if (cl) cl[k] = sl;
    }
}
__global__ void k_gather_d(long n, const int *__restrict__ perm, const double *__restrict__ in, double *__restrict__ out)
{ GS_LOOP(i, n) out[i] = perm ? in[perm[i]] : in[i]; }
__global__ void k_zero(long n, double *__restrict__ x) { GS_LOOP(i, n) x[i] = 0.0; }
__global__ void k_add_to(long n, double *__restrict__ x, const double *__restrict__ y) { GS_LOOP(i, n) x[i] += y[i]; }
__global__ void k_sub_to(long n, double *__restrict__ x, const double *__restrict__ y) { GS_LOOP(i, n) x[i] -= y[i]; }
__global__ void k_sub3(long n, double *__restrict__ r, const double *__restrict__ b, const double *__restrict__ w) { GS_LOOP(i, n) r[i] = b[i] - w[i]; }
// GAMGSolver::scale: sf = (source.field)/stabilise(Acf.field, vSmall); field = sf*field + (source - sf*Acf)/D
__global__ void k_scale_factor(double *scal)
{
    if (threadIdx.x || blockIdx.x) return;
    const double num = scal[S_TMP0], den = scal[S_TMP1];
    const double st = (den < 0) ? ((den > -1e-300) ? -1e-300 : den) : ((den < 1e-300) ? 1e-300 : den);
    scal[S_TMP2] = num / st;
}
__global__ void k_scale(long n, double *__restrict__ field, const double *__restrict__ acf, const double *__restrict__ source,
                        const double *__restrict__ D, const double *__restrict__ scal)
{
    const double sf = scal[S_TMP2];
    GS_LOOP(i, n) field[i] = sf * field[i] + (source[i] - sf * acf[i]) / D[i];
}
// normFactor partial: |Apsi - xRef*sumA| + |source - xRef*sumA|, xRef = scal[S_TMP0]/nGlobal
__global__ __launch_bounds__(256) void k_gamg_normf(long n, const double *__restrict__ Ax, const double *__restrict__ b, const double *__restrict__ sumA,
                                                    const double *__restrict__ scal, double nGlobal, double *__restrict__ partials)
{
    __shared__ double sm[4];
    const double xRef = scal[S_TMP0] / nGlobal;
    double acc = 0.0;
    GS_LOOP(i, n) { const double t = sumA[i] * xRef; acc += fabs(Ax[i] - t) + fabs(b[i] - t); }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
__global__ __launch_bounds__(1024) void k_sum_partials(int n, const double *__restrict__ partials, double *__restrict__ scal, int slot)
{
    __shared__ double sm[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) scal[slot] = r;
}

// sum of a few host values over the ranks (set-up decisions: cell counts), through the context's scalar slots
int ffm_host_allreduce_sum(ffm_ctx *c, double *v, int n)
{
    if (c->nRanks <= 1 && !c->comm) return FFM_OK;
    if (n > 4) return FFM_ERR_ARG;
    FFM_TRY(ffm_h2d(c, c->scal_d + S_TMP0, v, sizeof(double) * n));
    FFM_TRY(ffm_allreduce_slots(c, S_TMP0, n));
    return ffm_d2h(c, v, c->scal_d + S_TMP0, sizeof(double) * n);
}

inline int grid_of(long n) { return (int)std::max(1L, std::min((n + 255) / 256, (long)RED_BLOCKS)); }

}  // namespace

struct ffm_gamg {
    ffm_ctx *ctx = nullptr;
    std::vector<Level> lev;             // lev[0]: the caller's matrix; lev[k]: coarse level k-1 of the reference's numbering
    bool symmetric = true, haveMatrix = false, decomposed = false;
    int nPre = 0, nPost = 2, nFinest = 2, preMul = 1, maxPre = 4, postMul = 1, maxPost = 4;
    double *fRes = nullptr, *fCorr = nullptr, *fApsi = nullptr, *fSumA = nullptr;      // finest-level work (internal order)
    int *faceToNative = nullptr;            // [nFaces] caller face -> native coefficient index (ffm_gamg_set_matrix_native_d)
    double *upC = nullptr, *loC = nullptr;  // the finest coefficients gathered back to caller face order
    std::vector<ffm_perf> coarsestLog;
};

static void free_level(Level &L)
{
    hipFree(L.rStart); hipFree(L.rItem); hipFree(L.toCoarse); hipFree(L.cStart); hipFree(L.cItem); hipFree(L.iStart); hipFree(L.iItem);
    hipFree(L.fStart); hipFree(L.fItem); hipFree(L.dInt); hipFree(L.src); hipFree(L.corr); hipFree(L.acf); hipFree(L.pre); hipFree(L.tmp);
    if (L.owned) { hipFree(L.diag); hipFree(L.upper); hipFree(L.lower); if (L.A) ffm_ldu_destroy(L.A); }
}

extern "C" int ffm_gamg_destroy(ffm_gamg *G)
{
    if (!G) return FFM_OK;
    for (auto &L : G->lev) free_level(L);
    hipFree(G->fRes); hipFree(G->fCorr); hipFree(G->fApsi); hipFree(G->fSumA);
    hipFree(G->faceToNative); hipFree(G->upC); hipFree(G->loC);
    delete G;
    return FFM_OK;
}

// faceAreaPairGAMGAgglomeration: mag(cmptMultiply(Sf/sqrt(mag(Sf)), (1, 1.01, 1.02)))
extern "C" int ffm_gamg_face_area_pair_weights(int nFaces, const double *Sf, double *w)
{
    if (nFaces < 0 || (nFaces && (!Sf || !w))) return FFM_ERR_ARG;
    for (int f = 0; f < nFaces; f++) {
        const double x = Sf[3 * f], y = Sf[3 * f + 1], z = Sf[3 * f + 2];
        const double m = std::sqrt(std::sqrt((x * x + y * y) + z * z));
        const double a = x / m * 1.0, b = y / m * 1.01, c = z / m * 1.02;
        w[f] = std::sqrt((a * a + b * b) + c * c);
    }
    return FFM_OK;
}

// Build the level hierarchy for the matrix `finest` (created from the same lowerAddr / upperAddr) and one ffm_ldu per coarse level.
extern "C" int ffm_gamg_create(ffm_ctx *ctx, ffm_ldu *finest, int nCells, int nFaces, const int *lowerAddr, const int *upperAddr,
                               const double *faceWeights, int nCellsInCoarsestLevel, int mergeLevels, ffm_gamg **out)
{
    if (!ctx || !finest || !out || nCells < 1 || nFaces < 0 || (nFaces && (!lowerAddr || !upperAddr || !faceWeights)) || nCellsInCoarsestLevel < 1) {
        ffm_set_error("ffm_gamg_create: bad argument"); return FFM_ERR_ARG;
    }
    if (mergeLevels != 1) { ffm_set_error("GAMG: mergeLevels %d not offered (the reference's dictionaries use 1)", mergeLevels); return FFM_ERR_UNSUPPORTED; }
    if (finest->nCells != nCells) { ffm_set_error("GAMG: the matrix has %d cells (owned + ghost), the addressing %d", finest->nCells, nCells); return FFM_ERR_ARG; }
    const bool decomposed = finest->nCells > finest->nOwned;
    if (decomposed && finest->ghNbrRank.empty()) { ffm_set_error("GAMG: ghost cells without a ghost exchange (ffm_ldu_set_ghost_exchange first)"); return FFM_ERR_ARG; }
    FFM_HIP(hipSetDevice(ctx->device));
    ffm_gamg *G = new ffm_gamg; G->ctx = ctx;
    auto fail = [&](int rc) { ffm_gamg_destroy(G); return rc; };
    G->lev.emplace_back();
    { Level &L0 = G->lev[0]; L0.A = finest; L0.nCells = finest->nOwned; L0.nGhost = finest->nCells - finest->nOwned; L0.nFaces = nFaces; L0.l.assign(lowerAddr, lowerAddr + nFaces); L0.u.assign(upperAddr, upperAddr + nFaces); }
    std::vector<double> w(faceWeights, faceWeights + nFaces);
    // Tile plans for the coarse levels: when the finest matrix runs the tiled wavefront kernels, a coarse cell inherits the tile
    // (group) of its first fine cell.  On a hex box the pairs of the first few agglomerations lie inside the tiles, the coarse
    // levels keep <= 3 lower / upper neighbours per cell and an acyclic tile graph, and their smoother sweeps run as tiled
    // wavefronts instead of the dataflow sweeps (2 us per dependency level); wherever the planner cannot use the inherited groups
    // ffm_ldu_create_hint falls back to the level-scheduled form by itself.  FFM_GAMG_TILES=0 switches the inheritance off.
    std::vector<int> hint;                  // group label per cell of the current level (caller order); empty: no inheritance
    {
        const char *e = getenv("FFM_GAMG_TILES");
        if (!decomposed && finest->sweepMode == 2 && finest->grpCell && finest->nGroups > 0 && !(e && atoi(e) == 0)) {
            std::vector<int> gc((size_t)finest->nGroups + 1);
            FFM_HIP(hipStreamSynchronize(ctx->stream));              // the table was uploaded on the context's (non-blocking) stream
            FFM_TRY(ffm_d2h(ctx, gc.data(), finest->grpCell, sizeof(int) * gc.size()));
            hint.assign(nCells, 0);
            for (int g = 0; g < finest->nGroups; g++)
                for (int n = gc[g]; n < gc[g + 1] && n < nCells; n++) hint[finest->identity ? n : finest->h_newToOldCell[n]] = g;
        }
    }
    long tileMinCells = 2048;               // below this a level is a handful of tiles (measured: 2048 ... 8192 equal, 32768 slower)
    if (const char *e = getenv("FFM_GAMG_TILE_MIN_CELLS")) tileMinCells = atol(e);
    bool forward = ctx->gamgForward;        // pairGAMGAgglomeration::forward_: static upstream, here kept per context (= per run)
    const int maxLevels = 50;
    while ((int)G->lev.size() - 1 < maxLevels - 1) {
        // NOTE: G->lev may reallocate below; take no references across emplace_back
        const int k = (int)G->lev.size() - 1;
        const int nFine = G->lev[k].nCells, nFineGhost = G->lev[k].nGhost, nF = (int)G->lev[k].l.size();
        std::vector<int> cmap; int nCoarse = 0;
        if (!decomposed) pair_agglomerate(nFine, G->lev[k].l, G->lev[k].u, w, forward, cmap, nCoarse);
        else {
            // the rank's own cells over its internal faces (the cut faces belong to the processor interfaces, not to lduAddr())
            std::vector<int> li, ui; std::vector<double> wi;
            for (int f = 0; f < nF; f++) if (G->lev[k].u[f] < nFine) { li.push_back(G->lev[k].l[f]); ui.push_back(G->lev[k].u[f]); wi.push_back(w[f]); }
            pair_agglomerate(nFine, li, ui, wi, forward, cmap, nCoarse);
        }
        forward = !forward;
        {   // GAMGAgglomeration::continueAgglomerating: global cell counts (nCoarse >= nProcs*nCellsInCoarsestLevel and some agglomeration happened)
            double tot[2] = {(double)nCoarse, (double)nFine};
            if (decomposed) { FFM_TRY(ffm_host_allreduce_sum(ctx, tot, 2)); }
            if (tot[0] < (double)ctx->nRanks * nCellsInCoarsestLevel || !(tot[0] < tot[1])) break;
        }
        // ---- decomposed: the neighbours' restriction maps for the ghost cells, the coarse ghost cells and the coarse exchange lists
        int nCoarseGhost = 0;
        std::vector<int> cNbr, cSendCount, cSendCells, cRecvCount;
        if (decomposed) {
            ffm_ldu *Af = G->lev[k].A;
            const int nNbr = (int)Af->ghNbrRank.size();
            std::vector<double> x((size_t)nFine + nFineGhost, -1.0);
            for (int c = 0; c < nFine; c++) x[c] = (double)cmap[c];
            // through the level's own ghost exchange (caller order == internal order for the ghost range; owned values are gathered by the send list)
            {
                double *xd = nullptr;
                FFM_TRY(ffm_malloc_uninit(ctx, sizeof(double) * x.size(), (void **)&xd));
                std::vector<double> xi(x.size());
                const std::vector<int> &n2o = Af->h_newToOldCell;
                for (size_t i = 0; i < x.size(); i++) xi[i] = x[Af->identity ? i : n2o[i]];
                FFM_TRY(ffm_h2d(ctx, xd, xi.data(), sizeof(double) * xi.size()));
                FFM_TRY(ffm_ghost_exchange(Af, xd));
                FFM_TRY(ffm_d2h(ctx, xi.data(), xd, sizeof(double) * xi.size()));
                for (int g = 0; g < nFineGhost; g++) x[nFine + g] = xi[nFine + g];          // ghosts keep their place behind the owned cells in both orders
                FFM_TRY(ffm_free(ctx, xd));
            }
            cmap.resize((size_t)nFine + nFineGhost);
            cNbr = Af->ghNbrRank; cSendCount.assign(nNbr, 0); cRecvCount.assign(nNbr, 0);
            for (int q = 0; q < nNbr; q++) {
                // coarse ghost cells of neighbour q: its coarse cells in the order of first appearance along this rank's receive list
                std::unordered_map<int, int> seen;
                for (int g = Af->ghRecvOff[q]; g < Af->ghRecvOff[q + 1]; g++) {
                    const int remote = (int)x[nFine + g];
                    if (remote < 0) { ffm_set_error("GAMG: a ghost cell received no restriction address"); return fail(FFM_ERR_COMM); }
                    auto it = seen.find(remote);
                    if (it == seen.end()) { it = seen.emplace(remote, nCoarse + nCoarseGhost).first; nCoarseGhost++; cRecvCount[q]++; }
                    cmap[nFine + g] = it->second;
                }
                // what this rank sends to q on the coarse level: the coarse cells of its fine send list, first appearance (the mirror image)
                std::unordered_map<int, int> sent;
                for (int i = Af->ghSendOff[q]; i < Af->ghSendOff[q + 1]; i++) {
                    const int cc = cmap[Af->h_ghSendCaller[i]];
                    if (sent.emplace(cc, 1).second) { cSendCells.push_back(cc); cSendCount[q]++; }
                }
            }
        }
        std::vector<int> cl, cu, fra; std::vector<char> flip;
        agglomerate_addressing(G->lev[k].l, G->lev[k].u, cmap, nCoarse + nCoarseGhost, cl, cu, fra, flip, decomposed ? nCoarse : -1);
        const int nCF = (int)cl.size();
        // restrictFaceField of the weights
        std::vector<double> cw(nCF, 0.0);
        for (int f = 0; f < nF; f++) if (fra[f] >= 0) cw[fra[f]] += w[f];
        w.swap(cw);
        // the coarse matrix
        ffm_ldu *Ac = nullptr;
        int rc;
        if (!hint.empty()) {
            std::vector<int> ch(nCoarse, -1);
            for (int c = 0; c < nFine; c++) if (ch[cmap[c]] < 0) ch[cmap[c]] = hint[c];          // the first fine cell's tile
            hint.swap(ch);
            if (nCoarse < tileMinCells) hint.clear();
        }
        if (decomposed) {
            rc = ffm_ldu_create_ext(ctx, nCoarse, nCoarseGhost, nCF, cl.data(), cu.data(), &Ac);
            if (!rc) rc = ffm_ldu_set_ghost_exchange(Ac, (int)cNbr.size(), cNbr.data(), cSendCount.data(), cSendCells.data(), cRecvCount.data());
            if (!rc && !G->lev[k].A->ghTags.empty()) rc = ffm_ldu_set_exchange_tags(Ac, 1, (int)cNbr.size(), G->lev[k].A->ghTags.data());
            if (!rc) { double gc = (double)nCoarse; rc = ffm_host_allreduce_sum(ctx, &gc, 1); if (!rc) rc = ffm_ldu_set_global_cells(Ac, (long)gc); }
        }
        else if (!hint.empty()) rc = ffm_ldu_create_hint(ctx, nCoarse, 0, nCF, cl.data(), cu.data(), hint.data(), &Ac);
        else rc = ffm_ldu_create(ctx, nCoarse, nCF, cl.data(), cu.data(), &Ac);
        if (rc) return fail(rc);
        if (Ac->sweepMode != 2) hint.clear();           // the planner gave up here: no tiles further down either
        if (getenv("FFM_VERBOSE")) fprintf(stderr, "ffm gamg: coarse level %d: %d cells (+ %d ghost), %s sweeps\n", k + 1, nCoarse, nCoarseGhost, Ac->sweepMode == 2 ? "tiled" : "level-scheduled");
        G->lev.emplace_back();
        Level &Lf = G->lev[k], &Lc = G->lev[k + 1];
        Lc.A = Ac; Lc.owned = true; Lc.nCells = nCoarse; Lc.nGhost = nCoarseGhost; Lc.nFaces = nCF; Lc.l = cl; Lc.u = cu;
        Lf.nCoarse = nCoarse; Lf.nCoarseFaces = nCF;
        // ---- maps.  Caller-order CSRs for the coefficients (owned cells only: a ghost cell's row belongs to its own rank)
        std::vector<int> st, it;
        std::vector<int> cmapOwn(cmap.begin(), cmap.begin() + nFine);
        group_by(nCoarse, nFine, cmapOwn, st, it);
        if ((rc = up(ctx, &Lf.cStart, st)) || (rc = up(ctx, &Lf.cItem, it))) return fail(rc);
        std::vector<int> key(nF);
        for (int f = 0; f < nF; f++) key[f] = fra[f] < 0 ? -1 - fra[f] : -1;
        group_by(nCoarse, nF, key, st, it);
        if ((rc = up(ctx, &Lf.iStart, st)) || (rc = up(ctx, &Lf.iItem, it))) return fail(rc);
        std::vector<int> val(nF);
        for (int f = 0; f < nF; f++) { key[f] = fra[f] >= 0 ? fra[f] : -1; val[f] = f | (flip[f] ? (int)0x80000000 : 0); }
        group_by(nCF, nF, key, st, it, &val);
        if ((rc = up(ctx, &Lf.fStart, st)) || (rc = up(ctx, &Lf.fItem, it))) return fail(rc);
        // internal-order maps for the vectors: fine internal i = oldToNew_f[caller], coarse likewise (owned cells)
        const std::vector<int> &n2oF = Lf.A->h_newToOldCell, &n2oC = Lc.A->h_newToOldCell;
        std::vector<int> o2nF(nFine), o2nC(nCoarse);
        for (int i = 0; i < nFine; i++) o2nF[Lf.A->identity ? i : n2oF[i]] = i;
        for (int i = 0; i < nCoarse; i++) o2nC[Lc.A->identity ? i : n2oC[i]] = i;
        std::vector<int> toC(nFine), keyI(nFine), valI(nFine);
        for (int i = 0; i < nFine; i++) toC[i] = o2nC[cmap[Lf.A->identity ? i : n2oF[i]]];
        // restrict lists: per coarse internal cell its fine cells ascending in CALLER index, stored as internal indices
        for (int c = 0; c < nFine; c++) { keyI[c] = o2nC[cmap[c]]; valI[c] = o2nF[c]; }
        group_by(nCoarse, nFine, keyI, st, it, &valI);
        if ((rc = up(ctx, &Lf.rStart, st)) || (rc = up(ctx, &Lf.rItem, it)) || (rc = up(ctx, &Lf.toCoarse, toC))) return fail(rc);
        // coarse level storage
        const size_t nAll = (size_t)nCoarse + nCoarseGhost;           // vectors carry the ghost range behind the owned cells
        if ((rc = dalloc(&Lc.diag, nAll)) || (rc = dalloc(&Lc.upper, nCF)) || (rc = dalloc(&Lc.lower, nCF)) || (rc = dalloc(&Lc.dInt, nAll)) ||
            (rc = dalloc(&Lc.src, nAll)) || (rc = dalloc(&Lc.corr, nAll)) || (rc = dalloc(&Lc.acf, nAll)) || (rc = dalloc(&Lc.pre, nAll)) ||
            (rc = dalloc(&Lc.tmp, nAll))) return fail(rc);
        FFM_HIP(hipMemsetAsync(Lc.diag, 0, sizeof(double) * nAll, ctx->stream));     // (ghost rows: no coefficients of their own here)
    }
    ctx->gamgForward = forward;             // the next agglomeration of this run continues in the direction this one ended with
    // no coarse level (the first agglomeration already falls below nCellsInCoarsestLevel: small meshes / regions): ffm_gamg_solve_d then
    // runs the coarsest-level solver (PCG + DIC / PBiCGStab + DILU) on the fine matrix itself instead of failing the whole run
    int rc;
    if ((rc = dalloc(&G->lev[0].dInt, nCells)) || (rc = dalloc(&G->fRes, nCells)) || (rc = dalloc(&G->fCorr, nCells)) || (rc = dalloc(&G->fApsi, nCells)) ||
        (rc = dalloc(&G->fSumA, nCells))) return fail(rc);
    G->decomposed = decomposed;
    *out = G;
    return FFM_OK;
}

extern "C" int ffm_ctx_set_gamg_forward(ffm_ctx *c, int forward) { if (!c) return FFM_ERR_ARG; c->gamgForward = forward != 0; return FFM_OK; }
extern "C" int ffm_ctx_gamg_forward(const ffm_ctx *c) { return c ? (c->gamgForward ? 1 : 0) : FFM_ERR_ARG; }
extern "C" int ffm_gamg_nlevels(const ffm_gamg *G) { return G ? (int)G->lev.size() - 1 : FFM_ERR_ARG; }       // coarse levels
extern "C" int ffm_gamg_level_size(const ffm_gamg *G, int level, int *nCells, int *nFaces)
{
    if (!G || level < 0 || level >= (int)G->lev.size()) return FFM_ERR_ARG;
    if (nCells) *nCells = G->lev[level].nCells;
    if (nFaces) *nFaces = G->lev[level].nFaces;
    return FFM_OK;
}
// level 0: the caller's addressing; level k: coarse level k.  Tests.
extern "C" int ffm_gamg_get_level_addressing(const ffm_gamg *G, int level, int *l, int *u)
{
    if (!G || level < 0 || level >= (int)G->lev.size() || !l || !u) return FFM_ERR_ARG;
    std::copy(G->lev[level].l.begin(), G->lev[level].l.end(), l); std::copy(G->lev[level].u.begin(), G->lev[level].u.end(), u);
    return FFM_OK;
}
extern "C" int ffm_gamg_get_level_coeffs(ffm_gamg *G, int level, double *diag, double *upper, double *lower)
{
    if (!G || level < 0 || level >= (int)G->lev.size() || !G->haveMatrix) return FFM_ERR_ARG;
    const Level &L = G->lev[level];
    FFM_HIP(hipStreamSynchronize(G->ctx->stream));
    if (diag) FFM_TRY(ffm_d2h(G->ctx, diag, L.diag, sizeof(double) * L.nCells));
    if (upper) FFM_TRY(ffm_d2h(G->ctx, upper, L.upper, sizeof(double) * L.nFaces));
    if (lower) FFM_TRY(ffm_d2h(G->ctx, lower, G->symmetric ? L.upper : L.lower, sizeof(double) * L.nFaces));
    return FFM_OK;
}

extern "C" int ffm_gamg_set_sweeps(ffm_gamg *G, int nPreSweeps, int nPostSweeps, int nFinestSweeps)
{
    if (!G || nPreSweeps < 0 || nPostSweeps < 0 || nFinestSweeps < 0) return FFM_ERR_ARG;
    G->nPre = nPreSweeps; G->nPost = nPostSweeps; G->nFinest = nFinestSweeps;
    return FFM_OK;
}

// GAMGSolver::agglomerateMatrix for every coarse level from the finest coefficients in caller order (G->lev[0].diag/upper/lower)
static int agglomerate_levels(ffm_gamg *G)
{
    hipStream_t s = G->ctx->stream;
    for (size_t k = 0; k + 1 < G->lev.size(); k++) {
        Level &Lf = G->lev[k], &Lc = G->lev[k + 1];
        const double *fl = G->symmetric ? nullptr : Lf.lower;
        hipLaunchKernelGGL(k_agg_diag, dim3(grid_of(Lf.nCoarse)), dim3(256), 0, s, (long)Lf.nCoarse, (const int *)Lf.cStart, (const int *)Lf.cItem,
                           (const int *)Lf.iStart, (const int *)Lf.iItem, (const double *)Lf.diag, (const double *)Lf.upper, fl, Lc.diag);
        hipLaunchKernelGGL(k_agg_faces, dim3(grid_of(Lf.nCoarseFaces)), dim3(256), 0, s, (long)Lf.nCoarseFaces, (const int *)Lf.fStart, (const int *)Lf.fItem,
                           (const double *)Lf.upper, fl, Lc.upper, G->symmetric ? (double *)nullptr : Lc.lower);
        FFM_HIP(hipGetLastError());
        FFM_TRY(ffm_ldu_set_coeffs_d(Lc.A, Lc.diag, Lc.upper, G->symmetric ? nullptr : Lc.lower));
    }
    for (auto &L : G->lev) {
        hipLaunchKernelGGL(k_gather_d, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, (const int *)(L.A->identity ? nullptr : L.A->cellPerm),
                           (const double *)L.diag, L.dInt);
    }
    FFM_HIP(hipGetLastError());
    G->haveMatrix = true;
    return FFM_OK;
}

// diag / upper / lower: device arrays in the caller's cell / face order (lower null: symmetric); they are also handed to the
// finest ffm_ldu.  The caller keeps them alive until the next call.
extern "C" int ffm_gamg_set_matrix_d(ffm_gamg *G, const double *diag_d, const double *upper_d, const double *lower_d)
{
    if (!G || !diag_d || !upper_d) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(G->ctx->device));
    G->symmetric = lower_d == nullptr;
    Level &L0 = G->lev[0];
    L0.diag = const_cast<double *>(diag_d); L0.upper = const_cast<double *>(upper_d); L0.lower = const_cast<double *>(lower_d);
    FFM_TRY(ffm_ldu_set_coeffs_d(L0.A, diag_d, upper_d, lower_d));
    return agglomerate_levels(G);
}

// The same from coefficients in the library's NATIVE layout (what fvMatrix of include/ffmFoam.H and the compiled time step
// assemble: diag in the library's cell order, upper / lower by native face index; ffm_ldu_set_coeffs_native_d): the finest
// matrix takes them as they are, the agglomeration reads them through the caller-face -> native map.  Needs a matrix in the
// library's own cell order (ffm_renumber_levels).
extern "C" int ffm_gamg_set_matrix_native_d(ffm_gamg *G, const double *diag_d, const double *upperNative_d, const double *lowerNative_d)
{
    if (!G || !diag_d || !upperNative_d) return FFM_ERR_ARG;
    Level &L0 = G->lev[0];
    if (!L0.A->identity) { ffm_set_error("GAMG: native coefficient layout needs the library's cell order (ffm_renumber_levels)"); return FFM_ERR_UNSUPPORTED; }
    FFM_HIP(hipSetDevice(G->ctx->device));
    hipStream_t s = G->ctx->stream;
    if (!G->faceToNative) {
        FFM_TRY(up(G->ctx, &G->faceToNative, L0.A->h_callerToNative));
        FFM_TRY(dalloc(&G->upC, L0.nFaces)); FFM_TRY(dalloc(&G->loC, L0.nFaces));
    }
    G->symmetric = lowerNative_d == nullptr;
    hipLaunchKernelGGL(k_gather_d, dim3(grid_of(L0.nFaces)), dim3(256), 0, s, (long)L0.nFaces, (const int *)G->faceToNative, upperNative_d, G->upC);
    if (lowerNative_d) hipLaunchKernelGGL(k_gather_d, dim3(grid_of(L0.nFaces)), dim3(256), 0, s, (long)L0.nFaces, (const int *)G->faceToNative, lowerNative_d, G->loC);
    FFM_HIP(hipGetLastError());
    L0.diag = const_cast<double *>(diag_d); L0.upper = G->upC; L0.lower = lowerNative_d ? G->loC : nullptr;
    FFM_TRY(ffm_ldu_set_coeffs_native_d(L0.A, diag_d, upperNative_d, lowerNative_d));
    return agglomerate_levels(G);
}

// smoothers[k].smooth(psi, source, nSweeps) on matrix k: GaussSeidel, symGaussSeidel, DIC or DILU (psi += M^-1 (source - A psi))
static int smooth_level(ffm_gamg *G, int k, int smoother, double *psi, const double *b, int nSweeps)
{
    Level &L = G->lev[k];
    if (nSweeps < 1) return FFM_OK;
    if (smoother == FFM_GS || smoother == FFM_SYMGS) return ffm_gs_smooth_i(L.A, smoother == FFM_SYMGS, nSweeps, psi, b);
    hipStream_t s = G->ctx->stream;
    double *rA = (k == 0) ? G->fApsi : L.acf, *wA = (k == 0) ? G->fSumA : L.tmp;
    FFM_TRY(ffm_precond_setup_i(L.A, smoother));
    for (int sw = 0; sw < nSweeps; sw++) {
        FFM_TRY(ffm_k_residual(L.A, psi, b, rA));
        FFM_TRY(ffm_precond_apply_i(L.A, smoother, false, rA, wA));
        hipLaunchKernelGGL(k_add_to, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, psi, (const double *)wA);
    }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// GAMGSolver::scale on matrix k (acf: scratch of the level's size)
static int scale_level(ffm_gamg *G, int k, double *field, double *acf, const double *source)
{
    Level &L = G->lev[k]; ffm_ctx *c = G->ctx; hipStream_t s = c->stream;
    FFM_TRY(ffm_k_spmv(L.A, field, acf, false));
    FFM_TRY(ffm_k_dot(c, source, field, L.nCells, S_TMP0));
    FFM_TRY(ffm_k_dot(c, acf, field, L.nCells, S_TMP1));
    if (G->decomposed) FFM_TRY(ffm_allreduce_slots(c, S_TMP0, 2));            // gSumProd over the ranks
    hipLaunchKernelGGL(k_scale_factor, dim3(1), dim3(64), 0, s, c->scal_d);
    hipLaunchKernelGGL(k_scale, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, field, (const double *)acf, source, (const double *)L.dInt, (const double *)c->scal_d);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

static int restrict_to(ffm_gamg *G, int k, const double *ff, double *cf)       // matrix k -> k + 1
{
    Level &L = G->lev[k];
    hipLaunchKernelGGL(k_restrict, dim3(grid_of(L.nCoarse)), dim3(256), 0, G->ctx->stream, (long)L.nCoarse, (const int *)L.rStart, (const int *)L.rItem, ff, cf);
    return FFM_OK;
}
static int prolong_to(ffm_gamg *G, int k, const double *cf, double *ff)        // matrix k + 1 -> k
{
    Level &L = G->lev[k];
    hipLaunchKernelGGL(k_prolong, dim3(grid_of(L.nCells)), dim3(256), 0, G->ctx->stream, (long)L.nCells, (const int *)L.toCoarse, cf, ff);
    return FFM_OK;
}

// GAMGSolver::Vcycle.  Matrices 1 .. nC are the reference's coarse levels 0 .. nC-1: coarseSources[lev] = lev[lev+1].src etc.
static int vcycle(ffm_gamg *G, int smoother, double *psi, const double *source, double tol, double relTol)
{
    ffm_ctx *c = G->ctx; hipStream_t s = c->stream;
    const int nC = (int)G->lev.size() - 1, coarsest = nC - 1;
    const bool scaleCorr = G->symmetric;
    FFM_TRY(restrict_to(G, 0, G->fRes, G->lev[1].src));
    for (int lv = 0; lv < coarsest; lv++) {
        Level &L = G->lev[lv + 1];
        if (G->nPre) {
            hipLaunchKernelGGL(k_zero, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, L.corr);
            FFM_TRY(smooth_level(G, lv + 1, smoother, L.corr, L.src, std::min(G->nPre + G->preMul * lv, G->maxPre)));
            if (scaleCorr && lv < coarsest - 1) FFM_TRY(scale_level(G, lv + 1, L.corr, L.acf, L.src));
            FFM_TRY(ffm_k_spmv(L.A, L.corr, L.acf, false));
            hipLaunchKernelGGL(k_sub_to, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, L.src, (const double *)L.acf);
            FFM_HIP(hipMemcpyAsync(L.pre, L.corr, sizeof(double) * L.nCells, hipMemcpyDeviceToDevice, s));
        }
        FFM_TRY(restrict_to(G, lv + 1, L.src, G->lev[lv + 2].src));
    }
    {   // solveCoarsestLevel: PCG + DIC (symmetric) or PBiCGStab + DILU, to the solver's own tolerance / relTol
        Level &L = G->lev[nC];
        hipLaunchKernelGGL(k_zero, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, L.corr);
        ffm_perf pf;
        FFM_TRY(ffm_solve_internal_i(L.A, G->symmetric ? FFM_PCG : FFM_PBICGSTAB, G->symmetric ? FFM_DIC : FFM_DILU, tol, relTol, 0, 1000, 1, L.corr, L.src, &pf));
        G->coarsestLog.push_back(pf);
    }
    for (int lv = coarsest - 1; lv >= 0; lv--) {
        Level &L = G->lev[lv + 1];
        FFM_TRY(prolong_to(G, lv + 1, G->lev[lv + 2].corr, L.corr));
        if (scaleCorr && lv < coarsest - 1) FFM_TRY(scale_level(G, lv + 1, L.corr, L.acf, L.src));
        if (G->nPre) hipLaunchKernelGGL(k_add_to, dim3(grid_of(L.nCells)), dim3(256), 0, s, (long)L.nCells, L.corr, (const double *)L.pre);
        FFM_TRY(smooth_level(G, lv + 1, smoother, L.corr, L.src, std::min(G->nPost + G->postMul * lv, G->maxPost)));
    }
    Level &L0 = G->lev[0];
    FFM_TRY(prolong_to(G, 0, G->lev[1].corr, G->fCorr));
    if (scaleCorr) FFM_TRY(scale_level(G, 0, G->fCorr, G->fApsi, G->fRes));
    hipLaunchKernelGGL(k_add_to, dim3(grid_of(L0.nCells)), dim3(256), 0, s, (long)L0.nCells, psi, (const double *)G->fCorr);
    FFM_HIP(hipGetLastError());
    return smooth_level(G, 0, smoother, psi, source, G->nFinest);
}

// GAMGSolver::solve.  psi_d / source_d: device vectors in the caller's cell order of the finest matrix.
extern "C" int ffm_gamg_solve_d(ffm_gamg *G, int smoother, double tol, double relTol, int minIter, int maxIter, double *psi_d, const double *source_d,
                                ffm_perf *out)
{
    if (!G || !psi_d || !source_d || !out) { ffm_set_error("ffm_gamg_solve_d: null argument"); return FFM_ERR_ARG; }
    if (!G->haveMatrix) { ffm_set_error("ffm_gamg_solve_d: no coefficients (ffm_gamg_set_matrix_d)"); return FFM_ERR_ARG; }
    if (!(smoother == FFM_GS || smoother == FFM_SYMGS || smoother == FFM_DIC || smoother == FFM_DILU)) { ffm_set_error("GAMG: smoother %d not offered", smoother); return FFM_ERR_UNSUPPORTED; }
    if (smoother == FFM_DIC && !G->symmetric) { ffm_set_error("GAMG: the DIC smoother needs a symmetric matrix"); return FFM_ERR_UNSUPPORTED; }
    ffm_ctx *c = G->ctx; hipStream_t s = c->stream;
    FFM_HIP(hipSetDevice(c->device));
    memset(out, 0, sizeof(*out));
    G->coarsestLog.clear();
    Level &L0 = G->lev[0]; ffm_ldu *A = L0.A; const long N = L0.nCells; const int g = grid_of(N);
    double *psi = psi_d; const double *src = source_d;
    if (!A->identity) {
        const double *t; FFM_TRY(ffm_to_internal(A, source_d, 1, &t)); src = t;
        FFM_TRY(ffm_to_internal(A, psi_d, 2, &t)); psi = A->permIn[2];
    }
    // Apsi, normFactor, initial residual
    FFM_TRY(ffm_k_spmv(A, psi, G->fApsi, false));
    FFM_TRY(ffm_k_sumA(A, G->fSumA));
    FFM_TRY(ffm_k_sum(c, psi, N, S_TMP0));
    if (G->decomposed) FFM_TRY(ffm_allreduce_slots(c, S_TMP0, 1));             // xRef = gAverage(psi): global sum over the global cell count
    hipLaunchKernelGGL(k_gamg_normf, dim3(g), dim3(256), 0, s, N, (const double *)G->fApsi, src, (const double *)G->fSumA, (const double *)c->scal_d,
                       (double)(G->decomposed ? A->globalCells : N), c->partials_d);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, s, g, (const double *)c->partials_d, c->scal_d, (int)S_TMP1);
    hipLaunchKernelGGL(k_sub3, dim3(g), dim3(256), 0, s, N, G->fRes, src, (const double *)G->fApsi);
    FFM_TRY(ffm_k_summag(c, G->fRes, N, S_TMP0));
    if (G->decomposed) FFM_TRY(ffm_allreduce_slots(c, S_TMP0, 2));             // gSumMag(residual), the normFactor sum
    FFM_TRY(ffm_read_scalars(c));
    const double normFactor = c->scal_h[S_TMP1] + 1e-20;
    out->initialResidual = c->scal_h[S_TMP0] / normFactor;
    out->finalResidual = out->initialResidual;
    auto converged = [&]() { return out->finalResidual < tol || (relTol > 1e-20 && out->finalResidual < relTol * out->initialResidual); };
    if (G->lev.size() < 2) {             // no coarse level: the coarsest-level solver on the fine matrix (see ffm_gamg_create)
        ffm_perf pf;
        FFM_TRY(ffm_solve_internal_i(A, G->symmetric ? FFM_PCG : FFM_PBICGSTAB, G->symmetric ? FFM_DIC : FFM_DILU, tol, relTol, minIter, maxIter, 1, psi, src, &pf));
        G->coarsestLog.push_back(pf);
        *out = pf;
    } else
    if (minIter > 0 || !converged()) {
        do {
            FFM_TRY(vcycle(G, smoother, psi, src, tol, relTol));
            FFM_TRY(ffm_k_spmv(A, psi, G->fApsi, false));
            hipLaunchKernelGGL(k_sub3, dim3(g), dim3(256), 0, s, N, G->fRes, src, (const double *)G->fApsi);
            FFM_TRY(ffm_k_summag(c, G->fRes, N, S_TMP0));
            if (G->decomposed) FFM_TRY(ffm_allreduce_slots(c, S_TMP0, 1));
            FFM_TRY(ffm_read_scalars(c));
            out->finalResidual = c->scal_h[S_TMP0] / normFactor;
        } while ((++out->nIterations < maxIter && !converged()) || out->nIterations < minIter);
    }
    out->converged = converged() ? 1 : 0;
    if (!A->identity) FFM_TRY(ffm_from_internal(A, psi, psi_d));
    FFM_HIP(hipStreamSynchronize(s));
    for (auto &L : G->lev) { if (L.A->sweepMode == 2) FFM_TRY(ffm_tile_check_abort(L.A)); else FFM_TRY(ffm_flow_check_abort(L.A)); }
    return FFM_OK;
}

// the coarsest-level solves of the last ffm_gamg_solve_d (one per V-cycle)
extern "C" int ffm_gamg_coarsest_solves(const ffm_gamg *G, int maxN, ffm_perf *out)
{
    if (!G) return FFM_ERR_ARG;
    const int n = (int)G->coarsestLog.size();
    for (int i = 0; i < std::min(n, maxN); i++) if (out) out[i] = G->coarsestLog[i];
    return n;
}
