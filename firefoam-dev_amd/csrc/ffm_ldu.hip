// ffm_ldu.hip -- lduAddressing analysis (host), device LDU container and the
// lduMatrix kernels Amul / Tmul / sumA / residual.
//
// Replaces (OpenFOAM-dev @940e28f, not vendored in the reference):
//   src/OpenFOAM/matrices/lduMatrix/lduAddressing/lduAddressing.C
//   src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixATmul.C
// reached from the reference at solver/pEqn.H:5,39, solver/UEqn.H:19,
// solver/YEEqn.H:60,111.
//
// Data layout in HBM (all arrays in the *internal* cell numbering):
//   cells are ordered level-major: level(c) = longest path to c in the DAG
//   owner->neighbour, cells of one level contiguous and sorted by the caller's
//   index.  Every level-scheduled sweep (DIC/DILU/Gauss-Seidel) then touches a
//   contiguous cell range per level, and the SpMV runs in the same numbering so
//   no vector is permuted between SpMV and preconditioner.  The renumbering is a
//   topological order of the same DAG, so owner<neighbour is preserved and the
//   incomplete factorisations are the same operators as in the caller's
//   numbering.  A caller whose mesh is already level-major (ffm_renumber_levels)
//   pays no permutation pass at all.
//
//   Faces are stored as a sliced owner-ELL (slices of 64 cells = one wavefront,
//   slot-major inside a slice; see ffm_internal.hpp): every index / coefficient
//   load is unit-stride across the wave, the owner of a face is implicit, and a
//   symmetric coefficient is stored once -- the neighbour row re-reads it through
//   a packed (owner cell, slot) entry, which hits cache because neighbouring rows
//   are neighbours in memory.  HBM traffic of a symmetric Amul is therefore the
//   algorithmic 24 N + 16 F bytes plus padding.
//   Row form of the face loops: row c first adds its lower entries (faces whose
//   neighbour is c, in the caller's face order), then its upper slots (faces
//   owned by c, in the caller's face order) -- exactly the order in which the
//   serial face loop of the reference touches row c, so no atomics are needed
//   and every row sum is bitwise the reference's.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <atomic>
#include <numeric>

// ---------------------------------------------------------------- analysis ---
struct LduAnalysis {
    std::vector<int> newToOldCell, oldToNewCell, newToOldFace;
    std::vector<int> l, u;                 // new numbering, upper-triangular order
    std::vector<int> fwdLevelStart;        // [nLevels+1]
    std::vector<int> bwdLevelStart;        // [nBwd+1] into bwdOrder
    std::vector<int> bwdOrder;             // cells sorted by backward level
    bool identity = true, bwdContig = true;
    // group plan of the tiled sweeps (mode 2)
    int mode = 0, nGroups = 0; bool bwdIsReverse = false;
    std::vector<int> levNew, blNew;         // forward / backward level of every owned cell (new numbering)
    std::vector<int> grpCell, bwdCells;
};

// Sweep modes.  0 "levels": one launch per dependency level (level-major numbering).  2 "tile": tiled wavefront sweep
// (ffm_tile.hip).  Default (FFM_SWEEP unset or "auto"): tile when the caller gives a group hint or the mesh is a
// blockMesh-numbered box and the plan is feasible, levels otherwise.  FFM_SWEEP=tile also tiles un-hinted meshes (chunks of
// the cell order; tests).
enum { SWEEP_AUTO = 3 };
static int default_sweep_mode()
{
    const char *e = getenv("FFM_SWEEP");
    if (!e || e[0] == 'a' || e[0] == 'A') return SWEEP_AUTO;
    if (e[0] == 't' || e[0] == 'T' || e[0] == '2') return 2;
    return 0;
}

// blockMesh single-block numbering (c = i + nx*(j + ny*k), faces of c in the order +x, +y, +z): returns true and the box
static bool detect_box(int N, int F, const int *l, const int *u, int &nx, int &ny, int &nz)
{
    if (N < 8 || F < 3) return false;
    // nx: first cell without a face to c+1
    int f = 0; nx = 0;
    for (int c = 0; c < N; c++) {
        bool hasX = false;
        while (f < F && l[f] == c) { if (u[f] == c + 1) hasX = true; f++; }
        if (!hasX) { nx = c + 1; break; }
    }
    if (nx < 1 || N % nx) return false;
    // ny: number of rows until a row start has no face to c+nx
    std::vector<int> start(N + 1, 0);
    for (int q = 0; q < F; q++) start[l[q] + 1]++;
    for (int c = 0; c < N; c++) start[c + 1] += start[c];
    ny = 0;
    for (int j = 0; (long)j * nx < N; j++) {
        const int c = j * nx; bool hasY = false;
        for (int q = start[c]; q < start[c + 1]; q++) if (u[q] == c + nx) hasY = true;
        if (!hasY) { ny = j + 1; break; }
    }
    if (ny < 1 || (N / nx) % ny) return false;
    nz = N / nx / ny;
    if ((long)F != (long)(nx - 1) * ny * nz + (long)nx * (ny - 1) * nz + (long)nx * ny * (nz - 1)) return false;
    int q = 0;
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
        const int c = i + nx * (j + ny * k);
        if (i < nx - 1) { if (l[q] != c || u[q] != c + 1) return false; q++; }
        if (j < ny - 1) { if (l[q] != c || u[q] != c + nx) return false; q++; }
        if (k < nz - 1) { if (l[q] != c || u[q] != c + nx * ny) return false; q++; }
    }
    return q == F;
}

static int tile_edge()
{
    int t = 16;
    if (const char *e = getenv("FFM_TILE")) t = std::max(1, atoi(e));
    return t;
}

// ffm_renumber_* and ffm_ldu_create* are separate calls: the groups chosen by a renumbering (contiguous cell ranges of the
// new numbering) are remembered under a fingerprint of the renumbered addressing, so that creating the matrix from that
// addressing without a hint finds them again.
struct GroupMemo { unsigned long long key; int N, F; std::vector<int> grpCell; };
static std::vector<GroupMemo> g_groupMemo;
static unsigned long long addr_fingerprint(int N, int F, const int *l, const int *u)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)N << 32) ^ (unsigned)F;
    for (int f = 0; f < F; f++) { h ^= (unsigned long long)(unsigned)l[f] | ((unsigned long long)(unsigned)u[f] << 32); h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; }
    return h;
}
static void memo_put(int N, int F, const int *l, const int *u, const std::vector<int> &grpCell)
{
    GroupMemo m{addr_fingerprint(N, F, l, u), N, F, grpCell};
    for (auto &x : g_groupMemo) if (x.key == m.key && x.N == N && x.F == F) { x = m; return; }
    if (g_groupMemo.size() >= 8) g_groupMemo.erase(g_groupMemo.begin());
    g_groupMemo.push_back(std::move(m));
}
static const std::vector<int> *memo_get(int N, int F, const int *l, const int *u)
{
    if (g_groupMemo.empty()) return nullptr;
    const unsigned long long key = addr_fingerprint(N, F, l, u);
    for (auto &x : g_groupMemo) if (x.key == key && x.N == N && x.F == F) return &x.grpCell;
    return nullptr;
}

// nOwn < N: cells [nOwn, N) are ghost cells (copies of neighbour-rank cells).  They own no faces, stay at the end of the
// numbering in their given order, take no part in the level structure, and faces towards them are ignored by the
// backward levels (block-Jacobi sweeps).
static int analyse(int N, int nOwn, int F, const int *l, const int *u, bool renumber, bool sortByNewNeighbour, LduAnalysis &a,
                   const int *groupHint = nullptr, int forceMode = -1)
{
    a.mode = forceMode >= 0 ? forceMode : default_sweep_mode();
    FfmLapTimer lap_("analyse");
    std::vector<int> autoHint;
    const bool autoMode = a.mode == SWEEP_AUTO;
    if (a.mode == SWEEP_AUTO) {
        a.mode = 0;
        if (renumber && nOwn > 0 && (long)N * 24 < (1L << 32)) {      // (the tiled kernels use 32-bit byte offsets into their streams)
            int bx, by, bz;
            if (groupHint) a.mode = 2;
            else if (const std::vector<int> *gc = memo_get(N, F, l, u)) {
                autoHint.resize(nOwn);
                for (int g = 0; g + 1 < (int)gc->size(); g++) for (int c = (*gc)[g]; c < (*gc)[g + 1] && c < nOwn; c++) autoHint[c] = g;
                groupHint = autoHint.data(); a.mode = 2;
            } else if (N == nOwn && detect_box(N, F, l, u, bx, by, bz)) {
                const int T = tile_edge();
                autoHint.resize(nOwn);
                for (int c = 0; c < nOwn; c++) autoHint[c] = ffm_tile_label(((c / bx) % by) / T, (c / (bx * by)) / T);
                groupHint = autoHint.data(); a.mode = 2;
            }
        }
    }
    for (int f = 0; f < F; f++) {
        if (l[f] < 0 || u[f] >= N || l[f] >= u[f]) {
            ffm_set_error("LDU addressing: face %d has l=%d u=%d (need 0<=l<u<nCells=%d)", f, l[f], u[f], N);
            return FFM_ERR_ADDR;
        }
        if (f && l[f] < l[f - 1]) {
            ffm_set_error("LDU addressing: faces not sorted by owner at face %d", f);
            return FFM_ERR_ADDR;
        }
        if (l[f] >= nOwn) { ffm_set_error("LDU addressing: face %d is owned by a ghost cell", f); return FFM_ERR_ADDR; }
    }
    lap_.lap("hint + validation");
    // forward levels
    std::vector<int> lev(N, 0);
    for (int f = 0; f < F; f++) if (u[f] < nOwn) lev[u[f]] = std::max(lev[u[f]], lev[l[f]] + 1);
    int nLev = 0;
    for (int c = 0; c < nOwn; c++) nLev = std::max(nLev, lev[c] + 1);
    if (nOwn == 0) nLev = 0;
    a.newToOldCell.resize(N); a.oldToNewCell.resize(N);
    std::vector<int> grpOfOld;            // mode 1: group of every owned cell (caller numbering)
    if (renumber && a.mode >= 1) {
        // Groups.  With a hint (one label per owned cell, e.g. a 2-D tile of cell columns computed by the host from the cell
        // centres) the groups are the label classes, provided their dependency graph is acyclic; they are then ranked in a
        // topological order (ties by label).  Without a usable hint: contiguous chunks of the caller's cell order, which is
        // a topological order of the DAG, so cross-group dependencies point from lower to higher groups by construction.
        grpOfOld.assign(nOwn, 0);
        int G = 0;
        bool hinted = false;
        if (groupHint && nOwn > 0) {
            // the distinct labels in ascending order and every cell's index into them: a presence table where the labels span a small
            // range (tile labels do), a sort otherwise
            std::vector<int> labels, gid(nOwn);
            int labMin = groupHint[0], labMax = groupHint[0];
            for (int c = 1; c < nOwn; c++) { labMin = std::min(labMin, groupHint[c]); labMax = std::max(labMax, groupHint[c]); }
            if ((long)labMax - labMin < (1L << 24)) {
                std::vector<int> idx((size_t)(labMax - labMin) + 1, 0);
                for (int c = 0; c < nOwn; c++) idx[groupHint[c] - labMin] = 1;
                for (size_t i = 0; i < idx.size(); i++) if (idx[i]) { idx[i] = (int)labels.size(); labels.push_back(labMin + (int)i); } else idx[i] = -1;
                ffm_parallel_for(nOwn, [&](long lo, long hi) { for (long c = lo; c < hi; c++) gid[c] = idx[groupHint[c] - labMin]; });
            } else {
                labels.assign(groupHint, groupHint + nOwn);
                std::sort(labels.begin(), labels.end()); labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
                ffm_parallel_for(nOwn, [&](long lo, long hi) {
                    for (long c = lo; c < hi; c++) gid[c] = (int)(std::lower_bound(labels.begin(), labels.end(), groupHint[c]) - labels.begin()); });
            }
            const int nl = (int)labels.size();
            int nl2 = nl;
            // A label class that an internal wall (a sheet of baffle faces, cases/steckler/system/createBafflesDict) cuts into pieces that
            // are not connected inside the class has dependency levels that restart behind the wall: a level then holds cells of two
            // planes, is split over several entries, and the neighbours of a cell are no longer in the entries next to its own (the
            // ring window of the tiled Amul; the sweeps' LDS ring).  Every connected component of a class becomes a group of its own:
            // no new edges, so the group graph stays acyclic; classes in one piece (every tile of a box) are unchanged.
            static const bool splitComponents = !(getenv("FFM_TILE_SPLIT_COMPONENTS") && atoi(getenv("FFM_TILE_SPLIT_COMPONENTS")) == 0);
            if (splitComponents) {
                std::vector<int> parent(nOwn);
                std::iota(parent.begin(), parent.end(), 0);
                auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
                for (int f = 0; f < F; f++) if (u[f] < nOwn && gid[l[f]] == gid[u[f]]) { const int a_ = find(l[f]), b_ = find(u[f]); if (a_ != b_) parent[std::max(a_, b_)] = std::min(a_, b_); }
                // components of a class, numbered by their lowest cell; the first LARGE one keeps the class' id, the other large ones get
                // new ids; fragments (a few cells cut off by scattered baffle faces: fewer than 1024 cells or a sixteenth of the class)
                // stay with the class -- a group per fragment would only add tile hand-offs
                std::vector<int> compSize(nOwn, 0), classSize(nl, 0);
                for (int c = 0; c < nOwn; c++) { compSize[find(c)]++; classSize[gid[c]]++; }
                std::vector<int> firstRoot(nl, -1), newId(nOwn, -1);
                for (int c = 0; c < nOwn; c++) {
                    const int r = find(c);
                    if (newId[r] >= 0) continue;
                    const int g = gid[r];
                    const bool large = compSize[r] >= std::max(1024, classSize[g] / 16);
                    if (!large) newId[r] = g;
                    else if (firstRoot[g] < 0) { firstRoot[g] = r; newId[r] = g; }
                    else newId[r] = nl2++;
                }
                if (nl2 > nl) {
                    if (getenv("FFM_VERBOSE")) fprintf(stderr, "ffm: %d group labels in %d connected pieces (internal walls): one group per piece\n", nl, nl2);
                    for (int c = 0; c < nOwn; c++) gid[c] = newId[find(c)];
                }
            }
    lap_.lap("labels + components");
            std::vector<std::pair<int, int>> edges;
            for (int f = 0; f < F; f++) if (u[f] < nOwn && gid[l[f]] != gid[u[f]]) edges.emplace_back(gid[l[f]], gid[u[f]]);
            std::sort(edges.begin(), edges.end()); edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
            std::vector<int> indeg(nl2, 0), estart(nl2 + 1, 0);
            for (auto &e : edges) { indeg[e.second]++; estart[e.first + 1]++; }
            for (int i = 0; i < nl2; i++) estart[i + 1] += estart[i];
            std::vector<int> rank(nl2, -1), heap;
            auto cmp = [](int x, int y) { return x > y; };
            for (int i = 0; i < nl2; i++) if (!indeg[i]) heap.push_back(i);
            std::make_heap(heap.begin(), heap.end(), cmp);
            int done = 0;
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end(), cmp); const int x = heap.back(); heap.pop_back();
                rank[x] = done++;
                for (int k = estart[x]; k < estart[x + 1]; k++) { const int y = edges[k].second; if (--indeg[y] == 0) { heap.push_back(y); std::push_heap(heap.begin(), heap.end(), cmp); } }
            }
            if (done == nl2) { hinted = true; G = nl2; for (int c = 0; c < nOwn; c++) grpOfOld[c] = rank[gid[c]]; }
        }
        if (!hinted && autoMode) {           // unusable hint: level-scheduled sweeps
            if (getenv("FFM_VERBOSE") && groupHint) fprintf(stderr, "ffm: the group hint gives a cyclic group graph: level-scheduled sweeps\n");
            LduAnalysis b;
            FFM_TRY(analyse(N, nOwn, F, l, u, renumber, sortByNewNeighbour, b, nullptr, 0));
            a = std::move(b);
            return FFM_OK;
        }
        if (!hinted) {
            // at most 512 groups and at least 8192 cells per group
            int B = std::max(8192, (nOwn + 511) / 512);
            if (const char *e = getenv("FFM_PIPE_GROUP_CELLS")) B = std::max(1, atoi(e));     // tests: force many small groups
            G = nOwn ? (nOwn + B - 1) / B : 0;
            for (int c = 0; c < nOwn; c++) grpOfOld[c] = c / B;
        }
    lap_.lap("group graph");
        a.nGroups = G;
        a.grpCell.assign(G + 1, 0);
        for (int c = 0; c < nOwn; c++) a.grpCell[grpOfOld[c] + 1]++;
        for (int g = 0; g < G; g++) a.grpCell[g + 1] += a.grpCell[g];
        // new numbering: group-major, then level, then "edge class", then caller index (stable counting passes, least
        // significant key first).  Edge class: the cells of a level that have a neighbour in another group come first, grouped
        // by the lowest-ranked such group, so that the values a neighbouring tile gathers from this one (Amul, the FV row
        // kernels) are runs of consecutive cells instead of one 128-byte line per value (a 16 x 16 column tile: the two
        // y-edges of a level were 16 cells with stride 16)
        {
            std::vector<int> first(nOwn);
            std::iota(first.begin(), first.end(), 0);
            static const bool edgeOrder = !(getenv("FFM_TILE_EDGE_ORDER") && atoi(getenv("FFM_TILE_EDGE_ORDER")) == 0);
            if (edgeOrder && G > 1) {
                std::vector<int> key(nOwn, G), cntK(G + 2, 0);
                for (int f = 0; f < F; f++) {
                    if (u[f] >= nOwn) continue;
                    const int gl = grpOfOld[l[f]], gu = grpOfOld[u[f]];
                    if (gl != gu) { key[l[f]] = std::min(key[l[f]], gu); key[u[f]] = std::min(key[u[f]], gl); }
                }
                for (int c = 0; c < nOwn; c++) cntK[key[c] + 1]++;
                for (int i = 0; i <= G; i++) cntK[i + 1] += cntK[i];
                for (int c = 0; c < nOwn; c++) first[cntK[key[c]]++] = c;
            }
            std::vector<int> byLevel(nOwn), cnt(nLev + 1, 0);
            for (int c = 0; c < nOwn; c++) cnt[lev[c] + 1]++;
            for (int i = 0; i < nLev; i++) cnt[i + 1] += cnt[i];
            for (int i = 0; i < nOwn; i++) { const int c = first[i]; byLevel[cnt[lev[c]]++] = c; }
            std::vector<int>().swap(first);
            std::vector<int> pos(a.grpCell.begin(), a.grpCell.end() - (G ? 1 : 0));
            if (!G) pos.clear();
            for (int i = 0; i < nOwn; i++) { const int c = byLevel[i]; const int p = pos[grpOfOld[c]]++; a.newToOldCell[p] = c; a.oldToNewCell[c] = p; }
        }
    lap_.lap("cell order");
        for (int c = nOwn; c < N; c++) { a.newToOldCell[c] = c; a.oldToNewCell[c] = c; }
        a.fwdLevelStart.assign(nLev + 1, 0);
    } else if (renumber) {
        a.mode = 0;
        std::vector<int> start(nLev + 1, 0);
        for (int c = 0; c < nOwn; c++) start[lev[c] + 1]++;
        for (int i = 0; i < nLev; i++) start[i + 1] += start[i];
        a.fwdLevelStart = start;
        std::vector<int> pos(start.begin(), start.end() - (nLev ? 1 : 0));
        if (!nLev) pos.clear();
        for (int c = 0; c < nOwn; c++) { int p = pos[lev[c]]++; a.newToOldCell[p] = c; a.oldToNewCell[c] = p; }
        for (int c = nOwn; c < N; c++) { a.newToOldCell[c] = c; a.oldToNewCell[c] = c; }
    } else {
        a.mode = 0;
        // caller insists on its numbering: only legal if it is level-major already
        std::iota(a.newToOldCell.begin(), a.newToOldCell.end(), 0);
        a.oldToNewCell = a.newToOldCell;
        a.fwdLevelStart.assign(nLev + 1, 0);
        for (int c = 0; c < nOwn; c++) a.fwdLevelStart[lev[c] + 1]++;
        for (int i = 0; i < nLev; i++) a.fwdLevelStart[i + 1] += a.fwdLevelStart[i];
        for (int c = 1; c < nOwn; c++) if (lev[c] < lev[c - 1]) { ffm_set_error("numbering is not level-major"); return FFM_ERR_ARG; }
    }
    a.identity = true;
    for (int c = 0; c < N; c++) if (a.newToOldCell[c] != c) { a.identity = false; break; }
    // faces in the new numbering, sorted by (owner, neighbour)
    a.l.resize(F); a.u.resize(F); a.newToOldFace.resize(F);
    if (a.identity) {
        std::copy(l, l + F, a.l.begin()); std::copy(u, u + F, a.u.begin());
        std::iota(a.newToOldFace.begin(), a.newToOldFace.end(), 0);
    } else {
        std::vector<int> cnt(N + 1, 0);
        for (int f = 0; f < F; f++) cnt[a.oldToNewCell[l[f]] + 1]++;
        for (int c = 0; c < N; c++) cnt[c + 1] += cnt[c];
        std::vector<int> pos(cnt.begin(), cnt.end() - 1);
        for (int f = 0; f < F; f++) a.newToOldFace[pos[a.oldToNewCell[l[f]]]++] = f;
        // The counting sort leaves the faces of one owner in the caller's face order.  The device
        // layout keeps that order so every row is accumulated exactly as in the caller's face loop;
        // the public renumbering sorts by the new neighbour (a proper upper-triangular mesh).
        if (sortByNewNeighbour) ffm_parallel_for(N, [&](long lo, long hi) {
            for (long c = lo; c < hi; c++)
                std::sort(a.newToOldFace.begin() + cnt[c], a.newToOldFace.begin() + cnt[c + 1],
                          [&](int f1, int f2) { return a.oldToNewCell[u[f1]] < a.oldToNewCell[u[f2]]; });
        });
        std::atomic<int> flipped(0);
        ffm_parallel_for(F, [&](long lo, long hi) {
            for (long f = lo; f < hi; f++) {
                const int of = a.newToOldFace[f];
                a.l[f] = a.oldToNewCell[l[of]]; a.u[f] = a.oldToNewCell[u[of]];
                if (a.l[f] >= a.u[f]) flipped = 1;
            }
        });
        if (flipped) { ffm_set_error("internal: renumbering flipped a face"); return FFM_ERR_ADDR; }
    }
    lap_.lap("faces");
    // backward levels (new numbering)
    std::vector<int> bl(N, 0);
    for (int f = F - 1; f >= 0; f--) if (a.u[f] < nOwn) bl[a.l[f]] = std::max(bl[a.l[f]], bl[a.u[f]] + 1);
    int nB = 0;
    for (int c = 0; c < nOwn; c++) nB = std::max(nB, bl[c] + 1);
    if (nOwn == 0) nB = 0;
    a.bwdLevelStart.assign(nB + 1, 0);
    for (int c = 0; c < nOwn; c++) a.bwdLevelStart[bl[c] + 1]++;
    for (int i = 0; i < nB; i++) a.bwdLevelStart[i + 1] += a.bwdLevelStart[i];
    a.bwdOrder.resize(nOwn);
    {
        std::vector<int> pos(a.bwdLevelStart.begin(), a.bwdLevelStart.end() - (nB ? 1 : 0));
        if (!nB) pos.clear();
        for (int c = 0; c < nOwn; c++) a.bwdOrder[pos[bl[c]]++] = c;
    }
    a.bwdContig = true;
    for (int b = 0; b < nB && a.bwdContig; b++) {
        int s = a.bwdLevelStart[b], e = a.bwdLevelStart[b + 1];
        if (e > s && a.bwdOrder[e - 1] - a.bwdOrder[s] != e - s - 1) a.bwdContig = false;
    }
    lap_.lap("backward levels");
    if (a.mode >= 1) {
        const int G = a.nGroups;
        a.levNew.resize(nOwn); a.blNew.assign(bl.begin(), bl.begin() + nOwn);
        ffm_parallel_for(nOwn, [&](long lo, long hi) { for (long c = lo; c < hi; c++) a.levNew[c] = lev[a.newToOldCell[c]]; });
        // the group graph must be acyclic in the new numbering: every cross-group face points from a lower to a higher group
        // (new numbering: the group of cell c is the one whose range [grpCell[g], grpCell[g + 1]) holds it)
        std::vector<int> grpOfNew(nOwn);
        ffm_parallel_for(G, [&](long lo, long hi) { for (long g = lo; g < hi; g++) std::fill(grpOfNew.begin() + a.grpCell[g], grpOfNew.begin() + a.grpCell[g + 1], (int)g); });
        std::atomic<int> cyclic(0);
        ffm_parallel_for(F, [&](long lo, long hi) {
            for (long f = lo; f < hi; f++) if (a.u[f] < nOwn && grpOfNew[a.l[f]] > grpOfNew[a.u[f]]) cyclic = 1;
        });
        if (cyclic) { ffm_set_error("internal: group graph not acyclic"); return FFM_ERR_ADDR; }
        // backward order inside each group: by backward level, then descending cell index (a counting sort per group; groups in parallel)
        a.bwdCells.resize(nOwn);
        std::atomic<int> notReverse(0);
        auto groups = [&](long g0, long g1) {
            std::vector<int> cntB;
            for (long g = g0; g < g1; g++) {
                const int c0 = a.grpCell[g], c1 = a.grpCell[g + 1];
                if (c1 <= c0) continue;
                int bMin = bl[c0], bMax = bl[c0];
                for (int c = c0 + 1; c < c1; c++) { bMin = std::min(bMin, bl[c]); bMax = std::max(bMax, bl[c]); }
                cntB.assign((size_t)(bMax - bMin) + 2, 0);
                for (int c = c0; c < c1; c++) cntB[bl[c] - bMin + 1]++;
                for (size_t i = 1; i < cntB.size(); i++) cntB[i] += cntB[i - 1];
                for (int c = c1 - 1; c >= c0; c--) a.bwdCells[c0 + cntB[bl[c] - bMin]++] = c;      // descending cell index inside a level
                for (int p = c0; p < c1; p++) if (a.bwdCells[p] != c1 - 1 - (p - c0)) { notReverse = 1; break; }
            }
        };
        if (G >= 8 && nOwn >= (1 << 20)) {
            static const int nT = [] { const char *e = getenv("FFM_HOST_THREADS"); int t = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return std::max(1, std::min(t, 16)); }();
            std::vector<std::thread> th;
            for (int t = 0; t < nT; t++) th.emplace_back([&, t] { groups((long)G * t / nT, (long)G * (t + 1) / nT); });
            for (auto &x : th) x.join();
        } else groups(0, G);
        a.bwdIsReverse = !notReverse;
    }
    lap_.lap("group checks + backward order");
    if (a.mode == 2 && !ffm_tile_feasible(nOwn, F, a.l.data(), a.u.data())) {
        if (getenv("FFM_VERBOSE")) fprintf(stderr, "ffm: tiled sweeps not applicable (more than 3 lower or upper neighbours): level-scheduled sweeps\n");
        // the tiled sweeps cannot take this mesh / grouping: level-scheduled sweeps instead
        LduAnalysis b;
        FFM_TRY(analyse(N, nOwn, F, l, u, renumber, sortByNewNeighbour, b, nullptr, 0));
        a = std::move(b);
    }
    lap_.lap("feasibility");
    return FFM_OK;
}

// Group hint from cell centres, for hosts that have geometry but no structured indices (an OpenFOAM fvMesh): columns run
// along the axis that consecutive cell labels follow most often (the fastest index of a blockMesh block), the other two
// axes are cut into strips about `tileCells` cells wide (cell spacing estimated from the bounding box and the cell count).
// The result is only a hint: ffm_renumber_hint / ffm_ldu_create_hint check that the label classes form an acyclic group graph
// and fall back to the level-scheduled sweeps otherwise.
extern "C" int ffm_tile_hint_from_centres(int nCells, const double *C /* [3][nCells] */, int tileCells, int *hint)
{
    if (nCells < 0 || (nCells && (!C || !hint))) return FFM_ERR_ARG;
    if (nCells == 0) return FFM_OK;
    if (tileCells <= 0) tileCells = tile_edge();
    double lo[3], hi[3];
    for (int d = 0; d < 3; d++) { lo[d] = hi[d] = C[(size_t)d * nCells]; }
    long votes[3] = {0, 0, 0};
    for (int c = 0; c < nCells; c++) {
        for (int d = 0; d < 3; d++) { const double v = C[(size_t)d * nCells + c]; lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v); }
        if (c + 1 < nCells) {
            double best = -1; int bd = 0;
            for (int d = 0; d < 3; d++) { const double dv = std::fabs(C[(size_t)d * nCells + c + 1] - C[(size_t)d * nCells + c]); if (dv > best) { best = dv; bd = d; } }
            votes[bd]++;
        }
    }
    // the axis that changes between most consecutive labels is the column axis... unless it is the row-wrap axis: take the
    // axis with the most votes (the fastest index changes N - N/nx times, the others far less)
    int col = 0;
    for (int d = 1; d < 3; d++) if (votes[d] > votes[col]) col = d;
    const int a = (col + 1) % 3, b = (col + 2) % 3;
    double L[3];
    for (int d = 0; d < 3; d++) L[d] = std::max(hi[d] - lo[d], 1e-300);
    // cells per axis from N and the aspect ratios (cell centres span L = (n-1) h): n_d ~ (N L_d^2/(L_e L_f))^(1/3)
    auto cellsAlong = [&](int d) { const int e = (d + 1) % 3, f = (d + 2) % 3; return std::max(1.0, std::cbrt((double)nCells * L[d] * L[d] / (L[e] * L[f]))); };
    double ha = L[a] / std::max(cellsAlong(a) - 1.0, 1.0), hb = L[b] / std::max(cellsAlong(b) - 1.0, 1.0);
    // better where the numbering allows it: the smallest step of the coordinate between consecutive cells (a structured block wraps
    // its rows by exactly one spacing).  The estimate above is off by a fraction of a percent (centres span (n-1) h, not n h), enough
    // to put a 17th cell row into a tile: levels of more than 256 cells, split entries, no ring plan for the tiled Amul
    {
        double sa = 1e300, sb = 1e300;
        for (int c = 0; c + 1 < nCells; c++) {
            const double da = std::fabs(C[(size_t)a * nCells + c + 1] - C[(size_t)a * nCells + c]);
            const double db = std::fabs(C[(size_t)b * nCells + c + 1] - C[(size_t)b * nCells + c]);
            if (da > 1e-9 * L[a] && da < sa) sa = da;
            if (db > 1e-9 * L[b] && db < sb) sb = db;
        }
        if (sa < 1e300 && sa >= 0.5 * ha) ha = sa;          // (a step far below the estimate: graded or unstructured, keep the estimate)
        if (sb < 1e300 && sb >= 0.5 * hb) hb = sb;
    }
    for (int c = 0; c < nCells; c++) {
        const int ta = (int)std::floor((C[(size_t)a * nCells + c] - lo[a]) / (tileCells * ha) + 1e-9);
        const int tb = (int)std::floor((C[(size_t)b * nCells + c] - lo[b]) / (tileCells * hb) + 1e-9);
        hint[c] = ffm_tile_label(ta, tb);
    }
    return FFM_OK;
}

extern "C" int ffm_renumber_levels_ext(int nOwned, int nGhost, int nFaces, const int *l, const int *u,
                                       int *newToOldCell, int *newToOldFace)
{ return ffm_renumber_hint(nOwned, nGhost, nFaces, l, u, nullptr, newToOldCell, newToOldFace); }

extern "C" int ffm_renumber_hint(int nOwned, int nGhost, int nFaces, const int *l, const int *u, const int *groupHint,
                                 int *newToOldCell, int *newToOldFace)
{
    if (nOwned < 0 || nGhost < 0 || nFaces < 0 || (nFaces && (!l || !u))) return FFM_ERR_ARG;
    LduAnalysis a;
    FFM_TRY(analyse(nOwned + nGhost, nOwned, nFaces, l, u, true, true, a, groupHint));
    if (newToOldCell) std::copy(a.newToOldCell.begin(), a.newToOldCell.end(), newToOldCell);
    if (newToOldFace) std::copy(a.newToOldFace.begin(), a.newToOldFace.end(), newToOldFace);
    if (a.mode == 2) memo_put(nOwned + nGhost, nFaces, a.l.data(), a.u.data(), a.grpCell);
    return FFM_OK;
}

extern "C" int ffm_renumber_levels(int nCells, int nFaces, const int *l, const int *u,
                                   int *newToOldCell, int *newToOldFace)
{
    if (nCells < 0 || nFaces < 0 || (nFaces && (!l || !u))) return FFM_ERR_ARG;
    LduAnalysis a;
    FFM_TRY(analyse(nCells, nCells, nFaces, l, u, true, true, a));
    if (newToOldCell) std::copy(a.newToOldCell.begin(), a.newToOldCell.end(), newToOldCell);
    if (newToOldFace) std::copy(a.newToOldFace.begin(), a.newToOldFace.end(), newToOldFace);
    if (a.mode == 2) memo_put(nCells, nFaces, a.l.data(), a.u.data(), a.grpCell);
    return FFM_OK;
}

template <class T>
static int upload(ffm_ctx *c, T **dst, const std::vector<T> &v, size_t minCount = 1)
{
    size_t n = std::max(v.size(), minCount);
    FFM_HIP(hipMalloc((void **)dst, n * sizeof(T)));
    if (!v.empty()) FFM_HIP(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return FFM_OK;
}

extern "C" int ffm_ldu_create(ffm_ctx *ctx, int N, int F, const int *l, const int *u, ffm_ldu **out)
{ return ffm_ldu_create_ext(ctx, N, 0, F, l, u, out); }

extern "C" int ffm_ldu_create_ext(ffm_ctx *ctx, int nOwn, int nGhost, int F, const int *l, const int *u, ffm_ldu **out)
{ return ffm_ldu_create_hint(ctx, nOwn, nGhost, F, l, u, nullptr, out); }

static int ldu_create_impl(ffm_ctx *ctx, int nOwn, int nGhost, int F, const int *l, const int *u, const int *groupHint, int forceMode,
                           ffm_ldu **out);

extern "C" int ffm_ldu_create_hint(ffm_ctx *ctx, int nOwn, int nGhost, int F, const int *l, const int *u, const int *groupHint,
                                   ffm_ldu **out)
{
    if (!ctx || !out || nOwn < 0 || nGhost < 0 || F < 0 || (F && (!l || !u))) { ffm_set_error("ffm_ldu_create: bad argument"); return FFM_ERR_ARG; }
    FFM_HIP(hipSetDevice(ctx->device));
    FFM_TRY(ldu_create_impl(ctx, nOwn, nGhost, F, l, u, groupHint, -1, out));
    if ((*out)->sweepMode == 2 && !ffm_tile_usable(*out)) {
        // the tile planner gave up on this grouping (entry order, external references per entry, ghost faces before owned
        // ones): the same matrix with level-scheduled sweeps instead
        if (getenv("FFM_VERBOSE")) fprintf(stderr, "ffm: no tile plan for this mesh / grouping: level-scheduled sweeps\n");
        ffm_ldu_destroy(*out); *out = nullptr;
        FFM_TRY(ldu_create_impl(ctx, nOwn, nGhost, F, l, u, nullptr, 0, out));
    }
    return FFM_OK;
}

static int ldu_create_impl(ffm_ctx *ctx, int nOwn, int nGhost, int F, const int *l, const int *u, const int *groupHint, int forceMode,
                           ffm_ldu **out)
{
    const int N = nOwn + nGhost;
    LduAnalysis a;
    { FfmStageTimer tm_("ldu_create: analyse"); FFM_TRY(analyse(N, nOwn, F, l, u, true, false, a, groupHint, forceMode)); }
    FfmStageTimer tmRest_("ldu_create: layout+upload+tile");
    ffm_ldu *A = new ffm_ldu();
    A->ctx = ctx; A->nCells = N; A->nOwned = nOwn; A->nFaces = F; A->globalCells = nOwn;
    A->identity = a.identity; A->bwdContig = a.bwdContig;
    A->sweepMode = a.mode; A->nGroups = a.nGroups; A->bwdIsReverse = a.bwdIsReverse;
    A->nLevels = (int)a.fwdLevelStart.size() - 1; if (A->nLevels < 0) A->nLevels = 0;
    A->nBwdLevels = (int)a.bwdLevelStart.size() - 1; if (A->nBwdLevels < 0) A->nBwdLevels = 0;
    A->h_fwdLevelStart = a.fwdLevelStart; A->h_bwdLevelStart = a.bwdLevelStart;
    A->h_newToOldCell = a.newToOldCell; A->h_newToOldFace = a.newToOldFace;
    // derived addressing: sliced owner-ELL
    const int nSl = (nOwn + 63) / 64;                 // rows exist for owned cells only
    A->nSlices = nSl;
    std::vector<int> upCnt(N, 0), loCnt(N, 0);
    for (int f = 0; f < F; f++) { upCnt[a.l[f]]++; if (a.u[f] < nOwn) loCnt[a.u[f]]++; }
    std::vector<int> upOff(nSl + 1, 0), loOff(nSl + 1, 0);
    int uniform = -2, uniformLo = -2, maxW = 0;
    for (int sl = 0; sl < nSl; sl++) {
        int wu = 0, wl = 0;
        for (int c = sl * 64; c < std::min(nOwn, sl * 64 + 64); c++) { wu = std::max(wu, upCnt[c]); wl = std::max(wl, loCnt[c]); }
        // at most 32 upper and 32 lower faces per cell (the widest instantiation of the row kernels); of the upper faces, those towards OWNED
        // cells must sit in the first 16 slots (a lower entry packs owner << 4 | slot; checked below) -- the faces towards ghost cells, which
        // follow them in a decomposed matrix and have no lower entry, may use the slots above (coarse GAMG levels at rank boundaries)
        if (wu > 32 || wl > 32) { ffm_set_error("a cell has %d upper / %d lower faces (> 32): not supported by the sliced layout", wu, wl); delete A; return FFM_ERR_UNSUPPORTED; }
        upOff[sl + 1] = upOff[sl] + wu * 64; loOff[sl + 1] = loOff[sl] + wl * 64;
        uniform = (uniform == -2) ? wu : (uniform == wu ? wu : -1);
        uniformLo = (uniformLo == -2) ? wl : (uniformLo == wl ? wl : -1);
        maxW = std::max(maxW, std::max(wu, wl));
    }
    if ((long)nSl * 32 * 64 > 0x7fffffffL || N >= (1 << 27)) { ffm_set_error("mesh too large for int32 packed entries"); delete A; return FFM_ERR_UNSUPPORTED; }
    A->upTotal = upOff[nSl]; A->loTotal = loOff[nSl];
    A->h_upOff = upOff; A->h_loOff = loOff;
    A->upWidthUniform = (uniform >= 0) ? uniform : -1;
    A->loWidthUniform = (uniformLo >= 0) ? uniformLo : -1;
    A->maxW = maxW;
    std::vector<int> upNbr(std::max(A->upTotal, 1), -1), faceSrc(std::max(A->upTotal, 1), -1), loEnt(std::max(A->loTotal, 1), -1);
    A->h_callerToNative.assign(F, -1);
    {
        // upper slots: faces of one owner are consecutive in a.l (owner-sorted), caller order inside
        std::vector<int> slotOfFace(F);
        int prev = -1, slot = 0;
        for (int f = 0; f < F; f++) {
            const int c = a.l[f];
            slot = (c == prev) ? slot + 1 : 0; prev = c;
            slotOfFace[f] = slot;
            if (slot >= 16 && a.u[f] < nOwn) { ffm_set_error("a cell owns more than 16 faces towards owned cells (or they follow its faces towards ghost cells): not supported by the packed layout"); delete A; return FFM_ERR_UNSUPPORTED; }
            const int e = upOff[c >> 6] + slot * 64 + (c & 63);
            upNbr[e] = a.u[f]; faceSrc[e] = a.newToOldFace[f];
            A->h_callerToNative[a.newToOldFace[f]] = e;
        }
        // lower entries in the CALLER's face order (losort of the caller's addressing)
        std::vector<int> oldToNewFace(F);
        for (int f = 0; f < F; f++) oldToNewFace[a.newToOldFace[f]] = f;
        std::vector<int> fill(N, 0);
        for (int of = 0; of < F; of++) {
            const int f = oldToNewFace[of], c = a.u[f];
            if (c >= nOwn) continue;                      // ghost cells have no rows
            const int q = loOff[c >> 6] + fill[c]++ * 64 + (c & 63);
            loEnt[q] = (a.l[f] << 4) | slotOfFace[f];
        }
    }
    if (A->sweepMode == 0 && A->bwdContig) {
        // keep only the first cell of every backward level: ranges are [first, first+count)
        A->h_bwdFirstCell.resize(A->nBwdLevels);
        for (int b = 0; b < A->nBwdLevels; b++) {
            int s = a.bwdLevelStart[b];
            A->h_bwdFirstCell[b] = (a.bwdLevelStart[b + 1] > s) ? a.bwdOrder[s] : 0;
        }
    }
    int rc = FFM_OK;
    do {
        if ((rc = upload(ctx, &A->upOff, upOff))) break;
        if ((rc = upload(ctx, &A->loOff, loOff))) break;
        if ((rc = upload(ctx, &A->upNbr, upNbr))) break;
        if ((rc = upload(ctx, &A->loEnt, loEnt))) break;
        if ((rc = upload(ctx, &A->faceSrc, faceSrc))) break;
        if (A->sweepMode == 0 && !A->bwdContig && (rc = upload(ctx, &A->bwdOrder, a.bwdOrder))) break;
        if (A->sweepMode >= 1) {
            if ((rc = upload(ctx, &A->grpCell, a.grpCell))) break;
            if ((rc = upload(ctx, &A->bwdCells, a.bwdCells))) break;
            if (hipMalloc((void **)&A->sweepTicket, 2 * sizeof(unsigned int)) != hipSuccess) { rc = FFM_ERR_HIP; break; }
            hipMemsetAsync(A->sweepTicket, 0, 2 * sizeof(unsigned int), ctx->stream);
        }
        if (!A->identity && (rc = upload(ctx, &A->cellPerm, a.newToOldCell))) break;
        if (A->sweepMode == 2) {
            A->h_loEnt = loEnt; A->h_upNbr = upNbr;
            // Backward order of the tiled sweeps.  Where the backward dependency levels are the mirror image of the forward ones (a box)
            // the backward sweep walks the forward entries in reverse.  Where they are not (internal walls, unstructured graphs) it STILL
            // can: the reverse of a topological order is a topological order of the reversed graph, and the cells of one forward level are
            // never neighbours, so they form a valid backward entry as well -- neighbours that are then far away in the numbering go
            // through the mailboxes like any other external.  The sweeps of such meshes therefore take the mirror kernels (fused PCG
            // iteration, multi-system sweeps, no permutation passes); FFM_TILE_POS_BACKWARD=1 keeps the separate backward-level order in
            // "position space" (round 2's form).
            static const bool posBackward = getenv("FFM_TILE_POS_BACKWARD") && atoi(getenv("FFM_TILE_POS_BACKWARD")) != 0;
            if (a.bwdIsReverse || posBackward) rc = ffm_tile_build(A, a.levNew, a.blNew, a.grpCell, a.bwdIsReverse ? nullptr : &a.bwdCells);
            else {
                int maxLev = 0;
                for (int c = 0; c < nOwn; c++) maxLev = std::max(maxLev, a.levNew[c]);
                std::vector<int> blSym(nOwn);
                for (int c = 0; c < nOwn; c++) blSym[c] = maxLev - a.levNew[c];
                rc = ffm_tile_build(A, a.levNew, blSym, a.grpCell, nullptr);
            }
            A->h_loEnt.clear(); A->h_loEnt.shrink_to_fit(); A->h_upNbr.clear(); A->h_upNbr.shrink_to_fit();
            if (rc) break;
        }
        {
            // XCD-aware schedule (see ffm_internal.hpp): chunks of 256 rows, binned by the eighth of their dependency level
            const int nChunks = (nOwn + 255) / 256;
            std::vector<std::vector<int>> seq(8);
            if (A->sweepMode == 0 && A->nLevels > 0) {
                int L = 0;
                for (int ch = 0; ch < nChunks; ch++) {
                    const int c0 = ch * 256;
                    while (L + 1 < A->nLevels && a.fwdLevelStart[L + 1] <= c0) L++;
                    const long ls = a.fwdLevelStart[L], le = a.fwdLevelStart[L + 1];
                    int eighth = (le > ls) ? (int)(8L * (c0 - ls) / (le - ls)) : 0;
                    seq[std::min(std::max(eighth, 0), 7)].push_back(ch);
                }
            } else {
                // group-major numbering (tiles): each XCD streams through one contiguous eighth of the rows, so that the rows a
                // row gathers from (same or neighbouring tile) were fetched into the same L2
                for (int ch = 0; ch < nChunks; ch++) seq[std::min(7, (int)(8L * ch / std::max(nChunks, 1)))].push_back(ch);
            }
            size_t mx = 0;
            for (auto &q : seq) mx = std::max(mx, q.size());
            std::vector<int> sched(mx * 8, -1);
            for (int x = 0; x < 8; x++) for (size_t k = 0; k < seq[x].size(); k++) sched[k * 8 + x] = seq[x][k];
            A->nSched = (int)sched.size();
            if ((rc = upload(ctx, &A->rowSched, sched))) break;
        }
        size_t nb = sizeof(double) * (size_t)std::max(N, 1), fb = sizeof(double) * (size_t)std::max(A->upTotal, 1);
        if (hipMalloc((void **)&A->diag, nb) != hipSuccess || hipMalloc((void **)&A->upper, fb) != hipSuccess ||
            hipMalloc((void **)&A->rD, nb) != hipSuccess) { ffm_set_error("ffm_ldu_create: hipMalloc failed"); rc = FFM_ERR_HIP; break; }
        A->lower = A->upper; A->diagBuf = A->diag; A->upperBuf = A->upper;
        hipMemsetAsync(A->diag, 0, nb, ctx->stream); hipMemsetAsync(A->upper, 0, fb, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ffm_set_error("ffm_ldu_create: upload failed"); rc = FFM_ERR_HIP; break; }
    } while (0);
    if (rc) { ffm_ldu_destroy(A); return rc; }
    *out = A;
    return FFM_OK;
}

extern "C" int ffm_ldu_destroy(ffm_ldu *A)
{
    if (!A) return FFM_OK;
    hipSetDevice(A->ctx->device);
    hipStreamSynchronize(A->ctx->stream);
    for (auto &kv : A->graphs) hipGraphExecDestroy(kv.second);
    for (double *w : A->work) hipFree(w);
    hipFree(A->gsProd);
    for (int i = 0; i < 3; i++) hipFree(A->permIn[i]);
    hipFree(A->upOff); hipFree(A->loOff); hipFree(A->upNbr); hipFree(A->loEnt); hipFree(A->faceSrc);
    hipFree(A->bwdOrder); hipFree(A->cellPerm); hipFree(A->callerToNative); hipFree(A->smallFwdStart); hipFree(A->smallBwdRange); hipFree(A->flowOrder);
    hipFree(A->grpCell); hipFree(A->bwdCells); hipFree(A->sweepTicket); hipFree(A->rowSched);
    ffm_tile_free(A);
    hipFree(A->ghSendCells); hipFree(A->ghSendBuf); if (A->ghSendBuf_h) hipHostFree(A->ghSendBuf_h); if (A->ghRecvBuf_h) hipHostFree(A->ghRecvBuf_h);
    hipFree(A->diagBuf); hipFree(A->upperBuf); hipFree(A->lowerBuf); hipFree(A->rD);
    hipFree(A->ifFaceCells); hipFree(A->ifBou); hipFree(A->ifInt); hipFree(A->haloSend); hipFree(A->haloRecv);
    hipFree(A->ifCell); hipFree(A->ifCellStart); hipFree(A->ifItem);
    delete A;
    return FFM_OK;
}

extern "C" int ffm_ldu_ncells(const ffm_ldu *A) { return A ? A->nCells : FFM_ERR_ARG; }
extern "C" int ffm_ldu_sweep_mode(const ffm_ldu *A) { return !A ? FFM_ERR_ARG : A->sweepMode; }
extern "C" int ffm_ldu_nowned(const ffm_ldu *A) { return A ? A->nOwned : FFM_ERR_ARG; }
extern "C" int ffm_ldu_nfaces(const ffm_ldu *A) { return A ? A->nFaces : FFM_ERR_ARG; }
extern "C" int ffm_ldu_nlevels(const ffm_ldu *A) { return A ? A->nLevels : FFM_ERR_ARG; }
extern "C" int ffm_ldu_is_native_order(const ffm_ldu *A) { return A ? (A->identity ? 1 : 0) : FFM_ERR_ARG; }
extern "C" int ffm_ldu_get_cell_order(const ffm_ldu *A, int *newToOld)
{
    if (!A || !newToOld) return FFM_ERR_ARG;
    std::copy(A->h_newToOldCell.begin(), A->h_newToOldCell.end(), newToOld);
    return FFM_OK;
}
extern "C" int ffm_ldu_n_native_faces(const ffm_ldu *A) { return A ? A->upTotal : FFM_ERR_ARG; }
extern "C" int ffm_ldu_get_face_map(const ffm_ldu *A, int *callerToNative)
{
    if (!A || !callerToNative) return FFM_ERR_ARG;
    std::copy(A->h_callerToNative.begin(), A->h_callerToNative.end(), callerToNative);
    return FFM_OK;
}
extern "C" int ffm_ldu_set_global_cells(ffm_ldu *A, long g) { if (!A || g < A->nOwned) return FFM_ERR_ARG; A->globalCells = g; return FFM_OK; }

int ffm_ldu_work(ffm_ldu *A, int idx, double **out)
{
    if (idx < 0 || idx > 63) return FFM_ERR_ARG;          // 32 ..: the lanes of ffm_solve_multi_d
    if ((int)A->work.size() <= idx) A->work.resize(idx + 1, nullptr);
    if (!A->work[idx]) FFM_HIP(hipMalloc((void **)&A->work[idx], sizeof(double) * (size_t)std::max(A->nCells, 1)));
    *out = A->work[idx];
    return FFM_OK;
}

// ----------------------------------------------------- permutation kernels ---
__global__ void k_gather(long n, const int *__restrict__ perm, const double *__restrict__ src, double *__restrict__ dst)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[perm[i]];
}
__global__ void k_scatter(long n, const int *__restrict__ perm, const double *__restrict__ src, double *__restrict__ dst)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[perm[i]] = src[i];
}

static inline int stream_grid(long n) { long g = (n + 255) / 256; return (int)std::max(1L, std::min(g, (long)RED_BLOCKS)); }

int ffm_to_internal(ffm_ldu *A, const double *x_d, int slot, const double **out)
{
    if (A->identity) { *out = x_d; return FFM_OK; }
    if (!A->permIn[slot]) FFM_HIP(hipMalloc((void **)&A->permIn[slot], sizeof(double) * (size_t)std::max(A->nCells, 1)));
    hipLaunchKernelGGL(k_gather, dim3(stream_grid(A->nCells)), dim3(256), 0, A->ctx->stream, (long)A->nCells, A->cellPerm, x_d, A->permIn[slot]);
    *out = A->permIn[slot];
    return FFM_OK;
}
int ffm_from_internal(ffm_ldu *A, const double *xin, double *x_d)
{
    if (A->identity) {
        if (xin != x_d) FFM_HIP(hipMemcpyAsync(x_d, xin, sizeof(double) * A->nCells, hipMemcpyDeviceToDevice, A->ctx->stream));
        return FFM_OK;
    }
    hipLaunchKernelGGL(k_scatter, dim3(stream_grid(A->nCells)), dim3(256), 0, A->ctx->stream, (long)A->nCells, A->cellPerm, xin, x_d);
    return FFM_OK;
}

// gather with padding: dst[e] = (src_idx[e] >= 0) ? src[src_idx[e]] : 0   (LDU face order -> native faces)
__global__ void k_gather_faces(long n, const int *__restrict__ srcIdx, const double *__restrict__ src, double *__restrict__ dst)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int f = srcIdx[i];
        dst[i] = (f >= 0) ? src[f] : 0.0;
    }
}

extern "C" int ffm_ldu_set_coeffs_d(ffm_ldu *A, const double *diag_d, const double *upper_d, const double *lower_d)
{
    if (!A || !diag_d || (A->nFaces && !upper_d)) return FFM_ERR_ARG;
    hipStream_t s = A->ctx->stream;
    const size_t nb = sizeof(double) * A->nCells, fb = sizeof(double) * std::max(A->upTotal, 1);
    A->diag = A->diagBuf; A->upper = A->upperBuf;
    if (lower_d && lower_d != upper_d) {
        if (!A->lowerBuf) FFM_HIP(hipMalloc((void **)&A->lowerBuf, fb));
        A->lower = A->lowerBuf; A->symmetric = false;
    } else { A->lower = A->upper; A->symmetric = true; }
    if (A->identity) { if (nb) FFM_HIP(hipMemcpyAsync(A->diag, diag_d, nb, hipMemcpyDeviceToDevice, s)); }
    else hipLaunchKernelGGL(k_gather, dim3(stream_grid(A->nCells)), dim3(256), 0, s, (long)A->nCells, A->cellPerm, diag_d, A->diag);
    if (A->upTotal) {
        hipLaunchKernelGGL(k_gather_faces, dim3(stream_grid(A->upTotal)), dim3(256), 0, s, (long)A->upTotal, A->faceSrc, upper_d, A->upper);
        if (!A->symmetric)
            hipLaunchKernelGGL(k_gather_faces, dim3(stream_grid(A->upTotal)), dim3(256), 0, s, (long)A->upTotal, A->faceSrc, lower_d, A->lower);
    }
    FFM_HIP(hipGetLastError());
    A->coeffEpoch++; A->offDiagEpoch++;
    return FFM_OK;
}

extern "C" int ffm_ldu_set_coeffs(ffm_ldu *A, const double *diag, const double *upper, const double *lower)
{
    if (!A || !diag || (A->nFaces && !upper)) return FFM_ERR_ARG;
    // stage through temporary device buffers (host pointers may be pageable)
    double *d = nullptr, *u = nullptr, *l = nullptr;
    const size_t nb = sizeof(double) * std::max(A->nCells, 1), fb = sizeof(double) * std::max(A->nFaces, 1);
    FFM_HIP(hipMalloc((void **)&d, nb)); FFM_HIP(hipMalloc((void **)&u, fb));
    if (lower) FFM_HIP(hipMalloc((void **)&l, fb));
    FFM_TRY(ffm_h2d(A->ctx, d, diag, sizeof(double) * A->nCells));
    if (A->nFaces) FFM_TRY(ffm_h2d(A->ctx, u, upper, sizeof(double) * A->nFaces));
    if (lower && A->nFaces) FFM_TRY(ffm_h2d(A->ctx, l, lower, sizeof(double) * A->nFaces));
    FFM_HIP(hipDeviceSynchronize());           // the null-stream uploads have landed before the context's (non-blocking) stream reads them
    int rc = ffm_ldu_set_coeffs_d(A, d, u, l);
    hipStreamSynchronize(A->ctx->stream);
    hipFree(d); hipFree(u); hipFree(l);
    return rc;
}

// native-layout entry for hosts that assemble directly in the library's face layout
extern "C" int ffm_ldu_set_coeffs_native_d(ffm_ldu *A, const double *diag_d, const double *upper_d, const double *lower_d)
{
    if (!A || !diag_d || (A->upTotal && !upper_d)) return FFM_ERR_ARG;
    if (!A->identity) { ffm_set_error("native coefficient layout needs the library's cell order (ffm_renumber_levels)"); return FFM_ERR_UNSUPPORTED; }
    hipStream_t s = A->ctx->stream;
    const size_t nb = sizeof(double) * A->nCells, fb = sizeof(double) * A->upTotal;
    A->diag = A->diagBuf; A->upper = A->upperBuf;
    if (lower_d && lower_d != upper_d) {
        if (!A->lowerBuf) FFM_HIP(hipMalloc((void **)&A->lowerBuf, std::max(fb, sizeof(double))));
        A->lower = A->lowerBuf; A->symmetric = false;
        if (fb) FFM_HIP(hipMemcpyAsync(A->lower, lower_d, fb, hipMemcpyDeviceToDevice, s));
    } else { A->lower = A->upper; A->symmetric = true; }
    if (nb) FFM_HIP(hipMemcpyAsync(A->diag, diag_d, nb, hipMemcpyDeviceToDevice, s));
    if (fb) FFM_HIP(hipMemcpyAsync(A->upper, upper_d, fb, hipMemcpyDeviceToDevice, s));
    A->coeffEpoch++; A->offDiagEpoch++;
    return FFM_OK;
}

// Zero-copy form: the matrix uses the caller's device arrays (native layout) until the next set / bind call; the caller keeps
// them alive and unchanged meanwhile.  offDiagUnchanged != 0: upper / lower hold the same values as at the previous call
// (e.g. the components of a vector equation, which differ in the boundary diagonal only), so layouts derived from them are kept.
extern "C" int ffm_ldu_bind_coeffs_native_d(ffm_ldu *A, const double *diag_d, const double *upper_d, const double *lower_d,
                                            int offDiagUnchanged)
{
    if (!A || !diag_d || (A->upTotal && !upper_d)) return FFM_ERR_ARG;
    if (!A->identity) { ffm_set_error("native coefficient layout needs the library's cell order (ffm_renumber_levels)"); return FFM_ERR_UNSUPPORTED; }
    A->diag = const_cast<double *>(diag_d);
    A->upper = const_cast<double *>(upper_d ? upper_d : A->upperBuf);
    if (lower_d && lower_d != upper_d) { A->lower = const_cast<double *>(lower_d); A->symmetric = false; }
    else { A->lower = A->upper; A->symmetric = true; }
    A->coeffEpoch++;
    if (!offDiagUnchanged) A->offDiagEpoch++;
    return FFM_OK;
}

// after a bind: the matrix points at its own coefficient buffers again (their contents are whatever the last set call left), so that
// nothing in the library refers to the caller's arrays once they are gone
extern "C" int ffm_ldu_unbind_coeffs(ffm_ldu *A)
{
    if (!A) return FFM_ERR_ARG;
    A->diag = A->diagBuf; A->upper = A->upperBuf;
    if (A->symmetric || !A->lowerBuf) { A->lower = A->upper; A->symmetric = true; } else A->lower = A->lowerBuf;
    // (the layouts derived from the off-diagonal coefficients stay as they are: a following bind with offDiagUnchanged != 0 -- the same
    // values again, e.g. the next specie sharing its coefficient arrays -- finds them current; any other set / bind renews them)
    A->coeffEpoch++;
    return FFM_OK;
}

LduView ffm_view(const ffm_ldu *A)
{
    LduView v; v.N = A->nOwned; v.upOff = A->upOff; v.loOff = A->loOff; v.upNbr = A->upNbr; v.loEnt = A->loEnt;
    v.upW = A->upWidthUniform; v.loW = A->loWidthUniform; v.sched = A->rowSched; v.nSched = A->nSched;
    return v;
}

// ----------------------------------------------------------- lduMatrix::Amul ---
// One thread per row, grid-stride over a fixed grid; a wave = one slice, so all
// index/coefficient loads are unit-stride and the slot loops are wave-uniform.
// MODE 0: y = A x.  MODE 1: r = b - A x (lduMatrix::residual order).  MODE 2: s = sumA.
// MODE 3: y = A x and sumA (written through `partials`) in one pass over the coefficients -- the first two things every
// solver does with a freshly bound matrix (wA = A psi; normFactor needs sumA).
// DOT: additionally accumulates the block-partial of x[c]*y[c] (PCG's wApA).
template <int MODE, bool DOT, int W>
__global__ __launch_bounds__(256) void k_rows(LduView v, const double *__restrict__ diag,
                                              const double *__restrict__ upper, const double *__restrict__ lower,
                                              const double *__restrict__ x, const double *__restrict__ b,
                                              double *__restrict__ y, double *__restrict__ partials)
{
    __shared__ double sm[4];
    double dot = 0.0;
    for (int it = blockIdx.x; it < v.nSched; it += gridDim.x) {
        const int chunk = v.sched[it];
        const int c = chunk * 256 + (int)threadIdx.x;
        if (chunk < 0 || c >= v.N) continue;
        RowEnt<W> L, U;
        load_lower<W, true>(v, c, L);
        load_upper<W, false, true>(v, c, U);
        double al[W], au[W], xl[W], xu[W];
#pragma unroll
        for (int s = 0; s < W; s++) { al[s] = lower[L.f[s]]; au[s] = upper[U.f[s]]; }
        if (MODE != 2) {
#pragma unroll
            for (int s = 0; s < W; s++) { xl[s] = x[L.nb[s]]; xu[s] = x[U.nb[s]]; }
        }
        double xc = 0.0, acc;
        if (MODE != 2) xc = x[c];
        const double dc = __builtin_nontemporal_load(&diag[c]);
        double sum = dc;
        if (MODE == 0 || MODE == 3) acc = dc * xc;
        else if (MODE == 1) acc = __builtin_nontemporal_load(&b[c]) - dc * xc;
        else acc = dc;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) {
            if (MODE == 0 || MODE == 3) acc += al[s] * xl[s];
            else if (MODE == 1) acc -= al[s] * xl[s];
            else acc += al[s];
            if (MODE == 3) sum += al[s];
        }
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) {
            if (MODE == 0 || MODE == 3) acc += au[s] * xu[s];
            else if (MODE == 1) acc -= au[s] * xu[s];
            else acc += au[s];
            if (MODE == 3) sum += au[s];
        }
        __builtin_nontemporal_store(acc, &y[c]);
        if (MODE == 3) __builtin_nontemporal_store(sum, &partials[c]);
        if (DOT) dot += acc * xc;
    }
    if (DOT) {
        double r = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(1024) void k_sum_partials(int n, const double *__restrict__ partials, double *__restrict__ scal, int slot)
{
    __shared__ double sm[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
    double r = block_sum(acc, sm);
    if (threadIdx.x == 0) scal[slot] = r;
}

// grid of the row kernels: a multiple of 8 so that schedule entry i always meets blockIdx % 8 == i % 8
static inline int rows_grid(const ffm_ldu *A)
{
    // Fewer workgroups in flight on very large meshes: the rows in flight per XCD (grid/8 chunks of 256 rows, ~110 B each)
    // should not exceed the 4 MB L2 by much, or the neighbour gathers miss it (measured at 64 M rows: 1.16 ms with 1024
    // workgroups, 1.37 ms with 2048; below 10 M rows 2048 is marginally better)
    static const int cap = getenv("FFM_ROWS_GRID") ? std::min(atoi(getenv("FFM_ROWS_GRID")), RED_BLOCKS) : 0;
    int g = std::min(A->nSched, cap ? cap : (A->nOwned >= (32 << 20) ? 1024 : RED_BLOCKS)); g = (g + 7) & ~7; return std::max(g, 8);
}

int ffm_k_spmv(ffm_ldu *A, const double *x, double *y, bool transpose)
{
    const double *up = transpose ? A->lower : A->upper, *lo = transpose ? A->upper : A->lower;
    if (ffm_tile_amul_usable(A) && !getenv("FFM_NO_TILE_AMUL")) {
        // the tiled kernel computes the rows without their ghost faces: the ghost refresh overlaps it and is waited for by the tail
        if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange_begin(A, const_cast<double *>(x)));
        FFM_TRY(ffm_tile_amul(A, x, y, -1));
        return FFM_OK;
    }
    if (!transpose && ffm_tile_amul_asym_usable(A)) {
        if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange_begin(A, const_cast<double *>(x)));
        return ffm_tile_amul_asym(A, x, y);
    }
    if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange(A, const_cast<double *>(x)));   // refresh ghost columns
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows<0, false, W>), dim3(rows_grid(A)), dim3(256), 0, A->ctx->stream, ffm_view(A),
                                               A->diag, up, lo, x, (const double *)nullptr, y, (double *)nullptr));
    FFM_HIP(hipGetLastError());
    if (!A->ifaces.empty()) FFM_TRY(ffm_halo_update(A, x, y, transpose ? A->ifInt : A->ifBou, -1.0));
    return FFM_OK;
}

// y_i = A_i x_i for NF matrices with the off-diagonal coefficients upper / lower and their own diagonals (the lanes of
// ffm_solve_multi_d): addressing and coefficients are read once per row; per system the row sum is k_rows<0> / <3>, term for term.
template <int NF> struct RowsMulti { const double *diag[NF], *x[NF]; double *y[NF], *sum[NF]; };
template <int MODE, int NF, int W>
__global__ __launch_bounds__(256) void k_rows_m(LduView v, const double *__restrict__ upper, const double *__restrict__ lower, RowsMulti<NF> m)
{
    for (int it = blockIdx.x; it < v.nSched; it += gridDim.x) {
        const int chunk = v.sched[it];
        const int c = chunk * 256 + (int)threadIdx.x;
        if (chunk < 0 || c >= v.N) continue;
        RowEnt<W> L, U;
        load_lower<W, true>(v, c, L);
        load_upper<W, false, true>(v, c, U);
        double al[W], au[W];
#pragma unroll
        for (int s = 0; s < W; s++) { al[s] = lower[L.f[s]]; au[s] = upper[U.f[s]]; }
#pragma unroll
        for (int i = 0; i < NF; i++) {
            const double *__restrict__ x = m.x[i];
            double xl[W], xu[W];
#pragma unroll
            for (int s = 0; s < W; s++) { xl[s] = x[L.nb[s]]; xu[s] = x[U.nb[s]]; }
            const double xc = x[c], dc = __builtin_nontemporal_load(&m.diag[i][c]);
            double acc = dc * xc, sum = dc;
#pragma unroll
            for (int s = 0; s < W; s++) if (L.on[s]) { acc += al[s] * xl[s]; if (MODE == 3) sum += al[s]; }
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) { acc += au[s] * xu[s]; if (MODE == 3) sum += au[s]; }
            __builtin_nontemporal_store(acc, &m.y[i][c]);
            if (MODE == 3) __builtin_nontemporal_store(sum, &m.sum[i][c]);
        }
    }
}
template <int MODE, int NF>
static void rows_multi_launch(ffm_ldu *A, const double *const *diag, const double *const *x, double *const *y, double *const *sumA)
{
    RowsMulti<NF> m;
    for (int i = 0; i < NF; i++) { m.diag[i] = diag[i]; m.x[i] = x[i]; m.y[i] = y[i]; m.sum[i] = sumA ? sumA[i] : nullptr; }
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows_m<MODE, NF, W>), dim3(rows_grid(A)), dim3(256), 0, A->ctx->stream, ffm_view(A),
                                               (const double *)A->upper, (const double *)A->lower, m));
}
// n = 2 .. 4 systems; sumA != NULL: also sumA[i] = the row sums of system i (lduMatrix::sumA).  Matrices without coupled patches.
int ffm_k_spmv_multi(ffm_ldu *A, int n, const double *const *diag, const double *const *x, double *const *y, double *const *sumA)
{
    if (n < 2 || n > 4 || !A->ifaces.empty()) return FFM_ERR_ARG;
    if (!A->ghNbrRank.empty()) for (int i = 0; i < n; i++) FFM_TRY(ffm_ghost_exchange(A, const_cast<double *>(x[i])));
    if (sumA) { if (n == 2) rows_multi_launch<3, 2>(A, diag, x, y, sumA); else if (n == 3) rows_multi_launch<3, 3>(A, diag, x, y, sumA); else rows_multi_launch<3, 4>(A, diag, x, y, sumA); }
    else { if (n == 2) rows_multi_launch<0, 2>(A, diag, x, y, sumA); else if (n == 3) rows_multi_launch<0, 3>(A, diag, x, y, sumA); else rows_multi_launch<0, 4>(A, diag, x, y, sumA); }
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_k_spmv_dot(ffm_ldu *A, const double *x, double *y, int slot)
{
    if (!A->ifaces.empty()) {  // the halo term changes y after the row kernel: separate dot
        FFM_TRY(ffm_k_spmv(A, x, y, false));
        return ffm_k_dot(A->ctx, y, x, A->nOwned, slot);
    }
    if (ffm_tile_amul_usable(A) && !getenv("FFM_NO_TILE_AMUL")) {
        if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange_begin(A, const_cast<double *>(x)));
        const int rc = ffm_tile_amul(A, x, y, slot);
        if (rc == 1) return ffm_k_dot(A->ctx, y, x, A->nOwned, slot);      // ghost faces were added after the tiled kernel
        return rc;
    }
    if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange(A, const_cast<double *>(x)));
    const int g = rows_grid(A);
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows<0, true, W>), dim3(g), dim3(256), 0, A->ctx->stream, ffm_view(A), A->diag,
                                               A->upper, A->lower, x, (const double *)nullptr, y, A->ctx->partials_d));
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, A->ctx->stream, g, A->ctx->partials_d, A->ctx->scal_d, slot);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_k_residual(ffm_ldu *A, const double *x, const double *b, double *r)
{
    if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange(A, const_cast<double *>(x)));
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows<1, false, W>), dim3(rows_grid(A)), dim3(256), 0, A->ctx->stream, ffm_view(A),
                                               A->diag, A->upper, A->lower, x, b, r, (double *)nullptr));
    FFM_HIP(hipGetLastError());
    if (!A->ifaces.empty()) FFM_TRY(ffm_halo_update(A, x, r, A->ifBou, +1.0));
    return FFM_OK;
}

// y = A x and s = sumA of the same matrix: one pass where the row kernel does the Amul, two where the tiled kernel does
int ffm_k_spmv_sumA(ffm_ldu *A, const double *x, double *y, double *s)
{
    if ((ffm_tile_amul_usable(A) && !getenv("FFM_NO_TILE_AMUL")) || getenv("FFM_NO_FUSED_SUMA")) {
        FFM_TRY(ffm_k_spmv(A, x, y, false));
        return ffm_k_sumA(A, s);
    }
    if (!A->ghNbrRank.empty()) FFM_TRY(ffm_ghost_exchange(A, const_cast<double *>(x)));
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows<3, false, W>), dim3(rows_grid(A)), dim3(256), 0, A->ctx->stream, ffm_view(A),
                                               A->diag, A->upper, A->lower, x, (const double *)nullptr, y, s));
    FFM_HIP(hipGetLastError());
    if (!A->ifaces.empty()) {
        FFM_TRY(ffm_halo_update(A, x, y, A->ifBou, -1.0));
        FFM_TRY(ffm_halo_apply(A, s, A->ifBou, nullptr, -1.0));
    }
    return FFM_OK;
}

int ffm_k_sumA(ffm_ldu *A, double *s)
{
    FFM_DISPATCH_W(A->maxW, hipLaunchKernelGGL((k_rows<2, false, W>), dim3(rows_grid(A)), dim3(256), 0, A->ctx->stream, ffm_view(A),
                                               A->diag, A->upper, A->lower, (const double *)nullptr, (const double *)nullptr, s,
                                               (double *)nullptr));
    FFM_HIP(hipGetLastError());
    // coupled patches: sumA[faceCells] -= interfaceBouCoeffs
    if (!A->ifaces.empty()) FFM_TRY(ffm_halo_apply(A, s, A->ifBou, nullptr, -1.0));
    return FFM_OK;
}

// ------------------------------------------------------------- public entry ---
extern "C" int ffm_spmv(ffm_ldu *A, const double *x_d, double *y_d)
{
    if (!A || !x_d || !y_d) return FFM_ERR_ARG;
    const double *xi; FFM_TRY(ffm_to_internal(A, x_d, 0, &xi));
    if (A->identity) return ffm_k_spmv(A, xi, y_d, false);
    double *yi; FFM_TRY(ffm_ldu_work(A, 0, &yi));
    FFM_TRY(ffm_k_spmv(A, xi, yi, false));
    return ffm_from_internal(A, yi, y_d);
}
extern "C" int ffm_tmul(ffm_ldu *A, const double *x_d, double *y_d)
{
    if (!A || !x_d || !y_d) return FFM_ERR_ARG;
    const double *xi; FFM_TRY(ffm_to_internal(A, x_d, 0, &xi));
    if (A->identity) return ffm_k_spmv(A, xi, y_d, true);
    double *yi; FFM_TRY(ffm_ldu_work(A, 0, &yi));
    FFM_TRY(ffm_k_spmv(A, xi, yi, true));
    return ffm_from_internal(A, yi, y_d);
}
extern "C" int ffm_sumA(ffm_ldu *A, double *s_d)
{
    if (!A || !s_d) return FFM_ERR_ARG;
    if (A->identity) return ffm_k_sumA(A, s_d);
    double *si; FFM_TRY(ffm_ldu_work(A, 0, &si));
    FFM_TRY(ffm_k_sumA(A, si));
    return ffm_from_internal(A, si, s_d);
}
extern "C" int ffm_residual(ffm_ldu *A, const double *x_d, const double *b_d, double *r_d)
{
    if (!A || !x_d || !b_d || !r_d) return FFM_ERR_ARG;
    const double *xi, *bi;
    FFM_TRY(ffm_to_internal(A, x_d, 0, &xi)); FFM_TRY(ffm_to_internal(A, b_d, 1, &bi));
    if (A->identity) return ffm_k_residual(A, xi, bi, r_d);
    double *ri; FFM_TRY(ffm_ldu_work(A, 0, &ri));
    FFM_TRY(ffm_k_residual(A, xi, bi, ri));
    return ffm_from_internal(A, ri, r_d);
}

extern "C" int ffm_bench_spmv(ffm_ldu *A, const double *x_d, double *y_d, int reps, double *avg_ms)
{
    if (!A || !x_d || !y_d || reps < 1 || !avg_ms) return FFM_ERR_ARG;
    if (!A->identity) { ffm_set_error("ffm_bench_spmv needs a level-major (native order) LDU"); return FFM_ERR_UNSUPPORTED; }
    hipEvent_t e0, e1;
    FFM_HIP(hipEventCreate(&e0)); FFM_HIP(hipEventCreate(&e1));
    FFM_TRY(ffm_k_spmv(A, x_d, y_d, false));  // warm-up
    FFM_HIP(hipEventRecord(e0, A->ctx->stream));
    for (int i = 0; i < reps; i++) FFM_TRY(ffm_k_spmv(A, x_d, y_d, false));
    FFM_HIP(hipEventRecord(e1, A->ctx->stream));
    FFM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FFM_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return FFM_OK;
}
