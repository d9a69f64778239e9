// ffm_partition.cpp -- domain decomposition of an arbitrary LDU graph (host only): the replacement of decomposePar's
// partitioners for this path.  The reference decomposes with scotch (cases/steckler/system/decomposeParDict:18-20) or `simple`
// (cases/wallFireSpread2D/system/decomposeParDict:18-27), BASELINE.json names METIS; none of these libraries exists in this
// image, so two partitioners are written out here:
//   ffm_partition_rcb     recursive coordinate bisection of the cell centres (cuts the longest extent at the median; any
//                         number of parts: a set is split in the proportion floor(n/2) : ceil(n/2))
//   ffm_partition_graph   greedy graph growing on the cell graph (breadth-first regions of equal size grown one after the
//                         other from the frontier of what is already assigned), for meshes without usable geometry
// and the sub-domain builder that turns `part[cell]` into what ffm_ldu_create_ext / ffm_ldu_set_ghost_exchange /
// ffm_ldu_set_exchange_tags take (the ghost-cell form of csrc/ffm_comm.hip) and, for hosts that keep OpenFOAM's processor
// patches, the cut faces per neighbour rank in an order both sides agree on (ascending global face label):
//   * owned cells keep their relative (global) order, so owner < neighbour and the upper-triangular face order survive and
//     DIC / DILU / Gauss-Seidel on a sub-domain are the block-Jacobi restriction of the global operators;
//   * ghost cells follow, grouped by neighbour rank (ascending), inside a group by ascending global cell label; the sender
//     lists the same cells in the same order, so a receive lands directly in the ghost range;
//   * a cut face is a face of the sub-domain owned by its owned cell; where the global owner lies on the other rank the
//     face is flipped (local upper coefficient = global lower coefficient): `flip`.
#include "../../include/ffm.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <queue>
#include <vector>

void ffm_set_error(const char *fmt, ...);

namespace {
// split idx[lo,hi) into nParts parts numbered from `first` by recursive bisection of the centres
void rcb(const double *C, int N, std::vector<int> &idx, int lo, int hi, int first, int nParts, int *part)
{
    if (nParts <= 1 || hi - lo <= 1) { for (int i = lo; i < hi; i++) part[idx[i]] = first; return; }
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = lo; i < hi; i++) for (int d = 0; d < 3; d++) { const double v = C[(size_t)d * N + idx[i]]; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
    int ax = 0;
    for (int d = 1; d < 3; d++) if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
    const int nl = nParts / 2;
    const int cut = lo + (int)((long)(hi - lo) * nl / nParts);
    const double *c = C + (size_t)ax * N;
    // ties broken by the cell label: the partition is a function of the input alone
    std::nth_element(idx.begin() + lo, idx.begin() + cut, idx.begin() + hi, [&](int a, int b) { return c[a] < c[b] || (c[a] == c[b] && a < b); });
    rcb(C, N, idx, lo, cut, first, nl, part);
    rcb(C, N, idx, cut, hi, first + nl, nParts - nl, part);
}
}  // namespace

extern "C" int ffm_partition_rcb(int nCells, const double *C, int nParts, int *part)
{
    if (nCells < 0 || nParts < 1 || (nCells && (!C || !part))) return FFM_ERR_ARG;
    std::vector<int> idx(nCells);
    std::iota(idx.begin(), idx.end(), 0);
    rcb(C, nCells, idx, 0, nCells, 0, nParts, part);
    return FFM_OK;
}

extern "C" int ffm_partition_graph(int nCells, int nFaces, const int *l, const int *u, int nParts, int *part)
{
    if (nCells < 0 || nFaces < 0 || nParts < 1 || (nCells && !part) || (nFaces && (!l || !u))) return FFM_ERR_ARG;
    std::vector<int> start(nCells + 1, 0), adj(2 * (size_t)nFaces);
    for (int f = 0; f < nFaces; f++) {
        if (l[f] < 0 || u[f] < 0 || l[f] >= nCells || u[f] >= nCells) { ffm_set_error("ffm_partition_graph: face %d out of range", f); return FFM_ERR_ARG; }
        start[l[f] + 1]++; start[u[f] + 1]++;
    }
    for (int c = 0; c < nCells; c++) start[c + 1] += start[c];
    { std::vector<int> pos(start.begin(), start.end() - 1); for (int f = 0; f < nFaces; f++) { adj[pos[l[f]]++] = u[f]; adj[pos[u[f]]++] = l[f]; } }
    std::fill(part, part + nCells, -1);
    std::vector<int> frontier;                       // unassigned cells next to assigned ones, in the order they were met
    size_t fpos = 0;
    int assigned = 0, scan = 0;
    for (int p = 0; p < nParts; p++) {
        const int target = (int)(((long)nCells * (p + 1)) / nParts) - assigned;
        int got = 0;
        std::queue<int> q;
        while (got < target) {
            if (q.empty()) {                         // seed: the oldest frontier cell still free, else the first free cell
                int seed = -1;
                while (fpos < frontier.size()) { const int c = frontier[fpos++]; if (part[c] < 0) { seed = c; break; } }
                if (seed < 0) { while (scan < nCells && part[scan] >= 0) scan++; if (scan >= nCells) break; seed = scan; }
                part[seed] = p; got++; q.push(seed);
                continue;
            }
            const int c = q.front(); q.pop();
            for (int k = start[c]; k < start[c + 1] && got < target; k++) {
                const int n = adj[k];
                if (part[n] < 0) { part[n] = p; got++; q.push(n); }
            }
        }
        // what the region still touches becomes the frontier of the next parts
        while (!q.empty()) { const int c = q.front(); q.pop(); for (int k = start[c]; k < start[c + 1]; k++) if (part[adj[k]] < 0) frontier.push_back(adj[k]); }
        assigned += got;
    }
    for (int c = 0; c < nCells; c++) if (part[c] < 0) part[c] = nParts - 1;
    return FFM_OK;
}

struct ffm_subdomain {
    int rank = 0, nOwned = 0, nGhost = 0;
    std::vector<int> globalCell;                 // [nOwned + nGhost]
    std::vector<int> l, u, globalFace, flip;     // local faces in upper-triangular order
    std::vector<int> nbrRank, sendCount, recvCount, tags, sendCells;
    // cut faces per neighbour in ascending global face order (processor-patch form)
    std::vector<int> cutStart, cutCell, cutFace, cutFlip;
};

extern "C" int ffm_subdomain_create(int nCells, int nFaces, const int *l, const int *u, const int *part, int nParts, int rank,
                                    ffm_subdomain **out)
{
    if (!out || nCells < 0 || nFaces < 0 || nParts < 1 || rank < 0 || rank >= nParts || (nCells && !part) || (nFaces && (!l || !u))) return FFM_ERR_ARG;
    for (int c = 0; c < nCells; c++) if (part[c] < 0 || part[c] >= nParts) { ffm_set_error("ffm_subdomain_create: part[%d] = %d", c, part[c]); return FFM_ERR_ARG; }
    ffm_subdomain *S = new ffm_subdomain();
    S->rank = rank;
    std::vector<int> local(nCells, -1);
    for (int c = 0; c < nCells; c++) if (part[c] == rank) { local[c] = S->nOwned++; S->globalCell.push_back(c); }
    // ghosts: cells of other ranks that share a face with an owned cell; per neighbour rank, ascending global label
    std::vector<std::pair<int, int>> gh, snd;    // (rank, global cell): what this rank receives / sends
    for (int f = 0; f < nFaces; f++) {
        const int a = l[f], b = u[f];
        if (a < 0 || b >= nCells || a >= b) { ffm_set_error("ffm_subdomain_create: face %d has l=%d u=%d", f, a, b); delete S; return FFM_ERR_ADDR; }
        if (part[a] == rank && part[b] != rank) { gh.push_back({part[b], b}); snd.push_back({part[b], a}); }
        if (part[b] == rank && part[a] != rank) { gh.push_back({part[a], a}); snd.push_back({part[a], b}); }
    }
    auto uniq = [](std::vector<std::pair<int, int>> &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(gh); uniq(snd);
    for (auto &g : gh) { local[g.second] = S->nOwned + S->nGhost++; S->globalCell.push_back(g.second); }
    // neighbour entries: one per neighbour rank, ascending
    {
        size_t i = 0, j = 0;
        while (i < gh.size() || j < snd.size()) {
            const int r = std::min(i < gh.size() ? gh[i].first : nParts, j < snd.size() ? snd[j].first : nParts);
            int nr = 0, ns = 0;
            while (i < gh.size() && gh[i].first == r) { nr++; i++; }
            while (j < snd.size() && snd[j].first == r) { S->sendCells.push_back(local[snd[j].second]); ns++; j++; }
            S->nbrRank.push_back(r); S->recvCount.push_back(nr); S->sendCount.push_back(ns);
            S->tags.push_back(std::min(rank, r) * nParts + std::max(rank, r));
        }
    }
    // faces: both ends owned, or one end owned (the owned cell becomes the owner of the local face)
    struct LF { int lo, up, gf, flip; };
    std::vector<LF> F;
    std::vector<std::vector<LF>> cut(S->nbrRank.size());
    auto nbrIndex = [&](int r) { return (int)(std::lower_bound(S->nbrRank.begin(), S->nbrRank.end(), r) - S->nbrRank.begin()); };
    for (int f = 0; f < nFaces; f++) {
        const int a = l[f], b = u[f];
        const bool oa = part[a] == rank, ob = part[b] == rank;
        if (oa && ob) F.push_back({local[a], local[b], f, 0});
        else if (oa) { F.push_back({local[a], local[b], f, 0}); cut[nbrIndex(part[b])].push_back({local[a], local[b], f, 0}); }
        else if (ob) { F.push_back({local[b], local[a], f, 1}); cut[nbrIndex(part[a])].push_back({local[b], local[a], f, 1}); }
    }
    std::stable_sort(F.begin(), F.end(), [](const LF &x, const LF &y) { return x.lo < y.lo || (x.lo == y.lo && x.up < y.up); });
    for (const LF &x : F) { S->l.push_back(x.lo); S->u.push_back(x.up); S->globalFace.push_back(x.gf); S->flip.push_back(x.flip); }
    S->cutStart.push_back(0);
    for (auto &cq : cut) {                        // faces were visited in ascending global label already
        for (const LF &x : cq) { S->cutCell.push_back(x.lo); S->cutFace.push_back(x.gf); S->cutFlip.push_back(x.flip); }
        S->cutStart.push_back((int)S->cutCell.size());
    }
    *out = S;
    return FFM_OK;
}

extern "C" int ffm_subdomain_destroy(ffm_subdomain *S) { delete S; return FFM_OK; }
extern "C" int ffm_subdomain_sizes(const ffm_subdomain *S, int *nOwned, int *nGhost, int *nFaces, int *nNbr, int *nSend, int *nCut)
{
    if (!S) return FFM_ERR_ARG;
    if (nOwned) *nOwned = S->nOwned;
    if (nGhost) *nGhost = S->nGhost;
    if (nFaces) *nFaces = (int)S->l.size();
    if (nNbr) *nNbr = (int)S->nbrRank.size();
    if (nSend) *nSend = (int)S->sendCells.size();
    if (nCut) *nCut = (int)S->cutCell.size();
    return FFM_OK;
}
template <class T> static void cp(const std::vector<T> &v, T *dst) { if (dst && !v.empty()) std::memcpy(dst, v.data(), sizeof(T) * v.size()); }
extern "C" int ffm_subdomain_cells(const ffm_subdomain *S, int *globalCell) { if (!S) return FFM_ERR_ARG; cp(S->globalCell, globalCell); return FFM_OK; }
extern "C" int ffm_subdomain_faces(const ffm_subdomain *S, int *lowerAddr, int *upperAddr, int *globalFace, int *flip)
{ if (!S) return FFM_ERR_ARG; cp(S->l, lowerAddr); cp(S->u, upperAddr); cp(S->globalFace, globalFace); cp(S->flip, flip); return FFM_OK; }
extern "C" int ffm_subdomain_exchange(const ffm_subdomain *S, int *nbrRank, int *sendCount, int *sendCells, int *recvCount, int *tags)
{ if (!S) return FFM_ERR_ARG; cp(S->nbrRank, nbrRank); cp(S->sendCount, sendCount); cp(S->sendCells, sendCells); cp(S->recvCount, recvCount); cp(S->tags, tags); return FFM_OK; }
extern "C" int ffm_subdomain_cut_faces(const ffm_subdomain *S, int *cutStart, int *localCell, int *globalFace, int *flip)
{ if (!S) return FFM_ERR_ARG; cp(S->cutStart, cutStart); cp(S->cutCell, localCell); cp(S->cutFace, globalFace); cp(S->cutFlip, flip); return FFM_OK; }
