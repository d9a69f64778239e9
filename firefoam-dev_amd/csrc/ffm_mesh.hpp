// ffm_mesh.hpp -- the device mesh (fvMesh geometry + boundary tables) shared by the finite-volume translation units
// (ffm_fv.hip: one kernel per fvc:: / fvm:: operator; ffm_fused.hip: the fused assembly passes of the compiled time step).
#pragma once
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>

struct ffm_mesh {
    ffm_ldu *A = nullptr;
    ffm_ctx *ctx = nullptr;
    int N = 0, F = 0, nNat = 0, B = 0, nPatches = 0;
    std::vector<int> patchOff;       // [nPatches+1]
    // device geometry
    double *V = nullptr, *C[3] = {nullptr, nullptr, nullptr};
    double *Sf[3] = {nullptr, nullptr, nullptr}, *magSf = nullptr, *delta = nullptr, *w = nullptr;  // [nNat]
    double *Cf[3] = {nullptr, nullptr, nullptr};     // [nNat] face centres (optional: ffm_mesh_set_face_centres, needed by LUST)
    double *corr[3] = {nullptr, nullptr, nullptr};   // [nNat] nonOrthCorrectionVectors (optional: ffm_mesh_set_nonorth_correction)
    double *invT = nullptr;          // [6][N] inverse of surfaceSum(Sf (x) Sf / magSf), symmetric
    int *bCells = nullptr;           // [B] face cell of each boundary face
    double *bSf[3] = {nullptr, nullptr, nullptr}, *bMagSf = nullptr, *bDelta = nullptr;           // [B]
    int *cellB = nullptr;            // [N] index into the boundary-cell list, or -1
    int nBC = 0;
    int *bcStart = nullptr, *bcItem = nullptr;   // boundary-cell CSR: items in (patch, face) order
};

struct MeshView {
    LduView v;
    const double *V, *Sfx, *Sfy, *Sfz, *magSf, *delta, *w;
    const int *cellB, *bcStart, *bcItem;
    const double *bSfx, *bSfy, *bSfz;
};

static inline MeshView mview(const ffm_mesh *m)
{
    MeshView q; q.v = ffm_view(m->A); q.V = m->V; q.Sfx = m->Sf[0]; q.Sfy = m->Sf[1]; q.Sfz = m->Sf[2];
    q.magSf = m->magSf; q.delta = m->delta; q.w = m->w; q.cellB = m->cellB; q.bcStart = m->bcStart; q.bcItem = m->bcItem;
    q.bSfx = m->bSf[0]; q.bSfy = m->bSf[1]; q.bSfz = m->bSf[2];
    return q;
}

static inline int sgrid(long n) { long g = (n + 255) / 256; return (int)std::max(1L, std::min(g, (long)RED_BLOCKS)); }
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
#define LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3(sgrid(n)), dim3(256), 0, m->ctx->stream, __VA_ARGS__)
// Cell-row kernels walk the matrix' XCD-aware row schedule (chunks of 256 rows; entry i is served by a workgroup with
// blockIdx % 8 == i % 8, see ffm_internal.hpp): the rows a row gathers from are then fetched into the same XCD's L2.
#define CELL_SCHED(ci, q)                                                                              \
    for (long it_ = blockIdx.x, ci = 0; it_ < (q).v.nSched; it_ += gridDim.x)                           \
        if ((q).v.sched[it_] >= 0 && (ci = (long)(q).v.sched[it_] * 256 + threadIdx.x) < (q).v.N)
static inline int cgrid(const ffm_mesh *m)
{
    int g = std::min(m->A->nSched, m->A->nOwned >= (32 << 20) ? 1024 : RED_BLOCKS);
    g = (g + 7) & ~7;
    return std::max(g, 8);
}
#define LAUNCH_CELLS(kern, ...) hipLaunchKernelGGL(kern, dim3(cgrid(m)), dim3(256), 0, m->ctx->stream, __VA_ARGS__)

// ------------------------------------------------------------------ face kernels ---
// One thread per owner cell walks its upper slots; native face index e = base + s*64 + lane.
#define FOR_OWN_FACES(q, c, e, nb)                                                     \
    const int sl_ = (c) >> 6, lane_ = (c)&63;                                          \
    const int ub_ = up_base((q).v, sl_), uw_ = up_width((q).v, sl_);                   \
    for (int s_ = 0, e = ub_ + lane_, nb; s_ < uw_; s_++, e += 64)                     \
        if ((nb = (q).v.upNbr[e]) >= 0)

