// ffm_comm.hip -- the Pstream replacement: processor-patch halo exchange and
// scalar all-reduces, one process per GPU, RCCL over xGMI.
//
// Replaces (OpenFOAM-dev @940e28f, not vendored in the reference):
//   src/Pstream/mpi/UPstream.C, src/OpenFOAM/db/IOstreams/Pstreams/*  (reduce, gSum*)
//   src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixUpdateMatrixInterfaces.C
//   src/finiteVolume/fields/fvPatchFields/constraint/processor/processorFvPatchField.C
// Reference-side use: `mpirun -np 2 fireFoam -parallel`
// (cases/wallFireSpread2D/runParallel.sh:18); collectives C1-C7 of SURVEY 2.4.
//
// Halo (C1): one tiny pack kernel gathers psi[faceCells] of every processor
// patch into one contiguous send buffer; the patches are exchanged in a single
// ncclGroup of ncclSend/ncclRecv pairs on the context stream (xGMI gives every
// peer its own link, so the <=6 neighbours of a block decomposition proceed in
// parallel); an apply kernel then adds coeff*neighbourValue to the boundary
// cells, one thread per boundary cell, terms in (patch, face) order.
// Dots (C2, C3): the device scalars are all-reduced in place with
// ncclAllReduce on the same stream; no host round trip.
//
// A second transport (host callbacks) lets several ranks share one GPU, which
// is how the decomposed path is tested on a single-GPU box; it stages the halo
// through pinned host memory and calls back into the launcher (gloo/MPI).
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <rccl/rccl.h>
#include <algorithm>

#define FFM_NCCL(call)                                                           \
    do {                                                                         \
        ncclResult_t r_ = (call);                                                \
        if (r_ != ncclSuccess) {                                                 \
            ffm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call,           \
                          ncclGetErrorString(r_));                               \
            return FFM_ERR_COMM;                                                 \
        }                                                                        \
    } while (0)

// inside an ncclGroup: close the group before reporting the error, so that the next call does not find it open
#define FFM_NCCL_IN_GROUP(call)                                                  \
    do {                                                                         \
        ncclResult_t r_ = (call);                                                \
        if (r_ != ncclSuccess) {                                                 \
            ffm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call,           \
                          ncclGetErrorString(r_));                               \
            ncclGroupEnd();                                                      \
            return FFM_ERR_COMM;                                                 \
        }                                                                        \
    } while (0)

// posting order of the point-to-point messages of one exchange: RCCL matches several messages between the same pair of
// ranks in issue order, so both ranks must post the messages they share in the same relative order -- ascending pair tag
// (ffm_ldu_set_exchange_tags; default: the entry index, i.e. both sides list their common patches in the same order)
static std::vector<int> posting_order(int n, const std::vector<int> &tags)
{
    std::vector<int> o(n);
    for (int i = 0; i < n; i++) o[i] = i;
    if ((int)tags.size() == n) std::stable_sort(o.begin(), o.end(), [&](int a, int b) { return tags[a] < tags[b]; });
    return o;
}

extern "C" int ffm_ldu_set_exchange_tags(ffm_ldu *A, int kind, int n, const int *tags)
{
    if (!A || (kind != 0 && kind != 1) || n < 0 || (n && !tags)) return FFM_ERR_ARG;
    const int have = kind == 0 ? (int)A->ifaces.size() : (int)A->ghNbrRank.size();
    if (n != have) { ffm_set_error("ffm_ldu_set_exchange_tags: %d tags for %d entries", n, have); return FFM_ERR_ARG; }
    (kind == 0 ? A->ifaceTags : A->ghTags).assign(tags, tags + n);
    return FFM_OK;
}

extern "C" int ffm_comm_unique_id(void *uniqueId128)
{
    if (!uniqueId128) return FFM_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    FFM_NCCL(ncclGetUniqueId(&id));
    memcpy(uniqueId128, &id, sizeof(id));
    return FFM_OK;
}

extern "C" int ffm_comm_init(ffm_ctx *c, int rank, int nRanks, const void *uniqueId128)
{
    if (!c || rank < 0 || nRanks < 1 || rank >= nRanks) return FFM_ERR_ARG;
    c->rank = rank; c->nRanks = nRanks;
    // a single rank needs no communicator; FFM_FORCE_COMM=1 creates one anyway so that the RCCL calls of the halo exchange
    // (send / receive to itself) and of the reductions run on a one-GPU box (tests/test_rccl_single_rank_gpu.py)
    if (nRanks == 1 && !(uniqueId128 && getenv("FFM_FORCE_COMM"))) return FFM_OK;
    if (!uniqueId128) return FFM_ERR_ARG;
    FFM_HIP(hipSetDevice(c->device));
    ncclUniqueId id; memcpy(&id, uniqueId128, sizeof(id));
    ncclComm_t comm;
    FFM_NCCL(ncclCommInitRank(&comm, nRanks, id, rank));
    c->comm = (ncclComm *)comm;
    // second communicator (same ranks, same order) for the overlapped ghost exchange on commStream: see ffm_ctx::haloComm
    if (!getenv("FFM_NO_OVERLAP")) {
        ncclComm_t halo;
        FFM_NCCL(ncclCommSplit(comm, 0, rank, &halo, nullptr));
        c->haloComm = (ncclComm *)halo;
    }
    return FFM_OK;
}

extern "C" int ffm_comm_init_host(ffm_ctx *c, int rank, int nRanks, void *user, ffm_host_allreduce_fn ar,
                                  ffm_host_exchange_fn ex)
{
    if (!c || rank < 0 || nRanks < 1 || rank >= nRanks || (nRanks > 1 && (!ar || !ex))) return FFM_ERR_ARG;
    c->rank = rank; c->nRanks = nRanks; c->hostUser = user; c->hostAllreduce = ar; c->hostExchange = ex;
    return FFM_OK;
}

void ffm_comm_finalize_i(ffm_ctx *c)
{
    if (c->commStream) { hipStreamSynchronize(c->commStream); hipStreamDestroy(c->commStream); c->commStream = nullptr; }
    if (c->evPack) { hipEventDestroy(c->evPack); c->evPack = nullptr; }
    if (c->evRecv) { hipEventDestroy(c->evRecv); c->evRecv = nullptr; }
    if (c->haloComm) { ncclCommDestroy((ncclComm_t)c->haloComm); c->haloComm = nullptr; }
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
}

extern "C" int ffm_comm_rank(const ffm_ctx *c) { return c ? c->rank : FFM_ERR_ARG; }
extern "C" int ffm_comm_size(const ffm_ctx *c) { return c ? c->nRanks : FFM_ERR_ARG; }

int ffm_allreduce_slots(ffm_ctx *c, int firstSlot, int n)
{
    if (c->nRanks <= 1 && !c->comm) return FFM_OK;
    if (c->comm) {
        FFM_NCCL(ncclAllReduce(c->scal_d + firstSlot, c->scal_d + firstSlot, n, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
        return FFM_OK;
    }
    if (!c->hostAllreduce) { ffm_set_error("nRanks>1 but no communicator attached"); return FFM_ERR_COMM; }
    FFM_HIP(hipMemcpyAsync(c->scal_h + firstSlot, c->scal_d + firstSlot, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    c->hostAllreduce(c->hostUser, c->scal_h + firstSlot, n, 0);
    FFM_HIP(hipMemcpyAsync(c->scal_d + firstSlot, c->scal_h + firstSlot, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}

int ffm_allreduce_minmax(ffm_ctx *c, int slot, int isMax)
{
    if (c->nRanks <= 1 && !c->comm) return FFM_OK;
    if (c->comm) {
        FFM_NCCL(ncclAllReduce(c->scal_d + slot, c->scal_d + slot, 1, ncclDouble, isMax ? ncclMax : ncclMin, (ncclComm_t)c->comm, c->stream));
        return FFM_OK;
    }
    if (!c->hostAllreduce) { ffm_set_error("nRanks>1 but no communicator attached"); return FFM_ERR_COMM; }
    FFM_HIP(hipMemcpyAsync(c->scal_h + slot, c->scal_d + slot, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    c->hostAllreduce(c->hostUser, c->scal_h + slot, 1, isMax ? 2 : 1);
    FFM_HIP(hipMemcpyAsync(c->scal_d + slot, c->scal_h + slot, sizeof(double), hipMemcpyHostToDevice, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    return FFM_OK;
}

// --------------------------------------------------------------- interfaces ---
extern "C" int ffm_ldu_set_interfaces(ffm_ldu *A, int nPatches, const int *patchSizes, const int *const *faceCells,
                                      const double *const *bouCoeffs, const double *const *intCoeffs, const int *neighbRank)
{
    if (!A || nPatches < 0 || (nPatches && (!patchSizes || !faceCells || !bouCoeffs || !neighbRank))) return FFM_ERR_ARG;
    ffm_ctx *c = A->ctx;
    hipStreamSynchronize(c->stream);
    hipFree(A->ifFaceCells); hipFree(A->ifBou); hipFree(A->ifInt); hipFree(A->haloSend); hipFree(A->haloRecv);
    hipFree(A->ifCell); hipFree(A->ifCellStart); hipFree(A->ifItem);
    if (A->haloSend_h) { hipHostFree(A->haloSend_h); A->haloSend_h = nullptr; }
    if (A->haloRecv_h) { hipHostFree(A->haloRecv_h); A->haloRecv_h = nullptr; }
    A->ifFaceCells = nullptr; A->ifBou = A->ifInt = A->haloSend = A->haloRecv = nullptr;
    A->ifCell = A->ifCellStart = A->ifItem = nullptr;
    A->ifaces.clear(); A->ifaceTags.clear(); A->haloTotal = 0; A->nIfCells = 0;
    for (auto &kv : A->graphs) hipGraphExecDestroy(kv.second);   // halo buffers are baked into no graph, but be safe
    A->graphs.clear();
    if (!nPatches) return FFM_OK;
    // old -> new cell map
    std::vector<int> oldToNew(A->nCells);
    for (int i = 0; i < A->nCells; i++) oldToNew[A->h_newToOldCell[i]] = i;
    int total = 0;
    for (int p = 0; p < nPatches; p++) {
        if (patchSizes[p] < 0 || neighbRank[p] < 0 || neighbRank[p] >= std::max(c->nRanks, 1)) { ffm_set_error("interface %d: bad size or rank", p); return FFM_ERR_ARG; }
        ffm_iface f; f.size = patchSizes[p]; f.nbrRank = neighbRank[p]; f.offset = total; total += f.size;
        A->ifaces.push_back(f);
    }
    A->haloTotal = total;
    std::vector<int> fc(total); std::vector<double> bou(total), in(total);
    for (int p = 0; p < nPatches; p++) for (int i = 0; i < patchSizes[p]; i++) {
        const int oc = faceCells[p][i];
        if (oc < 0 || oc >= A->nCells) { ffm_set_error("interface %d: faceCell out of range", p); A->ifaces.clear(); A->haloTotal = 0; return FFM_ERR_ARG; }
        const int k = A->ifaces[p].offset + i;
        fc[k] = oldToNew[oc]; bou[k] = bouCoeffs[p][i];
        in[k] = (intCoeffs && intCoeffs[p]) ? intCoeffs[p][i] : bouCoeffs[p][i];
    }
    // group packed items by cell, keeping (patch, face) order inside a cell
    std::vector<int> order(total);
    for (int k = 0; k < total; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fc[a] < fc[b]; });
    std::vector<int> cells, start;
    for (int j = 0; j < total; j++) {
        if (j == 0 || fc[order[j]] != fc[order[j - 1]]) { cells.push_back(fc[order[j]]); start.push_back(j); }
    }
    start.push_back(total);
    A->nIfCells = (int)cells.size();
    const size_t tb = sizeof(double) * std::max(total, 1), ti = sizeof(int) * std::max(total, 1);
    FFM_HIP(hipMalloc((void **)&A->ifFaceCells, ti)); FFM_HIP(hipMalloc((void **)&A->ifBou, tb)); FFM_HIP(hipMalloc((void **)&A->ifInt, tb));
    FFM_HIP(hipMalloc((void **)&A->haloSend, tb)); FFM_HIP(hipMalloc((void **)&A->haloRecv, tb));
    FFM_HIP(hipMalloc((void **)&A->ifCell, sizeof(int) * std::max(A->nIfCells, 1)));
    FFM_HIP(hipMalloc((void **)&A->ifCellStart, sizeof(int) * (A->nIfCells + 1)));
    FFM_HIP(hipMalloc((void **)&A->ifItem, ti));
    FFM_HIP(hipHostMalloc((void **)&A->haloSend_h, tb, hipHostMallocDefault));
    FFM_HIP(hipHostMalloc((void **)&A->haloRecv_h, tb, hipHostMallocDefault));
    FFM_TRY(ffm_h2d(A->ctx, A->ifFaceCells, fc.data(), sizeof(int) * total));
    FFM_TRY(ffm_h2d(A->ctx, A->ifBou, bou.data(), sizeof(double) * total));
    FFM_TRY(ffm_h2d(A->ctx, A->ifInt, in.data(), sizeof(double) * total));
    FFM_TRY(ffm_h2d(A->ctx, A->ifCell, cells.data(), sizeof(int) * cells.size()));
    FFM_TRY(ffm_h2d(A->ctx, A->ifCellStart, start.data(), sizeof(int) * start.size()));
    FFM_TRY(ffm_h2d(A->ctx, A->ifItem, order.data(), sizeof(int) * total));
    return FFM_OK;
}

__global__ void k_halo_pack(int n, const int *__restrict__ fc, const double *__restrict__ x, double *__restrict__ send)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) send[i] = x[fc[i]];
}

// y[cell] (+/-)= coeff[item]*val[item] for the items of each boundary cell, in (patch, face) order
__global__ void k_halo_apply(int nCells, const int *__restrict__ cell, const int *__restrict__ start,
                             const int *__restrict__ item, const double *__restrict__ coeff,
                             const double *__restrict__ val, double *__restrict__ y, int subtract)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nCells; j += gridDim.x * blockDim.x) {
        const int c = cell[j];
        double acc = y[c];
        for (int q = start[j]; q < start[j + 1]; q++) {
            const int k = item[q];
            const double t = val ? coeff[k] * val[k] : coeff[k];
            acc = subtract ? acc - t : acc + t;
        }
        y[c] = acc;
    }
}

int ffm_halo_exchange(ffm_ldu *A, const double *x)
{
    ffm_ctx *c = A->ctx;
    if (A->ifaces.empty()) return FFM_OK;
    hipLaunchKernelGGL(k_halo_pack, dim3(std::max(1, std::min(ffm_grid(A->haloTotal, 256), 1024))), dim3(256), 0, c->stream,
                       A->haloTotal, A->ifFaceCells, x, A->haloSend);
    FFM_HIP(hipGetLastError());
    if (c->comm) {
        const std::vector<int> order = posting_order((int)A->ifaces.size(), A->ifaceTags);
        FFM_NCCL(ncclGroupStart());
        for (int i : order) {
            const ffm_iface &p = A->ifaces[i];
            if (!p.size) continue;
            FFM_NCCL_IN_GROUP(ncclSend(A->haloSend + p.offset, p.size, ncclDouble, p.nbrRank, (ncclComm_t)c->comm, c->stream));
            FFM_NCCL_IN_GROUP(ncclRecv(A->haloRecv + p.offset, p.size, ncclDouble, p.nbrRank, (ncclComm_t)c->comm, c->stream));
        }
        FFM_NCCL(ncclGroupEnd());
        return FFM_OK;
    }
    if (!c->hostExchange) { ffm_set_error("processor interfaces set but no communicator attached"); return FFM_ERR_COMM; }
    const size_t tb = sizeof(double) * A->haloTotal;
    FFM_HIP(hipMemcpyAsync(A->haloSend_h, A->haloSend, tb, hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    std::vector<int> sizes, ranks, offs;
    for (const ffm_iface &p : A->ifaces) { sizes.push_back(p.size); ranks.push_back(p.nbrRank); offs.push_back(p.offset); }
    c->hostExchange(c->hostUser, (int)A->ifaces.size(), sizes.data(), ranks.data(), offs.data(), A->haloSend_h, A->haloRecv_h);
    FFM_HIP(hipMemcpyAsync(A->haloRecv, A->haloRecv_h, tb, hipMemcpyHostToDevice, c->stream));
    return FFM_OK;
}

int ffm_halo_apply(ffm_ldu *A, double *y, const double *coeffs, const double *vals, double sign)
{
    if (!A->nIfCells) return FFM_OK;
    hipLaunchKernelGGL(k_halo_apply, dim3(std::max(1, std::min(ffm_grid(A->nIfCells, 256), 1024))), dim3(256), 0, A->ctx->stream,
                       A->nIfCells, A->ifCell, A->ifCellStart, A->ifItem, coeffs, vals, y, sign < 0 ? 1 : 0);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

int ffm_halo_update(ffm_ldu *A, const double *x, double *y, const double *coeffs, double sign)
{
    FFM_TRY(ffm_halo_exchange(A, x));
    return ffm_halo_apply(A, y, coeffs, A->haloRecv, sign);
}

// ------------------------------------------------------------ ghost-cell halo ---
// Native decomposed path: the rank's mesh carries one layer of ghost cells (copies of the neighbour
// ranks' boundary cells) appended after its owned cells, and the cut faces are ordinary faces
// between an owned and a ghost cell.  Every FV kernel and Amul then run unchanged; what this
// section adds is the refresh of the ghost entries of a cell field: one pack kernel gathers the
// owned cells each neighbour needs, one ncclGroup of Send/Recv moves them, and the receive lands
// directly in the (contiguous) ghost range of the field -- no unpack kernel.
extern "C" int ffm_ldu_set_ghost_exchange(ffm_ldu *A, int nNbr, const int *nbrRank, const int *sendCount,
                                          const int *sendCells, const int *recvCount)
{
    if (!A || nNbr < 0 || (nNbr && (!nbrRank || !sendCount || !sendCells || !recvCount))) return FFM_ERR_ARG;
    hipStreamSynchronize(A->ctx->stream);
    hipFree(A->ghSendCells); hipFree(A->ghSendBuf); A->ghSendCells = nullptr; A->ghSendBuf = nullptr;
    if (A->ghSendBuf_h) { hipHostFree(A->ghSendBuf_h); A->ghSendBuf_h = nullptr; }
    if (A->ghRecvBuf_h) { hipHostFree(A->ghRecvBuf_h); A->ghRecvBuf_h = nullptr; }
    A->ghNbrRank.assign(nbrRank, nbrRank + nNbr); A->ghTags.clear();
    A->ghSendOff.assign(nNbr + 1, 0); A->ghRecvOff.assign(nNbr + 1, 0);
    for (int q = 0; q < nNbr; q++) { A->ghSendOff[q + 1] = A->ghSendOff[q] + sendCount[q]; A->ghRecvOff[q + 1] = A->ghRecvOff[q] + recvCount[q]; }
    if (A->ghRecvOff[nNbr] != A->nCells - A->nOwned) { ffm_set_error("ghost exchange: receive counts (%d) != ghost cells (%d)", A->ghRecvOff[nNbr], A->nCells - A->nOwned); A->ghNbrRank.clear(); return FFM_ERR_ARG; }
    const int nSend = A->ghSendOff[nNbr];
    A->h_ghSendCaller.assign(sendCells, sendCells + nSend);
    for (int i = 0; i < nSend; i++) if (sendCells[i] < 0 || sendCells[i] >= A->nOwned) { ffm_set_error("ghost exchange: send cell out of range"); A->ghNbrRank.clear(); return FFM_ERR_ARG; }
    FFM_HIP(hipMalloc((void **)&A->ghSendCells, sizeof(int) * std::max(nSend, 1)));
    FFM_HIP(hipMalloc((void **)&A->ghSendBuf, sizeof(double) * std::max(nSend, 1)));
    {
        // the caller's cell labels -> the library's numbering (ghost cells keep their place behind the owned ones in both)
        std::vector<int> oldToNew(A->nCells), mapped(std::max(nSend, 1));
        for (int i = 0; i < A->nCells; i++) oldToNew[A->h_newToOldCell[i]] = i;
        for (int i = 0; i < nSend; i++) mapped[i] = oldToNew[sendCells[i]];
        FFM_TRY(ffm_h2d(A->ctx, A->ghSendCells, mapped.data(), sizeof(int) * nSend));
    }
    FFM_HIP(hipHostMalloc((void **)&A->ghSendBuf_h, sizeof(double) * std::max(nSend, 1), hipHostMallocDefault));
    FFM_HIP(hipHostMalloc((void **)&A->ghRecvBuf_h, sizeof(double) * std::max(A->ghRecvOff[nNbr], 1), hipHostMallocDefault));
    return FFM_OK;
}

// the send / receive group of one ghost refresh on stream s
static int ghost_group(ffm_ldu *A, double *ghost, hipStream_t s, ncclComm *useComm = nullptr)
{
    ffm_ctx *c = A->ctx;
    ncclComm_t comm = (ncclComm_t)(useComm ? useComm : c->comm);
    const int nNbr = (int)A->ghNbrRank.size();
    const std::vector<int> order = posting_order(nNbr, A->ghTags);
    FFM_NCCL(ncclGroupStart());
    for (int q : order) {
        const int ns = A->ghSendOff[q + 1] - A->ghSendOff[q], nr = A->ghRecvOff[q + 1] - A->ghRecvOff[q];
        if (ns) FFM_NCCL_IN_GROUP(ncclSend(A->ghSendBuf + A->ghSendOff[q], ns, ncclDouble, A->ghNbrRank[q], comm, s));
        if (nr) FFM_NCCL_IN_GROUP(ncclRecv(ghost + A->ghRecvOff[q], nr, ncclDouble, A->ghNbrRank[q], comm, s));
    }
    FFM_NCCL(ncclGroupEnd());
    return FFM_OK;
}

// Overlapped form for an operator whose interior work does not read the ghost entries (the tiled Amul: its ghost faces are
// added by a tail kernel): the pack kernel runs on the main stream, the RCCL group on a second stream behind an event, and
// ffm_ghost_exchange_end() makes the main stream wait for the receives just before the tail.  Same data, same arithmetic as the
// serial order -- only the interior rows no longer wait for the wire.  Host transport and FFM_NO_OVERLAP: the serial exchange.
int ffm_ghost_exchange_begin(ffm_ldu *A, double *x)
{
    ffm_ctx *c = A->ctx;
    static const bool noOverlap = getenv("FFM_NO_OVERLAP") != nullptr;
    const int nNbr = (int)A->ghNbrRank.size();
    if (!nNbr) return FFM_OK;
    if (!c->comm || !c->haloComm || noOverlap) return ffm_ghost_exchange(A, x);
    if (!c->commStream) {
        FFM_HIP(hipStreamCreateWithFlags(&c->commStream, hipStreamNonBlocking));
        FFM_HIP(hipEventCreateWithFlags(&c->evPack, hipEventDisableTiming));
        FFM_HIP(hipEventCreateWithFlags(&c->evRecv, hipEventDisableTiming));
    }
    const int nSend = A->ghSendOff[nNbr];
    if (nSend) hipLaunchKernelGGL(k_halo_pack, dim3(std::max(1, std::min(ffm_grid(nSend, 256), 1024))), dim3(256), 0, c->stream,
                                  nSend, A->ghSendCells, x, A->ghSendBuf);
    FFM_HIP(hipGetLastError());
    FFM_HIP(hipEventRecord(c->evPack, c->stream));
    FFM_HIP(hipStreamWaitEvent(c->commStream, c->evPack, 0));
    FFM_TRY(ghost_group(A, x + A->nOwned, c->commStream, c->haloComm));          // its own communicator: never queued behind / ahead of the main stream's all-reduces
    FFM_HIP(hipEventRecord(c->evRecv, c->commStream));
    A->ghPending = true;
    return FFM_OK;
}

int ffm_ghost_exchange_end(ffm_ldu *A)
{
    if (!A->ghPending) return FFM_OK;
    FFM_HIP(hipStreamWaitEvent(A->ctx->stream, A->ctx->evRecv, 0));
    A->ghPending = false;
    return FFM_OK;
}

int ffm_ghost_exchange(ffm_ldu *A, double *x)
{
    ffm_ctx *c = A->ctx;
    const int nNbr = (int)A->ghNbrRank.size();
    if (!nNbr) return FFM_OK;
    const int nSend = A->ghSendOff[nNbr], nRecv = A->ghRecvOff[nNbr];
    if (nSend) hipLaunchKernelGGL(k_halo_pack, dim3(std::max(1, std::min(ffm_grid(nSend, 256), 1024))), dim3(256), 0, c->stream,
                                  nSend, A->ghSendCells, x, A->ghSendBuf);
    FFM_HIP(hipGetLastError());
    double *ghost = x + A->nOwned;
    if (c->comm) return ghost_group(A, ghost, c->stream);
    if (!c->hostExchange2) { ffm_set_error("ghost cells set but no communicator attached"); return FFM_ERR_COMM; }
    FFM_HIP(hipMemcpyAsync(A->ghSendBuf_h, A->ghSendBuf, sizeof(double) * nSend, hipMemcpyDeviceToHost, c->stream));
    FFM_HIP(hipStreamSynchronize(c->stream));
    c->hostExchange2(c->hostUser, nNbr, A->ghNbrRank.data(), A->ghSendOff.data(), A->ghRecvOff.data(), A->ghSendBuf_h, A->ghRecvBuf_h);
    FFM_HIP(hipMemcpyAsync(ghost, A->ghRecvBuf_h, sizeof(double) * nRecv, hipMemcpyHostToDevice, c->stream));
    return FFM_OK;
}

extern "C" int ffm_halo_refresh_d(ffm_ldu *A, double *field_d)
{
    if (!A || !field_d) return FFM_ERR_ARG;
    return ffm_ghost_exchange(A, field_d);
}

extern "C" int ffm_comm_set_host_exchange2(ffm_ctx *c, ffm_host_exchange2_fn fn) { if (!c) return FFM_ERR_ARG; c->hostExchange2 = fn; return FFM_OK; }
