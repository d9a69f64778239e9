"""The RCCL transport on real hardware, as far as a one-GPU box allows: a ONE-rank RCCL communicator (FFM_FORCE_COMM=1) under
the ghost-cell exchange and the reductions.  The rank is its own neighbour on both ends of a 1-D chain, so ncclGroupStart /
ncclSend / ncclRecv / ncclGroupEnd move real data (rank 0 -> rank 0) and ncclAllReduce runs inside every dot product of the
solver.  What this checks is the call sequence, counts, types, stream and buffer offsets of csrc/ffm_comm.hip -- the multi-GPU
runs themselves are the driver's."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_ghost_exchange_and_allreduce_through_a_one_rank_rccl_communicator(O, ffm):
    os.environ["FFM_FORCE_COMM"] = "1"
    try:
        ctx = ffm.Context(0)
        ctx.comm_init_rccl(0, 1, ffm.Context.comm_unique_id())
    finally:
        del os.environ["FFM_FORCE_COMM"]
    L = ffm.lib()
    N = 300
    # chain 0-1-...-(N-1) plus two ghost cells: N next to cell 0, N+1 next to cell N-1; faces in upper-triangular order
    faces = sorted([(i, i + 1) for i in range(N - 1)] + [(0, N), (N - 1, N + 1)])
    l = np.array([f[0] for f in faces], np.int32); u = np.array([f[1] for f in faces], np.int32)
    h = C.c_void_p()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    assert L.ffm_ldu_create_ext(ctx.h, N, 2, len(l), ip(l), ip(u), C.byref(h)) == 0, L.ffm_last_error()
    A = ffm.lduMatrix.__new__(ffm.lduMatrix)
    A.ctx, A.h, A.nCells, A.nFaces = ctx, h, N + 2, len(l)
    assert A.native_order
    # both neighbours are rank 0: message 1 = cell 0, message 2 = cell N-1; receives are matched in order, so ghost N gets
    # cell 0 and ghost N+1 gets cell N-1
    nbr = np.array([0, 0], np.int32); cnt = np.array([1, 1], np.int32); send = np.array([0, N - 1], np.int32)
    assert L.ffm_ldu_set_ghost_exchange(h, 2, ip(nbr), ip(cnt), ip(send), ip(cnt)) == 0, L.ffm_last_error()
    rng = lambda seed, n: O.hash_u(seed, np.arange(n))
    upper = -(0.5 + rng(1, len(l)))
    diag = np.zeros(N + 2)
    np.add.at(diag, l, -upper); np.add.at(diag, u, -upper); diag[:N] += 0.3
    A.set_coeffs(diag, upper)
    # the operator the rank sees: ghost columns are copies of their source cells
    src = np.arange(N + 2); src[N] = 0; src[N + 1] = N - 1
    M = np.zeros((N, N))
    M[np.arange(N), np.arange(N)] = diag[:N]
    for f, (a, b) in enumerate(zip(l, u)):
        M[a, src[b]] += upper[f]
        if b < N:
            M[b, a] += upper[f]
    x = rng(2, N + 2); x[N:] = 1e300                        # ghosts hold rubbish until the exchange
    y = A.Amul(ctx.to_device(x)).cpu().numpy()
    assert np.abs(y[:N] - M @ x[:N]).max() <= 1e-13 * np.abs(M @ x[:N]).max()
    b = np.zeros(N + 2); b[:N] = rng(3, N) - 0.5
    psi = ctx.zeros(N + 2)
    perf = A.solve(psi, ctx.to_device(b), solver="PCG", preconditioner="DIC", tolerance=1e-12, relTol=0.0)
    assert perf["converged"]
    ref = np.linalg.solve(M, b[:N])
    assert np.linalg.norm(psi.cpu().numpy()[:N] - ref) <= 1e-9 * np.linalg.norm(ref)
    # reductions through ncclAllReduce (sum over one rank = the local value)
    out = C.c_double()
    xd = ctx.to_device(x[:N].copy())
    assert L.ffm_reduce_sum(ctx.h, C.c_void_p(xd.data_ptr()), N, C.byref(out)) == 0
    assert abs(out.value - x[:N].sum()) <= 1e-12 * abs(x[:N].sum())
    assert L.ffm_reduce_max(ctx.h, C.c_void_p(xd.data_ptr()), N, C.byref(out)) == 0 and out.value == x[:N].max()
    A.close(); ctx.close()


def test_plume_step_with_every_reduction_through_rccl(O, ffm):
    """a whole time step of the plume case on a context whose reductions all go through ncclAllReduce (one rank): every solver's
    dot products, normalisation factors and residual norms, multi-slot all-reduces included -- bitwise the same fields and the
    same iteration counts as without a communicator"""
    plain = ffm.Context(0)
    os.environ["FFM_FORCE_COMM"] = "1"
    try:
        viaRccl = ffm.Context(0)
        viaRccl.comm_init_rccl(0, 1, ffm.Context.comm_unique_id())
    finally:
        del os.environ["FFM_FORCE_COMM"]
    a, b = ffm.Plume(plain, (10, 12, 9)), ffm.Plume(viaRccl, (10, 12, 9))
    for _ in range(2):
        a.step(); b.step()
        assert [(n, p["nIterations"]) for n, p in a.solves()] == [(n, p["nIterations"]) for n, p in b.solves()]
    for name in ("rho", "p", "p_rgh", "T", "Ux", "Uy", "Uz", "C3H8", "O2"):
        assert np.array_equal(a.field(name), b.field(name)), name
    a.close(); b.close(); plain.close(); viaRccl.close()


def test_overlapped_ghost_exchange_of_the_tiled_amul(O, ffm):
    """the tiled Amul computes the rows without their ghost faces while the RCCL send / receive group of the
    ghost refresh runs on a second stream; the ghost-face tail waits for it.  One rank that is its own neighbour (a box whose +x
    layer of ghost cells mirrors its x = 0 layer, i.e. a periodic coupling): real ncclSend / ncclRecv on the communication stream,
    tile numbering with a ghost-face tail.  Amul against the operator written out in numpy, PCG against a sparse direct solve --
    ghosts hold rubbish before every Amul, so a tail that did not wait for the receive would show."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    os.environ["FFM_FORCE_COMM"] = "1"
    try:
        ctx = ffm.Context(0)
        ctx.comm_init_rccl(0, 1, ffm.Context.comm_unique_id())
    finally:
        del os.environ["FFM_FORCE_COMM"]
    nx, ny, nz = 24, 40, 36
    N, l, u = O.hex_ldu(nx, ny, nz)
    c = np.arange(N); i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
    last = c[i == nx - 1]; first = c[i == 0]                 # same (j, k) order
    G = len(last)
    fl = np.concatenate([l, last]); fu = np.concatenate([u, N + np.arange(G)])
    order = np.lexsort((fu, fl)); fl, fu = fl[order].astype(np.int32), fu[order].astype(np.int32)
    hint = (j // 16 + 1000 * (k // 16)).astype(np.int32)
    cOrd, fOrd = ffm.renumber_levels(N, l, u, groupHint=hint)     # owned cells in tile order; ghosts keep their place behind
    oldToNew = np.empty(N + G, np.int64); oldToNew[cOrd] = np.arange(N); oldToNew[N:] = N + np.arange(G)
    l2, u2 = oldToNew[fl], oldToNew[fu]
    o2 = np.lexsort((u2, l2)); l2, u2 = l2[o2].astype(np.int32), u2[o2].astype(np.int32)
    A = ffm.lduMatrix(ctx, N, l2, u2, groupHint=hint[cOrd], nGhost=G)
    assert A.sweep_mode == 2 and A.native_order
    A.set_ghost_exchange([0], [G], oldToNew[first], [G])
    nF = len(l2)
    upper = -(0.5 + O.hash_u(1, np.arange(nF)))
    diag = np.zeros(N + G); np.add.at(diag, l2, -upper); own = u2 < N; np.add.at(diag, u2[own], -upper[own]); diag[:N] += 0.2
    A.set_coeffs(diag, upper)
    src = np.arange(N + G); src[N:] = oldToNew[first]        # ghost column = its source cell
    M = sp.coo_matrix((np.concatenate([diag[:N], upper, upper[own]]),
                       (np.concatenate([np.arange(N), l2, u2[own]]), np.concatenate([np.arange(N), src[u2], l2[own]]))), shape=(N, N)).tocsr()
    x = O.hash_u(2, np.arange(N + G)); x[N:] = 1e300
    for _ in range(3):                                        # back-to-back launches: send buffer and events are reused
        y = A.Amul(ctx.to_device(x)).cpu().numpy()
        assert np.abs(y[:N] - M @ x[:N]).max() <= 1e-13 * np.abs(M @ x[:N]).max()
    b = np.zeros(N + G); b[:N] = O.hash_u(3, np.arange(N)) - 0.5
    psi = ctx.zeros(N + G)
    perf = A.solve(psi, ctx.to_device(b), solver="PCG", preconditioner="DIC", tolerance=1e-12, relTol=0.0)
    assert perf["converged"]
    ref = spl.spsolve(M.tocsc(), b[:N])
    assert np.linalg.norm(psi.cpu().numpy()[:N] - ref) <= 1e-9 * np.linalg.norm(ref)
    A.close(); ctx.close()
