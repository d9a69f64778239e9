"""ffm_solve_multi_d -- the segregated solves of a vector equation / the species equations with common off-diagonal coefficients
(fvMatrix::solveSegregated; solver/UEqn.H:19-30, solver/YEEqn.H:43-60) advanced in lock step with ONE tiled sweep per preconditioner
application for all of them (csrc/ffm_tile.hip: k_tile_m) -- against one ffm_solve_d per system: every system must get the SAME solve,
bit for bit (solution, residuals, iteration count), whatever the others do: systems that converge at once, after half an iteration, after
several; 2, 3, 4 and 5 systems (the fifth runs on its own); DILU on an asymmetric and DIC on a symmetric matrix; and the solutions solve
their systems (checked with the oracle's Amul)."""
import numpy as np
import pytest

from ffm_import import ffm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ffm.Context(0)
    yield c
    c.close()


def Amul(l, u, d, upper, lower, x):
    """lduMatrix::Amul (OpenFOAM-dev lduMatrixATmul.C): y = diag x; y[u] += lower x[l]; y[l] += upper x[u]"""
    y = d * x
    np.add.at(y, u, lower * x[l]); np.add.at(y, l, upper * x[u])
    return y


def build(ctx, n, asym, seed):
    """a convection-diffusion-like matrix on an n[0] x n[1] x n[2] box in the library's cell order (native coefficient layout)"""
    rng = np.random.default_rng(seed)
    N, l, u = ffm.hexmesh.hex_ldu(*n)
    F = l.size
    cOrd, fOrd = ffm.renumber_levels(N, l, u)
    l2, u2, _ = ffm.hexmesh.apply_renumbering(N, l, u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    assert A.sweep_mode == 2 and A.native_order
    g = rng.uniform(0.5, 1.5, F)
    phi = rng.normal(0.0, 0.6, F) if asym else np.zeros(F)
    upper = -g + 0.5 * phi
    lower = -g - 0.5 * phi
    base = np.zeros(N)
    np.add.at(base, l2, -lower); np.add.at(base, u2, -upper)          # negSumDiag
    return A, N, l2, u2, upper, lower, base, rng


@pytest.mark.parametrize("n,nSys,asym", [((24, 20, 18), 3, True), ((33, 17, 29), 4, True), ((20, 20, 20), 2, True), ((20, 16, 24), 4, False), ((18, 18, 18), 5, True)])
def test_lock_step_solves_equal_separate_solves(ctx, n, nSys, asym):
    A, N, l, u, upper, lower, base, rng = build(ctx, n, asym, seed=sum(n) + nSys)
    fmap = A.face_map()
    nat = lambda f: (lambda out: (out.__setitem__(fmap, f), out)[1])(np.zeros(A.nNative))
    up_d, lo_d = ctx.to_device(nat(upper)), (ctx.to_device(nat(lower)) if asym else None)
    diags, srcs, psi0 = [], [], []
    for i in range(nSys):
        # different diagonal dominance => different iteration counts; system 1 starts from its solution (converged at once)
        d = base * (1.0 + 0.02 * 4.0 ** i * rng.uniform(0.5, 1.5, N)) + (10.0 if i == 2 else 0.02) * rng.uniform(0.0, 1.0, N)
        x = rng.standard_normal(N)
        b = Amul(l, u, d, upper, lower, x) if i == 1 else rng.standard_normal(N)
        diags.append(d); srcs.append(b); psi0.append(x if i == 1 else np.zeros(N))
    kw = dict(solver="PBiCGStab", preconditioner="DILU" if asym else "DIC", tolerance=1e-9, relTol=0.0)
    dev = lambda a: ctx.to_device(a)
    host = lambda t: (ctx.sync(), t.cpu().numpy())[1]
    # one solve per system
    ref, refPerf = [], []
    for i in range(nSys):
        dd, pp, ss = dev(diags[i]), dev(psi0[i]), dev(srcs[i])
        A.bind_coeffs_native(dd, up_d, lo_d)
        refPerf.append(A.solve(pp, ss, **kw)); ref.append(host(pp))
    # all at once
    dd, pp, ss = [dev(a) for a in diags], [dev(a) for a in psi0], [dev(a) for a in srcs]
    perf = A.solve_multi(dd, up_d, lo_d, pp, ss, **kw)
    counts = [p["nIterations"] for p in perf]
    assert counts == [p["nIterations"] for p in refPerf], (counts, refPerf)
    assert counts[1] == 0 and len(set(counts)) >= 2, counts                   # the systems really stop at different times
    for i in range(nSys):
        assert np.array_equal(host(pp[i]).view(np.uint64), ref[i].view(np.uint64)), (i, np.abs(host(pp[i]) - ref[i]).max())
        for key in ("initialResidual", "finalResidual", "converged"):
            assert perf[i][key] == refPerf[i][key], (i, key, perf[i], refPerf[i])
        r = srcs[i] - Amul(l, u, diags[i], upper, lower, host(pp[i]))
        assert np.abs(r).sum() <= 1e-6 * np.abs(srcs[i]).sum() + 1e-12
    # and the matrix is usable for an ordinary solve afterwards (reciprocal diagonal recomputed)
    d0, p0, s0 = dev(diags[0]), dev(psi0[0]), dev(srcs[0])
    A.bind_coeffs_native(d0, up_d, lo_d)
    again = A.solve(p0, s0, **kw)
    assert again["nIterations"] == refPerf[0]["nIterations"] and np.array_equal(host(p0).view(np.uint64), ref[0].view(np.uint64))
    A.close()


def test_other_selections_run_one_after_the_other(ctx):
    A, N, l, u, upper, lower, base, rng = build(ctx, (16, 16, 16), True, seed=5)
    fmap = A.face_map()
    nat = lambda f: (lambda out: (out.__setitem__(fmap, f), out)[1])(np.zeros(A.nNative))
    up_d, lo_d = ctx.to_device(nat(upper)), ctx.to_device(nat(lower))
    diags = [base * 1.1 + 0.1, base * 1.3 + 0.2]; srcs = [rng.standard_normal(N) for _ in range(2)]
    kw = dict(solver="smoothSolver", preconditioner="symGaussSeidel", tolerance=1e-8, relTol=0.0, maxIter=50)
    ref = []
    for i in range(2):
        dd, pp, ss = ctx.to_device(diags[i]), ctx.zeros(N), ctx.to_device(srcs[i])
        A.bind_coeffs_native(dd, up_d, lo_d)
        perf = A.solve(pp, ss, solver="smoothSolver", smoother="symGaussSeidel", tolerance=1e-8, maxIter=50)
        ctx.sync(); ref.append((pp.cpu().numpy(), perf))
    dd, pp, ss = [ctx.to_device(a) for a in diags], [ctx.zeros(N) for _ in range(2)], [ctx.to_device(a) for a in srcs]
    perf = A.solve_multi(dd, up_d, lo_d, pp, ss, **kw)
    ctx.sync()
    for i in range(2):
        assert perf[i]["nIterations"] == ref[i][1]["nIterations"] and np.array_equal(pp[i].cpu().numpy(), ref[i][0])
    A.close()
