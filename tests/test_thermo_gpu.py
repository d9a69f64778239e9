"""SURVEY 8(f) N2 on the device: thermo.correct() (hePsiThermo of the steckler case's janaf / sutherland / perfectGas mixture),
the reference's eddyDissipationModel::correct and kEqn's nut / alphat as streaming kernels (csrc/ffm_thermo.hip) against the oracle
(oracle/thermo.py, oracle/steckler_case.py -- pinned on the reference's start-up numbers and the golden log's first time step):
on the states of that first step and on a synthetic flame sheet (every species present, 300-2300 K, both janaf ranges)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DATA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "steckler_case_data.json")))


def _species(TH):
    tab = {n: {k: (np.array(v) if isinstance(v, list) else v) for k, v in d.items()} for n, d in DATA["table"].items()}
    return TH.Species(DATA["species"], tab), tab


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_thermo_correct_and_he_match_the_oracle(ffm, ctx):
    from oracle import thermo as TH
    sp, tab = _species(TH)
    th = ffm.Thermo(ctx, DATA["species"], tab, TH.RR)
    n = 20000
    x = np.linspace(0.0, 1.0, n)
    # a mixing line from air to products to fuel, temperatures across Tcommon = 1000 K
    Yf = np.clip(2.0 * x - 1.0, 0, 1); Yp = 1.0 - np.abs(2.0 * x - 1.0)
    Y = np.zeros((5, n)); idx = {s: i for i, s in enumerate(DATA["species"])}
    Y[idx["C3H8"]] = Yf; Y[idx["H2O"]] = 0.1 * Yp; Y[idx["CO2"]] = 0.18 * Yp
    Y[idx["O2"]] = 0.233 * (1 - Yf - 0.28 * Yp); Y[idx["N2"]] = 1.0 - Y.sum(axis=0)
    Y[:, :50] = 0.0; Y[idx["O2"], :50] = 1.0                                       # pure O2: the later species add nothing
    T = 300.0 + 2000.0 * Yp + 40.0 * np.sin(37 * x)
    p = np.full(n, 101325.0)
    mx = sp.mixture(Y)
    he = mx.Hs(p, T)
    T0 = T * (1.0 + 0.08 * np.cos(11 * x))                                          # the "old" temperature the iteration starts from
    Tr = mx.THs(he, p, T0)
    D = lambda a: ctx.to_device(np.ascontiguousarray(a, np.float64))
    Yd = [D(Y[i]) for i in range(5)]
    Td, psi, mu, al = D(T0), ctx.empty(n), ctx.empty(n), ctx.empty(n)
    th.correct(Yd, D(he), D(p), Td, psi, mu, al)
    assert _rel(Td.cpu().numpy(), Tr) < 1e-13 and np.abs(Tr - T).max() < 0.3          # converged to the Newton tolerance T0*1e-4
    assert _rel(psi.cpu().numpy(), mx.psi(p, Tr)) < 1e-13
    assert _rel(mu.cpu().numpy(), mx.mu(p, Tr)) < 1e-13 and _rel(al.cpu().numpy(), mx.alphah(p, Tr)) < 1e-13
    hd = ctx.empty(n)
    th.he(Yd, D(T), hd)
    assert _rel(hd.cpu().numpy(), he) < 1e-13
    th.close()


def test_first_step_thermo_edc_and_nut_of_the_steckler_case(ffm, ctx):
    """the states of the golden log's first time step (oracle/steckler_case.py): thermo.correct() after the enthalpy solve on the
    cells and on every patch, the EDC source, nut / alphat"""
    from oracle import steckler_case as SC, thermo as TH
    sp, tab = _species(TH)
    th = ffm.Thermo(ctx, DATA["species"], tab, TH.RR)
    D = lambda a: ctx.to_device(np.ascontiguousarray(a, np.float64))
    c = SC.StecklerCase()
    c.hook = None
    c.hydrostatic_init(); c.correct_nut()
    c.psi0, c.p0, c.p_rgh0, c.phi0 = c.psi.copy(), c.p.copy(), c.p_rgh.copy(), c.phi.copy()
    c.dpdt = np.zeros(c.m.nCells); c.K = np.zeros(c.m.nCells)
    c.step_begin(); c.U_eqn()
    before = {}
    orig = c.thermo_correct

    def spy():              # the state the reference's thermo.correct() call of solver/YEEqn.H:114 starts from
        before.update(he=c.he.copy(), T=c.T.copy(), heb=[h.copy() for h in c.heb], Tb=[t.copy() for t in c.Tb])
        orig()
    c.thermo_correct = spy
    c.YE_eqn(True)
    assert before
    N = c.m.nCells
    Yd = [D(c.Y[i]) for i in range(5)]
    Td, psi, mu, al = D(before["T"]), ctx.empty(N), ctx.empty(N), ctx.empty(N)
    th.correct(Yd, D(before["he"]), D(c.p), Td, psi, mu, al)
    assert _rel(Td.cpu().numpy(), c.T) < 1e-14 and _rel(psi.cpu().numpy(), c.psi) < 1e-14
    assert _rel(mu.cpu().numpy(), c.mu) < 1e-14 and _rel(al.cpu().numpy(), c.alpha) < 1e-14
    allT = np.concatenate([c.T] + list(c.Tb))                                            # min / max of a volScalarField include its patch values
    assert abs(allT.max() - 300.49) < 0.005 and abs(allT.min() - 298.15) < 0.005         # the golden log's min / max(T) of this step
    for q, pn in enumerate(SC.PATCHES):
        nf = c.m.patches[q].size
        Yb = [D(c.Yb[i][q]) for i in range(5)]
        if pn in c.FIXES_T or (c.baffle_fixed and pn.startswith("baffle")):
            hd = ctx.empty(nf); th.he(Yb, D(c.Tb[q]), hd)
            assert _rel(hd.cpu().numpy(), c.heb[q]) < 1e-14, pn
        else:
            Tq, pq = D(before["Tb"][q]), ctx.empty(nf)
            th.correct(Yb, D(before["heb"][q]), D(c.pb[q]), Tq, pq, None, None)
            assert _rel(Tq.cpu().numpy(), c.Tb[q]) < 1e-14 and _rel(pq.cpu().numpy(), c.psib[q]) < 1e-14, pn
    # EDC on a burning state: fuel around the room's centre, the step's rho / alpha / delta, k raised in the flame
    L = ffm.lib()
    Yf = np.clip(0.3 - np.linalg.norm(c.m.C - c.m.C.mean(axis=0), axis=1), 0, None); Yo = c.Y[c.iO2]
    k = c.k * (1.0 + 50.0 * Yf)
    eps = c.Ce * k * np.sqrt(k) / c.delta
    rt = np.maximum(4.0 * eps / np.maximum(k, 1e-15), 0.0 * c.alpha / c.rho / c.delta ** 2)
    w_ref = c.rho * np.minimum(Yf, Yo / c.rx.s) / c.dt / 1.0 * (1.0 - np.exp(-1.0 * c.dt * rt))
    wd, qd = ctx.empty(N), ctx.empty(N)
    P = lambda t: C.c_void_p(t.data_ptr())
    keep = [D(c.rho), D(k), D(c.delta), D(c.alpha), D(Yf), D(Yo)]
    ctx._ready()
    assert L.ffm_edc_correct_d(ctx.h, N, *[P(t) for t in keep], float(c.rx.s), float(c.dt), float(c.Ce), 4.0, 0.0, 1.0, float(c.rx.qFuel), P(wd), P(qd)) == 0
    assert w_ref.max() > 0 and _rel(wd.cpu().numpy(), w_ref) < 1e-13 and _rel(qd.cpu().numpy(), c.rx.qFuel * w_ref) < 1e-13
    nut, alt = ctx.empty(N), ctx.empty(N)
    assert L.ffm_les_keqn_nut_d(ctx.h, N, float(c.Ck), float(c.Prt), P(keep[1]), P(keep[2]), P(keep[0]), P(nut), P(alt)) == 0
    nr = c.Ck * np.sqrt(k) * c.delta
    assert _rel(nut.cpu().numpy(), nr) < 1e-15 and _rel(alt.cpu().numpy(), c.rho * nr / c.Prt) < 1e-15
    th.close()


def test_thermo_correct_through_the_handle_of_the_foam_layer(ffm, ctx):
    """`thermo.correct()` as solver/YEEqn.H:114 calls it, on the handle include/fireFoamHandles.H gives the reference's files
    (hePsiThermoJanaf over ffm_thermo_*): the steckler mesh with all its patches, state of the golden log's first time step --
    T, psi, mu, alpha, he and rho = psi*p on the cells and on every patch face against the oracle's hePsiThermo::calculate
    (patches that fix T evaluate he from T, the others T from he)."""
    from oracle import steckler_case as SC, thermo as TH
    sp, tab = _species(TH)
    th = ffm.Thermo(ctx, DATA["species"], tab, TH.RR)
    c = SC.StecklerCase()
    c.hook = None
    c.hydrostatic_init(); c.correct_nut()
    c.psi0, c.p0, c.p_rgh0, c.phi0 = c.psi.copy(), c.p.copy(), c.p_rgh.copy(), c.phi.copy()
    c.dpdt = np.zeros(c.m.nCells); c.K = np.zeros(c.m.nCells)
    c.step_begin(); c.U_eqn()
    before = {}
    orig = c.thermo_correct

    def spy():
        before.update(he=c.he.copy(), T=c.T.copy(), heb=[h.copy() for h in c.heb], Tb=[t.copy() for t in c.Tb])
        orig()
    c.thermo_correct = spy
    c.YE_eqn(True)
    m = c.m
    N = m.nCells
    cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
    l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
    A = ffm.lduMatrix(ctx, N, l2, u2)
    patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
    mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
    B = sum(p.size for p in m.patches)
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    dp = C.POINTER(C.c_double)
    h = lambda a: np.ascontiguousarray(a, np.float64)
    keep = []

    def P(a):
        a = h(a); keep.append(a)
        return a.ctypes.data_as(dp)

    def PP(arrs):
        arrs = [h(a) for a in arrs]; keep.append(arrs)
        arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr)
        return arr
    cat = np.concatenate
    fixes = cat([np.full(p.size, 1.0 if (pn in c.FIXES_T or (c.baffle_fixed and pn.startswith("baffle"))) else 0.0) for pn, p in zip(SC.PATCHES, m.patches)])
    outC, outB = np.empty((6, N)), np.empty((6, B))
    lib.b1_thermo_handle.restype = C.c_int
    lib.b1_thermo_handle.argtypes = [C.c_void_p] * 4 + [C.c_int, C.POINTER(dp), C.POINTER(dp)] + [dp] * 9
    ctx._ready()
    rc = lib.b1_thermo_handle(ctx.h, A.h, mesh.h, th.h, 5, PP([c.Y[i][cOrd] for i in range(5)]), PP([cat(c.Yb[i]) for i in range(5)]),
                              P(before["he"][cOrd]), P(cat(before["heb"])), P(c.p[cOrd]), P(cat(c.pb)), P(before["T"][cOrd]), P(cat(before["Tb"])),
                              P(fixes), outC.ctypes.data_as(dp), outB.ctypes.data_as(dp))
    assert rc == 0
    inv = np.empty(N, np.int64); inv[cOrd] = np.arange(N)
    for k, (cell, bnd) in enumerate(((c.T, c.Tb), (c.psi, c.psib), (c.mu, c.mub), (c.alpha, c.alphab), (c.he, c.heb), (c.psi * c.p, [a * b for a, b in zip(c.psib, c.pb)]))):
        assert _rel(outC[k][inv], cell) < 1e-14, k
        assert _rel(outB[k], cat(bnd)) < 1e-14, k
    mesh.close(); A.close(); th.close()
