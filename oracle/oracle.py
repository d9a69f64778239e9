"""ORACLE binding (test infrastructure, not product code).

ctypes wrapper around oracle/_build/libffo.so, the plain-C CPU restatement of
the OpenFOAM-dev algorithms on fireFoam's hot path (see oracle/ffo.h for the
reference file:line citations).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libffo.so")

PCG, PBICGSTAB, PBICG, DIAGONAL, SMOOTH = 0, 1, 2, 3, 4
NONE, DIC, DILU, GS, SYMGS, DIAGONALP = 0, 1, 2, 3, 4, 5


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Perf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("nIterations", C.c_int), ("converged", C.c_int), ("singular", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Controls(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double),
                ("minIter", C.c_int), ("maxIter", C.c_int), ("nSweeps", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                          C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))


class Comm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("nRanks", C.c_int),
                ("allreduce_sum", ALLREDUCE_FN), ("exchange", EXCHANGE_FN)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.ffo_ldu_create.restype = vp
        L.ffo_ldu_create.argtypes = [C.c_int, C.c_int, ip, ip]
        L.ffo_ldu_destroy.argtypes = [vp]
        L.ffo_ldu_set_coeffs.argtypes = [vp, dp, dp, dp]
        L.ffo_ldu_set_global_cells.argtypes = [vp, C.c_long]
        L.ffo_ldu_set_interfaces.argtypes = [vp, C.c_int, ip, C.POINTER(ip), C.POINTER(dp), C.POINTER(dp)]
        for name in ("ffo_amul", "ffo_tmul"):
            getattr(L, name).argtypes = [vp, dp, dp, vp]
        L.ffo_sumA.argtypes = [vp, dp]
        L.ffo_residual.argtypes = [vp, dp, dp, dp, vp]
        L.ffo_norm_factor.restype = C.c_double
        L.ffo_norm_factor.argtypes = [vp, dp, dp, dp, dp, vp]
        for name in ("ffo_dic_calc_rD", "ffo_dilu_calc_rD"):
            getattr(L, name).argtypes = [vp, dp]
        for name in ("ffo_dic_precondition", "ffo_dilu_precondition", "ffo_dilu_preconditionT"):
            getattr(L, name).argtypes = [vp, dp, dp, dp]
        L.ffo_gs_smooth.argtypes = [vp, dp, dp, C.c_int, C.c_int, vp]
        L.ffo_solve.restype = C.c_int
        L.ffo_solve.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Controls), dp, dp, C.POINTER(Perf), vp]
        L.ffo_hash_u.restype = C.c_double
        L.ffo_hash_u.argtypes = [C.c_uint64, C.c_uint64]
        L.ffo_hex_counts.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.ffo_hex_ldu.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
        L.ffo_solve_multi.restype = C.c_int
        L.ffo_solve_multi.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(ip), C.POINTER(ip), C.c_int, C.c_int,
                                      C.POINTER(Controls), C.POINTER(dp), C.POINTER(dp), C.POINTER(Perf)]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def hash_u(seed, idx):
    """Vectorised SURVEY 8(d) hash; bit-identical to ffo_hash_u."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ idx) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hex_ldu(nx, ny, nz):
    nC, nF = C.c_long(), C.c_long()
    lib().ffo_hex_counts(nx, ny, nz, C.byref(nC), C.byref(nF))
    l = np.empty(nF.value, np.int32)
    u = np.empty(nF.value, np.int32)
    lib().ffo_hex_ldu(nx, ny, nz, _i(l), _i(u))
    return nC.value, l, u


class Ldu:
    """Oracle lduMatrix (serial or one rank of a decomposed case)."""

    def __init__(self, nCells, l, u):
        self.l, self.u = i32(l), i32(u)
        self.nCells, self.nFaces = int(nCells), len(self.l)
        self.h = lib().ffo_ldu_create(self.nCells, self.nFaces, _i(self.l), _i(self.u))
        if not self.h:
            raise ValueError("invalid LDU addressing (need l<u, owner-sorted)")
        self._keep = []
        self.comm = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().ffo_ldu_destroy(self.h)
            self.h = None

    def set_coeffs(self, diag, upper, lower=None):
        diag, upper = f64(diag), f64(upper)
        lo = None if lower is None else f64(lower)
        lib().ffo_ldu_set_coeffs(self.h, _d(diag), _d(upper), None if lo is None else _d(lo))
        return self

    def set_interfaces(self, faceCells, bouCoeffs, intCoeffs=None):
        n = len(faceCells)
        fc = [i32(a) for a in faceCells]
        bc = [f64(a) for a in bouCoeffs]
        ic = bc if intCoeffs is None else [f64(a) for a in intCoeffs]
        sizes = (C.c_int * n)(*[len(a) for a in fc])
        fcp = (C.POINTER(C.c_int) * n)(*[_i(a) for a in fc])
        bcp = (C.POINTER(C.c_double) * n)(*[_d(a) for a in bc])
        icp = (C.POINTER(C.c_double) * n)(*[_d(a) for a in ic])
        lib().ffo_ldu_set_interfaces(self.h, n, sizes, fcp, bcp, icp)
        return self

    def set_global_cells(self, g):
        lib().ffo_ldu_set_global_cells(self.h, int(g))
        return self

    def _c(self):
        return C.byref(self.comm) if self.comm is not None else None

    def amul(self, x):
        x = f64(x); y = np.empty_like(x)
        lib().ffo_amul(self.h, _d(x), _d(y), self._c()); return y

    def tmul(self, x):
        x = f64(x); y = np.empty_like(x)
        lib().ffo_tmul(self.h, _d(x), _d(y), self._c()); return y

    def sumA(self):
        s = np.empty(self.nCells)
        lib().ffo_sumA(self.h, _d(s)); return s

    def residual(self, x, b):
        x, b = f64(x), f64(b); r = np.empty_like(x)
        lib().ffo_residual(self.h, _d(x), _d(b), _d(r), self._c()); return r

    def norm_factor(self, x, b):
        x, b = f64(x), f64(b)
        Ax = self.amul(x); tmp = np.empty_like(x)
        return lib().ffo_norm_factor(self.h, _d(x), _d(b), _d(Ax), _d(tmp), self._c())

    def dic_rD(self):
        rD = np.empty(self.nCells); lib().ffo_dic_calc_rD(self.h, _d(rD)); return rD

    def dilu_rD(self):
        rD = np.empty(self.nCells); lib().ffo_dilu_calc_rD(self.h, _d(rD)); return rD

    def dic_precondition(self, rD, r):
        rD, r = f64(rD), f64(r); w = np.empty_like(r)
        lib().ffo_dic_precondition(self.h, _d(rD), _d(r), _d(w)); return w

    def dilu_precondition(self, rD, r, transpose=False):
        rD, r = f64(rD), f64(r); w = np.empty_like(r)
        fn = lib().ffo_dilu_preconditionT if transpose else lib().ffo_dilu_precondition
        fn(self.h, _d(rD), _d(r), _d(w)); return w

    def gs_smooth(self, psi, b, nSweeps=1, sym=True):
        psi = f64(psi).copy(); b = f64(b)
        lib().ffo_gs_smooth(self.h, _d(psi), _d(b), nSweeps, 1 if sym else 0, self._c()); return psi

    def solve(self, solver, precond, psi, source, tolerance=1e-6, relTol=0.0,
              minIter=0, maxIter=1000, nSweeps=1):
        psi = f64(psi).copy(); source = f64(source)
        k = Controls(tolerance, relTol, minIter, maxIter, nSweeps)
        perf = Perf()
        rc = lib().ffo_solve(self.h, solver, precond, C.byref(k), _d(psi), _d(source), C.byref(perf), self._c())
        if rc:
            raise RuntimeError("ffo_solve rc=%d" % rc)
        return psi, perf.as_dict()


def solve_multi(ldus, nbrRank, nbrPatch, solver, precond, psis, sources, tolerance=1e-6,
                relTol=0.0, minIter=0, maxIter=1000, nSweeps=1):
    """Run P sub-domain LDUs as P pthreads (block-Jacobi, halo exchange)."""
    P = len(ldus)
    psis = [f64(p).copy() for p in psis]
    sources = [f64(s) for s in sources]
    nr = [i32(a) for a in nbrRank]
    npch = [i32(a) for a in nbrPatch]
    hs = (C.c_void_p * P)(*[a.h for a in ldus])
    nrp = (C.POINTER(C.c_int) * P)(*[_i(a) for a in nr])
    npp = (C.POINTER(C.c_int) * P)(*[_i(a) for a in npch])
    pp = (C.POINTER(C.c_double) * P)(*[_d(a) for a in psis])
    sp = (C.POINTER(C.c_double) * P)(*[_d(a) for a in sources])
    perfs = (Perf * P)()
    k = Controls(tolerance, relTol, minIter, maxIter, nSweeps)
    rc = lib().ffo_solve_multi(P, hs, nrp, npp, solver, precond, C.byref(k), pp, sp, perfs)
    if rc:
        raise RuntimeError("ffo_solve_multi rc=%d" % rc)
    return psis, [p.as_dict() for p in perfs]
