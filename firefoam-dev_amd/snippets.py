"""Host-side binding of libffm_refsnippets.so -- the reference's unchanged solver/rhoEqn.H, UEqn.H, YEEqn.H, pEqn.H (and phrghEqn.H,
setMultiRegionDeltaT.H ...) compiled against include/ffmFoam.H (examples/fireFoam_snippets.C).  The library exists only where the
reference was mounted at build time (it travels to the GPU box as a built file).  `SnippetCase` mirrors `struct snippetCase`; tests
fill it from the oracle's state, bench.py from the compiled plume case (from_plume) to time the SAME case through the class layer."""
import ctypes as C
import os

import numpy as np

from . import binding as B

_dp, _dpp = C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_double))


class SnippetCase(C.Structure):
    _fields_ = ([("deltaT", C.c_double)] + [(k, C.c_double) for k in ("RR", "Cp", "Tref", "pRef", "mu", "Pr", "sO2", "HC", "tau")]
                + [(k, C.c_int) for k in ("nSpecies", "inertIndex", "fuelIndex", "o2Index")]
                + [("W", _dp), ("nu", _dp)]
                + [("rho", _dp), ("U", _dp), ("p", _dp), ("p_rgh", _dp), ("h", _dp), ("Y", _dpp)]
                + [("K", _dp), ("dpdt", _dp), ("phiF", _dp), ("phiB", _dp)]
                + [("gh", _dp), ("ghfF", _dp), ("ghfB", _dp)]
                + [("fU", _dp), ("refU", _dp), ("fixesU", _dp)]
                + [("fY", _dp), ("refY", _dpp), ("fH", _dp), ("refH", _dp)]
                + [("fluxMaskP", _dp), ("totalMaskP", _dp), ("ph_rgh_b", _dp), ("p_rghB", _dp)]
                + [("rhoOut", _dp), ("UOut", _dp), ("pOut", _dp), ("p_rghOut", _dp), ("hOut", _dp), ("YOut", _dpp), ("TOut", _dp), ("KOut", _dp)]
                + [("dpdtOut", _dp), ("phiOutF", _dp), ("phiOutB", _dp), ("p_rghBOut", _dp), ("nIterOut", C.POINTER(C.c_int)), ("nIterCap", C.c_int)]
                + [("radiationFreq", C.c_int), ("kAbs", C.c_double), ("sigmaSB", C.c_double), ("dAve", _dp), ("omega", _dp), ("GOut", _dp)]
                + [("psiB", _dp), ("resOut", _dp)]
                + [("adjustTimeStep", C.c_int), ("maxCo", C.c_double), ("maxDeltaT", C.c_double), ("dtOut", _dp), ("emptyDirections", C.c_int)]
                + [("wallFireSelection", C.c_int), ("gamg", C.c_void_p)]
                + [("pyro", C.c_void_p), ("pyroCols", C.c_int), ("pyroMap", C.POINTER(C.c_int)), ("pyroQin", _dp)]
                + [(k, C.c_double) for k in ("pyroEmissivity", "pyroAbsorptivity", "pyroHocSolid", "pyroQFuel")]
                + [(k, C.c_int) for k in ("fvdomReal", "radNPhi", "radNTheta", "radMaxIter", "radDivScheme")]
                + [(k, C.c_double) for k in ("radTolerance", "radEhrr1", "radEhrr2")] + [("radMlrMask", _dp), ("radMlrMask2", _dp), ("radEmissivity", _dp)]
                + [("qinOut", _dp), ("radItersOut", C.POINTER(C.c_int)), ("pyroInStep", C.c_int), ("pyroMaxDi", C.c_double)])


# constants of the synthetic plume case (csrc/ffm_plume.hip; the stand-ins documented in DESIGN section 0)
PLUME = dict(RR=8314.47, Cp=1005.0, Tref=298.15, pRef=101325.0, mu=1.8e-5, Pr=0.7, sO2=3.6282945, HC=46357151.0, tau=0.05)
WMOL = np.array([31.9988, 18.0153, 44.0962, 44.01, 28.0134])
NU = np.array([-3.6282945, 4 * 18.0153 / 44.0962, -1.0, 3 * 44.01 / 44.0962, 0.0])
SPECIES = ["O2", "H2O", "C3H8", "CO2", "N2"]


def libpath():
    return os.path.join(os.path.dirname(B.libpath()), "libffm_refsnippets.so")


def load():
    """the library, or None where it was not built (no reference at build time)"""
    if not os.path.exists(libpath()):
        return None
    B.lib()
    lib = C.CDLL(libpath())
    argt = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]
    lib.firefoam_snippets_create.restype = C.c_void_p; lib.firefoam_snippets_create.argtypes = argt
    lib.firefoam_snippets_advance.restype = C.c_int; lib.firefoam_snippets_advance.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
    lib.firefoam_snippets_destroy.argtypes = [C.c_void_p]
    return lib


class FromPlume:
    """The state of a compiled plume case (ffm.Plume, single block, before its first step) as a SnippetCase on the plume's own matrix
    and mesh: the same case, the same start state, the same device mesh -- advanced by the reference's equation files."""

    def __init__(self, plume):
        self.keep = []
        N, F = plume.nCells, plume.nFaces
        P = lambda a: self._p(a)
        raw = plume.raw
        kind = raw("kind")
        Bn = len(kind)
        fixed = (kind < 1.5).astype(np.float64)
        fU = np.concatenate([raw("fStaticU%d" % c) for c in range(3)]); refU = np.concatenate([raw("refU%d" % c) for c in range(3)])
        U = np.stack([raw("Ux"), raw("Uy"), raw("Uz")])
        nit = (C.c_int * 32)(); self.keep.append(nit)
        self.nit = nit
        self.cs = SnippetCase(
            deltaT=1e-3, nSpecies=5, inertIndex=4, fuelIndex=2, o2Index=0, W=P(WMOL), nu=P(NU),
            rho=P(raw("rho")), U=P(U), p=P(raw("p")), p_rgh=P(raw("p_rgh")), h=P(raw("h")), Y=self._pp([raw(s) for s in SPECIES]),
            K=P(raw("K")), dpdt=P(raw("dpdt")), phiF=P(raw("phi")), phiB=P(raw("phib")), gh=P(raw("gh")), ghfF=P(raw("ghf")), ghfB=P(raw("ghfB")),
            fU=P(fU), refU=P(refU), fixesU=P(fixed), fY=P(raw("fStaticS")), refY=self._pp([raw("refY%d" % i) for i in range(5)]),
            fH=P(raw("fStaticH")), refH=P(raw("refH")), fluxMaskP=P(fixed), totalMaskP=P(1.0 - fixed), ph_rgh_b=P(raw("ph_rgh_b")), p_rghB=P(raw("ph_rgh_b")),
            nIterOut=nit, nIterCap=32, **PLUME)
        self.N, self.F, self.B = N, F, Bn

    def _p(self, a):
        a = np.ascontiguousarray(a, np.float64); self.keep.append(a)
        return a.ctypes.data_as(_dp)

    def _pp(self, arrs):
        arrs = [np.ascontiguousarray(a, np.float64) for a in arrs]; self.keep.append(arrs)
        arr = (_dp * len(arrs))(*[a.ctypes.data_as(_dp) for a in arrs]); self.keep.append(arr)
        return arr

    def outputs(self):
        """host arrays for everything a download writes (advance(download=1)); returns them by name"""
        N, F, Bn = self.N, self.F, self.B
        out = dict(rho=np.empty(N), U=np.empty((3, N)), p=np.empty(N), p_rgh=np.empty(N), h=np.empty(N), T=np.empty(N), K=np.empty(N), dpdt=np.empty(N),
                   phi=np.empty(F), phib=np.empty(Bn), p_rghB=np.empty(Bn), Y=[np.empty(N) for _ in range(5)])
        self.keep.append(out)
        cs = self.cs
        for key, fld in (("rho", "rhoOut"), ("U", "UOut"), ("p", "pOut"), ("p_rgh", "p_rghOut"), ("h", "hOut"), ("T", "TOut"), ("K", "KOut"), ("dpdt", "dpdtOut"),
                         ("phi", "phiOutF"), ("phib", "phiOutB"), ("p_rghB", "p_rghBOut")):
            setattr(cs, fld, out[key].ctypes.data_as(_dp))
        yo = (_dp * 5)(*[a.ctypes.data_as(_dp) for a in out["Y"]]); self.keep.append(yo)
        cs.YOut = yo
        return out
