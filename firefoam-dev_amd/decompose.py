"""Host-side domain decomposition over the C ABI (ffm_partition_*, ffm_subdomain_*: csrc/ffm_partition.cpp) -- decomposePar's role
for this path (reference: scotch / simple, cases/steckler/system/decomposeParDict:18-20, cases/wallFireSpread2D/system/
decomposeParDict:18-27).  Works on any LDU graph: the baffled steckler room, unstructured meshes."""
import ctypes as C

import numpy as np

from . import binding as B


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def partition_rcb(centres, nParts):
    """centres[nCells][3] -> part[nCells] (recursive coordinate bisection)"""
    Cc = np.ascontiguousarray(np.asarray(centres, np.float64).T)
    part = np.empty(Cc.shape[1], np.int32)
    B._check(B.lib().ffm_partition_rcb(Cc.shape[1], Cc.ctypes.data_as(C.POINTER(C.c_double)), int(nParts), _ip(part)), "ffm_partition_rcb")
    return part


def partition_graph(nCells, l, u, nParts):
    """greedy graph growing on the LDU graph -> part[nCells]"""
    l = np.ascontiguousarray(l, np.int32); u = np.ascontiguousarray(u, np.int32)
    part = np.empty(nCells, np.int32)
    B._check(B.lib().ffm_partition_graph(int(nCells), len(l), _ip(l), _ip(u), int(nParts), _ip(part)), "ffm_partition_graph")
    return part


class SubDomain:
    """One rank's share of a decomposed LDU graph in the ghost-cell form of ffm_ldu_create_ext (see include/ffm.h)."""

    def __init__(self, nCells, l, u, part, nParts, rank):
        l = np.ascontiguousarray(l, np.int32); u = np.ascontiguousarray(u, np.int32); part = np.ascontiguousarray(part, np.int32)
        h = C.c_void_p()
        B._check(B.lib().ffm_subdomain_create(int(nCells), len(l), _ip(l), _ip(u), _ip(part), int(nParts), int(rank), C.byref(h)), "ffm_subdomain_create")
        n = [C.c_int() for _ in range(6)]
        B.lib().ffm_subdomain_sizes(h, *[C.byref(x) for x in n])
        self.nOwned, self.nGhost, self.nFaces, nNbr, nSend, nCut = [x.value for x in n]
        I = lambda k: np.empty(max(k, 1), np.int32)
        self.gcell = I(self.nOwned + self.nGhost); B.lib().ffm_subdomain_cells(h, _ip(self.gcell))
        self.gcell = self.gcell[:self.nOwned + self.nGhost]
        self.l, self.u, self.gface, self.flip = I(self.nFaces), I(self.nFaces), I(self.nFaces), I(self.nFaces)
        B.lib().ffm_subdomain_faces(h, _ip(self.l), _ip(self.u), _ip(self.gface), _ip(self.flip))
        self.l, self.u, self.gface, self.flip = (a[:self.nFaces] for a in (self.l, self.u, self.gface, self.flip))
        self.nbrRank, self.sendCount, self.recvCount, self.tags, self.sendCells = I(nNbr), I(nNbr), I(nNbr), I(nNbr), I(nSend)
        B.lib().ffm_subdomain_exchange(h, _ip(self.nbrRank), _ip(self.sendCount), _ip(self.sendCells), _ip(self.recvCount), _ip(self.tags))
        self.nbrRank, self.sendCount, self.recvCount, self.tags = (a[:nNbr] for a in (self.nbrRank, self.sendCount, self.recvCount, self.tags))
        self.sendCells = self.sendCells[:nSend]
        self.cutStart, self.cutCell, self.cutFace, self.cutFlip = I(nNbr + 1), I(nCut), I(nCut), I(nCut)
        B.lib().ffm_subdomain_cut_faces(h, _ip(self.cutStart), _ip(self.cutCell), _ip(self.cutFace), _ip(self.cutFlip))
        self.cutStart = self.cutStart[:nNbr + 1]
        self.cutCell, self.cutFace, self.cutFlip = (a[:nCut] for a in (self.cutCell, self.cutFace, self.cutFlip))
        B.lib().ffm_subdomain_destroy(h)
        self.rank, self.nParts, self.globalCells = rank, nParts, int(nCells)

    def coeffs(self, diag, upper, lower=None):
        """the rank's coefficients from the global ones: diag[nOwned + nGhost] (0 on ghosts), upper / lower [nFaces] with the
        cut faces whose global owner is the ghost flipped"""
        d = np.zeros(self.nOwned + self.nGhost); d[:self.nOwned] = np.asarray(diag)[self.gcell[:self.nOwned]]
        up_g = np.asarray(upper)[self.gface]; lo_g = up_g if lower is None else np.asarray(lower)[self.gface]
        fl = self.flip.astype(bool)
        return d, np.where(fl, lo_g, up_g), (None if lower is None else np.where(fl, up_g, lo_g))

    def field(self, glob):
        """a global cell field on the rank's owned + ghost cells"""
        return np.asarray(glob)[self.gcell]

    def interface_form(self, upper, lower=None):
        """OpenFOAM's processor-patch form: the LDU of the owned cells alone + per neighbour (faceCells, interfaceBouCoeffs =
        -coefficient of the cut face in the owned cell's row), faces of a patch in ascending global face label on both sides"""
        keep = self.u < self.nOwned
        pat = []
        for q in range(len(self.nbrRank)):
            s, e = self.cutStart[q], self.cutStart[q + 1]
            gf, fl = self.cutFace[s:e], self.cutFlip[s:e].astype(bool)
            up_g = np.asarray(upper)[gf]; lo_g = up_g if lower is None else np.asarray(lower)[gf]
            pat.append((self.cutCell[s:e].copy(), -np.where(fl, lo_g, up_g)))
        return keep, pat
