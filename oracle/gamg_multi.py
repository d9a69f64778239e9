"""ORACLE (test infrastructure): GAMG on a decomposed mesh the way OpenFOAM runs it without a processorAgglomerator (the default;
cases/wallFireSpread2D/system/fvSolution:36-60 on BASELINE config 5's four ranks): one rank of P, in OpenFOAM's processor-patch form.
  * every rank agglomerates its own cells over its internal faces (pairGAMGAgglomeration on the rank's lduAddressing; oracle/gamg.py);
  * GAMGAgglomeration::continueAgglomerating is global: sum(nCoarse) >= nProcs*nCellsInCoarsestLevel and sum(nCoarse) < sum(nFine);
  * the processor interfaces are agglomerated with the cells (GAMGInterface / processorGAMGInterface): along a patch's fine faces, in
    their order, every new pair (local coarse cell, neighbour's coarse cell) opens a coarse interface face -- the neighbour's restrict
    addressing of its patch-face cells is exchanged once per level -- and the coarse interface coefficients are the sums over the
    fine faces of a coarse face (GAMGSolver::agglomerateMatrix / GAMGInterface::agglomerateCoeffs);
  * smoothers update the interfaces before every sweep (the C oracle's gs_smooth / residual / amul with a communicator), DIC / DILU are
    block-Jacobi, the coarsest level is solved by PCG + DIC / PBiCGStab + DILU over all ranks, the correction scale factors, the
    normalisation factor and the residual norms are global sums.
`comm` supplies allreduce(array) -> summed array and exchange(list of arrays, one per neighbour) -> list of received arrays, plus the
ffo communicator for the C kernels (oracle.Comm).  Restated from OpenFOAM-dev's GAMG sources as this builder recalls them; PARITY
UNPINNED by reference data.  Only tests/ may import this module."""
import numpy as np

from . import gamg, oracle as O


class AgglomerationMulti:
    def __init__(self, nOwned, l, u, faceWeights, patchFaceCells, comm, nRanks, nCellsInCoarsestLevel=10, maxLevels=50, forward=True):
        """l, u, faceWeights: the rank's INTERNAL faces; patchFaceCells[q]: local cell of every face of the interface towards neighbour q"""
        self.nCells = [int(nOwned)]
        self.l, self.u = [np.asarray(l, np.int32)], [np.asarray(u, np.int32)]
        self.restrictMap, self.faceRestrict, self.faceFlip = [], [], []
        self.patchFaceCells = [[np.asarray(fc, np.int32) for fc in patchFaceCells]]
        self.patchRestrict = []                     # per level, per patch: fine patch face -> coarse patch face
        w = np.asarray(faceWeights, float)
        while len(self.restrictMap) < maxLevels - 1:
            nFine = self.nCells[-1]
            cmap, nCoarse = gamg.pair_agglomerate(nFine, self.l[-1], self.u[-1], w, forward)
            forward = not forward
            tot = comm.allreduce(np.array([float(nCoarse), float(nFine)]))
            if tot[0] < nRanks * nCellsInCoarsestLevel or not tot[0] < tot[1]:
                break
            cl, cu, fra, flip = gamg.agglomerate_addressing(self.l[-1], self.u[-1], cmap, nCoarse)
            # interfaces: the neighbour's restrict addressing of the patch-face cells, then the coarse faces by first appearance
            mine = [cmap[fc].astype(np.float64) for fc in self.patchFaceCells[-1]]
            theirs = comm.exchange(mine)
            cfc, prs = [], []
            for loc, rem in zip(mine, theirs):
                seen, cells, pr = {}, [], np.empty(len(loc), np.int64)
                for i, pair in enumerate(zip(loc.astype(np.int64), rem.astype(np.int64))):
                    k = seen.get(pair)
                    if k is None:
                        k = seen[pair] = len(cells); cells.append(pair[0])
                    pr[i] = k
                cfc.append(np.asarray(cells, np.int32)); prs.append(pr)
            self.restrictMap.append(cmap); self.faceRestrict.append(fra); self.faceFlip.append(flip)
            self.nCells.append(nCoarse); self.l.append(cl); self.u.append(cu)
            self.patchFaceCells.append(cfc); self.patchRestrict.append(prs)
            cw = np.zeros(len(cl))
            for f in range(len(fra)):
                if fra[f] >= 0:
                    cw[fra[f]] += w[f]
            w = cw
        self.forward_after = forward
        self.nLevels = len(self.restrictMap)

    restrict = gamg.Agglomeration.restrict
    prolong = gamg.Agglomeration.prolong


class GAMGSolverMulti(gamg.GAMGSolver):
    def __init__(self, agg, diag, upper, lower, patchBou, patchInt, comm, globalCells, smoother="GaussSeidel", **kw):
        """patchBou[q] / patchInt[q]: interfaceBouCoeffs / interfaceIntCoeffs of the finest level (None: symmetric, = bou)"""
        self.comm, self.globalCells = comm, globalCells
        self.agg, self.smoother = agg, smoother
        self.nPre, self.nPost, self.nFinest = kw.get("nPreSweeps", 0), kw.get("nPostSweeps", 2), kw.get("nFinestSweeps", 2)
        self.preMul, self.maxPre, self.postMul, self.maxPost = 1, 4, 1, 4
        self.symmetric = lower is None
        self.scaleCorrection = self.symmetric
        self.coef = [(np.asarray(diag, float), np.asarray(upper, float), None if lower is None else np.asarray(lower, float))]
        bou = [[np.asarray(b, float) for b in patchBou]]
        inn = [None if patchInt is None else [np.asarray(b, float) for b in patchInt]]
        for lev in range(agg.nLevels):
            self.coef.append(gamg.agglomerate_matrix(agg, lev, *self.coef[-1]))
            cb, ci = [], []
            for q, pr in enumerate(agg.patchRestrict[lev]):
                n = len(agg.patchFaceCells[lev + 1][q])
                b = np.zeros(n); i_ = np.zeros(n)
                for f in range(len(pr)):                      # GAMGInterface::agglomerateCoeffs: sum over the fine faces, in their order
                    b[pr[f]] += bou[-1][q][f]
                    if inn[-1] is not None:
                        i_[pr[f]] += inn[-1][q][f]
                cb.append(b); ci.append(i_)
            bou.append(cb); inn.append(None if inn[-1] is None else ci)
        self.A = []
        nGlob = comm.allreduce(np.array([float(n) for n in agg.nCells]))
        for lev, (d, up, lo) in enumerate(self.coef):
            A = O.Ldu(agg.nCells[lev], agg.l[lev], agg.u[lev]).set_coeffs(d, up, lo)
            A.set_interfaces(agg.patchFaceCells[lev], bou[lev], inn[lev])
            A.set_global_cells(int(nGlob[lev]))
            A.comm = comm.ffo
            self.A.append(A)
        self.rD = [None] * len(self.A)
        if smoother in ("DILU", "DIC"):
            self.rD = [a.dic_rD() if smoother == "DIC" else a.dilu_rD() for a in self.A]

    def scale(self, k, field, source):
        A = self.A[k]
        Acf = A.amul(field)
        num = den = 0.0
        for i in range(len(field)):
            num += source[i] * field[i]; den += Acf[i] * field[i]
        num, den = self.comm.allreduce(np.array([num, den]))
        sf = num / (den if abs(den) >= gamg.VSMALL else (gamg.VSMALL if den >= 0 else -gamg.VSMALL))
        return sf * field + (source - sf * Acf) / self.coef[k][0]

    def solve(self, psi, source, tolerance=1e-6, relTol=0.0, minIter=0, maxIter=1000):
        A = self.A[0]
        psi = np.asarray(psi, float).copy(); source = np.asarray(source, float)
        self.coarsest_log = []
        normFactor = A.norm_factor(psi, source)                    # global (the C oracle reduces over the communicator)
        res = source - A.amul(psi)
        gsum = lambda v: float(self.comm.allreduce(np.array([np.abs(v).sum()]))[0])
        perf = dict(initialResidual=gsum(res) / normFactor, nIterations=0)
        perf["finalResidual"] = perf["initialResidual"]

        def converged():
            return perf["finalResidual"] < tolerance or (relTol > 1e-20 and perf["finalResidual"] < relTol * perf["initialResidual"])
        if minIter > 0 or not converged():
            while True:
                psi = self.vcycle(psi, source, res, tolerance, relTol)
                res = source - A.amul(psi)
                perf["finalResidual"] = gsum(res) / normFactor
                perf["nIterations"] += 1
                if not ((perf["nIterations"] < maxIter and not converged()) or perf["nIterations"] < minIter):
                    break
        perf["converged"] = bool(converged())
        return psi, perf
