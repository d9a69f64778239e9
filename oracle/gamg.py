"""ORACLE (test infrastructure): OpenFOAM-dev's GAMG solver as the reference's cases select it --
    p_rgh / ph_rgh:  solver GAMG; smoother GaussSeidel; cacheAgglomeration true; nCellsInCoarsestLevel 10; agglomerator
                     faceAreaPair; mergeLevels 1            (cases/wallFireSpread2D/system/fvSolution:36-60, cases/pyrolysis1D)
    Ii:              solver GAMG; smoother DILU; the same agglomeration   (cases/steckler/system/fvSolution:63-73)
The algorithm lives in OpenFOAM-dev (library `OpenFOAM`, src/OpenFOAM/matrices/lduMatrix/solvers/GAMG; the reference's
golden log names build dev-tmp-0c4175bec707, 2017-08), which is NOT part of the reference tree: what follows restates, from
the published sources as this builder knows them,
    faceAreaPairGAMGAgglomeration       face weights |Sf/sqrt|Sf| * (1, 1.01, 1.02)|
    pairGAMGAgglomeration::agglomerate  greedy pairing by the largest face weight, cell order reversed from level to level
                                        (static forward_), unmatched cells join the best neighbouring cluster
    GAMGAgglomeration::continueAgglomerating, agglomerateLduAddressing (coarse faces in order of discovery per coarse owner,
                                        faceFlipMap), restrictField / restrictFaceField / prolongField
    GAMGSolver::agglomerateMatrix       coarse diag = sum of fine diags + interior faces (upper + lower), coarse upper / lower
    GAMGSolver::solve / Vcycle / scale / solveCoarsestLevel   nPreSweeps 0, nPostSweeps 2 (+1 per level, max 4), nFinestSweeps 2,
                                        scaleCorrection for symmetric matrices, interpolateCorrection off, coarsest level by
                                        PCG+DIC (symmetric) or PBiCGStab+DILU to the solver's own tolerance / relTol
    GaussSeidelSmoother, DILUSmoother, DICSmoother
PARITY UNPINNED: the reference holds no output of a GAMG solve made with its current dictionaries that this path could
reproduce without the full fvDOM boundary conditions (cases/steckler/original/linux64/log.fireFoam:183-214 are the ILambda
solves), so the tests check the multigrid identities (Galerkin coarse operators, the restriction / prolongation pair,
convergence to the PCG solution) and the HIP path against this restatement.  Serial only.  Only tests/ may import this."""
import numpy as np

from . import oracle as O

GREAT = 1e15
VSMALL = 1e-300


def face_area_pair_weights(Sf):
    """faceAreaPairGAMGAgglomeration: mag(cmptMultiply(Sf/sqrt(mag(Sf)), (1, 1.01, 1.02)))"""
    Sf = np.asarray(Sf, float)
    m = np.sqrt(np.sqrt((Sf * Sf).sum(axis=1)))
    v = Sf / m[:, None] * np.array([1.0, 1.01, 1.02])
    return np.sqrt((v * v).sum(axis=1))


def pair_agglomerate(nFine, l, u, w, forward):
    """pairGAMGAgglomeration::agglomerate(nCoarseCells, addressing, faceWeights) -> (coarseCellMap, nCoarse)"""
    nF = len(l)
    nNbrs = np.zeros(nFine, np.int64)
    np.add.at(nNbrs, u, 1); np.add.at(nNbrs, l, 1)
    off = np.concatenate(([0], np.cumsum(nNbrs)))
    cellFaces = np.empty(2 * nF, np.int64)
    cnt = np.zeros(nFine, np.int64)
    for f in range(nF):
        c = u[f]; cellFaces[off[c] + cnt[c]] = f; cnt[c] += 1
    for f in range(nF):
        c = l[f]; cellFaces[off[c] + cnt[c]] = f; cnt[c] += 1
    cmap = np.full(nFine, -1, np.int64)
    nCoarse = 0
    order = range(nFine) if forward else range(nFine - 1, -1, -1)
    for c in order:
        if cmap[c] >= 0:
            continue
        match, best = -1, -GREAT
        for f in cellFaces[off[c]:off[c + 1]]:
            if cmap[u[f]] < 0 and cmap[l[f]] < 0 and w[f] > best:
                match, best = f, w[f]
        if match >= 0:
            cmap[u[match]] = nCoarse; cmap[l[match]] = nCoarse
            nCoarse += 1
        else:
            match, best = -1, -GREAT
            for f in cellFaces[off[c]:off[c + 1]]:
                if w[f] > best:
                    match, best = f, w[f]
            if match >= 0:
                cmap[c] = max(cmap[u[match]], cmap[l[match]])
    for c in order:
        if cmap[c] < 0:
            cmap[c] = nCoarse; nCoarse += 1
    if not forward:
        cmap = (nCoarse - 1) - cmap
    return cmap, nCoarse


def agglomerate_addressing(l, u, rmap, nCoarse):
    """GAMGAgglomeration::agglomerateLduAddressing -> coarse (l, u), faceRestrictAddr, faceFlipMap"""
    nF = len(l)
    fra = np.empty(nF, np.int64)
    cellFaces = [[] for _ in range(nCoarse)]       # per coarse owner: provisional coarse faces in order of discovery
    initNbr = []
    for f in range(nF):
        ru, rl = rmap[u[f]], rmap[l[f]]
        if ru == rl:
            fra[f] = -(ru + 1)
            continue
        own, nei = (rl, ru) if ru > rl else (ru, rl)
        for cf in cellFaces[own]:
            if initNbr[cf] == nei:
                fra[f] = cf
                break
        else:
            cellFaces[own].append(len(initNbr)); fra[f] = len(initNbr); initNbr.append(nei)
    nCF = len(initNbr)
    cl = np.empty(nCF, np.int32); cu = np.empty(nCF, np.int32); fmap = np.empty(nCF, np.int64)
    k = 0
    for cc in range(nCoarse):
        for cf in cellFaces[cc]:
            cl[k] = cc; cu[k] = initNbr[cf]; fmap[cf] = k; k += 1
    flip = np.zeros(nF, bool)
    for f in range(nF):
        if fra[f] >= 0:
            fra[f] = fmap[fra[f]]
            flip[f] = rmap[u[f]] < rmap[l[f]]
    return cl, cu, fra, flip


class Agglomeration:
    """pairGAMGAgglomeration::agglomerate(mesh, faceWeights): the level hierarchy (cacheAgglomeration: built once per mesh)"""

    def __init__(self, nCells, l, u, faceWeights, nCellsInCoarsestLevel=10, mergeLevels=1, maxLevels=50, forward=True):
        assert mergeLevels == 1, "mergeLevels 1 (both reference dictionaries)"
        self.nCells = [int(nCells)]
        self.l, self.u = [np.asarray(l, np.int32)], [np.asarray(u, np.int32)]
        self.restrictMap, self.faceRestrict, self.faceFlip = [], [], []
        w = np.asarray(faceWeights, float)
        while len(self.restrictMap) < maxLevels - 1:
            nFine = self.nCells[-1]
            cmap, nCoarse = pair_agglomerate(nFine, self.l[-1], self.u[-1], w, forward)
            forward = not forward
            if nCoarse < nCellsInCoarsestLevel or not nCoarse < nFine:        # continueAgglomerating (one process)
                break
            cl, cu, fra, flip = agglomerate_addressing(self.l[-1], self.u[-1], cmap, nCoarse)
            self.restrictMap.append(cmap); self.faceRestrict.append(fra); self.faceFlip.append(flip)
            self.nCells.append(nCoarse); self.l.append(cl); self.u.append(cu)
            cw = np.zeros(len(cl))                                             # restrictFaceField
            for f in range(len(fra)):
                if fra[f] >= 0:
                    cw[fra[f]] += w[f]
            w = cw
        self.forward_after = forward
        self.nLevels = len(self.restrictMap)        # number of coarse levels

    def restrict(self, lev, ff):
        cf = np.zeros(self.nCells[lev + 1])
        rm = self.restrictMap[lev]
        for i in range(len(ff)):                    # restrictField: sequential sum in fine-cell order
            cf[rm[i]] += ff[i]
        return cf

    def prolong(self, lev, cf):
        return cf[self.restrictMap[lev]]


def agglomerate_matrix(agg, lev, diag, upper, lower):
    """GAMGSolver::agglomerateMatrix: fine level `lev` (0 = the solver's matrix) -> coarse (diag, upper, lower or None)"""
    cd = agg.restrict(lev, diag)
    fra, flip = agg.faceRestrict[lev], agg.faceFlip[lev]
    nCF = len(agg.l[lev + 1])
    cu = np.zeros(nCF)
    if lower is not None:
        clo = np.zeros(nCF)
        for f in range(len(fra)):
            cf = fra[f]
            if cf >= 0:
                if not flip[f]:
                    cu[cf] += upper[f]; clo[cf] += lower[f]
                else:
                    cu[cf] += lower[f]; clo[cf] += upper[f]
            else:
                cd[-1 - cf] += upper[f] + lower[f]
        return cd, cu, clo
    for f in range(len(fra)):
        cf = fra[f]
        if cf >= 0:
            cu[cf] += upper[f]
        else:
            cd[-1 - cf] += 2 * upper[f]
    return cd, cu, None


class GAMGSolver:
    def __init__(self, agg, diag, upper, lower=None, smoother="GaussSeidel", nPreSweeps=0, nPostSweeps=2, nFinestSweeps=2,
                 preSweepsLevelMultiplier=1, maxPreSweeps=4, postSweepsLevelMultiplier=1, maxPostSweeps=4):
        self.agg, self.smoother = agg, smoother
        self.nPre, self.nPost, self.nFinest = nPreSweeps, nPostSweeps, nFinestSweeps
        self.preMul, self.maxPre, self.postMul, self.maxPost = preSweepsLevelMultiplier, maxPreSweeps, postSweepsLevelMultiplier, maxPostSweeps
        self.symmetric = lower is None
        self.scaleCorrection = self.symmetric
        self.coef = [(np.asarray(diag, float), np.asarray(upper, float), None if lower is None else np.asarray(lower, float))]
        for lev in range(agg.nLevels):
            self.coef.append(agglomerate_matrix(agg, lev, *self.coef[-1]))
        self.A = []
        for lev, (d, up, lo) in enumerate(self.coef):
            self.A.append(O.Ldu(agg.nCells[lev], agg.l[lev], agg.u[lev]).set_coeffs(d, up, lo))
        self.rD = [None] * len(self.A)
        if smoother in ("DILU", "DIC"):
            self.rD = [a.dic_rD() if smoother == "DIC" else a.dilu_rD() for a in self.A]

    def smooth(self, k, psi, b, nSweeps):
        """smoothers[k]: k = 0 the finest matrix, k = lev + 1 the coarse level lev"""
        A = self.A[k]
        if self.smoother == "GaussSeidel":
            return A.gs_smooth(psi, b, nSweeps, sym=False)
        if self.smoother == "symGaussSeidel":
            return A.gs_smooth(psi, b, nSweeps, sym=True)
        psi = psi.copy()
        for _ in range(nSweeps):            # DILUSmoother / DICSmoother: psi += M^-1 (b - A psi)
            rA = A.residual(psi, b)
            psi = psi + (A.dic_precondition(self.rD[k], rA) if self.smoother == "DIC" else A.dilu_precondition(self.rD[k], rA))
        return psi

    def scale(self, k, field, source):
        """GAMGSolver::scale on matrix k"""
        A = self.A[k]
        Acf = A.amul(field)
        num = den = 0.0
        for i in range(len(field)):
            num += source[i] * field[i]; den += Acf[i] * field[i]
        sf = num / (den if abs(den) >= VSMALL else (VSMALL if den >= 0 else -VSMALL))
        return sf * field + (source - sf * Acf) / self.coef[k][0]

    def solve_coarsest(self, source, tolerance, relTol):
        A = self.A[-1]
        if self.symmetric:
            x, pf = A.solve(O.PCG, O.DIC, np.zeros(len(source)), source, tolerance=tolerance, relTol=relTol)
        else:
            x, pf = A.solve(O.PBICGSTAB, O.DILU, np.zeros(len(source)), source, tolerance=tolerance, relTol=relTol)
        self.coarsest_log.append(pf)
        return x

    def vcycle(self, psi, source, finestResidual, tolerance, relTol):
        agg = self.agg
        nC = agg.nLevels                        # coarse levels 0 .. nC-1 live on matrices 1 .. nC
        coarsest = nC - 1
        src = [None] * nC; corr = [None] * nC
        src[0] = agg.restrict(0, finestResidual)
        for lev in range(coarsest):
            if self.nPre:
                corr[lev] = self.smooth(lev + 1, np.zeros_like(src[lev]), src[lev], min(self.nPre + self.preMul * lev, self.maxPre))
                if self.scaleCorrection and lev < coarsest - 1:
                    corr[lev] = self.scale(lev + 1, corr[lev], src[lev])
                src[lev] = src[lev] - self.A[lev + 1].amul(corr[lev])
            src[lev + 1] = agg.restrict(lev + 1, src[lev])
        corr[coarsest] = self.solve_coarsest(src[coarsest], tolerance, relTol)
        for lev in range(coarsest - 1, -1, -1):
            pre = corr[lev] if self.nPre else None
            corr[lev] = agg.prolong(lev + 1, corr[lev + 1])
            if self.scaleCorrection and lev < coarsest - 1:
                corr[lev] = self.scale(lev + 1, corr[lev], src[lev])
            if self.nPre:
                corr[lev] = corr[lev] + pre
            corr[lev] = self.smooth(lev + 1, corr[lev], src[lev], min(self.nPost + self.postMul * lev, self.maxPost))
        fc = agg.prolong(0, corr[0])
        if self.scaleCorrection:
            fc = self.scale(0, fc, finestResidual)
        psi = psi + fc
        return self.smooth(0, psi, source, self.nFinest)

    def solve(self, psi, source, tolerance=1e-6, relTol=0.0, minIter=0, maxIter=1000):
        A = self.A[0]
        psi = np.asarray(psi, float).copy(); source = np.asarray(source, float)
        self.coarsest_log = []
        Apsi = A.amul(psi)
        normFactor = A.norm_factor(psi, source)
        res = source - Apsi
        perf = dict(initialResidual=float(np.abs(res).sum() / normFactor), nIterations=0)
        perf["finalResidual"] = perf["initialResidual"]

        def converged():
            return perf["finalResidual"] < tolerance or (relTol > 1e-20 and perf["finalResidual"] < relTol * perf["initialResidual"])

        if minIter > 0 or not converged():
            if self.agg.nLevels == 0:
                raise ValueError("no coarse level: the mesh is below nCellsInCoarsestLevel")
            while True:
                psi = self.vcycle(psi, source, res, tolerance, relTol)
                res = source - A.amul(psi)
                perf["finalResidual"] = float(np.abs(res).sum() / normFactor)
                perf["nIterations"] += 1
                if not ((perf["nIterations"] < maxIter and not converged()) or perf["nIterations"] < minIter):
                    break
        perf["converged"] = bool(converged())
        return psi, perf
