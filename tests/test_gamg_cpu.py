"""The GAMG oracle (oracle/gamg.py): multigrid identities and convergence on a hexahedral box.  PARITY UNPINNED by reference
data (see the oracle's header): these tests check what the algorithm must satisfy whatever the details."""
import numpy as np
import pytest

from ffm_import import ffm


def _box(n, h=0.05):
    H = ffm.hexmesh
    blk = H.HexBlock(n)
    s = H.synth_p_rgh(blk, h=h)
    d = blk.u.astype(np.int64) - blk.l
    axis = np.where(d == 1, 0, np.where(d == n[0], 1, 2))                # blockMesh numbering: x fastest
    Sf = np.zeros((len(blk.l), 3))
    Sf[np.arange(len(blk.l)), axis] = h * h * np.array([1.0, 1.3, 0.7])[axis]       # an anisotropic box: unequal weights
    return blk, s, Sf


def test_pair_agglomeration_and_galerkin_coarse_operators(O):
    from oracle import gamg
    blk, s, Sf = _box((12, 10, 8))
    w = gamg.face_area_pair_weights(Sf)
    agg = gamg.Agglomeration(blk.nCells, blk.l, blk.u, w)
    # pairs: every level at most halves the cell count (a few single cells join clusters), stops above nCellsInCoarsestLevel
    assert agg.nLevels >= 5 and agg.nCells[-1] >= 10
    for a, b in zip(agg.nCells[:-1], agg.nCells[1:]):
        assert a / 3.0 <= b <= (a + 1) // 2 + a // 8
    lo = s["upper"] * (1.0 + 0.3 * (ffm.hexmesh.hash_u(0xA1, blk.gface) - 0.5))
    for lower in (None, lo):
        G = gamg.GAMGSolver(agg, s["diag"], s["upper"], lower, smoother="GaussSeidel")
        for lev in range(agg.nLevels):
            # coarse addressing: owner-sorted upper-triangular; every coarse face collects >= 1 fine face
            cl, cu = agg.l[lev + 1], agg.u[lev + 1]
            assert np.all(cl < cu) and np.all(np.diff(cl) >= 0)
            assert set(agg.faceRestrict[lev][agg.faceRestrict[lev] >= 0]) == set(range(len(cl)))
            # Galerkin: A_c x_c = R A_f P x_c
            xc = ffm.hexmesh.hash_u(0x77 + lev, np.arange(agg.nCells[lev + 1]))
            lhs = G.A[lev + 1].amul(xc)
            rhs = agg.restrict(lev, G.A[lev].amul(agg.prolong(lev, xc)))
            assert np.abs(lhs - rhs).max() <= 1e-12 * np.abs(lhs).max()


@pytest.mark.parametrize("smoother,asym", [("GaussSeidel", False), ("DIC", False), ("DILU", True), ("GaussSeidel", True)])
def test_gamg_converges_to_the_krylov_solution(O, smoother, asym):
    from oracle import gamg
    blk, s, Sf = _box((14, 12, 10))
    agg = gamg.Agglomeration(blk.nCells, blk.l, blk.u, gamg.face_area_pair_weights(Sf))
    lo = s["upper"] * (1.0 + 0.3 * (ffm.hexmesh.hash_u(0xA1, blk.gface) - 0.5)) if asym else None
    G = gamg.GAMGSolver(agg, s["diag"], s["upper"], lo, smoother=smoother)
    psi, pf = G.solve(np.zeros(blk.nCells), s["source"], tolerance=1e-10)
    assert pf["converged"] and 2 <= pf["nIterations"] <= 40
    A = O.Ldu(blk.nCells, blk.l, blk.u).set_coeffs(s["diag"], s["upper"], lo)
    ref, pk = A.solve(O.PBICGSTAB if asym else O.PCG, O.DILU if asym else O.DIC, np.zeros(blk.nCells), s["source"], tolerance=1e-13)
    assert np.linalg.norm(psi - ref) <= 1e-7 * np.linalg.norm(ref)
    # the coarsest level was solved once per V-cycle, on >= nCellsInCoarsestLevel cells
    assert len(G.coarsest_log) == pf["nIterations"] and agg.nCells[-1] >= 10
    # relTol stops earlier
    _, pr = G.solve(np.zeros(blk.nCells), s["source"], tolerance=1e-10, relTol=0.01)
    assert pr["nIterations"] < pf["nIterations"] and pr["finalResidual"] < 0.01 * pr["initialResidual"]
