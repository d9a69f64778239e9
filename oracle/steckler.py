"""ORACLE (test infrastructure): the hydrostatic initialisation of the steckler case
(reference solver/phrghEqn.H:1-62 driven from solver/createFields.H:100-104), restated with the
numpy FV operators of oracle/fv.py and the C solvers of oracle/ffo_solvers.c.  This is the one
place where the reference holds golden numbers for the hot path
(cases/steckler/original/linux64/log.fireFoam:92-101), so it is the oracle's parity pin.

Case data used (all reference file:line):
  mesh 30x15x20 on (-2 0 -2)-(4 3 2) m        cases/steckler/constant/polyMesh/blockMeshDict:40-52
  compartment shell (cell centres in box)     cases/steckler/system/topoSetDictCompartment:441-521
  doorway removed, box (0 0 -.5)-(10 1 .5)    cases/steckler/system/topoSetDictCompartment:505-514
  shell faces -> wall baffle pair             cases/steckler/system/createBafflesDict:11-58
  ph_rgh: top fixedValue 0, rest fixedFluxPressure   cases/steckler/0/ph_rgh.orig:22-65
  g (0 -9.81 0), hRef 3, pRef 101325          cases/steckler/constant/{g,hRef,pRef}
  T 298.15, Y O2 0.23301 / N2 0.76699, W from cases/steckler/constant/thermo.compressibleGas
  PCG + DIC, tolerance 1e-6, relTol 0.01      cases/steckler/system/fvSolution:29-46
  nHydrostaticCorrectors 5                    cases/steckler/system/fvSolution:91-92
  boundary mixture: the inert specie's file gives N2 `calculated; value uniform 0` on top, sides, base, burner and floor
  (cases/steckler/0/N2:24-51), and nothing evaluates that boundary before the hydrostatic initialisation, so the patch-face
  mixture there is O2 alone (0.23301 of it): W_b = W_O2, psi_b = W_O2/(RR T) -- 10.9 % above the interior's.  On the
  baffle patches T = 300 K and O2/N2 = 0.232/0.768 (cases/steckler/0/T:51-82, 0/include/wallBafflePatches).  On
  fixedFluxPressure patches the boundary density cancels between the Laplacian's boundary source and div(phig); on `top`
  (fixedValue) it is the coefficient rhof_b*|Sf|*deltaCoeffs of the top cell layer.
The doorway box touches face centres at z = +-0.5 exactly; those faces are inside the box (boundBox::contains is inclusive).
With both, the five solves reproduce the golden log DIGIT FOR DIGIT (all 8 printed digits of every residual and of every
gMax-gMin, and every iteration count): tests/test_golden_log_cpu.py.
"""
import numpy as np

from . import fv, oracle as O

RR = 6.0221417930e26 * 1.38065e-23      # OpenFOAM-dev etc/controlDict SI constants: NA*k = 8314.47 J/(kmol K)


def build_mesh(door_k=(7, 12), refine=1):
    """refine = r: every cell of the 30 x 15 x 20 base mesh split r x r x r (the `refineMesh` step of the reference's larger
    runs; BASELINE config 2 = r 4 = 576 000 cells); room, baffles and doorway keep their place."""
    r = int(refine)
    n = (30 * r, 15 * r, 20 * r)
    base = fv.HexMesh(n, (-2, 0, -2), (4, 3, 2))
    i, j, k = base.ijk
    inside = (i >= 3 * r) & (i <= 17 * r - 1) & (j <= 11 * r - 1) & (k >= 3 * r) & (k <= 17 * r - 1)
    l, u, fd = base.l, base.u, base.fdir
    baffle = inside[l] != inside[u]
    door = (fd == 0) & (i[l] == 17 * r - 1) & (j[l] <= 5 * r - 1) & (k[l] >= door_k[0] * r) & (k[l] <= (door_k[1] + 1) * r - 1) & baffle
    m = fv.HexMesh(n, (-2, 0, -2), (4, 3, 2), baffle=baffle & ~door, baffle_name="baffle1DWall")
    m.set_patches([("top", ["ymax"]), ("sides", ["zmax", "zmin", "xmax", "xmin"]), ("base", ["ymin"]),
                   ("baffle1DWall_master", ["baffle1DWall_master"]), ("baffle1DWall_slave", ["baffle1DWall_slave"])])
    return m


def hydrostatic_initialisation(solve, nCorr=5, mesh=None):
    """solve(mesh, diag, upper, source, psi0) -> (psi, perf dict).  Returns the per-corrector
    records and the final ph_rgh."""
    m = mesh or build_mesh()
    g = np.array([0.0, -9.81, 0.0]); hRef = 3.0; pRef = 101325.0
    ghRef = -np.linalg.norm(g) * hRef
    gh = m.C @ g - ghRef
    ghf = m.Cf @ g - ghRef
    ghb = [p.Cf @ g - ghRef for p in m.patches]
    W = 1.0 / (0.23301 / 31.9988 + 0.76699 / 28.0134)
    psi = 1.0 / ((RR / W) * 298.15)
    # patch-face mixtures (see the header): O2 alone where the N2 file leaves the boundary value 0, the wall values on the baffles
    Wbaffle = 1.0 / (0.232 / 31.9988 + 0.768 / 28.0134)
    psib = [1.0 / ((RR / Wbaffle) * 300.0) if pp.name.startswith("baffle") else 1.0 / ((RR / 31.9988) * 298.15) for pp in m.patches]
    rho = np.full(m.nCells, psi * 101325.0)
    rhob = [np.full(p.size, q * 101325.0) for p, q in zip(m.patches, psib)]
    ph = np.zeros(m.nCells); phb = [np.zeros(p.size) for p in m.patches]
    p = ph + rho * gh + pRef; pb = [a + r * gg + pRef for a, r, gg in zip(phb, rhob, ghb)]
    rho = psi * p; rhob = [q * x for q, x in zip(psib, pb)]
    recs = []
    for _ in range(nCorr):
        rhof, rhofb = fv.interpolate(m, rho, rhob)
        sg, sgb = fv.snGrad(m, rho, rhob)
        phig = -rhof * ghf * sg * m.magSf
        phigb = [-rf * gf * s * pp.magSf for rf, gf, s, pp in zip(rhofb, ghb, sgb, m.patches)]
        bcs = []
        for q, pp in enumerate(m.patches):     # constrainPressure: U = 0 at t = 0
            bcs.append(("fixedValue", 0.0) if pp.name == "top" else ("fixedGradient", phigb[q] / (pp.magSf * rhofb[q])))
        M = fv.laplacian(m, rhof, rhofb, bcs)
        M.source += m.V * fv.surface_integrate(m, phig, phigb)
        d, s = M.solve_ready()
        ph, perf = solve(m, d, M.upper, s, ph)
        phb = [np.zeros(pp.size) if bcs[q][0] == "fixedValue" else ph[pp.faceCells] + bcs[q][1] / pp.deltaCoeffs
               for q, pp in enumerate(m.patches)]
        p = ph + rho * gh + pRef; pb = [a + r * gg + pRef for a, r, gg in zip(phb, rhob, ghb)]
        rho = psi * p; rhob = [q * x for q, x in zip(psib, pb)]
        recs.append(dict(perf, variation=float(ph.max() - ph.min())))
    return recs, ph


def oracle_solve(m, diag, upper, source, psi0):
    return O.Ldu(m.nCells, m.l, m.u).set_coeffs(diag, upper).solve(O.PCG, O.DIC, psi0, source, tolerance=1e-6, relTol=0.01)
