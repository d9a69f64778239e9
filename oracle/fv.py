"""ORACLE (test infrastructure): finite-volume operators on hex-box meshes, numpy restatement of
OpenFOAM-dev @940e28f (not vendored in /root/reference):
  src/finiteVolume/finiteVolume/fvc/{fvcSurfaceIntegrate,fvcSnGrad,fvcFlux,fvcReconstruct}.C
  src/finiteVolume/finiteVolume/laplacianSchemes/gaussLaplacianScheme/gaussLaplacianScheme.C
  src/finiteVolume/finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionScheme.C
  src/finiteVolume/finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtScheme.C
  src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C (addBoundaryDiag/Source, solveSegregated)
  src/finiteVolume/cfdTools/general/constrainPressure/constrainPressure.C
  src/finiteVolume/fields/fvPatchFields/basic/{fixedValue,fixedGradient,zeroGradient,mixed}/*.C
selected by the reference in cases/steckler/system/fvSchemes:18-76 and called from
solver/phrghEqn.H:25-56, solver/pEqn.H:3-44, solver/UEqn.H:3-30, solver/YEEqn.H:43-111.
Only tests/ may import this module.
"""
import numpy as np


class Patch:
    def __init__(self, name, faceCells, Sf, Cf, delta):
        self.name = name
        self.faceCells = np.asarray(faceCells, np.int64)
        self.Sf = np.asarray(Sf, np.float64).reshape(-1, 3)
        self.Cf = np.asarray(Cf, np.float64).reshape(-1, 3)
        self.magSf = np.linalg.norm(self.Sf, axis=1)
        self.deltaCoeffs = np.asarray(delta, np.float64)       # 1/|d| , d = Cf - C[faceCell]
        self.size = len(self.faceCells)


class HexMesh:
    """blockMesh single block (nx,ny,nz) on [lo,hi], natural numbering.  `baffle`
    (bool per natural internal face) turns internal faces into a wall pair, as createBaffles does
    (reference cases/steckler/system/createBafflesDict:11-58): the face leaves the internal list
    (remaining faces keep upper-triangular order) and appears once in patch `<name>_master`
    (owner-side cell) and once in `<name>_slave` (neighbour-side cell)."""

    def __init__(self, n, lo, hi, baffle=None, baffle_name="baffle"):
        nx, ny, nz = n
        self.n = n
        lo = np.asarray(lo, float); hi = np.asarray(hi, float)
        d = (hi - lo) / np.array(n, float)
        self.d = d
        N = nx * ny * nz
        self.nCells = N
        k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        i, j, k = i.ravel(), j.ravel(), k.ravel()
        self.ijk = (i, j, k)
        self.C = np.stack([lo[0] + (i + 0.5) * d[0], lo[1] + (j + 0.5) * d[1], lo[2] + (k + 0.5) * d[2]], axis=1)
        self.V = np.full(N, d[0] * d[1] * d[2])
        c = np.arange(N)
        hx, hy, hz = i < nx - 1, j < ny - 1, k < nz - 1
        cnt = hx.astype(int) + hy + hz
        start = np.concatenate(([0], np.cumsum(cnt)))
        F = start[-1]
        l = np.empty(F, np.int64); u = np.empty(F, np.int64); fd = np.empty(F, np.int64)
        px = start[:-1][hx]; l[px] = c[hx]; u[px] = c[hx] + 1; fd[px] = 0
        py = (start[:-1] + hx)[hy]; l[py] = c[hy]; u[py] = c[hy] + nx; fd[py] = 1
        pz = (start[:-1] + hx + hy)[hz]; l[pz] = c[hz]; u[pz] = c[hz] + nx * ny; fd[pz] = 2
        area = np.array([d[1] * d[2], d[0] * d[2], d[0] * d[1]])
        Sf = np.zeros((F, 3)); Sf[np.arange(F), fd] = area[fd]
        Cf = self.C[l].copy(); Cf[np.arange(F), fd] += 0.5 * d[fd]
        self.patches = []
        keep = np.ones(F, bool)
        if baffle is not None:
            baffle = np.asarray(baffle, bool)
            keep = ~baffle
        self.natural_face = np.nonzero(keep)[0]
        self.l, self.u, self.fdir = l[keep], u[keep], fd[keep]
        self.Sf, self.Cf = Sf[keep], Cf[keep]
        self.magSf = np.linalg.norm(self.Sf, axis=1)
        self.nFaces = len(self.l)
        self.weights = np.full(self.nFaces, 0.5)
        self.deltaCoeffs = 1.0 / d[self.fdir]                 # = nonOrthDeltaCoeffs on a hex box
        # boundary patches of the block: (name, axis, side)
        self._bdefs = {}
        for name, ax, side in (("xmin", 0, 0), ("xmax", 0, 1), ("ymin", 1, 0), ("ymax", 1, 1), ("zmin", 2, 0), ("zmax", 2, 1)):
            idx = (i, j, k)[ax]
            sel = c[idx == (0 if side == 0 else n[ax] - 1)]
            S = np.zeros((len(sel), 3)); S[:, ax] = area[ax] * (1 if side else -1)
            Cb = self.C[sel].copy(); Cb[:, ax] += (0.5 if side else -0.5) * d[ax]
            self._bdefs[name] = Patch(name, sel, S, Cb, np.full(len(sel), 2.0 / d[ax]))
        if baffle is not None and baffle.any():
            bl, bu, bd = l[baffle], u[baffle], fd[baffle]
            S = Sf[baffle]; Cb = Cf[baffle]
            self._bdefs[baffle_name + "_master"] = Patch(baffle_name + "_master", bl, S, Cb, 2.0 / d[bd])
            self._bdefs[baffle_name + "_slave"] = Patch(baffle_name + "_slave", bu, -S, Cb, 2.0 / d[bd])

    def set_patches(self, spec):
        """spec: list of (patchName, [block-side or baffle names]) in OpenFOAM patch order."""
        self.patches = []
        for name, parts in spec:
            ps = [self._bdefs[p] for p in parts]
            self.patches.append(Patch(name, np.concatenate([p.faceCells for p in ps]), np.concatenate([p.Sf for p in ps]),
                                      np.concatenate([p.Cf for p in ps]), np.concatenate([p.deltaCoeffs for p in ps])))
        return self

    def patch(self, name):
        return next(p for p in self.patches if p.name == name)


# ------------------------------------------------------------------ fvc operators ---
def interpolate(mesh, vf, bvals):
    """fvc::interpolate, linear: face = w*P + (1-w)*N; boundary faces take the patch value."""
    w = mesh.weights
    return w * vf[mesh.l] + (1.0 - w) * vf[mesh.u], [np.asarray(b, float) for b in bvals]


def snGrad(mesh, vf, bvals):
    """fvc::snGrad, uncorrected: deltaCoeffs*(N - P); boundary: deltaCoeffs_b*(value_b - cell)."""
    internal = mesh.deltaCoeffs * (vf[mesh.u] - vf[mesh.l])
    return internal, [p.deltaCoeffs * (np.asarray(b, float) - vf[p.faceCells]) for p, b in zip(mesh.patches, bvals)]


def surface_integrate(mesh, ssf, bssf):
    """fvc::surfaceIntegrate == fvc::div(ssf): owner += , neighbour -= in face order, then the
    boundary faces patch by patch, then divide by V."""
    out = np.zeros(mesh.nCells)
    np.add.at(out, mesh.l, ssf)
    np.subtract.at(out, mesh.u, ssf)
    for p, b in zip(mesh.patches, bssf):
        np.add.at(out, p.faceCells, b)
    return out / mesh.V


class ScalarMatrix:
    """fvScalarMatrix on a HexMesh: A psi = source, with per-patch internalCoeffs/boundaryCoeffs."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.diag = np.zeros(mesh.nCells)
        self.upper = np.zeros(mesh.nFaces)
        self.lower = None
        self.source = np.zeros(mesh.nCells)
        self.internalCoeffs = [np.zeros(p.size) for p in mesh.patches]
        self.boundaryCoeffs = [np.zeros(p.size) for p in mesh.patches]

    def neg_sum_diag(self):
        lo = self.upper if self.lower is None else self.lower
        np.subtract.at(self.diag, self.mesh.l, lo)
        np.subtract.at(self.diag, self.mesh.u, self.upper)

    def solve_ready(self):
        """fvMatrix::solveSegregated preamble: diag + addBoundaryDiag, source + addBoundarySource."""
        d = self.diag.copy(); s = self.source.copy()
        for p, ic, bc in zip(self.mesh.patches, self.internalCoeffs, self.boundaryCoeffs):
            np.add.at(d, p.faceCells, ic)
            np.add.at(s, p.faceCells, bc)
        return d, s


def laplacian(mesh, gamma_f, gamma_b, bcs):
    """fvm::laplacian(gamma, vf), Gauss linear uncorrected (cases/steckler/system/fvSchemes:63-66).
    bcs[patch] = ('fixedValue', value) | ('fixedGradient', gradient) | ('zeroGradient',)."""
    M = ScalarMatrix(mesh)
    M.upper = gamma_f * mesh.magSf * mesh.deltaCoeffs
    M.neg_sum_diag()
    for q, p in enumerate(mesh.patches):
        pGamma = gamma_b[q] * p.magSf
        bc = bcs[q]
        if bc[0] == "fixedValue":
            gi = -p.deltaCoeffs * np.ones(p.size); gb = p.deltaCoeffs * bc[1]
        elif bc[0] == "fixedGradient":
            gi = np.zeros(p.size); gb = np.asarray(bc[1], float) * np.ones(p.size)
        else:
            gi = np.zeros(p.size); gb = np.zeros(p.size)
        M.internalCoeffs[q] = pGamma * gi
        M.boundaryCoeffs[q] = -pGamma * gb
    return M
