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


def shear(mesh, M):
    """Map the geometry of a HexMesh (patches already set) by the affine map x -> M x: a non-orthogonal test mesh.  Addressing
    is unchanged; centres map with M, area vectors with det(M) M^-T, volumes with det(M); weights, nonOrthDeltaCoeffs and
    nonOrthCorrectionVectors follow surfaceInterpolation::makeWeights / makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors
    (OpenFOAM-dev src/finiteVolume/interpolation/surfaceInterpolation/surfaceInterpolation/surfaceInterpolation.C)."""
    M = np.asarray(M, float); det = np.linalg.det(M); cof = det * np.linalg.inv(M).T
    mesh.C = mesh.C @ M.T; mesh.Cf = mesh.Cf @ M.T; mesh.Sf = mesh.Sf @ cof.T; mesh.V = mesh.V * det
    mesh.magSf = np.linalg.norm(mesh.Sf, axis=1)
    own = np.abs(np.einsum("fd,fd->f", mesh.Sf, mesh.Cf - mesh.C[mesh.l]))
    nei = np.abs(np.einsum("fd,fd->f", mesh.Sf, mesh.C[mesh.u] - mesh.Cf))
    mesh.weights = nei / (own + nei)
    d = mesh.C[mesh.u] - mesh.C[mesh.l]
    nf = mesh.Sf / mesh.magSf[:, None]
    mesh.deltaCoeffs = 1.0 / np.maximum(np.einsum("fd,fd->f", nf, d), 0.05 * np.linalg.norm(d, axis=1))     # nonOrthDeltaCoeffs
    mesh.nonOrthCorrectionVectors = nf - d * mesh.deltaCoeffs[:, None]
    for p in mesh.patches:
        p.Cf = p.Cf @ M.T; p.Sf = p.Sf @ cof.T; p.magSf = np.linalg.norm(p.Sf, axis=1)
        nfb = p.Sf / p.magSf[:, None]
        p.deltaCoeffs = 1.0 / np.einsum("fd,fd->f", nfb, p.Cf - mesh.C[p.faceCells])
    return mesh


def snGrad_correction(mesh, gradvf):
    """correctedSnGrad<Type>::correction(vf) = nonOrthCorrectionVectors & linear-interpolate(grad(vf)) on the internal faces; zero
    on non-coupled boundary faces (OpenFOAM-dev .../snGradSchemes/correctedSnGrad/correctedSnGrad.C).  `corrected` snGrad =
    snGrad(...) + this; `Gauss linear corrected` laplacian = fvm_laplacian(...) with source -= V*div(gamma_f*magSf*this)
    (gaussLaplacianScheme::fvmLaplacian)."""
    w = mesh.weights[:, None]
    gf = w * gradvf[mesh.l] + (1.0 - w) * gradvf[mesh.u]
    c = mesh.nonOrthCorrectionVectors
    return (c[:, 0] * gf[:, 0] + c[:, 1] * gf[:, 1]) + c[:, 2] * gf[:, 2]


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
    # a cell meets its faces in face order: first those where it is the neighbour, then those it owns
    np.subtract.at(out, mesh.u, ssf)
    np.add.at(out, mesh.l, ssf)
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
        np.subtract.at(self.diag, self.mesh.u, self.upper)
        np.subtract.at(self.diag, self.mesh.l, lo)

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


# =====================================================================================
# Everything below: operators for the full outer iteration (UEqn / YEEqn / pEqn / rhoEqn).
# Boundary conditions are carried in OpenFOAM's `mixed` form (mixedFvPatchField.C):
#   value_b = f*refValue + (1-f)*(cell + refGrad/delta)
#   valueInternalCoeffs = 1-f ; valueBoundaryCoeffs = f*ref + (1-f)*refGrad/delta
#   gradientInternalCoeffs = -f*delta ; gradientBoundaryCoeffs = f*delta*ref + (1-f)*refGrad
# fixedValue (f=1), zeroGradient (f=0, refGrad=0), fixedGradient (f=0), inletOutlet
# (f = 1-pos0(phi_b)) are special cases.
# =====================================================================================
class MixedBC:
    """Per-patch lists of arrays f, ref, refGrad (one entry per boundary face)."""

    def __init__(self, mesh, f=None, ref=None, refGrad=None):
        z = lambda: [np.zeros(p.size) for p in mesh.patches]
        self.f = f if f is not None else z()
        self.ref = ref if ref is not None else z()
        self.refGrad = refGrad if refGrad is not None else z()

    def values(self, mesh, vf):
        return [f * r + (1.0 - f) * (vf[p.faceCells] + g / p.deltaCoeffs)
                for p, f, r, g in zip(mesh.patches, self.f, self.ref, self.refGrad)]

    def value_internal(self):
        return [1.0 - f for f in self.f]

    def value_boundary(self, mesh):
        return [f * r + (1.0 - f) * g / p.deltaCoeffs for p, f, r, g in zip(mesh.patches, self.f, self.ref, self.refGrad)]

    def grad_internal(self, mesh):
        return [-f * p.deltaCoeffs for p, f in zip(mesh.patches, self.f)]

    def grad_boundary(self, mesh):
        return [f * p.deltaCoeffs * r + (1.0 - f) * g for p, f, r, g in zip(mesh.patches, self.f, self.ref, self.refGrad)]


def pos0(x):
    return (x >= 0).astype(float)


def surface_sum(mesh, ssf, bssf):
    """fvc::surfaceSum: owner += , neighbour += , boundary += (no division by V)."""
    out = np.zeros(mesh.nCells)
    np.add.at(out, mesh.u, ssf)
    np.add.at(out, mesh.l, ssf)
    for p, b in zip(mesh.patches, bssf):
        np.add.at(out, p.faceCells, b)
    return out


def grad(mesh, vf, bvals):
    """fvc::grad, Gauss linear (cases/steckler/system/fvSchemes:23-26): (1/V) sum_f Sf*vf_f."""
    ff, fb = interpolate(mesh, vf, bvals)
    out = np.zeros((mesh.nCells, 3))
    for d in range(3):
        out[:, d] = surface_integrate(mesh, mesh.Sf[:, d] * ff, [p.Sf[:, d] * b for p, b in zip(mesh.patches, fb)])
    return out


def reconstruct(mesh, ssf, bssf):
    """fvc::reconstruct(ssf) = inv(surfaceSum(Sf (x) Sf/magSf)) & surfaceSum((Sf/magSf)*ssf)."""
    N = mesh.nCells
    T = np.zeros((N, 3, 3)); v = np.zeros((N, 3))
    for a in range(3):
        v[:, a] = surface_sum(mesh, mesh.Sf[:, a] / mesh.magSf * ssf, [p.Sf[:, a] / p.magSf * b for p, b in zip(mesh.patches, bssf)])
        for b_ in range(3):
            T[:, a, b_] = surface_sum(mesh, mesh.Sf[:, a] * mesh.Sf[:, b_] / mesh.magSf,
                                      [p.Sf[:, a] * p.Sf[:, b_] / p.magSf for p in mesh.patches])
    # 2-D / 1-D meshes (empty patches): OpenFOAM's inv(tensorField) finds the missing directions from the first cell, adds 1 on
    # those diagonals before inverting and takes it off afterwards (src/OpenFOAM/fields/Fields/symmTensorField/symmTensorField.C)
    scale = (T[0] ** 2).sum()
    rm = np.array([T[0, a, a] ** 2 / scale < 1e-15 for a in range(3)], dtype=float) if scale > 0 else np.zeros(3)
    E = np.diag(rm)
    return np.einsum("nab,nb->na", np.linalg.inv(T + E) - E, v)


def limited_limiter(mesh, scheme, phi, vf, gradvf, k=1.0, bounds=(0.0, 1.0)):
    """The limiter field of limitedLinear k / limitedLinear01 k (limitedSurfaceInterpolationScheme::limiter: NVDTVD::r,
    limitedLinearLimiter, LimitedLimiter): 1 = linear, 0 = upwind."""
    P, Nn = vf[mesh.l], vf[mesh.u]
    d = mesh.C[mesh.u] - mesh.C[mesh.l]
    gradf = Nn - P
    gradcf = np.where(phi > 0, np.einsum("fd,fd->f", d, gradvf[mesh.l]), np.einsum("fd,fd->f", d, gradvf[mesh.u]))
    big = np.abs(gradcf) >= 1000.0 * np.abs(gradf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # OpenFOAM's sign(s) is (s >= 0) ? 1 : -1 (never 0): in a uniform region (gradf == gradcf == 0) r = 1999, the limiter
        # is 1 and the weights are the linear ones (NVDTVD::r, src/finiteVolume/.../limitedSchemes/LimitedScheme/NVDTVD.H)
        sgn = lambda a: np.where(a >= 0, 1.0, -1.0)
        r = np.where(big, 2.0 * 1000.0 * sgn(gradcf) * sgn(gradf) - 1.0, 2.0 * (gradcf / gradf) - 1.0)
    twoByk = 2.0 / max(k, 1e-15)
    lim = np.maximum(np.minimum(twoByk * r, 1.0), 0.0)
    if scheme == "limitedLinear01":
        lo, hi = bounds
        off = ((phi > 0) & ((P < lo) | (Nn > hi))) | ((phi < 0) & ((Nn < lo) | (P > hi)))
        lim = np.where(off, 0.0, lim)
    elif scheme != "limitedLinear":
        raise ValueError(scheme)
    return lim


def limited_weights(mesh, scheme, phi, vf, gradvf, k=1.0, bounds=(0.0, 1.0)):
    """Face weights of upwind / linear / limitedLinear k / limitedLinear01 k
    (limitedSurfaceInterpolationScheme::weights = limiter*linear + (1 - limiter)*upwind)."""
    w = mesh.weights
    if scheme == "linear":
        return w.copy()
    if scheme == "upwind":
        return pos0(phi)
    lim = limited_limiter(mesh, scheme, phi, vf, gradvf, k, bounds)
    return lim * w + (1.0 - lim) * pos0(phi)


def filtered_linear2V_weights(mesh, phi, U, gradU, k, l):
    """filteredLinear2V k l for a vector field (cases/wallFireSpread2D/system/fvSchemes:41 `div(phi,U) Gauss filteredLinear2V
    0.2 0.05`; OpenFOAM-dev .../limitedSchemes/filteredLinear2/filteredLinear2V.H, LimitedScheme<vector, ..., null>): ONE limiter per
    face for all components, from the face difference of the vector and twice the cell gradients projected on it,
        gradfV = U_N - U_P;  df = gradfV & gradfV;  tcP = 2*(gradfV & (d & gradU_P));  tcN = 2*(gradfV & (d & gradU_N))
        limiter = (l+1) - k*min(max(df - tcP, 0), max(df - tcN, 0))/(max(|tcP|, |tcN|) + SMALL)      (df > 0)
                = (l+1) - k*min(max(tcP - df, 0), max(tcN - df, 0))/(max(|tcP|, |tcN|) + SMALL)      (otherwise)
        limiter = max(min(limiter, 1), 0);  weight = limiter*w_linear + (1 - limiter)*pos0(phi)
    U[nCells][3]; gradU[nCells][3][3] with gradU[c][i][j] = d_i U_j (fvc::grad(U)); d = C_N - C_P.  The upstream source is not
    in the reference tree: 'parity unpinned' by reference data (no golden log of the two cases that select it exists)."""
    SMALL = 1.0e-15
    l1 = l + 1.0
    gradfV = U[mesh.u] - U[mesh.l]
    df = (gradfV[:, 0] * gradfV[:, 0] + gradfV[:, 1] * gradfV[:, 1]) + gradfV[:, 2] * gradfV[:, 2]
    d = mesh.C[mesh.u] - mesh.C[mesh.l]

    def tc(g):                                       # 2*(gradfV & (d & g)),  (d & g)_j = d_x g_xj + d_y g_yj + d_z g_zj
        dg = [(d[:, 0] * g[:, 0, j] + d[:, 1] * g[:, 1, j]) + d[:, 2] * g[:, 2, j] for j in range(3)]
        return 2.0 * ((gradfV[:, 0] * dg[0] + gradfV[:, 1] * dg[1]) + gradfV[:, 2] * dg[2])
    tcP, tcN = tc(gradU[mesh.l]), tc(gradU[mesh.u])
    den = np.maximum(np.abs(tcP), np.abs(tcN)) + SMALL
    lim = np.where(df > 0,
                   l1 - k * np.minimum(np.maximum(df - tcP, 0.0), np.maximum(df - tcN, 0.0)) / den,
                   l1 - k * np.minimum(np.maximum(tcP - df, 0.0), np.maximum(tcN - df, 0.0)) / den)
    lim = np.maximum(np.minimum(lim, 1.0), 0.0)
    return lim * mesh.weights + (1.0 - lim) * pos0(phi)


def lust_weights(mesh, phi):
    """LUST<Type>::weights (OpenFOAM-dev src/finiteVolume/interpolation/surfaceInterpolation/schemes/LUST/LUST.H):
    0.75*linear weights + 0.25*upwind weights; reference selection cases/steckler/system/fvSchemes:32."""
    return 0.75 * mesh.weights + 0.25 * pos0(phi)


def lust_correction(mesh, phi, gradvf):
    """LUST<Type>::correction = 0.25*linearUpwind<Type>::correction: per internal face (Cf - C_c) & grad(vf)_c with
    c = owner if phi > 0 else neighbour (linearUpwind.C); zero on non-coupled patches.  One scalar component."""
    cell = np.where(phi > 0, mesh.l, mesh.u)
    d = mesh.Cf - mesh.C[cell]
    g = gradvf[cell]
    return 0.25 * ((d[:, 0] * g[:, 0] + d[:, 1] * g[:, 1]) + d[:, 2] * g[:, 2])


def flip_maps(mesh, axis):
    """The mirror image of a hex box in one axis as a renaming of its cells and faces: cm[c] = the cell that takes c's place,
    fm[f] = the face that takes f's place, swap[f] = owner and neighbour change roles there.  The renamed matrix
    B[r, c] = A[cm[r], cm[c]] has the sparsity of A; an upwind matrix whose direction has one component of the other sign
    becomes triangular in the cell order when that axis is flipped (direction-ordered ray solves of the fvDOM stand-in)."""
    nx, ny, nz = mesh.n
    i, j, k = (a.copy() for a in mesh.ijk)
    if axis == 0:
        i = nx - 1 - i
    elif axis == 1:
        j = ny - 1 - j
    else:
        k = nz - 1 - k
    cm = i + nx * (j + ny * k)
    o, n = cm[mesh.l], cm[mesh.u]
    swap = o > n
    o, n = np.minimum(o, n), np.maximum(o, n)
    key = mesh.l.astype(np.int64) * mesh.nCells + mesh.u
    order = np.argsort(key)
    pos = np.searchsorted(key[order], o.astype(np.int64) * mesh.nCells + n)
    fm = order[pos]
    assert np.array_equal(mesh.l[fm], o) and np.array_equal(mesh.u[fm], n)
    return cm, fm, swap


def linear_upwind_correction(mesh, phi, gradvf):
    """linearUpwind<Type>::correction (OpenFOAM-dev .../schemes/linearUpwind/linearUpwind.C): (Cf - C_c) & grad(vf)_c, c the upwind
    cell, zero on non-coupled patches; its weights are upwind's.  Reference selection: `div(Ji,Ii_h) Gauss linearUpwind
    grad(Ii_h)`, cases/wallFireSpread2D/system/fvSchemes:58."""
    cell = np.where(phi > 0, mesh.l, mesh.u)
    d = mesh.Cf - mesh.C[cell]
    g = gradvf[cell]
    return (d[:, 0] * g[:, 0] + d[:, 1] * g[:, 1]) + d[:, 2] * g[:, 2]


class Matrix:
    """fvMatrix<Type> with nc components sharing diag/upper/lower (Type = scalar: nc=1, vector: nc=3)."""

    def __init__(self, mesh, nc=1):
        self.mesh, self.nc = mesh, nc
        self.diag = np.zeros(mesh.nCells)
        self.upper = np.zeros(mesh.nFaces)
        self.lower = np.zeros(mesh.nFaces)
        self.source = np.zeros((nc, mesh.nCells))
        self.internalCoeffs = [np.zeros((nc, p.size)) for p in mesh.patches]
        self.boundaryCoeffs = [np.zeros((nc, p.size)) for p in mesh.patches]

    def neg_sum_diag(self):
        np.subtract.at(self.diag, self.mesh.u, self.upper)
        np.subtract.at(self.diag, self.mesh.l, self.lower)

    def __iadd__(self, o):
        self.diag += o.diag; self.upper += o.upper; self.lower += o.lower; self.source += o.source
        for q in range(len(self.internalCoeffs)):
            self.internalCoeffs[q] += o.internalCoeffs[q]; self.boundaryCoeffs[q] += o.boundaryCoeffs[q]
        return self

    def __isub__(self, o):
        self.diag -= o.diag; self.upper -= o.upper; self.lower -= o.lower; self.source -= o.source
        for q in range(len(self.internalCoeffs)):
            self.internalCoeffs[q] -= o.internalCoeffs[q]; self.boundaryCoeffs[q] -= o.boundaryCoeffs[q]
        return self

    def add_su(self, su):
        """`M == su` / `M - su` for a volField su: source += V*su."""
        self.source += self.mesh.V * np.atleast_2d(su)
        return self

    def add_vol(self, su):
        """`M += su` for a volField su (fvMatrix::operator+=(volField)): source -= V*su.  Used by gaussConvectionScheme::fvmDiv
        for corrected() interpolation schemes: fvm += fvc::surfaceIntegrate(faceFlux*correction(vf))."""
        self.source -= self.mesh.V * np.atleast_2d(su)
        return self

    def relax(self, alpha, psi_prev):
        """fvMatrix<Type>::relax(alpha) (OpenFOAM-dev src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C; reference call sites
        solver/UEqn.H:13, solver/YEEqn.H:56,107; equation factors cases/wallFireSpread2D/system/fvSolution:194-200), all patches
        non-coupled: D += cmptMax|internalCoeffs|; D = max(|D|, sumMagOffDiag); D /= alpha; D -= cmptMin(internalCoeffs);
        source += (D - D0)*psi.prevIter()."""
        if alpha <= 0:
            return self
        m = self.mesh
        D0 = self.diag.copy()
        sumOff = np.zeros(m.nCells)
        np.add.at(sumOff, m.u, np.abs(self.lower))          # rows in face order: faces of which the cell is the neighbour first
        np.add.at(sumOff, m.l, np.abs(self.upper))
        D = self.diag
        for p, ic in zip(m.patches, self.internalCoeffs):
            np.add.at(D, p.faceCells, np.abs(ic).max(axis=0))
        D[:] = np.maximum(np.abs(D), sumOff)
        D /= alpha
        for p, ic in zip(m.patches, self.internalCoeffs):
            np.subtract.at(D, p.faceCells, ic.min(axis=0))
        self.source += (D - D0) * np.atleast_2d(psi_prev)
        return self

    def solve_system(self, cmpt=0):
        d = self.diag.copy(); s = self.source[cmpt].copy()
        for p, ic, bc in zip(self.mesh.patches, self.internalCoeffs, self.boundaryCoeffs):
            np.add.at(d, p.faceCells, ic[cmpt]); np.add.at(s, p.faceCells, bc[cmpt])
        return d, s

    def A(self):
        """fvMatrix::A(): (diag + cmptAv boundary diag)/V."""
        d = self.diag.copy()
        for p, ic in zip(self.mesh.patches, self.internalCoeffs):
            np.add.at(d, p.faceCells, ic.mean(axis=0))
        return d / self.mesh.V

    def H(self, psi):
        """fvMatrix::H(): (source + boundarySource - sum offdiag*psi_nb + (avgBD - BD_cmpt)*psi)/V."""
        m = self.mesh
        out = np.zeros((self.nc, m.nCells))
        for c in range(self.nc):
            bd = np.zeros(m.nCells); bda = np.zeros(m.nCells)
            for p, ic in zip(m.patches, self.internalCoeffs):
                np.add.at(bd, p.faceCells, ic[c]); np.add.at(bda, p.faceCells, ic.mean(axis=0))
            h = (bda - bd) * psi[c]
            hl = np.zeros(m.nCells)
            np.subtract.at(hl, m.u, self.lower * psi[c][m.l])
            np.subtract.at(hl, m.l, self.upper * psi[c][m.u])
            h += hl + self.source[c]
            for p, bc in zip(m.patches, self.boundaryCoeffs):
                np.add.at(h, p.faceCells, bc[c])
            out[c] = h / m.V
        return out

    def flux(self, psi, cmpt=0):
        """fvMatrix::flux(): internal upper*psi_u - lower*psi_l; boundary internalCoeffs*psi_c - boundaryCoeffs."""
        m = self.mesh
        fi = self.upper * psi[m.u] - self.lower * psi[m.l]
        fb = [ic[cmpt] * psi[p.faceCells] - bc[cmpt] for p, ic, bc in zip(m.patches, self.internalCoeffs, self.boundaryCoeffs)]
        return fi, fb


def fvm_ddt(mesh, rDeltaT, rho, rho0, vf0):
    """EulerDdtScheme::fvmDdt(rho, vf): diag = rDeltaT*rho*V; source = rDeltaT*rho0*vf0*V."""
    vf0 = np.atleast_2d(vf0)
    M = Matrix(mesh, vf0.shape[0])
    M.diag = rDeltaT * rho * mesh.V
    M.source = rDeltaT * rho0 * vf0 * mesh.V
    return M


def fvm_div(mesh, phi, phib, w, bcs):
    """gaussConvectionScheme::fvmDiv; bcs = list of MixedBC, one per component."""
    M = Matrix(mesh, len(bcs))
    M.lower = -w * phi
    M.upper = M.lower + phi
    M.neg_sum_diag()
    for c, bc in enumerate(bcs):
        vi, vb = bc.value_internal(), bc.value_boundary(mesh)
        for q in range(len(mesh.patches)):
            M.internalCoeffs[q][c] = phib[q] * vi[q]
            M.boundaryCoeffs[q][c] = -phib[q] * vb[q]
    return M


def fvm_laplacian(mesh, gamma_f, gamma_b, bcs):
    """gaussLaplacianScheme::fvmLaplacianUncorrected; bcs = list of MixedBC, one per component."""
    M = Matrix(mesh, len(bcs))
    M.upper = gamma_f * mesh.magSf * mesh.deltaCoeffs
    M.lower = M.upper.copy()
    M.neg_sum_diag()
    for c, bc in enumerate(bcs):
        gi, gb = bc.grad_internal(mesh), bc.grad_boundary(mesh)
        for q, p in enumerate(mesh.patches):
            pG = gamma_b[q] * p.magSf
            M.internalCoeffs[q][c] = pG * gi[q]
            M.boundaryCoeffs[q][c] = -pG * gb[q]
    return M
