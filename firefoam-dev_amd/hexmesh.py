"""Synthetic hex-box meshes in blockMesh numbering, block decomposition and the
SURVEY 8(d) synthetic p_rgh matrix -- host-side tooling for tests and bench.py.

blockMesh single block (reference cases/steckler/constant/polyMesh/blockMeshDict:50):
cell c = i + nx*(j + ny*k); internal faces in upper-triangular order: for c
ascending the faces (c,c+1), (c,c+nx), (c,c+nx*ny) that exist.  A sub-block of a
global box (structured decomposition, cf. `simpleCoeffs`/`hierarchicalCoeffs` in
cases/wallFireSpread2D/system/decomposeParDict:18-27) gets a local LDU plus one
processor interface per neighbouring block, with the cut faces listed in global
face order on both sides.
"""
import numpy as np


def hash_u(seed, idx):
    """u(seed, idx) = (splitmix64(seed ^ idx) >> 11) * 2^-53  in [0,1)   (SURVEY 8d)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ idx) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def owner_start(i, j, k, nx, ny, nz):
    """Number of internal faces owned by cells before cell (i,j,k) of an nx*ny*nz box (closed form)."""
    i = np.asarray(i, np.int64); j = np.asarray(j, np.int64); k = np.asarray(k, np.int64)
    c = i + nx * (j + ny * k)
    fx = (j + ny * k) * (nx - 1) + np.minimum(i, nx - 1)
    fy = k * nx * (ny - 1) + np.minimum(j, ny - 1) * nx + np.where(j < ny - 1, i, 0)
    fz = np.minimum(c, nx * ny * (nz - 1))
    return fx + fy + fz


def hex_ldu(nx, ny, nz):
    """(nCells, lowerAddr, upperAddr) of the whole box, natural numbering."""
    b = HexBlock((nx, ny, nz))
    return b.nCells, b.l, b.u


class HexBlock:
    """Cells [lo, hi) of a global (nx,ny,nz) box, numbered locally like a blockMesh block."""

    def __init__(self, glob, lo=(0, 0, 0), hi=None, rank=0, grid=None):
        self.G = tuple(int(v) for v in glob)
        self.lo = tuple(int(v) for v in lo)
        self.hi = tuple(int(v) for v in (hi if hi is not None else glob))
        self.rank, self.grid = rank, grid
        nx, ny, nz = (self.hi[d] - self.lo[d] for d in range(3))
        self.n = (nx, ny, nz)
        self.nCells = nx * ny * nz
        kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()        # local cell order: i fastest
        self.i, self.j, self.k = ii, jj, kk
        hx, hy, hz = (ii < nx - 1), (jj < ny - 1), (kk < nz - 1)
        cnt = hx.astype(np.int64) + hy + hz
        start = np.concatenate(([0], np.cumsum(cnt)))
        F = int(start[-1])
        c = np.arange(self.nCells, dtype=np.int64)
        l = np.empty(F, np.int32); u = np.empty(F, np.int32); fdir = np.empty(F, np.int8)
        px = start[:-1][hx]; l[px] = c[hx]; u[px] = c[hx] + 1; fdir[px] = 0
        py = (start[:-1] + hx)[hy]; l[py] = c[hy]; u[py] = c[hy] + nx; fdir[py] = 1
        pz = (start[:-1] + hx + hy)[hz]; l[pz] = c[hz]; u[pz] = c[hz] + nx * ny; fdir[pz] = 2
        self.l, self.u, self.fdir, self.nFaces = l, u, fdir, F
        # global ids
        GX, GY, GZ = self.G
        gi, gj, gk = ii + self.lo[0], jj + self.lo[1], kk + self.lo[2]
        self.gi, self.gj, self.gk = gi, gj, gk
        self.gcell = (gi + GX * (gj + GY * gk)).astype(np.int64)
        # global natural face id of every local internal face
        gos = owner_start(gi, gj, gk, GX, GY, GZ)
        ghx, ghy = (gi < GX - 1).astype(np.int64), (gj < GY - 1).astype(np.int64)
        gface = np.empty(F, np.int64)
        gface[px] = gos[hx]
        gface[py] = (gos + ghx)[hy]
        gface[pz] = (gos + ghx + ghy)[hz]
        self.gface = gface
        self._gos, self._ghx, self._ghy = gos, ghx, ghy

    # -- processor interfaces of a block decomposition -------------------------------
    def interfaces(self):
        """List of dicts {nbr (block coords), faceCells (local), gface, owner_side} for the six
        possible neighbours, in the order -x,+x,-y,+y,-z,+z restricted to existing ones.  Faces of
        each interface are sorted by global face id, which is the same order on both sides."""
        nx, ny, nz = self.n
        GX, GY, GZ = self.G
        out = []
        c = np.arange(self.nCells, dtype=np.int64)
        for d, (loc, n_d, G_d) in enumerate(((self.i, nx, GX), (self.j, ny, GY), (self.k, nz, GZ))):
            for side in (0, 1):
                if side == 0 and self.lo[d] == 0:
                    continue
                if side == 1 and self.hi[d] == G_d:
                    continue
                sel = (loc == 0) if side == 0 else (loc == n_d - 1)
                cells = c[sel]
                if side == 1:   # this block owns the cut face (its cell is the global owner)
                    gf = self._gos[sel] + (0 if d == 0 else self._ghx[sel] if d == 1 else self._ghx[sel] + self._ghy[sel])
                else:           # neighbour block's cell (one step back in direction d) owns it
                    gi, gj, gk = self.gi[sel].copy(), self.gj[sel].copy(), self.gk[sel].copy()
                    (gi, gj, gk)[d][:] -= 1
                    gos = owner_start(gi, gj, gk, GX, GY, GZ)
                    ghx, ghy = (gi < GX - 1).astype(np.int64), (gj < GY - 1).astype(np.int64)
                    gf = gos + (0 if d == 0 else ghx if d == 1 else ghx + ghy)
                order = np.argsort(gf, kind="stable")
                out.append({"dir": d, "side": side, "faceCells": cells[order].astype(np.int32),
                            "gface": gf[order], "owner_side": side == 1})
        return out


def decompose(glob, grid):
    """Structured block decomposition: returns [HexBlock] in rank order r = bx + gx*(by + gy*bz) and,
    per rank, the neighbour rank of each interface (same order as HexBlock.interfaces())."""
    gx, gy, gz = grid
    cuts = [np.linspace(0, glob[d], grid[d] + 1).astype(int) for d in range(3)]
    blocks, nbr = [], []
    for bz in range(gz):
        for by in range(gy):
            for bx in range(gx):
                b = (bx, by, bz)
                lo = tuple(int(cuts[d][b[d]]) for d in range(3))
                hi = tuple(int(cuts[d][b[d] + 1]) for d in range(3))
                blocks.append(HexBlock(glob, lo, hi, rank=bx + gx * (by + gy * bz), grid=grid))
    for r, blk in enumerate(blocks):
        bx, by, bz = r % gx, (r // gx) % gy, r // (gx * gy)
        ranks = []
        for itf in blk.interfaces():
            b = [bx, by, bz]
            b[itf["dir"]] += 1 if itf["side"] == 1 else -1
            ranks.append(b[0] + gx * (b[1] + gy * b[2]))
        nbr.append(ranks)
    return blocks, nbr


def grid_for(nRanks):
    """Block grid used by bench.py for N = 1, 2, 4, 8 GPUs (2x2x2 at 8)."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(nRanks) or (nRanks, 1, 1)


def synth_p_rgh(block, h=0.05, dt=1e-3):
    """SURVEY 8(d) synthetic pressure matrix of `block` (a sub-block or the whole box).

    rhorAUf_f = 1e-3*(0.5+u(0xF1,f)); upper_f = -rhorAUf_f*h; psi_c = 1.17e-5*(0.9+0.2*u(0xF2,c));
    diag_c = sum|upper| (all faces of c, cut faces included) + psi_c*h^3/dt
             (+ 2*rhorAUf_b*h on the j = ny-1 'top' layer: fixedValue 0, delta_b = 2/h,
                rhorAUf_b = 1e-3*(0.5+u(0xF5,c)));
    source_c = h^3*(2u(0xF3,c)-1); x_c = u(0xF4,c).  f, c are GLOBAL natural face / cell ids, so a
    decomposed case is the same global problem.  Returns dict with diag, upper, source, x and, per
    interface, bouCoeffs (= -offdiag coefficient, OpenFOAM sign convention)."""
    GX, GY, GZ = block.G
    up = -(1e-3 * (0.5 + hash_u(0xF1, block.gface))) * h
    diag = np.zeros(block.nCells)
    np.add.at(diag, block.l, -up)
    np.add.at(diag, block.u, -up)
    itfs = block.interfaces()
    bou = []
    for itf in itfs:
        cf = -(1e-3 * (0.5 + hash_u(0xF1, itf["gface"]))) * h     # off-diagonal coefficient of the cut face
        np.add.at(diag, itf["faceCells"], -cf)
        bou.append(-cf)                                            # interfaceBouCoeffs = -coefficient
    psi_c = 1.17e-5 * (0.9 + 0.2 * hash_u(0xF2, block.gcell))
    diag += psi_c * h ** 3 / dt
    top = block.gj == GY - 1
    diag[top] += 2.0 * (1e-3 * (0.5 + hash_u(0xF5, block.gcell[top]))) * h
    source = h ** 3 * (2.0 * hash_u(0xF3, block.gcell) - 1.0)
    x = hash_u(0xF4, block.gcell)
    return {"diag": diag, "upper": up, "source": source, "x": x, "interfaces": itfs, "bouCoeffs": bou}


def apply_renumbering(nCells, l, u, cellOrder, faceOrder):
    """Renumber an LDU graph with new->old permutations (as returned by ffm_renumber_levels)."""
    oldToNew = np.empty(nCells, np.int64)
    oldToNew[cellOrder] = np.arange(nCells)
    return oldToNew[l[faceOrder]].astype(np.int32), oldToNew[u[faceOrder]].astype(np.int32), oldToNew


def block_of_rank(glob, grid, rank):
    """Block [lo,hi) of `rank` (r = bx + gx*(by + gy*bz)) and its six neighbour ranks (-x,+x,-y,+y,-z,+z; -1 = none)."""
    gx, gy, gz = grid
    b = (rank % gx, (rank // gx) % gy, rank // (gx * gy))
    cuts = [np.linspace(0, glob[d], grid[d] + 1).astype(int) for d in range(3)]
    lo = tuple(int(cuts[d][b[d]]) for d in range(3))
    hi = tuple(int(cuts[d][b[d] + 1]) for d in range(3))
    nbr = []
    for d in range(3):
        for side in (-1, 1):
            bb = list(b); bb[d] += side
            nbr.append(-1 if bb[d] < 0 or bb[d] >= grid[d] else bb[0] + gx * (bb[1] + gy * bb[2]))
    return lo, hi, nbr
