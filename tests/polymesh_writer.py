"""Test helper: write an oracle HexMesh (patches set, before any shear) as an OpenFOAM ascii constant/polyMesh directory, the way
blockMesh (+ createPatch) would: points, faces (vertex order so that the area vector points from owner to neighbour / out of the
domain), owner, neighbour, boundary.  M: optional affine map of the points (non-orthogonal test meshes)."""
import os

import numpy as np

HEADER = """/*--------------------------------*- C++ -*----------------------------------*\\
  =========                 |
  \\\\      /  F ield         | OpenFOAM: The Open Source CFD Toolbox
\\*---------------------------------------------------------------------------*/
FoamFile
{
    version     2.0;
    format      ascii;
    class       %s;
    location    "constant/polyMesh";
    object      %s;
}
// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //

"""


def _face_vertices(nx, ny, i, j, k, axis, outward_positive):
    """point ids of the face of cell (i,j,k) on its + side (outward_positive) or - side along `axis`"""
    pid = lambda a, b, c: a + (nx + 1) * (b + (ny + 1) * c)
    if axis == 0:
        a = i + 1 if outward_positive else i
        v = [pid(a, j, k), pid(a, j + 1, k), pid(a, j + 1, k + 1), pid(a, j, k + 1)]
    elif axis == 1:
        b = j + 1 if outward_positive else j
        v = [pid(i, b, k), pid(i, b, k + 1), pid(i + 1, b, k + 1), pid(i + 1, b, k)]
    else:
        c = k + 1 if outward_positive else k
        v = [pid(i, j, c), pid(i + 1, j, c), pid(i + 1, j + 1, c), pid(i, j + 1, c)]
    return v if outward_positive else v[::-1]


def write_polymesh(path, mesh, lo, M=None, patch_types=None):
    os.makedirs(path, exist_ok=True)
    nx, ny, nz = mesh.n
    d = mesh.d
    kk, jj, ii = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    pts = np.stack([lo[0] + ii.ravel() * d[0], lo[1] + jj.ravel() * d[1], lo[2] + kk.ravel() * d[2]], axis=1)
    if M is not None:
        pts = pts @ np.asarray(M, float).T
    ci, cj, ck = mesh.ijk
    faces, owner, neighbour = [], [], []
    for f in range(mesh.nFaces):                                    # internal faces: the + side of the owner
        c = mesh.l[f]
        faces.append(_face_vertices(nx, ny, ci[c], cj[c], ck[c], int(mesh.fdir[f]), True))
        owner.append(int(c)); neighbour.append(int(mesh.u[f]))
    bnd = []
    for p in mesh.patches:
        start = len(faces)
        for c, S in zip(p.faceCells, p.Sf):
            axis = int(np.argmax(np.abs(S)))
            faces.append(_face_vertices(nx, ny, ci[c], cj[c], ck[c], axis, S[axis] > 0))
            owner.append(int(c))
        t = (patch_types or {}).get(p.name, "patch")
        bnd.append((p.name,) + (t if isinstance(t, tuple) else (t, {})) + (p.size, start))
    with open(os.path.join(path, "points"), "w") as f:
        f.write(HEADER % ("vectorField", "points"))
        f.write("%d\n(\n" % len(pts))
        for x in pts:
            f.write("(%.17g %.17g %.17g)\n" % tuple(x))
        f.write(")\n")
    with open(os.path.join(path, "faces"), "w") as f:
        f.write(HEADER % ("faceList", "faces"))
        f.write("%d\n(\n" % len(faces))
        for v in faces:
            f.write("%d(%s)\n" % (len(v), " ".join(map(str, v))))
        f.write(")\n")
    for name, arr in (("owner", owner), ("neighbour", neighbour)):
        with open(os.path.join(path, name), "w") as f:
            f.write((HEADER % ("labelList", name)).replace("    location", '    note        "nPoints:%d  nCells:%d  nFaces:%d  nInternalFaces:%d";\n    location'
                                                           % (len(pts), mesh.nCells, len(faces), mesh.nFaces)))
            f.write("%d\n(\n" % len(arr))
            f.write("\n".join(map(str, arr)))
            f.write("\n)\n")
    with open(os.path.join(path, "boundary"), "w") as f:
        f.write(HEADER % ("polyBoundaryMesh", "boundary"))
        f.write("%d\n(\n" % len(bnd))
        for name, typ, extra, n, start in bnd:           # extra: further keywords (a processor patch's myProcNo / neighbProcNo)
            f.write("    %s\n    {\n        type            %s;\n" % (name, typ))
            if typ == "wall":
                f.write("        inGroups        1(wall);\n")
            f.write("        nFaces          %d;\n        startFace       %d;\n" % (n, start))
            for k, v in extra.items():
                f.write("        %-15s %s;\n" % (k, v))
            f.write("    }\n")
        f.write(")\n")
