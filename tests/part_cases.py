"""Meshes and matrices of the partitioner tests (shared by the pytest process and the rank workers, so that every rank builds
the same global problem): the baffled steckler room (cases/steckler geometry, oracle/steckler.py) and a randomly relabelled box."""
import numpy as np

import common


def build(O, meshName, asym=0.0):
    if meshName == "steckler":
        from oracle import steckler
        m = steckler.build_mesh()
        N, l, u, centres = m.nCells, m.l.astype(np.int32), m.u.astype(np.int32), m.C
    elif meshName == "dag_random":
        n = 10
        N, l, u = common.random_dag_mesh(O, n)
        # centres of the relabelled box (RCB needs geometry; the graph partitioner does not use them)
        rng = np.random.RandomState(5); perm = rng.permutation(N)
        c = np.arange(N); ijk = np.stack([c % n, (c // n) % n, c // (n * n)], axis=1).astype(float)
        centres = np.empty((N, 3)); centres[perm] = ijk
    else:
        raise ValueError(meshName)
    diag, up, lo = common.laplacian_like(O, N, l, u, seed=3, asym=asym, shift=0.05)
    source = O.hash_u(0xF3, np.arange(N)) - 0.4
    return N, l, u, centres, diag, up, lo, source


def partition(ffm, name, N, l, u, centres, nParts):
    if name == "rcb":
        return ffm.decompose.partition_rcb(centres, nParts)
    return ffm.decompose.partition_graph(N, l, u, nParts)
