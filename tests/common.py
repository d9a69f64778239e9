"""Shared builders for the parity tests (same seeded inputs for oracle and HIP path)."""
import numpy as np


def laplacian_like(O, N, l, u, seed=1, asym=0.0, shift=0.01):
    """Diagonally dominant LDU coefficients on any addressing; asym>0 makes lower != upper."""
    F = len(l)
    up = -(0.5 + O.hash_u(seed, np.arange(F)))
    lo = up * (1.0 + asym * (O.hash_u(seed + 100, np.arange(F)) - 0.3)) if asym else None
    diag = np.zeros(N)
    np.add.at(diag, l, -up)
    np.add.at(diag, u, -(lo if lo is not None else up))
    diag += shift * (1.0 + O.hash_u(seed + 200, np.arange(N)))
    return diag, up, lo


def random_dag_mesh(O, n, seed=5):
    """A hex box whose cells are randomly relabelled (faces re-oriented to l<u and re-sorted):
    a general unstructured LDU graph whose dependency levels are not hyperplanes."""
    N, l, u = O.hex_ldu(n, n, n)
    rng = np.random.RandomState(seed)
    perm = rng.permutation(N)
    a, b = perm[l], perm[u]
    l2, u2 = np.minimum(a, b), np.maximum(a, b)
    order = np.lexsort((u2, l2))
    return N, l2[order].astype(np.int32), u2[order].astype(np.int32)


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def free_port():
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests on 127.0.0.1): asked from the kernel, so that two
    test cases run one after the other never meet on a port that is still in TIME_WAIT"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
