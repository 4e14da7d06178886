import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def canon_groups_np(xt_MSKD):
    """Sort the K axis of (M,S,K,D) lexicographically (order inside a group is unspecified)."""
    a = np.asarray(xt_MSKD)
    M, S, K, D = a.shape
    flat = a.reshape(M * S, K, D)
    out = np.empty_like(flat)
    for i in range(flat.shape[0]):
        keys = tuple(flat[i, :, d] for d in reversed(range(D)))
        out[i] = flat[i][np.lexsort(keys)]
    return out.reshape(M, S, K, D)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_rel_rows(a, b):
    """max over rows of ||a_i-b_i|| / ||b_i|| (features: atol scaled by row norm, SURVEY hard part 4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = np.linalg.norm(a - b, axis=-1)
    den = np.maximum(np.linalg.norm(b, axis=-1), 1e-30)
    return float((num / den).max())
