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


def synth_clip(seed, dt, P=900, Kp=300, R1=500, R2=200):
    """The four (rows, 8) clouds `__getitem__` loads for one video (cn3D_data_set.py:105-116), synthetic; the same
    formula as tools/make_goldens.py: synth_clip, whose arguments tests/golden/views.npz records in `cases`."""
    r = np.random.RandomState(seed)
    pts = (r.rand(P, 8) - 0.5).astype(dt)
    pts[::3, 4] = 0
    pts[1::4, 7] = 0
    return pts, (r.rand(Kp, 8) - 0.5).astype(dt), (r.rand(R1, 8) - 0.5).astype(dt), (r.rand(R2, 8) - 0.5).astype(dt)


def golden_view_clips(g):
    """The clips of views.npz, rebuilt from the recorded generator arguments, in stream order, with their tags."""
    out = []
    for tag, (cseed, is64, P, Kp, R1, R2) in zip("abcd", g["cases"].tolist()):
        out.append((tag, synth_clip(cseed, np.float64 if is64 else np.float32, P, Kp, R1, R2)))
    return out
