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


# ---- plain torch-fp64 evaluation of the encoder on grouped rows, optionally with the KERNEL's max-pool routing -----------
def forward64(x_rows, centers, sd, G, S, K, dev, routing=None, grad=False, dtype=None):
    """cn3d_model_conbag.py:213-234 in fp64 (matmuls, train-mode BN with batch statistics, ReLU, the three max-pools) on
    (P,D) grouped rows / (M*S,3) centres.  Returns (x, x_global, stats, q, ties).

    `routing` = {"sa_arg" (M*S,256), "seg_arg" (M,1024), "view_arg" (B,1024)}: the argmax indices the HIP forward chose.
    Each max-pool then GATHERS at those indices instead of taking its own maximum, so the autograd graph routes the gradient
    exactly as the kernels do and a comparison of gradients no longer depends on which side of a numerical near-tie each
    arithmetic lands on.  `ties[name]` = the largest distance of a gathered value from the true fp64 maximum, relative to
    max(|max|, mean |activation|): the test that every differing decision WAS a tie.
    `grad`: parameters require grad (q[k].grad after a backward).  `dtype` (default float64): torch.float32 gives the
    "plain torch fp32 with the same routing" twin -- the conditioning yardstick of a gradient comparison."""
    import torch
    dtype = torch.float64 if dtype is None else dtype
    q = {k: torch.as_tensor(v).to(dev).to(dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    if grad:
        for k in q:
            if "running" not in k:
                q[k].requires_grad_(True)
    stats, ties = {}, {}

    def bn_train(y, gamma, beta, key):
        P = y.shape[0]
        mean, var = y.mean(0), y.var(0, unbiased=False)
        stats[key] = (mean.detach(), (var * (P / (P - 1.0))).detach())
        return (y - mean) / torch.sqrt(var + 1e-5) * gamma + beta

    def pool(h3, idx, name):                                                   # (R, L, C) -> (R, C) over L
        mx = h3.max(dim=1).values
        if idx is None:
            return mx
        got = torch.gather(h3, 1, idx.long().view(h3.shape[0], 1, h3.shape[2])).squeeze(1)
        with torch.no_grad():
            scale = torch.maximum(mx.abs(), h3.abs().mean())
            ties[name] = float(((mx - got).abs() / scale).max())
            ties[name + "_flips"] = int((idx.long().view(h3.shape[0], h3.shape[2]) != h3.argmax(dim=1)).sum())
        return got

    r = routing or {}

    def relu(z, name):
        """ReLU, or -- with the kernel's own decisions in `routing` -- z * mask.  Every decision that differs from z > 0 must
        be a numerical tie: |z| there <= 1e-5 of the layer's mean |z| (recorded in ties["relu_*"])."""
        mask = r.get(name)
        if mask is None:
            return torch.relu(z)
        with torch.no_grad():
            diff = mask.view(z.shape) != (z > 0)
            ties[name + "_flips"] = int(diff.sum())
            ties[name] = float((z.abs() * diff).max() / z.abs().mean())
        return z * mask.view(z.shape)

    h = x_rows.to(dtype)
    for i, li in enumerate((0, 3, 6)):                                         # net3DV_1 (:43-58)
        W = q[f"net3DV_1.{li}.weight"].reshape(q[f"net3DV_1.{li}.weight"].shape[0], -1)
        y = h @ W.t() + q[f"net3DV_1.{li}.bias"]
        z = bn_train(y, q[f"net3DV_1.{li + 1}.weight"], q[f"net3DV_1.{li + 1}.bias"], f"net3DV_1.{li + 1}")
        del y
        # the third layer's ReLU sits behind the max-pool in the kernels (monotone per channel): its decision is pinned there
        h = relu(z, f"relu_sa{i + 1}") if i < 2 else z
        del z
    MS = h.shape[0] // K
    pooled = pool(h.view(MS, K, 256), r.get("sa_arg"), "sa")
    del h
    pooled = relu(pooled, "relu_sa3")
    h = torch.cat((centers.to(dtype), pooled), dim=1)                           # :219
    for i, li in enumerate((0, 3, 6)):                                         # net3DV_3 (:61-77)
        W = q[f"net3DV_3.{li}.weight"].reshape(q[f"net3DV_3.{li}.weight"].shape[0], -1)
        y = h @ W.t() + q[f"net3DV_3.{li}.bias"]
        z = bn_train(y, q[f"net3DV_3.{li + 1}.weight"], q[f"net3DV_3.{li + 1}.bias"], f"net3DV_3.{li + 1}")
        h = relu(z, f"relu_t{i + 1}") if i < 2 else z
    M = MS // S
    B = M // G
    x_pre = relu(pool(h.view(M, S, 1024), r.get("seg_arg"), "seg"), "relu_t3")   # :222 (max of relu = relu of max)
    # :225-226 (rows are view-major g*B+b): max over all G*S local features of a clip = max over the views of the view maxima
    xg_pre = pool(x_pre.view(G, B, 1024).transpose(0, 1), r.get("view_arg"), "view")

    fcm = r.get("relu_fc")                                                     # (M + B, 1024): view rows, then clip rows

    def head(t, key, mask):                                                    # netR_FC (:201-207), two BN calls (:228-229)
        y = t @ q["netR_FC.0.weight"].t() + q["netR_FC.0.bias"]
        z = bn_train(y, q["netR_FC.1.weight"], q["netR_FC.1.bias"], key)
        if mask is not None:
            r["relu_" + key] = mask
        a = relu(z, "relu_" + key)
        return a @ q["netR_FC.3.weight"].t() + q["netR_FC.3.bias"]
    r = dict(r)
    x = head(x_pre, "fc_a", None if fcm is None else fcm[:M])
    xg = head(xg_pre, "fc_b", None if fcm is None else fcm[M:])
    return x, xg, stats, q, ties


class routing_taps:
    """with routing_taps() as r: <HIP forward> -> r = {"sa_arg", "seg_arg", "view_arg"}: the argmax tensors of the three
    max-pools of that forward (facl_amd/_lib.py: tap)."""

    def __enter__(self):
        from facl_amd import _lib
        self.prev, _lib.TAPS = _lib.TAPS, {}
        return _lib.TAPS

    def __exit__(self, *exc):
        from facl_amd import _lib
        _lib.TAPS = self.prev
        return False
