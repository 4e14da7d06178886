"""Closed-form ("formula") parameters so that goldens never ship a 9.4 MB state_dict.

Both tools/make_goldens.py (which loads them into the reference's own
``PointNet_Plus_fine``) and the tests (which load them into ``facl_amd``'s
``PointNet_Plus``) rebuild exactly the same 52-key state_dict from this formula.
Key names / shapes follow /root/reference/training_code/cn3d_model_conbag.py:43-91.
"""
import numpy as np

CONV1 = [64, 64, 256]            # nstates_plus_1, cn3d_model_conbag.py:15
CONV3 = [256, 512, 1024]         # nstates_plus_3[0:3], cn3d_model_conbag.py:17


def _hash_uniform(n, key):
    """n floats in [-1,1): a closed-form integer hash (splitmix-style) of (key, i) -- the same
    values on every platform, decorrelated like a random init (structured sin() weights made
    BatchNorm badly conditioned: huge channel means relative to their spread)."""
    i = np.arange(n, dtype=np.uint64)
    off = (int(key) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        z = i + np.uint64(off)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return 2.0 * u - 1.0


def _wave(n, a, b, scale):
    """kept name; now hash noise scaled to +-scale (a,b only seed the hash)."""
    key = int(a * 1e6) * 1000003 + int(b * 1e3)
    return (scale * _hash_uniform(n, key)).astype(np.float32)


def state_dict_shapes(D):
    """Ordered (key, shape) list of the 52-key reference state_dict."""
    out = []
    cin = D
    for li, cout in zip((0, 3, 6), CONV1):
        out += [(f"net3DV_1.{li}.weight", (cout, cin, 1, 1)), (f"net3DV_1.{li}.bias", (cout,))]
        b = li + 1
        out += [(f"net3DV_1.{b}.weight", (cout,)), (f"net3DV_1.{b}.bias", (cout,)),
                (f"net3DV_1.{b}.running_mean", (cout,)), (f"net3DV_1.{b}.running_var", (cout,)),
                (f"net3DV_1.{b}.num_batches_tracked", ())]
        cin = cout
    cin = 3 + 256
    for li, cout in zip((0, 3, 6), CONV3):
        out += [(f"net3DV_3.{li}.weight", (cout, cin, 1, 1)), (f"net3DV_3.{li}.bias", (cout,))]
        b = li + 1
        out += [(f"net3DV_3.{b}.weight", (cout,)), (f"net3DV_3.{b}.bias", (cout,)),
                (f"net3DV_3.{b}.running_mean", (cout,)), (f"net3DV_3.{b}.running_var", (cout,)),
                (f"net3DV_3.{b}.num_batches_tracked", ())]
        cin = cout
    out += [("netR_FC.0.weight", (1024, 1024)), ("netR_FC.0.bias", (1024,)),
            ("netR_FC.1.weight", (1024,)), ("netR_FC.1.bias", (1024,)),
            ("netR_FC.1.running_mean", (1024,)), ("netR_FC.1.running_var", (1024,)),
            ("netR_FC.1.num_batches_tracked", ()),
            ("netR_FC.3.weight", (512, 1024)), ("netR_FC.3.bias", (512,)),
            ("mapping.weight", (64, 512))]
    return out


def formula_state_dict(D, neg_gamma=False, seed=0):
    """numpy state_dict; conv/linear weights ~ N(0, 2/fan_in)-ish magnitude, BN gamma in
    [0.4,1.6] (every 5th channel negated when ``neg_gamma``: exercises the min-pool branch
    of a fused BN+ReLU+max kernel), non-trivial running stats."""
    sd = {}
    for n, (key, shape) in enumerate(state_dict_shapes(D)):
        cnt = int(np.prod(shape)) if shape else 1
        a = 0.7310585 + 0.0137 * n + 0.001 * seed
        b = 0.31 * n + seed
        if key.endswith("num_batches_tracked"):
            v = np.array(3, dtype=np.int64)
        elif key.endswith("running_mean"):
            v = _wave(cnt, a, b, 0.1)
        elif key.endswith("running_var"):
            v = (1.0 + 0.5 * _hash_uniform(cnt, 7919 * n + seed)).astype(np.float32)
        elif len(shape) == 1 and (".1." in key or ".4." in key or ".7." in key) and key.endswith("weight"):
            v = (1.0 + 0.6 * _hash_uniform(cnt, 104729 * n + seed)).astype(np.float32)
            if neg_gamma:
                v[::5] = -v[::5]
        elif len(shape) == 1 and (".1." in key or ".4." in key or ".7." in key) and key.endswith("bias"):
            v = _wave(cnt, a, b, 0.2)
        elif key.endswith("bias"):
            v = _wave(cnt, a, b, 0.1)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = _wave(cnt, a, b, 1.0 / np.sqrt(fan_in))      # = torch's kaiming_uniform(a=sqrt(5)) bound
        sd[key] = v.reshape(shape)
    return sd
