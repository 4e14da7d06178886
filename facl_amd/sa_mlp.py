"""Set-abstraction point-MLP (net3DV_1) as HIP passes: host-side orchestration.

Replaces nn.Sequential(Conv2d,BN2d,ReLU x3, MaxPool2d((1,K))) of cn3d_model_conbag.py:43-58.
The train-mode BatchNorm statistics are global over all positions, so the forward is a short
pipeline of kernels with (C,2) fp64 sum buffers between them; under DDP those buffers are the
SyncBN all-reduce points (``reduce_fn``).
"""
import torch

from . import _lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
UNIT = 64


class _Workspace:
    """One scratch buffer per device for the partial-sum rows of the reducing kernels."""
    _ws = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        if key not in cls._ws:
            n = _lib.load_library().facl_ws_bytes()
            cls._ws[key] = torch.empty(n, dtype=torch.uint8, device=device)
        return cls._ws[key]


def _bn_finalize(sums, C, count, gamma, beta, running_mean, running_var, momentum=BN_MOMENTUM):
    lib = _lib.load_library()
    bnc = torch.empty((5, C), dtype=torch.float32, device=sums.device)
    _lib.check(lib.facl_bn_finalize(_lib.ptr(sums), C, float(count), _lib.ptr(gamma), _lib.ptr(beta), BN_EPS,
                                    momentum, _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(bnc),
                                    _lib.stream()), "facl_bn_finalize")
    return bnc


def _bn_eval(C, gamma, beta, running_mean, running_var):
    lib = _lib.load_library()
    bnc = torch.empty((5, C), dtype=torch.float32, device=gamma.device)
    _lib.check(lib.facl_bn_eval_consts(C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                       _lib.ptr(running_var), BN_EPS, _lib.ptr(bnc), _lib.stream()),
               "facl_bn_eval_consts")
    return bnc


def sa_mlp_forward(x_rows, p, training, reduce_fn=None, update_running=True):
    """x_rows (P,D) contiguous fp32 (P = groups*64).  ``p``: dict with W1,b1,g1,be1,rm1,rv1, ...3.
    Returns (pooled (P/64,256), ctx) where ctx carries what the backward needs.
    ``reduce_fn(t)`` (optional) all-reduces an fp64 tensor in place (SyncBN); counts are then the
    global ones: ``reduce_fn`` must also be applied to the count, handled here."""
    lib = _lib.load_library()
    _lib.require_cuda(x_rows)
    P, D = x_rows.shape
    if P % UNIT:
        raise ValueError("P must be a multiple of 64 (knn_K = 64)")
    nunits = P // UNIT
    dev = x_rows.device
    st = _lib.stream()
    ws = _Workspace.get(dev)
    f64 = dict(dtype=torch.float64, device=dev)
    count = float(P)
    W1 = p["W1"].reshape(64, D)
    W2 = p["W2"].reshape(64, 64)
    W3 = p["W3"].reshape(256, 64)
    ctx = {}
    if training:
        mom = torch.empty(D + D * D + 1, **f64)
        mom[-1] = count
        _lib.check(lib.facl_sa_x_moments(_lib.ptr(x_rows), P, D, _lib.ptr(mom), _lib.ptr(ws), st), "facl_sa_x_moments")
        if reduce_fn is not None:
            reduce_fn(mom)
            count = float(mom[-1].item())
        sums1 = torch.empty((64, 2), **f64)
        _lib.check(lib.facl_bn1_sums_from_moments(_lib.ptr(mom), count, D, _lib.ptr(W1), _lib.ptr(p["b1"]),
                                                  _lib.ptr(sums1), st), "facl_bn1_sums_from_moments")
        rm, rv = (p["rm1"], p["rv1"]) if update_running else (None, None)
        bnc1 = _bn_finalize(sums1, 64, count, p["g1"], p["be1"], rm, rv)
        ctx["mom"] = mom
    else:
        bnc1 = _bn_eval(64, p["g1"], p["be1"], p["rm1"], p["rv1"])
    l1tab = torch.empty((64, 8), dtype=torch.float32, device=dev)
    _lib.check(lib.facl_sa_l1tab(_lib.ptr(W1), _lib.ptr(p["b1"]), D, _lib.ptr(bnc1[2]), _lib.ptr(bnc1[3]),
                                 _lib.ptr(l1tab), st), "facl_sa_l1tab")
    y2f = torch.empty(nunits * UNIT * 64, dtype=torch.float32, device=dev)
    sums2 = torch.empty((64, 2), **f64) if training else None
    _lib.check(lib.facl_sa_fwd2(_lib.ptr(x_rows), nunits, D, _lib.ptr(l1tab), _lib.ptr(W2), _lib.ptr(p["b2"]),
                                _lib.ptr(y2f), _lib.ptr(sums2), _lib.ptr(ws), st), "facl_sa_fwd2")
    if training:
        if reduce_fn is not None:
            reduce_fn(sums2)
        rm, rv = (p["rm2"], p["rv2"]) if update_running else (None, None)
        bnc2 = _bn_finalize(sums2, 64, count, p["g2"], p["be2"], rm, rv)
    else:
        bnc2 = _bn_eval(64, p["g2"], p["be2"], p["rm2"], p["rv2"])
    sgn3 = torch.where(p["g3"] < 0, -1.0, 1.0).to(torch.float32)
    ymax = torch.empty((nunits, 256), dtype=torch.float32, device=dev)
    arg = torch.empty((nunits, 256), dtype=torch.uint8, device=dev)
    sums3 = torch.empty((256, 2), **f64) if training else None
    _lib.check(lib.facl_sa_fwd3(_lib.ptr(y2f), nunits, _lib.ptr(bnc2[2]), _lib.ptr(bnc2[3]), _lib.ptr(W3),
                                _lib.ptr(p["b3"]), _lib.ptr(sgn3), _lib.ptr(ymax), _lib.ptr(arg), _lib.ptr(sums3),
                                _lib.ptr(ws), st), "facl_sa_fwd3")
    if training:
        if reduce_fn is not None:
            reduce_fn(sums3)
        # statistics were taken of sgn3*y3: flip the channel sums back (sum of squares is unchanged)
        sums3[:, 0] *= sgn3.double()
        rm, rv = (p["rm3"], p["rv3"]) if update_running else (None, None)
        bnc3 = _bn_finalize(sums3, 256, count, p["g3"], p["be3"], rm, rv)
    else:
        bnc3 = _bn_eval(256, p["g3"], p["be3"], p["rm3"], p["rv3"])
    pooled = torch.empty((nunits, 256), dtype=torch.float32, device=dev)
    _lib.check(lib.facl_sa_pool(_lib.ptr(ymax), nunits, 256, _lib.ptr(bnc3[2]), _lib.ptr(bnc3[3]), _lib.ptr(pooled), st),
               "facl_sa_pool")
    ctx.update(y2f=y2f, ymax=ymax, arg=arg, bnc1=bnc1, bnc2=bnc2, bnc3=bnc3, sgn3=sgn3, l1tab=l1tab, count=count,
               nunits=nunits, D=D)
    return pooled, ctx
