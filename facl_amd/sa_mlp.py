"""Set-abstraction point-MLP (net3DV_1) as HIP passes: host-side orchestration.

Replaces nn.Sequential(Conv2d,BN2d,ReLU x3, MaxPool2d((1,K))) of cn3d_model_conbag.py:43-58.
The train-mode BatchNorm statistics are global over all positions, so the forward is a short
pipeline of kernels with (C,2) fp64 sum buffers between them; under DDP those buffers are the
SyncBN all-reduce points (``reduce_fn``).
"""
import torch

from . import _lib

_FWD_H3 = __import__("os").environ.get("FACL_FWD_H3", "1") != "0"
# eval mode through the ONE-kernel path (csrc/sa_eval.hip); FACL_EVAL_FUSED=0: the training passes with folded constants (A/B)
_EVAL_FUSED = __import__("os").environ.get("FACL_EVAL_FUSED", "1") != "0"
_EXTRA_LAUNCHES = int(__import__("os").environ.get("FACL_EXTRA_LAUNCHES", "0"))
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
UNIT = 64


class _Workspace:
    """Scratch for the partial-sum rows of the reducing kernels: one buffer per (device, stream).  A buffer is only ever
    touched by launches on the stream it belongs to, so calls on different streams (or threads, each with its own current
    stream) never share scratch; within one stream the launches are ordered."""
    _ws = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        if key not in cls._ws:
            n = _lib.load_library().facl_ws_bytes()
            cls._ws[key] = torch.empty(n, dtype=torch.uint8, device=device)
            cls._ws[key][-4096:].zero_()                 # ticket counters of the single-launch reduction (include/facl_hip.h)
        if _lib.POISON:
            cls._ws[key][:-4096].fill_(0xFF)             # NaN bytes: every partial row a reduction reads must be rewritten
        return cls._ws[key]


def _bn_finalize(sums, C, count, gamma, beta, running_mean, running_var, momentum=BN_MOMENTUM, aamax=None, zamax=None):
    """`aamax` (one row of _lib.amax_buffers): receives the bound of the layer's activation, the fp16x3 scale of the GEMM that
    consumes it (|gamma| sqrt(count - 1) sigma invstd + |beta|: csrc/finalize.hip)."""
    lib = _lib.load_library()
    bnc = _lib.empty((5, C), dtype=torch.float32, device=sums.device)
    _lib.check(lib.facl_bn_finalize(_lib.ptr(sums), C, float(count), _lib.ptr(gamma), _lib.ptr(beta), BN_EPS,
                                    momentum, _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(bnc),
                                    _lib.ptr(aamax), _lib.ptr(zamax), _lib.stream()), "facl_bn_finalize")
    return bnc


def act_amax_eval(y_rows, bnc, aamax):
    """Eval mode: running statistics say nothing about the data, so the activation maximum max relu(scale y + shift) is
    MEASURED (one streaming pass over the layer's raw output) into `aamax`."""
    lib = _lib.load_library()
    R, C = y_rows.shape
    _lib.check(lib.facl_rows_act_amax(_lib.ptr(y_rows), R, C, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(aamax), _lib.stream()),
               "facl_rows_act_amax")


_FRAG_ORDER = {}


def _frag_channel_order(dev):
    """Channel of every float of ONE position-pair... of a unit tile in the fragment layout (csrc/common.h), as a (4096,) index
    tensor: element ((ct*2 + rt)*4 + r4)*256 + lane*4 + e holds channel 32 rt + 8 r4 + 4 (lane >> 5) + e."""
    key = (dev.type, dev.index)
    if key not in _FRAG_ORDER:
        i = torch.arange(4096, device=dev)
        e, lane, r4, rt = i & 3, (i >> 2) & 63, (i >> 8) & 3, (i >> 10) & 1
        _FRAG_ORDER[key] = 32 * rt + 8 * r4 + 4 * (lane >> 5) + e
    return _FRAG_ORDER[key]


def _bn_eval(C, gamma, beta, running_mean, running_var):
    lib = _lib.load_library()
    bnc = _lib.empty((5, C), dtype=torch.float32, device=gamma.device)
    _lib.check(lib.facl_bn_eval_consts(C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                       _lib.ptr(running_var), BN_EPS, _lib.ptr(bnc), _lib.stream()),
               "facl_bn_eval_consts")
    return bnc


def _running_update(bnc, rm, rv, n_true):
    """Running-statistics update with an explicit element count (used when positions are replicated: the batch
    mean / biased variance are unaffected by uniform replication, the unbiased factor n/(n-1) is not)."""
    mean = bnc[0]
    var = 1.0 / (bnc[1].double() ** 2) - BN_EPS
    unb = (var * (n_true / max(n_true - 1.0, 1.0))).float()
    rm.mul_(1 - BN_MOMENTUM).add_(mean, alpha=BN_MOMENTUM)
    rv.mul_(1 - BN_MOMENTUM).add_(unb, alpha=BN_MOMENTUM)


def sa_mlp_forward(x_rows, p, training, reduce_fn=None, update_running=True, K=64, precision="f32"):
    """x_rows (P,D) contiguous fp32, P = groups*K rows.  ``p``: dict with W1,b1,g1,be1,rm1,rv1, ...3.
    Returns (pooled (groups,256), ctx) where ctx carries what the backward needs.
    ``reduce_fn(t)`` (optional) all-reduces an fp64 tensor in place (SyncBN).

    The kernels work on units of 64 positions.  K = 64 (the reference's live value, utils_my.py:260) maps one
    group to one unit.  Other K are served around the same kernels (cold path, tensor glue):
      * K = 64*R: a group spans R units; their per-unit maxima are merged before the pooling step and the
        sparse backward values are routed to the winning unit;
      * K | 64: every group is replicated 64/K times inside its unit.  Uniform replication leaves the batch mean,
        the biased variance, the max and (with P' = replicated count used consistently) every gradient unchanged;
        only the unbiased running-variance factor needs the true count."""
    lib = _lib.load_library()
    _lib.require_cuda(x_rows)
    for _ in range(_EXTRA_LAUNCHES):                 # experiment knob: what does one more tiny dependent kernel cost inside the step?
        _bn_eval(64, p["g1"], p["be1"], p["rm1"], p["rv1"])
    rep, R = 1, 1
    if K != UNIT:
        if K > UNIT and K % UNIT == 0:
            R = K // UNIT
        elif 0 < K < UNIT and UNIT % K == 0:
            rep = UNIT // K
            G_, D_ = x_rows.shape[0] // K, x_rows.shape[1]
            x_rows = x_rows.view(G_, K, D_).repeat(1, rep, 1).reshape(G_ * UNIT, D_).contiguous()
        else:
            raise NotImplementedError("knn_K must divide 64 or be a multiple of 64 (got %d)" % K)
    P, D = x_rows.shape
    if P % UNIT:
        raise ValueError("P must be a multiple of 64 (knn_K = 64)")
    if D not in (3, 4):
        raise NotImplementedError("INPUT_FEATURE_NUM must be 3 or 4 (got %d)" % D)
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
    # fp16x3 operand maxima (csrc/common.h): [0] max|x| (eval), [1] bound of a1, [2] bound of a2, [3] max(pooled)
    # training: [1], [2] are stored by facl_bn_finalize and [3] is zeroed by BN3's finalize call right before facl_sa_pool
    # raises it -- no fill launch; eval measures with atomics and starts from zeros
    amax = _lib.amax_buffers(4, dev, zero=not training)
    if training:
        mom = _lib.empty(D + D * D, **f64)
        _lib.check(lib.facl_sa_x_moments(_lib.ptr(x_rows), P, D, _lib.ptr(mom), _lib.ptr(ws), st), "facl_sa_x_moments")
        ctx["mom_local"] = mom
        if reduce_fn is not None:
            mom = reduce_fn(mom.clone())
            count = float(P) * reduce_fn.world_size          # every rank holds the same number of positions
        sums1 = _lib.empty((64, 2), **f64)
        direct = update_running and rep == 1
        n_true = count / rep
        rm, rv = (p["rm1"], p["rv1"]) if direct else (None, None)
        # sums of y1 from the moments -> BatchNorm-1 constants (+ running statistics, + the bound of a1) -> folded layer-1 table:
        # one single-wave launch (facl_sa_bn1_chain; bit-identical to facl_bn1_sums_from_moments + facl_bn_finalize + facl_sa_l1tab)
        bnc1 = _lib.empty((5, 64), dtype=torch.float32, device=dev)
        l1tab = _lib.empty((64, 8), dtype=torch.float32, device=dev)
        _lib.check(lib.facl_sa_bn1_chain(_lib.ptr(mom), count, D, _lib.ptr(W1), _lib.ptr(p["b1"]), _lib.ptr(p["g1"]), _lib.ptr(p["be1"]),
                                         BN_EPS, BN_MOMENTUM, _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(sums1), _lib.ptr(bnc1),
                                         _lib.ptr(amax[1]), _lib.ptr(l1tab), st), "facl_sa_bn1_chain")
        if update_running and not direct:
            _running_update(bnc1, p["rm1"], p["rv1"], n_true)
        ctx["mom"] = mom
    else:
        bnc1 = _bn_eval(64, p["g1"], p["be1"], p["rm1"], p["rv1"])
        xa = amax[0]                                       # eval: the bound of a1 follows from max|x| (facl_sa_l1tab)
        _lib.check(lib.facl_absmax(_lib.ptr(x_rows), x_rows.numel(), _lib.ptr(xa), st), "facl_absmax")
        l1tab = _lib.empty((64, 8), dtype=torch.float32, device=dev)
        _lib.check(lib.facl_sa_l1tab(_lib.ptr(W1), _lib.ptr(p["b1"]), D, _lib.ptr(bnc1[2]), _lib.ptr(bnc1[3]),
                                     _lib.ptr(l1tab), _lib.ptr(xa), _lib.ptr(amax[1]), st), "facl_sa_l1tab")
    if not training and _EVAL_FUSED and precision in ("f32", "x3b") and _FWD_H3:
        # eval mode = the extraction path (extract_motion_feature.py:143-221): every BatchNorm is a constant affine, so the whole
        # block is ONE kernel per unit, x -> pooled, with nothing stored in between (csrc/sa_eval.hip)
        bnc2 = _bn_eval(64, p["g2"], p["be2"], p["rm2"], p["rv2"])
        bnc3 = _bn_eval(256, p["g3"], p["be3"], p["rm3"], p["rv3"])
        pooled_u = _lib.empty((nunits, 256), dtype=torch.float32, device=dev)
        with _lib.timed("facl_sa_eval"):
            _lib.check(lib.facl_sa_eval(_lib.ptr(x_rows), nunits, D, _lib.ptr(l1tab), _lib.ptr(W2), _lib.ptr(p["b2"]),
                                        _lib.ptr(bnc2[2]), _lib.ptr(bnc2[3]), _lib.ptr(W3), _lib.ptr(p["b3"]), _lib.ptr(bnc3[2]),
                                        _lib.ptr(bnc3[3]), _lib.ptr(pooled_u), _lib.ptr(amax[1]), st), "facl_sa_eval")
        if R > 1:      # a group spans R units: BN + ReLU are monotone per channel, so the group's feature is the maximum of its units'
            pooled_u = pooled_u.view(nunits // R, R, 256).max(dim=1).values.contiguous()
        return pooled_u, {"amax": None, "nunits": nunits, "D": D, "R": R, "x_rows": x_rows}
    y2f = _lib.empty(nunits * UNIT * 64, dtype=torch.float32, device=dev)
    sums2 = _lib.empty((64, 2), **f64) if training else None
    with _lib.timed("facl_sa_fwd2"):
        _lib.check(lib.facl_sa_fwd2(_lib.ptr(x_rows), nunits, D, _lib.ptr(l1tab), _lib.ptr(W2), _lib.ptr(p["b2"]),
                                    _lib.ptr(y2f), _lib.ptr(sums2), _lib.ptr(ws), _lib.ptr(amax[1]), st), "facl_sa_fwd2")
    if training:
        if reduce_fn is not None:
            reduce_fn(sums2)
        rm, rv = (p["rm2"], p["rv2"]) if direct else (None, None)
        bnc2 = _bn_finalize(sums2, 64, count, p["g2"], p["be2"], rm, rv, aamax=amax[2])
        if update_running and not direct:
            _running_update(bnc2, p["rm2"], p["rv2"], n_true)
    else:
        bnc2 = _bn_eval(64, p["g2"], p["be2"], p["rm2"], p["rv2"])
        # eval: measured maximum of a2.  The fragment layout keeps 4 consecutive channels per float4 and the 64 channels of a
        # position in 16 float4s whose channel base is a function of the float4's index alone -- the row kernel only needs the
        # per-float4 constants in that order
        lay = _frag_channel_order(dev)
        sc_l, sh_l = bnc2[2][lay].contiguous(), bnc2[3][lay].contiguous()
        _lib.check(lib.facl_rows_act_amax(_lib.ptr(y2f), nunits, 4096, _lib.ptr(sc_l), _lib.ptr(sh_l), _lib.ptr(amax[2]), st),
                   "facl_rows_act_amax")
    sgn3 = p["g3"]                                                    # the kernels take sign(gamma3) themselves (sign(0) = +1)
    ymax = _lib.empty((nunits, 256), dtype=torch.float32, device=dev)
    arg = _lib.empty((nunits, 256), dtype=torch.uint8, device=dev)
    sums3 = _lib.empty((256, 2), **f64) if training else None
    with _lib.timed("facl_sa_fwd3"):
        # dense configuration: fp16-input 64->256 layer; "x3": the opt-in three-product variant (tail.precision)
        # "f32" / "x3b": the fp32-grade forward -- fp16x3 (csrc/common.h; FACL_FWD_H3=0 selects bf16x6 for A/B)
        f32_fwd3 = lib.facl_sa_fwd3_h3 if _FWD_H3 else lib.facl_sa_fwd3
        fwd3 = {"f32": f32_fwd3, "f16": lib.facl_sa_fwd3_f16, "x3": lib.facl_sa_fwd3_x3, "x3b": f32_fwd3}[precision]
        extra = (_lib.ptr(amax[2]),) if fwd3 is lib.facl_sa_fwd3_h3 else ()
        _lib.check(fwd3(_lib.ptr(y2f), nunits, _lib.ptr(bnc2[2]), _lib.ptr(bnc2[3]), _lib.ptr(W3),
                                    _lib.ptr(p["b3"]), _lib.ptr(sgn3), _lib.ptr(ymax), _lib.ptr(arg), _lib.ptr(sums3),
                                    _lib.ptr(ws), *extra, st), "facl_sa_fwd3")
    _lib.tap("sa_arg", arg)
    if training:
        if reduce_fn is not None:
            reduce_fn(sums3)
        rm, rv = (p["rm3"], p["rv3"]) if direct else (None, None)
        bnc3 = _bn_finalize(sums3, 256, count, p["g3"], p["be3"], rm, rv, zamax=amax[3])
        if update_running and not direct:
            _running_update(bnc3, p["rm3"], p["rv3"], n_true)
    else:
        bnc3 = _bn_eval(256, p["g3"], p["be3"], p["rm3"], p["rv3"])
    ngroups, ymax_g, rsel = nunits, ymax, None
    if R > 1:                                    # a group spans R units: merge their maxima (first unit wins ties)
        ngroups = nunits // R
        ymax_g, rsel = ymax.view(ngroups, R, 256).max(dim=1)
        ymax_g = ymax_g.contiguous()
    pooled = _lib.empty((ngroups, 256), dtype=torch.float32, device=dev)
    _lib.check(lib.facl_sa_pool(_lib.ptr(ymax_g), ngroups, 256, _lib.ptr(bnc3[2]), _lib.ptr(bnc3[3]), _lib.ptr(pooled),
                                _lib.ptr(amax[3]), st), "facl_sa_pool")
    if _lib.TAPS is not None and K == UNIT:          # tests only: the ReLU decisions of the three layers (tie-proof parity)
        xd, tb = x_rows.double(), l1tab.double()
        v = (tb[:, 0] * xd[:, :1] + tb[:, 4]).float().double()                # the kernels' sequential fp32 FMA chain
        for i in range(1, D):
            v = (tb[:, i] * xd[:, i:i + 1] + v).float().double()
        _lib.TAPS["relu_sa1"] = v > 0
        y2n = y2f.view(nunits, 2, 2, 4, 2, 32, 4).permute(0, 1, 5, 2, 3, 4, 6).reshape(nunits * UNIT, 64)   # fragment layout -> (P, 64)
        _lib.tap_relu("relu_sa2", y2n, bnc2[2], bnc2[3])
        _lib.tap_relu("relu_sa3", a=pooled)
    ctx["amax"] = amax                               # [1], [2]: the backward's operand scales; [3]: max(pooled) for the next GEMM
    ctx.update(y2f=y2f, ymax=ymax_g, arg=arg, bnc1=bnc1, bnc2=bnc2, bnc3=bnc3, sgn3=sgn3, l1tab=l1tab, count=count,
               nunits=nunits, D=D, R=R, rsel=rsel, ngroups=ngroups, x_rows=x_rows)
    return pooled, ctx


def sa_mlp_backward(ctx, dpooled, x_rows, p, reduce_fn=None):
    x_rows = ctx["x_rows"]                       # the (possibly replicated) rows the forward actually ran on
    return _sa_mlp_backward(ctx, dpooled, x_rows, p, reduce_fn)


def _sa_mlp_backward(ctx, dpooled, x_rows, p, reduce_fn=None):
    """Gradients of the 12 parameters of net3DV_1 given dL/dpooled (P/64,256).

    Four heavy passes (csrc/sa_bwd.hip) with three fp64 closed-form assembly kernels between them
    (csrc/finalize.hip).  ``reduce_fn`` all-reduces the BN-backward sums (SyncBN); parameter gradients stay
    LOCAL sums (the data-parallel wrapper averages them)."""
    lib = _lib.load_library()
    dev = x_rows.device
    st = _lib.stream()
    ws = _Workspace.get(dev)
    f64 = dict(dtype=torch.float64, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    nunits, D, P = ctx["nunits"], ctx["D"], float(ctx["count"])
    bnc1, bnc2, bnc3 = ctx["bnc1"], ctx["bnc2"], ctx["bnc3"]
    W1 = p["W1"].reshape(64, D)
    W2 = p["W2"].reshape(64, 64)
    W3 = p["W3"].reshape(256, 64)
    dpooled = dpooled.contiguous()
    ptr = _lib.ptr

    # ---- pass 0: sparse values + (dbeta3, dgamma3)
    ngroups, R = ctx["ngroups"], ctx["R"]
    coef = _lib.empty((ngroups, 256), **f32)
    sums0 = _lib.empty((256, 2), **f64)
    _lib.check(lib.facl_sa_bwd0(ptr(dpooled), ptr(ctx["ymax"]), ngroups, ptr(bnc3), ptr(coef), ptr(sums0), ptr(ws), st),
               "facl_sa_bwd0")
    if R > 1:                                    # route each group's sparse value to the unit that won the max
        cu = torch.zeros((ngroups, R, 256), **f32)
        cu.scatter_(1, ctx["rsel"].unsqueeze(1), coef.unsqueeze(1))
        coef = cu.view(nunits, 256)
    sums0_l = sums0
    if reduce_fn is not None:
        sums0 = reduce_fn(sums0.clone())
    G3, h3 = _lib.empty((64, 64), **f32), _lib.empty(64, **f32)
    _lib.check(lib.facl_sa_bwd_consts3(ptr(sums0), ptr(bnc3), ptr(W3), ptr(p["b3"]), P, ptr(G3), ptr(h3), st),
               "facl_sa_bwd_consts3")

    # ---- pass 1: dz2 + (dbeta2, dgamma2)
    dz2f = _lib.empty_like(ctx["y2f"])
    sums1 = _lib.empty((64, 2), **f64)
    with _lib.timed("facl_sa_bwd1"):
        _lib.check(lib.facl_sa_bwd1(ptr(ctx["y2f"]), nunits, ptr(bnc2), ptr(G3), ptr(h3), ptr(W3), ptr(coef),
                                    ptr(ctx["arg"]), ptr(dz2f), ptr(sums1), ptr(ws), ptr(ctx["amax"][2]), st), "facl_sa_bwd1")
    sums1_l = sums1
    if reduce_fn is not None:
        sums1 = reduce_fn(sums1.clone())

    # ---- pass 2: dy2, da1, dz1, dW2, R1 -- launched right behind pass 1 and walking the units backwards: the ends of dz2 / y2
    # that pass 1 touched last are still in the memory-side cache (FACL_BWD2_REV=0: front to back)
    bw2 = _lib.empty((4, 64), **f32)
    _lib.check(lib.facl_sa_bwd_consts2(ptr(sums1), ptr(bnc2), P, ptr(bw2), st), "facl_sa_bwd_consts2")
    out2 = _lib.empty(64 * 64 + 8 * 64, **f64)
    with _lib.timed("facl_sa_bwd2"):
        _lib.check(lib.facl_sa_bwd2(ptr(dz2f), ptr(ctx["y2f"]), ptr(x_rows), nunits, D, ptr(bw2), ptr(W2), ptr(ctx["l1tab"]),
                                    ptr(out2), ptr(ws), ptr(ctx["amax"][1]), st), "facl_sa_bwd2")
    # ---- layer-3 weight gradient ingredients: sparse gather, Gram, sum a2
    out3 = _lib.empty(256 * 64 + 64 * 64 + 64, **f64)
    with _lib.timed("facl_sa_bwd_w3"):
        _lib.check(lib.facl_sa_bwd_w3(ptr(ctx["y2f"]), nunits, ptr(bnc2), ptr(coef), ptr(ctx["arg"]), ptr(out3), ptr(ws),
                                      ptr(ctx["amax"][2]), st), "facl_sa_bwd_w3")

    R1_g = out2[4096:]
    if reduce_fn is not None:
        R1_g = reduce_fn(R1_g.clone())

    g = {"W3": _lib.empty_like(p["W3"]), "g3": _lib.empty(256, **f32), "be3": _lib.empty(256, **f32),
         "W2": _lib.empty_like(p["W2"]), "g2": _lib.empty(64, **f32), "be2": _lib.empty(64, **f32),
         "W1": _lib.empty_like(p["W1"]), "g1": _lib.empty(64, **f32), "be1": _lib.empty(64, **f32),
         # d(bias) of a conv that feeds a train-mode BN is identically zero: None (the parameter is left untouched)
         "b1": None, "b2": None, "b3": None}
    _lib.check(lib.facl_sa_bwd_final(ptr(out3), ptr(sums0), ptr(sums0_l), ptr(bnc3), ptr(W3), ptr(p["b3"]), ptr(out2),
                                     ptr(sums1_l), ptr(R1_g), ptr(ctx["mom_local"]), ptr(bnc1), ptr(W1), ptr(p["b1"]),
                                     D, P, ptr(g["W3"]), ptr(g["g3"]), ptr(g["be3"]), ptr(g["W2"]), ptr(g["g2"]),
                                     ptr(g["be2"]), ptr(g["W1"]), ptr(g["g1"]), ptr(g["be1"]), st), "facl_sa_bwd_final")
    return g


_PARAM_ORDER = ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3", "g3", "be3")


class SAMLPFunction(torch.autograd.Function):
    """pooled = net3DV_1(x) with HIP forward and backward.  ``state`` carries the BN buffers
    (updated in place in training mode), the mode flag and the optional SyncBN reduce_fn."""

    @staticmethod
    def forward(ctx, x_rows, state, *params):
        p = dict(zip(_PARAM_ORDER, [t.detach().contiguous() for t in params]))
        p.update(state["buffers"])
        pooled, c = sa_mlp_forward(x_rows, p, state["training"], state.get("reduce_fn"), K=state.get("K", UNIT),
                                   precision=state.get("precision", "f32"))
        # max(pooled): the fp16x3 scale of the GEMM that consumes the features (the fused eval kernel keeps none: measured there)
        state["pooled_amax"] = None if c.get("amax") is None else c["amax"][3]
        ctx.c, ctx.p, ctx.x_rows, ctx.reduce_fn = c, p, x_rows, state.get("reduce_fn")
        ctx.training = state["training"]
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        g = sa_mlp_backward(ctx.c, dpooled, ctx.x_rows, ctx.p, ctx.reduce_fn)
        return (None, None) + tuple(g[k] for k in _PARAM_ORDER)
