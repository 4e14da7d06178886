"""Encoder tail (net3DV_3, my_max_pool, netR_FC: cn3d_model_conbag.py:61-88, :199-207).

Per layer: the dense 1x1 channel contraction runs on the hand-written fp32 MFMA GEMM (csrc/gemm.hip: forward with
fused bias / centre term / BN column statistics, dgrad, split-K wgrad); train-mode BN finalisation (fp64 sums,
SyncBN hook), BN+ReLU apply, BN+ReLU+max over the S centroids without materialising the activation, and the
matching backward passes are the row kernels of csrc/rows.hip.
"""
import threading

import torch

from . import _lib
from .sa_mlp import BN_EPS, BN_MOMENTUM, _Workspace, _bn_eval, _bn_finalize, act_amax_eval


# ---- arithmetic of the dense contractions -------------------------------------------------------------------------------
# "f32": fp32-grade results (exact 3-way bf16 split on the bf16 MFMA, or the fp32 MFMA with FACL_GEMM_F32=1);
# "f16": fp16-input MFMA with fp32 accumulation (dense configuration);
# "x3":  opt-in "bf16x3" -- two bf16 pieces per operand, three products per multiply-add (~1e-5 relative on a GEMM result,
#        inside the north_star's 1e-4 for features / loss; half the MFMA work).  Never the default.
# "x3b": "x3" in the BACKWARD GEMMs only (dgrad / wgrad): forward, features and loss are those of "f32" to the bit, the
#        gradients carry ~1e-5 relative rounding (far below what Adam's update noise of the reference itself is).
# The precision in force when a layer's FORWARD runs is recorded on its autograd context and used again by its backward GEMMs.
PRECISIONS = ("f32", "f16", "x3", "x3b")
_SUFFIX = {"f32": "", "f16": "_f16", "x3": "_x3", "x3b": ""}
_LABEL = {"f32": "", "f16": " f16", "x3": " x3", "x3b": ""}
_SCOPE = threading.local()                  # per-thread scope of `with precision(...)`: read by FORWARD passes only


class precision:
    """``with precision("x3b"): ...`` -- the arithmetic the forward passes entered inside the block record on their
    autograd context.  Thread-local; the backward never consults it (each pass hands its recorded value to its GEMMs
    explicitly), so an exception or an out-of-order backward cannot leave a stale mode behind."""

    def __init__(self, p):
        if p not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.p = p

    def __enter__(self):
        stack = _SCOPE.__dict__.setdefault("stack", [])
        stack.append(self.p)
        return self

    def __exit__(self, *exc):
        _SCOPE.stack.pop()
        return False


def current_precision():
    stack = getattr(_SCOPE, "stack", None)
    return stack[-1] if stack else "f32"


def backward_precision(p):
    """Arithmetic of the backward GEMMs of a pass whose forward ran under ``p`` ("x3b" = three products there only)."""
    return "x3" if p == "x3b" else p


def _fn(lib, name, prec):
    return getattr(lib, name + _SUFFIX[prec])


def gemm_fwd(a, W, bias, want_stats=False, centers=None, Wc=None, prec=None):
    """y = a W^T + bias [+ centers Wc^T] on the hand-written fp32 MFMA GEMM (csrc/gemm.hip); optional fused
    column (sum, sumsq) for the BatchNorm that follows."""
    lib = _lib.load_library()
    M, K = a.shape
    N = W.shape[0]
    y = _lib.empty((M, N), dtype=torch.float32, device=a.device)
    sums = _lib.empty((N, 2), dtype=torch.float64, device=a.device) if want_stats else None
    ws = _Workspace.get(a.device)
    prec = current_precision() if prec is None else prec
    with _lib.timed("facl_gemm_fwd %dx%dx%d%s" % (M, K, N, _LABEL[prec])):
        _lib.check(_fn(lib, "facl_gemm_fwd", prec)(_lib.ptr(a), M, K, _lib.ptr(W), W.stride(0), N, _lib.ptr(bias), None, None,
                                     _lib.ptr(centers), _lib.ptr(Wc), 3 if Wc is not None else 0, _lib.ptr(y),
                                     _lib.ptr(sums), _lib.ptr(ws), _lib.stream()), "facl_gemm_fwd")
    return y, sums


def gemm_dgrad(dy, W, prec=None):
    """da = dy W   (W (N,K) row-major, possibly a column slice of a wider matrix)."""
    lib = _lib.load_library()
    M, N = dy.shape
    K = W.shape[1]
    da = _lib.empty((M, K), dtype=torch.float32, device=dy.device)
    prec = current_precision() if prec is None else prec
    with _lib.timed("facl_gemm_dgrad %dx%dx%d%s" % (M, N, K, _LABEL[prec])):
        _lib.check(_fn(lib, "facl_gemm_dgrad", prec)(_lib.ptr(dy), M, N, W.data_ptr(), W.stride(0), K, _lib.ptr(da), _lib.stream()),
                   "facl_gemm_dgrad")
    return da


def gemm_wgrad(dy, a, prec=None, amax=None, amax_b=None):
    """dW = dy^T a, contraction over the rows split into slices (deterministic slice-order sum).  `amax` (the device-side
    max|dy| buffer of facl_rows_bwd_apply_amax) + `amax_b` (the bound of max|a| its forward GEMM used): fp16x3 arithmetic
    where the 128x128-tile kernel serves the shape."""
    lib = _lib.load_library()
    M, N = dy.shape
    K = a.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    nz = max(1, min((M + 255) // 256, 512 // tiles))     # tiles * nz = one resident wave of workgroups (2 per CU)
    dW = _lib.empty((N, K), dtype=torch.float32, device=dy.device)
    slices = _lib.empty(nz * N * K, dtype=torch.float32, device=dy.device)
    prec = current_precision() if prec is None else prec
    if amax is not None and amax_b is not None and prec == "f32":
        with _lib.timed("facl_gemm_wgrad %dx%dx%d h3" % (M, N, K)):
            rc = lib.facl_gemm_wgrad_h3(_lib.ptr(dy), _lib.ptr(a), M, N, K, a.stride(0), None, None, _lib.ptr(amax), _lib.ptr(amax_b),
                                        _lib.ptr(dW), _lib.ptr(slices), nz, _lib.stream())
        if rc != -4:
            _lib.check(rc, "facl_gemm_wgrad_h3")
            return dW
    with _lib.timed("facl_gemm_wgrad %dx%dx%d%s" % (M, N, K, _LABEL[prec])):
        _lib.check(_fn(lib, "facl_gemm_wgrad", prec)(_lib.ptr(dy), _lib.ptr(a), M, N, K, a.stride(0), _lib.ptr(dW), _lib.ptr(slices), nz,
                                       _lib.stream()), "facl_gemm_wgrad")
    return dW


def _stats(y, ws):
    lib = _lib.load_library()
    R, C = y.shape
    sums = _lib.empty((C, 2), dtype=torch.float64, device=y.device)
    _lib.check(lib.facl_rows_stats(_lib.ptr(y), R, C, _lib.ptr(sums), _lib.ptr(ws), _lib.stream()), "facl_rows_stats")
    return sums


def _bn_bwd_consts(sums, C, count, reduce_fn):
    """(dbeta, dgamma) fp64 sums -> fp32 parameter gradients of THIS rank + kk (2,C) = SyncBN-reduced sums / P."""
    lib = _lib.load_library()
    f32 = dict(dtype=torch.float32, device=sums.device)
    dbeta, dgamma, kk = _lib.empty(C, **f32), _lib.empty(C, **f32), _lib.empty((2, C), **f32)
    sums_g = reduce_fn(sums.clone()) if reduce_fn is not None else sums
    _lib.check(lib.facl_bn_bwd_consts(_lib.ptr(sums), _lib.ptr(sums_g), C, float(count), _lib.ptr(dbeta), _lib.ptr(dgamma),
                                      _lib.ptr(kk), _lib.stream()), "facl_bn_bwd_consts")
    return dbeta, dgamma, kk


def _forward_bn_consts(y, bn, training, reduce_fn, ws, sums=None, aamax=None):
    """Statistics -> (5,C) constants; updates running stats / num_batches_tracked in training mode.  `aamax` (optional):
    receives the bound (training: BatchNorm's own; eval: measured) of the layer's activation for the fp16x3 GEMM behind it."""
    R, C = y.shape
    if not training:
        bnc = _bn_eval(C, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var)
        if aamax is not None:
            act_amax_eval(y, bnc, aamax)
        return bnc, float(R)
    if sums is None:
        sums = _stats(y, ws)
    count = float(R)
    if reduce_fn is not None:
        reduce_fn(sums)
        count = count * reduce_fn.world_size                  # equal shards (no host round trip for the count)
    bnc = _bn_finalize(sums, C, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, aamax=aamax)
    bn.count_batch()
    return bnc, count


class _LinearBNReLU(torch.autograd.Function):
    """relu(bn(h W^T + b)); with ``centers`` the input is the (never materialised) torch.cat((yt, xt), 1) of
    cn3d_model_conbag.py:219: W's first 3 columns act on the centroid xyz, the rest on ``h``."""

    @staticmethod
    def forward(ctx, h, W, b, gamma, beta, bn, training, reduce_fn, centers=None):
        ctx.prec = current_precision()
        lib = _lib.load_library()
        _lib.require_cuda(h)
        ws = _Workspace.get(h.device)
        h = h.contiguous()
        if centers is not None:
            Wc, Wh = W[:, :3].contiguous(), W[:, 3:].contiguous()
            centers = centers.contiguous()
        else:
            Wc, Wh = None, W.contiguous()
        y, sums = gemm_fwd(h, Wh, b, want_stats=training, centers=centers, Wc=Wc, prec=ctx.prec)
        bnc, count = _forward_bn_consts(y, bn, training, reduce_fn, ws, sums)
        ctx.centers = centers
        ctx.Wh = Wh
        R, C = y.shape
        a = _lib.empty_like(y)
        _lib.check(lib.facl_rows_bn_relu(_lib.ptr(y), R, C, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(a),
                                         _lib.stream()), "facl_rows_bn_relu")
        _lib.tap_relu("relu_t%d" % (1 if centers is not None else 2), a=a)
        ctx.save_for_backward(h, W, y, bnc)
        ctx.count, ctx.reduce_fn, ctx.training = count, reduce_fn, training
        return a

    @staticmethod
    def backward(ctx, da):
        bp = backward_precision(ctx.prec)   # the backward GEMMs run in the arithmetic the forward recorded
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        lib = _lib.load_library()
        h, W, y, bnc = ctx.saved_tensors
        ws = _Workspace.get(y.device)
        R, C = y.shape
        da = da.contiguous()
        sums = _lib.empty((C, 2), dtype=torch.float64, device=y.device)
        _lib.check(lib.facl_rows_bwd_stats(_lib.ptr(da), _lib.ptr(y), R, C, _lib.ptr(bnc), _lib.ptr(sums), _lib.ptr(ws),
                                           _lib.stream()), "facl_rows_bwd_stats")
        dbeta, dgamma, kk = _bn_bwd_consts(sums, C, ctx.count, ctx.reduce_fn)   # parameter gradients stay local sums
        dy = _lib.empty_like(y)
        _lib.check(lib.facl_rows_bwd_apply(_lib.ptr(da), _lib.ptr(y), R, C, _lib.ptr(bnc), _lib.ptr(kk), _lib.ptr(dy),
                                           _lib.stream()), "facl_rows_bwd_apply")
        dW = gemm_wgrad(dy, h, prec=bp)
        if ctx.centers is not None:                                     # xyz columns (C,3): one streaming pass over dy
            dWc = _lib.empty((C, 3), dtype=torch.float64, device=y.device)
            _lib.check(lib.facl_rows_center_wgrad(_lib.ptr(dy), _lib.ptr(ctx.centers), R, C, _lib.ptr(dWc), _lib.ptr(ws),
                                                  _lib.stream()), "facl_rows_center_wgrad")
            dW = torch.cat((dWc.float(), dW), dim=1)
        dh = gemm_dgrad(dy, ctx.Wh, prec=bp) if ctx.needs_input_grad[0] else None
        # d(bias) is identically zero in front of a train-mode BN: None leaves the parameter untouched
        return dh, dW, None, dgamma, dbeta, None, None, None, None


class _LinearBNSegmax(torch.autograd.Function):
    """x_pre = max over the S centroids of relu(bn(h W^T + b)) -- cn3d_model_conbag.py:71-73 + :199/:222."""

    @staticmethod
    def forward(ctx, h, W, b, gamma, beta, bn, training, reduce_fn, S):
        ctx.prec = current_precision()
        lib = _lib.load_library()
        _lib.require_cuda(h)
        ws = _Workspace.get(h.device)
        h = h.contiguous()
        W = W.contiguous()
        R, C = h.shape[0], W.shape[0]
        M = R // S
        xpre = _lib.empty((M, C), dtype=torch.float32, device=h.device)
        arg = _lib.empty((M, C), dtype=torch.int32, device=h.device)
        fused = False
        if S == 64 and R % 64 == 0:
            # max over the 64 centroid rows of each cloud inside the GEMM epilogue (sign(gamma) is known before the
            # statistics are; BN + ReLU are monotone per channel): y is not re-read by a pooling pass
            sgn = gamma.detach()                              # the kernel takes sign(gamma) itself (sign(0) = +1)
            y = _lib.empty((R, C), dtype=torch.float32, device=h.device)
            sums = _lib.empty((C, 2), dtype=torch.float64, device=h.device) if training else None
            ymax = _lib.empty((M, C), dtype=torch.float32, device=h.device)
            with _lib.timed("facl_gemm_fwd %dx%dx%d%s" % (R, h.shape[1], C, _LABEL[ctx.prec])):
                rc = _fn(lib, "facl_gemm_fwd_segmax", ctx.prec)(_lib.ptr(h), R, h.shape[1], _lib.ptr(W), W.stride(0), C, _lib.ptr(b),
                                              _lib.ptr(sgn), _lib.ptr(y), _lib.ptr(sums), _lib.ptr(ymax), _lib.ptr(arg),
                                              _lib.ptr(ws), _lib.stream())
            if rc == 0:
                fused = True
                bnc, count = _forward_bn_consts(y, bn, training, reduce_fn, ws, sums)
                _lib.check(lib.facl_sa_pool(_lib.ptr(ymax), M, C, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(xpre), None,
                                            _lib.stream()), "facl_sa_pool")
            elif rc != -4:                                   # FACL_E_CONFIG: too small for the fused kernel
                _lib.check(rc, "facl_gemm_fwd_segmax")
        if not fused:
            y, sums = gemm_fwd(h, W, b, want_stats=training, prec=ctx.prec)
            bnc, count = _forward_bn_consts(y, bn, training, reduce_fn, ws, sums)
            _lib.check(lib.facl_rows_segmax(_lib.ptr(y), M, S, C, _lib.ptr(bnc), _lib.ptr(xpre), _lib.ptr(arg),
                                            _lib.stream()), "facl_rows_segmax")
        _lib.tap("seg_arg", arg)
        _lib.tap_relu("relu_t3", a=xpre)
        ctx.save_for_backward(h, W, y, bnc, xpre, arg)
        ctx.count, ctx.reduce_fn, ctx.training, ctx.S = count, reduce_fn, training, S
        ctx.mark_non_differentiable(arg)
        return xpre, arg

    @staticmethod
    def backward(ctx, dxpre, _darg):
        bp = backward_precision(ctx.prec)   # the backward GEMMs run in the arithmetic the forward recorded
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        lib = _lib.load_library()
        h, W, y, bnc, xpre, arg = ctx.saved_tensors
        ws = _Workspace.get(y.device)
        R, C = y.shape
        S = ctx.S
        M = R // S
        dxpre = dxpre.contiguous()
        sums = _lib.empty((C, 2), dtype=torch.float64, device=y.device)
        _lib.check(lib.facl_segmax_bwd_stats(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(y), _lib.ptr(arg), M, S, C,
                                             _lib.ptr(bnc), _lib.ptr(sums), _lib.ptr(ws), _lib.stream()),
                   "facl_segmax_bwd_stats")
        dbeta, dgamma, kk = _bn_bwd_consts(sums, C, ctx.count, ctx.reduce_fn)   # parameter gradients stay local sums
        dy = _lib.empty_like(y)
        _lib.check(lib.facl_segmax_bwd_apply(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(y), _lib.ptr(arg), M, S, C,
                                             _lib.ptr(bnc), _lib.ptr(kk), _lib.ptr(dy), _lib.stream()),
                   "facl_segmax_bwd_apply")
        dW = gemm_wgrad(dy, h, prec=bp)
        dh = gemm_dgrad(dy, W, prec=bp)
        return dh, dW, None, dgamma, dbeta, None, None, None, None


def linear_bn_relu(h, affine, bn, training, reduce_fn=None, centers=None):
    """relu(bn([centers | h] W^T + b)) over rows; W is (Cout,Cin[,1,1])."""
    W = affine.weight.view(affine.weight.shape[0], -1)
    return _LinearBNReLU.apply(h, W, affine.bias, bn.weight, bn.bias, bn, training, reduce_fn, centers)


class _Linear(torch.autograd.Function):
    """Plain h W^T + b on the MFMA GEMMs (netR_FC.3, cn3d_model_conbag.py:206)."""

    @staticmethod
    def forward(ctx, h, W, b):
        ctx.prec = current_precision()
        h, W = h.contiguous(), W.contiguous()
        y, _ = gemm_fwd(h, W, b, prec=ctx.prec)
        ctx.save_for_backward(h, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        bp = backward_precision(ctx.prec)   # the backward GEMMs run in the arithmetic the forward recorded
        h, W = ctx.saved_tensors
        dy = dy.contiguous()
        return gemm_dgrad(dy, W, prec=bp), gemm_wgrad(dy, h, prec=bp), dy.sum(0)


_FC_FUSED = __import__("os").environ.get("FACL_FC_FUSED", "1") != "0"          # A/B switch: 0 = the single-segment kernels of rounds 2-4


class _FCHead(torch.autograd.Function):
    """gobaol_max_pool (cn3d_model_conbag.py:225-226) + netR_FC (Linear -> BatchNorm1d -> ReLU -> Linear, :201-207) applied
    to the per-view rows AND to the per-clip rows (:228-229) in one pass.

    x_pre (G*B, C) view-major -> ONE stacked output (G*B + B, dim): rows [0, G*B) = netR_FC(x_pre) = ``x``, rows
    [G*B, G*B + B) = netR_FC(max over the G views) = ``x_global``.  The two Linear layers run ONCE on the stacked rows (the
    second call alone has only B rows -- 16 workgroups for a 1024x1024 layer, latency-bound), while BatchNorm keeps the
    reference's two separate batch statistics and its two sequential running-statistics updates, segment by segment.
    Keeping the stacked tensor whole lets the loss run ONE similarity GEMM on it (utils_my._ContrastivePair) and spares
    the cat / split launches of both directions; in the backward the view-max gradient is scattered straight into the
    rows of dL/dx_pre (no (G*B, C) zero tensor, no autograd accumulation kernel).

    Round 4 (second session): this block is ~50 launches of a few microseconds each around nine ~20 us GEMMs, i.e. launch latency.  The
    two-segment BatchNorm is three launches per direction (csrc/fchead.hip: fp64 slice statistics
    -> both finalisations -> one apply; backward: slice sums -> constants + parameter gradients of both segments
    -> one apply) instead of nine / eleven, the view maximum fills the stacked input in the launch that reads it, and the
    bias gradient is one column-sum launch."""

    @staticmethod
    def forward(ctx, x_pre, G, W1, b1, gamma, beta, bn, W2, b2, training, reduce_fn):
        ctx.prec = current_precision()
        lib = _lib.load_library()
        _lib.require_cuda(x_pre)
        ws = _Workspace.get(x_pre.device)
        x_pre = x_pre.contiguous()
        M, Cin = x_pre.shape
        B = M // G
        h = _lib.empty((M + B, Cin), dtype=torch.float32, device=x_pre.device)
        arg = _lib.empty((B, Cin), dtype=torch.int32, device=x_pre.device)
        _lib.check(lib.facl_viewmax_stack(_lib.ptr(x_pre), G, B, Cin, _lib.ptr(h), _lib.ptr(arg), _lib.stream()), "facl_viewmax_stack")
        _lib.tap("view_arg", arg)
        W1, W2 = W1.contiguous(), W2.contiguous()
        R, C = M + B, W1.shape[0]
        segs = ((0, M), (M, R))
        fused = _FC_FUSED and training and M % 32 == 0 and C % 4 == 0
        ctx.fused = fused
        if fused:
            f32 = dict(dtype=torch.float32, device=x_pre.device)
            world = 1 if reduce_fn is None else reduce_fn.world_size
            counts = [float(M) * world, float(B) * world]
            sums2 = part = None
            na = nb = 0
            # (statistics out of the first GEMM's epilogue -- fp32 sums over 32 rows, tried -- save one more
            # launch but lose the clip segment's variance: B = 4..32 rows, var = E[y^2] - mean^2 of fp32-rounded sums put
            # x_global 1.4e-4 from fp64 on the C1 golden (gpurun_out/r5a_tests.log).  The slice statistics stay fp64.)
            y = gemm_fwd(h, W1, b1, prec=ctx.prec)[0]
            if reduce_fn is not None:
                sums2 = _lib.empty((2, C, 2), dtype=torch.float64, device=y.device)
            _lib.check(lib.facl_fc_bn_stats(_lib.ptr(y), M, R, C, _lib.ptr(sums2), _lib.ptr(ws), _lib.stream()), "facl_fc_bn_stats")
            if reduce_fn is not None:
                reduce_fn(sums2)                                 # ONE SyncBN all-reduce for the pair
            else:
                part, na, nb = ws, M // 32, (B + 31) // 32
            a = _lib.empty_like(y)
            bnc2 = _lib.empty((2, 5, C), **f32)
            _lib.check(lib.facl_fc_bn_apply(_lib.ptr(y), M, R, C, _lib.ptr(sums2), _lib.ptr(part), na, nb, counts[0], counts[1],
                                            _lib.ptr(gamma.detach()), _lib.ptr(beta.detach()), BN_EPS, BN_MOMENTUM,
                                            _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(bnc2), _lib.ptr(a),
                                            _lib.stream()), "facl_fc_bn_apply")
            bn.count_batch()
            bn.count_batch()
            bncs = [bnc2]                                        # one (2, 5, C) tensor: both segments' constants
        else:
            y, _ = gemm_fwd(h, W1, b1, prec=ctx.prec)
            a = _lib.empty_like(y)
            bncs, counts = [], []
            if training:
                # both segments' statistics first, ONE SyncBN all-reduce for the pair, then the two finalisations in the
                # reference's order (view rows :228, clip rows :229: the running statistics are updated twice)
                sums2 = _lib.empty((2, C, 2), dtype=torch.float64, device=y.device)
                for i, (r0, r1) in enumerate(segs):
                    _lib.check(lib.facl_rows_stats(_lib.ptr(y[r0:r1]), r1 - r0, C, _lib.ptr(sums2[i]), _lib.ptr(ws), _lib.stream()),
                               "facl_rows_stats")
                world = 1
                if reduce_fn is not None:
                    reduce_fn(sums2)
                    world = reduce_fn.world_size
            for i, (r0, r1) in enumerate(segs):
                if training:
                    count = float(r1 - r0) * world
                    bnc = _bn_finalize(sums2[i], C, count, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var)
                    bn.count_batch()
                else:
                    bnc, count = _forward_bn_consts(y[r0:r1], bn, False, None, ws)
                _lib.check(lib.facl_rows_bn_relu(_lib.ptr(y[r0:r1]), r1 - r0, C, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]),
                                                 _lib.ptr(a[r0:r1]), _lib.stream()), "facl_rows_bn_relu")
                bncs.append(bnc)
                counts.append(count)
        _lib.tap_relu("relu_fc", a=a)
        out, _ = gemm_fwd(a, W2, b2, prec=ctx.prec)
        ctx.save_for_backward(h, W1, y, a, W2, arg, *bncs)
        ctx.segs, ctx.counts, ctx.reduce_fn, ctx.training, ctx.G = segs, counts, reduce_fn, training, G
        return out

    @staticmethod
    def backward(ctx, dout):
        bp = backward_precision(ctx.prec)   # the backward GEMMs run in the arithmetic the forward recorded
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        lib = _lib.load_library()
        h, W1, y, a, W2, arg, *bncs = ctx.saved_tensors
        bnc2 = bncs[0] if ctx.fused else None
        bnc_a, bnc_b = (bnc2[0], bnc2[1]) if ctx.fused else bncs
        ws = _Workspace.get(y.device)
        R, C = y.shape
        M = ctx.segs[0][1]
        B = R - M
        dout = dout.contiguous()
        f32 = dict(dtype=torch.float32, device=y.device)
        # the weight / bias gradients are leaves: they run on the side stream beside the dgrad -> BatchNorm-backward chain
        # (not under data parallelism: the chain holds a collective, and a graph segment must not end with an open branch)
        side = ctx.reduce_fn is None
        with _lib.fork(enabled=side) as f2:
            dW2 = gemm_wgrad(dout, a, prec=bp)
            if dout.shape[1] % 4 == 0:
                db2 = _lib.empty(dout.shape[1], **f32)
                _lib.check(lib.facl_col_sums(_lib.ptr(dout), R, dout.shape[1], _lib.ptr(db2), _lib.stream()), "facl_col_sums")
            else:
                db2 = dout.sum(0)
        dact = gemm_dgrad(dout, W2, prec=bp)
        dy = _lib.empty_like(y)
        if ctx.fused:
            sums2 = sums2_g = None
            if ctx.reduce_fn is not None:
                sums2 = _lib.empty((2, C, 2), dtype=torch.float64, device=y.device)
            _lib.check(lib.facl_fc_bn_bwd_stats(_lib.ptr(dact), _lib.ptr(y), M, R, C, _lib.ptr(bnc2), _lib.ptr(sums2), _lib.ptr(ws),
                                                _lib.stream()), "facl_fc_bn_bwd_stats")
            if ctx.reduce_fn is not None:
                sums2_g = ctx.reduce_fn(sums2)                   # the local slice sums stay in the workspace
            dgamma, dbeta, kk2 = _lib.empty(C, **f32), _lib.empty(C, **f32), _lib.empty((2, 2, C), **f32)
            _lib.check(lib.facl_fc_bn_bwd_apply(_lib.ptr(dact), _lib.ptr(y), M, R, C, _lib.ptr(bnc2), _lib.ptr(sums2_g), _lib.ptr(ws),
                                                ctx.counts[0], ctx.counts[1], _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(kk2),
                                                _lib.ptr(dy), _lib.stream()), "facl_fc_bn_bwd_apply")
        else:
            sums2 = _lib.empty((2, C, 2), dtype=torch.float64, device=y.device)
            for i, ((r0, r1), bnc) in enumerate(zip(ctx.segs, (bnc_a, bnc_b))):
                _lib.check(lib.facl_rows_bwd_stats(_lib.ptr(dact[r0:r1]), _lib.ptr(y[r0:r1]), r1 - r0, C, _lib.ptr(bnc),
                                                   _lib.ptr(sums2[i]), _lib.ptr(ws), _lib.stream()), "facl_rows_bwd_stats")
            sums2_g = ctx.reduce_fn(sums2.clone()) if ctx.reduce_fn is not None else sums2       # one all-reduce for the pair
            dbe, dga, kk = _lib.empty((2, C), **f32), _lib.empty((2, C), **f32), _lib.empty((2, 2, C), **f32)
            for i, ((r0, r1), bnc, count) in enumerate(zip(ctx.segs, (bnc_a, bnc_b), ctx.counts)):
                _lib.check(lib.facl_bn_bwd_consts(_lib.ptr(sums2[i]), _lib.ptr(sums2_g[i]), C, float(count), _lib.ptr(dbe[i]),
                                                  _lib.ptr(dga[i]), _lib.ptr(kk[i]), _lib.stream()), "facl_bn_bwd_consts")
                _lib.check(lib.facl_rows_bwd_apply(_lib.ptr(dact[r0:r1]), _lib.ptr(y[r0:r1]), r1 - r0, C, _lib.ptr(bnc),
                                                   _lib.ptr(kk[i]), _lib.ptr(dy[r0:r1]), _lib.stream()), "facl_rows_bwd_apply")
            dgamma, dbeta = dga[0] + dga[1], dbe[0] + dbe[1]
        with _lib.fork(enabled=side) as f1:
            dW1 = gemm_wgrad(dy, h, prec=bp)
        dh = gemm_dgrad(dy, W1, prec=bp)
        # dL/dx_pre = dh[:M] + (dh[M:] routed to the winning view's row of each (clip, channel))
        _lib.check(lib.facl_viewmax_bwd_add(dh[M:].data_ptr(), _lib.ptr(arg), ctx.G, B, dh.shape[1], _lib.ptr(dh), _lib.stream()),
                   "facl_viewmax_bwd_add")
        f2.join(dW2, db2)
        f1.join(dW1)
        # d(bias of the first Linear) is identically zero in front of a train-mode BN: None leaves it untouched
        return dh[:M], None, dW1, None, dgamma, dbeta, None, dW2, db2, None, None


def fc_head(x_pre, G, affine1, bn, affine2, training, reduce_fn=None):
    """Stacked [netR_FC(x_pre) ; netR_FC(gobaol_max_pool(x_pre))] ((G*B + B, dim)): the reference's two BatchNorm calls,
    ONE pass over each Linear layer."""
    return _FCHead.apply(x_pre, G, affine1.weight, affine1.bias, bn.weight, bn.bias, bn, affine2.weight, affine2.bias,
                         training, reduce_fn)


class _NormalizeMap(torch.autograd.Function):
    """x_nor = F.normalize(x, p=2, dim=1), code = mapping(x_nor) (cn3d_model_conbag.py:231-232) as one HIP kernel.  The
    live loss does not use these outputs; the backward (SwAV branch, user code) is the closed form in tensor algebra."""

    @staticmethod
    def forward(ctx, x, Wm, lazy=False):
        lib = _lib.load_library()
        _lib.require_cuda(x)
        x, Wm = x.contiguous(), Wm.contiguous()
        M, C = x.shape
        K = Wm.shape[0]
        xn = _lib.empty_like(x)
        code = _lib.empty((M, K), dtype=torch.float32, device=x.device)
        # `lazy`: the caller (the training step, whose loss does not read these outputs) joins the side stream itself
        # (_lib.join_pending) -- the kernel then runs beside the loss block instead of in front of it
        with _lib.fork(enabled=lazy) as f:
            _lib.check(lib.facl_normalize_map(_lib.ptr(x), M, C, _lib.ptr(Wm), K, _lib.ptr(xn), _lib.ptr(code), _lib.stream()),
                       "facl_normalize_map")
        if f.on:
            _lib._PENDING.append((f, (xn, code)))
        ctx.save_for_backward(x, xn, Wm)
        return xn, code

    @staticmethod
    def backward(ctx, dxn, dcode):
        x, xn, Wm = ctx.saved_tensors
        dxn_t = dcode @ Wm if dxn is None else dxn + dcode @ Wm
        dWm = dcode.t() @ xn
        nrm = x.norm(dim=1, keepdim=True).clamp_min(1e-12)
        dx = (dxn_t - xn * (dxn_t * xn).sum(1, keepdim=True)) / nrm
        return dx, dWm, None


def normalize_map(x, mapping_weight, lazy=False):
    """lazy=True: the kernel runs on the side stream and the CALLER must call _lib.join_pending() before x_nor / code are
    read (or at the end of its step); the default joins nothing because nothing was forked."""
    return _NormalizeMap.apply(x, mapping_weight, lazy)


class _ViewMax(torch.autograd.Function):
    """gobaol_max_pool over the gost*S local features of a clip (cn3d_model_conbag.py:225-226) = max over the G views of
    the per-view maxima (rows are view-major: g*B+b); first view wins ties, like the reference's MaxPool over the
    concatenated sequence."""

    @staticmethod
    def forward(ctx, x_pre, G):
        lib = _lib.load_library()
        _lib.require_cuda(x_pre)
        x_pre = x_pre.contiguous()
        B, C = x_pre.shape[0] // G, x_pre.shape[1]
        out = _lib.empty((B, C), dtype=torch.float32, device=x_pre.device)
        arg = _lib.empty((B, C), dtype=torch.int32, device=x_pre.device)
        _lib.check(lib.facl_viewmax_fwd(_lib.ptr(x_pre), G, B, C, _lib.ptr(out), _lib.ptr(arg), _lib.stream()), "facl_viewmax_fwd")
        ctx.save_for_backward(arg)
        ctx.G = G
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load_library()
        arg, = ctx.saved_tensors
        B, C = arg.shape
        dout = dout.contiguous()
        dx = _lib.empty((ctx.G * B, C), dtype=torch.float32, device=dout.device)
        _lib.check(lib.facl_viewmax_bwd(_lib.ptr(dout), _lib.ptr(arg), ctx.G, B, C, _lib.ptr(dx), _lib.stream()), "facl_viewmax_bwd")
        return dx, None


def view_max(x_pre, G):
    return _ViewMax.apply(x_pre, G)


def linear(h, affine):
    return _Linear.apply(h, affine.weight, affine.bias)


def linear_bn_relu_segmax(h, affine, bn, training, S, reduce_fn=None):
    """(M*S,Cin) -> (M,Cout): the last per-centroid layer fused with the max over each cloud's S centroids."""
    W = affine.weight.view(affine.weight.shape[0], -1)
    xpre, _ = _LinearBNSegmax.apply(h, W, affine.bias, bn.weight, bn.bias, bn, training, reduce_fn, S)
    return xpre


# ---- net3DV_3 as ONE pass structure on the row-streamed GEMMs (csrc/gemm_rs.hip) ----------------------------------------
def rs_planes(W, transposed, centers_cols=None, half=False):
    """Fragment-ordered bf16 planes of a weight matrix view (rows x cols, any leading dimension) for facl_gemm_rs_fwd
    (transposed = False) / facl_gemm_rs_dgrad (True).  `centers_cols`: (N,3) view whose columns join as the centre k-step."""
    lib = _lib.load_library()
    N, K = W.shape
    nb = lib.facl_gemm_rs_planes_bytes(K if transposed else N, N if transposed else K, 0 if centers_cols is None else 1)
    planes = _lib.empty(nb, dtype=torch.uint8, device=W.device)
    _lib.check(lib.facl_gemm_rs_planes(W.data_ptr(), W.stride(0), N, K, 1 if transposed else 0,
                                       None if centers_cols is None else centers_cols.data_ptr(),
                                       0 if centers_cols is None else centers_cols.stride(0), 1 if half else 0,
                                       _lib.ptr(planes), _lib.stream()), "facl_gemm_rs_planes")
    return planes


def rs_planes_multi(jobs, absmax=None):
    """`jobs`: list of (W view, transposed, centre-column view or None[, half]) -> list of plane buffers, ONE launch.
    half = True: fp16x3 planes (forward arithmetic of csrc/common.h), else bf16x6 planes.
    `absmax` = (tensor, amax row): the same launch raises the amax row to max|tensor| (the centroid coordinates)."""
    jobs = [(j + (False,))[:4] for j in jobs]
    import ctypes
    lib = _lib.load_library()
    n = len(jobs)
    outs = []
    for W, tr, xc, _h in jobs:
        N, K = W.shape
        nb = lib.facl_gemm_rs_planes_bytes(K if tr else N, N if tr else K, 0 if xc is None else 1)
        outs.append(_lib.empty(nb, dtype=torch.uint8, device=W.device))
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    _lib.check(lib.facl_gemm_rs_planes_multi(
        n, vp(*[j[0].data_ptr() for j in jobs]), ip(*[j[0].stride(0) for j in jobs]), ip(*[j[0].shape[0] for j in jobs]),
        ip(*[j[0].shape[1] for j in jobs]), ip(*[1 if j[1] else 0 for j in jobs]),
        vp(*[None if j[2] is None else j[2].data_ptr() for j in jobs]),
        ip(*[0 if j[2] is None else j[2].stride(0) for j in jobs]), ip(*[1 if j[3] else 0 for j in jobs]),
        vp(*[o.data_ptr() for o in outs]), None if absmax is None else _lib.ptr(absmax[0]),
        0 if absmax is None else absmax[0].numel(), None if absmax is None else _lib.ptr(absmax[1]), _lib.stream()),
        "facl_gemm_rs_planes_multi")
    return outs


# Forward arithmetic of the row-streamed GEMMs: fp16x3 (two fp16 planes of pre-scaled operands, three products: the same
# fp32-GEMM accuracy at half the MFMA work, csrc/common.h) unless FACL_FWD_H3=0 selects bf16x6 (A/B, bit-identity tests)
FWD_H3 = __import__("os").environ.get("FACL_FWD_H3", "1") != "0"
# Backward arithmetic of the row-streamed dgrad / weight-gradient kernels: fp16x3 with the gradient operand's power-of-two
# scale taken per launch from max|dy| (a device scalar its producer kernel maintains); FACL_BWD_H3=0: bf16x6
BWD_H3 = __import__("os").environ.get("FACL_BWD_H3", "1") != "0"


def _rs_fwd(a, planes, N, bias, pro, centers, want_stats, seg_sgn, ws, half=False, amax_a=None):
    lib = _lib.load_library()
    M, K = a.shape
    y = _lib.empty((M, N), dtype=torch.float32, device=a.device)
    sums = _lib.empty((N, 2), dtype=torch.float64, device=a.device) if want_stats else None
    ymax = arg = None
    if seg_sgn is not None:
        ymax = _lib.empty((M // 64, N), dtype=torch.float32, device=a.device)
        arg = _lib.empty((M // 64, N), dtype=torch.int32, device=a.device)
    ps, pt = (pro[2], pro[3]) if pro is not None else (None, None)
    with _lib.timed("facl_gemm_rs_fwd %dx%dx%d%s" % (M, K, N, " h3" if half else "")):
        _lib.check(lib.facl_gemm_rs_fwd(_lib.ptr(a), M, K, _lib.ptr(planes), 1 if half else 0, _lib.ptr(amax_a), N, _lib.ptr(bias), _lib.ptr(ps), _lib.ptr(pt),
                                        _lib.ptr(centers), _lib.ptr(y), _lib.ptr(sums), _lib.ptr(seg_sgn), _lib.ptr(ymax),
                                        _lib.ptr(arg), _lib.ptr(ws), _lib.stream()), "facl_gemm_rs_fwd")
    return y, sums, ymax, arg


_FUSE_BNSTATS = __import__("os").environ.get("FACL_RS_BNSTATS", "1") != "0"      # A/B switch: 0 = separate facl_rows_bwd_stats pass


def _rs_dgrad(dy, W, prec, planes=None, bn_y=None, bn_c=None, ws=None, amax=None):
    """da = dy W on the row-streamed kernel (fp32-grade arithmetic) or, for the opt-in backward precision, the staged one.
    With (bn_y, bn_c) also the BatchNorm-backward column sums of the layer that consumes da: returns (da, sums or None).
    `amax` (the _lib.AMAX_WORDS int32 buffer holding the bits of max|dy|): fp16x3 arithmetic; `planes` must then be fp16x3 planes."""
    if prec != "f32":
        return gemm_dgrad(dy, W, prec=prec), None
    lib = _lib.load_library()
    M, N = dy.shape
    K = W.shape[1]
    half = 0 if amax is None else 1
    if planes is None:
        planes = rs_planes(W, True, half=bool(half))
    da = _lib.empty((M, K), dtype=torch.float32, device=dy.device)
    sums = None
    with _lib.timed("facl_gemm_rs_dgrad %dx%dx%d%s" % (M, N, K, " h3" if half else "")):
        if bn_y is not None and _FUSE_BNSTATS:
            sums = _lib.empty((K, 2), dtype=torch.float64, device=dy.device)
            _lib.check(lib.facl_gemm_rs_dgrad_bnstats(_lib.ptr(dy), M, N, _lib.ptr(planes), half, _lib.ptr(amax), K, _lib.ptr(da),
                                                      _lib.ptr(bn_y), _lib.ptr(bn_c), _lib.ptr(sums), _lib.ptr(ws), _lib.stream()),
                       "facl_gemm_rs_dgrad_bnstats")
        else:
            _lib.check(lib.facl_gemm_rs_dgrad(_lib.ptr(dy), M, N, _lib.ptr(planes), half, _lib.ptr(amax), K, _lib.ptr(da),
                                              _lib.stream()), "facl_gemm_rs_dgrad")
    return da, sums


_WGRAD_RS = __import__("os").environ.get("FACL_WGRAD_RS", "1") != "0"            # A/B switch


def _wgrad_pro(dy, y, bnc, prec, amax=None, amax_b=None):
    """dW = dy^T relu(bn(y)) with the activation recomputed while staged; falls back to a materialised activation.
    `amax`: bits of max|dy|, `amax_b`: the bound of max relu(bn(y)) the forward used -> fp16x3 arithmetic where the
    register-streamed kernel serves the shape."""
    lib = _lib.load_library()
    M, N = dy.shape
    K = y.shape[1]
    dW = _lib.empty((N, K), dtype=torch.float32, device=dy.device)
    nzr = lib.facl_gemm_rs_wgrad_slices(M, N, K) if (prec == "f32" and _WGRAD_RS and y.is_contiguous()) else 0
    if nzr > 0:                                                         # the widest layer: register-streamed kernel
        slices = _lib.empty(nzr * N * K, dtype=torch.float32, device=dy.device)
        with _lib.timed("facl_gemm_rs_wgrad %dx%dx%d%s" % (M, N, K, "" if amax is None else " h3")):
            _lib.check(lib.facl_gemm_rs_wgrad(_lib.ptr(dy), _lib.ptr(y), M, N, K, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]),
                                              _lib.ptr(amax), _lib.ptr(amax_b), _lib.ptr(dW), _lib.ptr(slices), _lib.stream()),
                       "facl_gemm_rs_wgrad")
        return dW
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    nz = max(1, min((M + 255) // 256, 512 // tiles))
    slices = _lib.empty(nz * N * K, dtype=torch.float32, device=dy.device)
    if amax is not None and prec == "f32":                              # fp16x3 on the LDS-staged kernel (the narrower layers)
        with _lib.timed("facl_gemm_wgrad %dx%dx%d h3" % (M, N, K)):
            rc = lib.facl_gemm_wgrad_h3(_lib.ptr(dy), _lib.ptr(y), M, N, K, y.stride(0), _lib.ptr(bnc[2]), _lib.ptr(bnc[3]),
                                        _lib.ptr(amax), _lib.ptr(amax_b), _lib.ptr(dW), _lib.ptr(slices), nz, _lib.stream())
        if rc != -4:
            _lib.check(rc, "facl_gemm_wgrad_h3")
            return dW
    fn = lib.facl_gemm_wgrad_pro_x3 if prec == "x3" else lib.facl_gemm_wgrad_pro
    with _lib.timed("facl_gemm_wgrad %dx%dx%d%s" % (M, N, K, _LABEL[prec])):
        rc = fn(_lib.ptr(dy), _lib.ptr(y), M, N, K, y.stride(0), _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(dW),
                _lib.ptr(slices), nz, _lib.stream())
    if rc == -4:                                                        # FACL_E_CONFIG: shape not served -> materialise a
        a = _lib.empty_like(y)
        _lib.check(lib.facl_rows_bn_relu(_lib.ptr(y), M, K, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(a), _lib.stream()),
                   "facl_rows_bn_relu")
        return gemm_wgrad(dy, a, prec=prec)
    _lib.check(rc, "facl_gemm_wgrad_pro")
    return dW


def net3dv3_supported(P, widths, S, prec):
    """True when the three per-centroid layers (cn3d_model_conbag.py:61-77) run on the row-streamed kernels."""
    import os
    if os.environ.get("FACL_TAIL_RS", "1") == "0" or prec not in ("f32", "x3b") or S != 64 or P % 64:
        return False
    lib = _lib.load_library()
    c0, c1, c2, c3 = widths
    return all(lib.facl_gemm_rs_supported(P, k, n) == 1 for k, n in ((c0, c1), (c1, c2), (c2, c3))) and max(c1, c2) <= 512 \
        and all(lib.facl_gemm_rs_supported(P, n, k) == 1 for k, n in ((c0, c1), (c1, c2), (c2, c3)))


class _Net3DV3(torch.autograd.Function):
    """x_pre = my_max_pool(net3DV_3(cat(centres, pooled)))  (cn3d_model_conbag.py:61-77, :219-223) as explicit passes:
    three row-streamed forward GEMMs whose prologue applies the previous layer's BatchNorm + ReLU (the activations a1,
    a2 are never written; `k_rows_bn_relu` is gone), the centre columns as one more k-step, statistics and the max over each
    cloud's 64 centroids in the epilogues; backward = BN-backward row passes + row-streamed dgrads + weight gradients whose
    activation operand is recomputed from the raw layer output while it is staged."""

    @staticmethod
    def forward(ctx, pooled, centers, S, training, reduce_fn, bns, pooled_amax, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, g3, be3):
        ctx.prec = current_precision()
        lib = _lib.load_library()
        _lib.require_cuda(pooled)
        ws = _Workspace.get(pooled.device)
        pooled, centers = pooled.contiguous(), centers.contiguous()
        W1, W2, W3 = W1.contiguous(), W2.contiguous(), W3.contiguous()
        P = pooled.shape[0]
        # all weight planes of the step in one launch: forward planes now, dgrad planes kept for the backward (the weights
        # do not change between the two)
        h3 = FWD_H3
        jobs = [(W1[:, 3:], False, W1[:, :3], h3), (W2, False, None, h3), (W3, False, None, h3)]
        want_bwd = training and backward_precision(ctx.prec) == "f32"
        if want_bwd:
            jobs += [(W3, True, None, BWD_H3), (W2, True, None, BWD_H3), (W1[:, 3:], True, None, BWD_H3)]
        # fp16x3 operand maxima of the three row operands (csrc/common.h): [0] max(pooled, |centres|) -- the pooled features
        # bring their exact maximum from facl_sa_pool when the caller hands it over; the centres' joins it inside the planes
        # launch --, [1] / [2] the bounds of a1 / a2 (training: stored by facl_bn_finalize, no fill needed; eval: measured)
        amax = _lib.amax_buffers(3, pooled.device, zero=(not training) or pooled_amax is None)
        if pooled_amax is not None:
            a0 = pooled_amax
        else:
            a0 = amax[0]
            _lib.check(lib.facl_absmax(_lib.ptr(pooled), pooled.numel(), _lib.ptr(a0), _lib.stream()), "facl_absmax")
        pl = rs_planes_multi(jobs, absmax=(centers, a0))
        ctx.bwd_planes = pl[3:] if want_bwd else None
        ctx.bwd_h3 = want_bwd and BWD_H3
        y1, sums1, _, _ = _rs_fwd(pooled, pl[0], W1.shape[0], b1, None, centers, training, None, ws, h3, a0)
        bnc1, count = _forward_bn_consts(y1, bns[0], training, reduce_fn, ws, sums1, aamax=amax[1])
        y2, sums2, _, _ = _rs_fwd(y1, pl[1], W2.shape[0], b2, bnc1, None, training, None, ws, h3, amax[1])
        bnc2, _ = _forward_bn_consts(y2, bns[1], training, reduce_fn, ws, sums2, aamax=amax[2])
        y3, sums3, ymax, arg = _rs_fwd(y2, pl[2], W3.shape[0], b3, bnc2, None, training, g3.detach(), ws, h3, amax[2])
        ctx.act_amax = (a0, amax[1], amax[2])
        bnc3, _ = _forward_bn_consts(y3, bns[2], training, reduce_fn, ws, sums3)
        M, C = P // S, W3.shape[0]
        xpre = _lib.empty((M, C), dtype=torch.float32, device=pooled.device)
        _lib.check(lib.facl_sa_pool(_lib.ptr(ymax), M, C, _lib.ptr(bnc3[2]), _lib.ptr(bnc3[3]), _lib.ptr(xpre), None, _lib.stream()),
                   "facl_sa_pool")
        _lib.tap("seg_arg", arg)
        _lib.tap_relu("relu_t1", y1, bnc1[2], bnc1[3])
        _lib.tap_relu("relu_t2", y2, bnc2[2], bnc2[3])
        _lib.tap_relu("relu_t3", a=xpre)
        ctx.save_for_backward(pooled, centers, W1, W2, W3, y1, y2, y3, bnc1, bnc2, bnc3, xpre, arg, ymax)
        ctx.count, ctx.reduce_fn, ctx.training, ctx.S = count, reduce_fn, training, S
        return xpre

    @staticmethod
    def backward(ctx, dxpre):
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        bp = backward_precision(ctx.prec)
        lib = _lib.load_library()
        pooled, centers, W1, W2, W3, y1, y2, y3, bnc1, bnc2, bnc3, xpre, arg, ymax = ctx.saved_tensors
        ws = _Workspace.get(y1.device)
        st = _lib.stream()
        P, S = y1.shape[0], ctx.S
        M = P // S
        f64 = dict(dtype=torch.float64, device=y1.device)
        # ---- layer 3: max-pool + BN backward on the raw output y3
        C3 = y3.shape[1]
        dxpre = dxpre.contiguous()
        sums = _lib.empty((C3, 2), **f64)
        # y3 at the argmax = sign(gamma3) * ymax exactly: the sums come from the (M, C3) maxima the forward kept, not from a gather
        # fp16x3 backward: max|dy| of each of the three gradient tensors, maintained by the kernel that writes it; the words are
        # zeroed by the statistics launch in front of those kernels (no fill launch)
        h3 = ctx.bwd_h3 and bp == "f32"
        amax = _lib.empty((3, _lib.AMAX_WORDS), dtype=torch.int32, device=y1.device) if h3 else None
        _lib.check(lib.facl_segmax_bwd_stats_ymax(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(ymax), M, C3, _lib.ptr(bnc3),
                                                  _lib.ptr(sums), _lib.ptr(ws), _lib.ptr(amax), 3 * _lib.AMAX_WORDS if h3 else 0, st),
                   "facl_segmax_bwd_stats")
        dbe3, dga3, kk = _bn_bwd_consts(sums, C3, ctx.count, ctx.reduce_fn)
        dy = _lib.empty_like(y3)
        am = (lambda i: amax[i]) if h3 else (lambda i: None)
        _lib.check(lib.facl_segmax_bwd_apply_amax(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(y3), _lib.ptr(arg), M, S, C3,
                                                  _lib.ptr(bnc3), _lib.ptr(kk), _lib.ptr(dy), _lib.ptr(am(0)), st),
                   "facl_segmax_bwd_apply")
        bpl = ctx.bwd_planes if ctx.bwd_planes is not None else (None, None, None)
        a0b, a1b, a2b = ctx.act_amax                       # the activation bounds the forward GEMMs used
        dW3 = _wgrad_pro(dy, y2, bnc2, bp, am(0), a2b)
        # dgrad + the statistics pass of the next BatchNorm backward in one kernel (the tile is still in registers)
        da, sums = _rs_dgrad(dy, W3, bp, bpl[0], y2, bnc2, ws, am(0))
        # ---- layers 2 and 1: BN backward rows passes, weight gradient with the recomputed activation, dgrad
        grads = []
        for li, (y, bnc, yin, bnc_in, W) in enumerate(((y2, bnc2, y1, bnc1, W2), (y1, bnc1, None, None, W1)), 1):
            C = y.shape[1]
            if sums is None:
                sums = _lib.empty((C, 2), **f64)
                _lib.check(lib.facl_rows_bwd_stats(_lib.ptr(da), _lib.ptr(y), P, C, _lib.ptr(bnc), _lib.ptr(sums), _lib.ptr(ws), st),
                           "facl_rows_bwd_stats")
            dbe, dga, kk = _bn_bwd_consts(sums, C, ctx.count, ctx.reduce_fn)
            dy = _lib.empty_like(y)
            _lib.check(lib.facl_rows_bwd_apply_amax(_lib.ptr(da), _lib.ptr(y), P, C, _lib.ptr(bnc), _lib.ptr(kk), _lib.ptr(dy),
                                                    _lib.ptr(am(li)), st), "facl_rows_bwd_apply")
            if yin is not None:
                dW = _wgrad_pro(dy, yin, bnc_in, bp, am(li), a1b)
                da, sums = _rs_dgrad(dy, W, bp, bpl[1], yin, bnc_in, ws, am(li))
            else:                                                       # first layer: input = pooled | centres
                dWh = gemm_wgrad(dy, pooled, prec=bp, amax=am(li), amax_b=a0b)
                dWc = _lib.empty((C, 3), **f64)
                _lib.check(lib.facl_rows_center_wgrad(_lib.ptr(dy), _lib.ptr(centers), P, C, _lib.ptr(dWc), _lib.ptr(ws), st),
                           "facl_rows_center_wgrad")
                dW = torch.cat((dWc.float(), dWh), dim=1)
                da = _rs_dgrad(dy, W[:, 3:], bp, bpl[2], amax=am(li))[0] if ctx.needs_input_grad[0] else None
            grads.append((dW, dga, dbe))
        (dW2, dga2, dbe2), (dW1, dga1, dbe1) = grads
        # d(bias) of a conv in front of a train-mode BN is identically zero: None leaves the parameter untouched
        return (da, None, None, None, None, None, None, dW1, None, dga1, dbe1, dW2, None, dga2, dbe2, dW3, None, dga3, dbe3)


def net3dv3(pooled, centers, net, training, S, reduce_fn=None, pooled_amax=None):
    """`net` = the nn.Sequential-like net3DV_3 container (indices 0,1 / 3,4 / 6,7 = affine, BatchNorm of the three layers).
    `pooled_amax`: the _lib.amax_buffers row facl_sa_pool raised to max(pooled) (else it is measured here)."""
    view = lambda a: a.weight.view(a.weight.shape[0], -1)
    return _Net3DV3.apply(pooled, centers, S, training, reduce_fn, (net[1], net[4], net[7]), pooled_amax,
                          view(net[0]), net[0].bias, net[1].weight, net[1].bias,
                          view(net[3]), net[3].bias, net[4].weight, net[4].bias,
                          view(net[6]), net[6].bias, net[7].weight, net[7].bias)
