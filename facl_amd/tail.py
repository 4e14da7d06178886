"""Encoder tail (net3DV_3, my_max_pool, netR_FC: cn3d_model_conbag.py:61-88, :199-207).

Per layer: one plain library GEMM (rocBLAS via torch.mm -- the 1x1 conv over (M*S) centroid rows IS a
dense GEMM) + hand-written HIP kernels for everything around it: train-mode BN statistics (fp64 sums,
SyncBN hook), BN+ReLU apply, BN+ReLU+max over the S centroids without materialising the activation,
and the matching backward passes (csrc/rows.hip).
"""
import torch

from . import _lib
from .sa_mlp import BN_EPS, BN_MOMENTUM, _Workspace, _bn_eval, _bn_finalize


def _stats(y, ws):
    lib = _lib.load_library()
    R, C = y.shape
    sums = torch.empty((C, 2), dtype=torch.float64, device=y.device)
    _lib.check(lib.facl_rows_stats(_lib.ptr(y), R, C, _lib.ptr(sums), _lib.ptr(ws), _lib.stream()), "facl_rows_stats")
    return sums


def _forward_bn_consts(y, bn, training, reduce_fn, ws):
    """Statistics -> (5,C) constants; updates running stats / num_batches_tracked in training mode."""
    R, C = y.shape
    if not training:
        return _bn_eval(C, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var), float(R)
    sums = _stats(y, ws)
    count = float(R)
    if reduce_fn is not None:
        reduce_fn(sums)
        count = count * reduce_fn.world_size                  # equal shards (no host round trip for the count)
    bnc = _bn_finalize(sums, C, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var)
    bn.count_batch()
    return bnc, count


class _LinearBNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, W, b, gamma, beta, bn, training, reduce_fn):
        lib = _lib.load_library()
        _lib.require_cuda(h)
        ws = _Workspace.get(h.device)
        h = h.contiguous()
        y = torch.addmm(b, h, W.t())                                   # library GEMM
        bnc, count = _forward_bn_consts(y, bn, training, reduce_fn, ws)
        R, C = y.shape
        a = torch.empty_like(y)
        _lib.check(lib.facl_rows_bn_relu(_lib.ptr(y), R, C, _lib.ptr(bnc[2]), _lib.ptr(bnc[3]), _lib.ptr(a),
                                         _lib.stream()), "facl_rows_bn_relu")
        ctx.save_for_backward(h, W, y, bnc)
        ctx.count, ctx.reduce_fn, ctx.training = count, reduce_fn, training
        return a

    @staticmethod
    def backward(ctx, da):
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        lib = _lib.load_library()
        h, W, y, bnc = ctx.saved_tensors
        ws = _Workspace.get(y.device)
        R, C = y.shape
        da = da.contiguous()
        sums = torch.empty((C, 2), dtype=torch.float64, device=y.device)
        _lib.check(lib.facl_rows_bwd_stats(_lib.ptr(da), _lib.ptr(y), R, C, _lib.ptr(bnc), _lib.ptr(sums), _lib.ptr(ws),
                                           _lib.stream()), "facl_rows_bwd_stats")
        sl = sums.float()                                               # parameter gradients stay local sums
        dbeta, dgamma = sl[:, 0], sl[:, 1]
        if ctx.reduce_fn is not None:
            ctx.reduce_fn(sums)
            sl = sums.float()
        kk = (sl.t() * (1.0 / ctx.count)).contiguous()                  # (2,C): dbeta/P, dgamma/P
        dy = torch.empty_like(y)
        _lib.check(lib.facl_rows_bwd_apply(_lib.ptr(da), _lib.ptr(y), R, C, _lib.ptr(bnc), _lib.ptr(kk), _lib.ptr(dy),
                                           _lib.stream()), "facl_rows_bwd_apply")
        dW = dy.t() @ h                                                 # library GEMMs
        dh = dy @ W if ctx.needs_input_grad[0] else None
        # d(bias) is identically zero in front of a train-mode BN: None leaves the parameter untouched
        return dh, dW, None, dgamma, dbeta, None, None, None


class _LinearBNSegmax(torch.autograd.Function):
    """x_pre = max over the S centroids of relu(bn(h W^T + b)) -- cn3d_model_conbag.py:71-73 + :199/:222."""

    @staticmethod
    def forward(ctx, h, W, b, gamma, beta, bn, training, reduce_fn, S):
        lib = _lib.load_library()
        _lib.require_cuda(h)
        ws = _Workspace.get(h.device)
        h = h.contiguous()
        y = torch.addmm(b, h, W.t())
        bnc, count = _forward_bn_consts(y, bn, training, reduce_fn, ws)
        R, C = y.shape
        M = R // S
        xpre = torch.empty((M, C), dtype=torch.float32, device=y.device)
        arg = torch.empty((M, C), dtype=torch.int32, device=y.device)
        _lib.check(lib.facl_rows_segmax(_lib.ptr(y), M, S, C, _lib.ptr(bnc), _lib.ptr(xpre), _lib.ptr(arg), _lib.stream()),
                   "facl_rows_segmax")
        ctx.save_for_backward(h, W, y, bnc, xpre, arg)
        ctx.count, ctx.reduce_fn, ctx.training, ctx.S = count, reduce_fn, training, S
        ctx.mark_non_differentiable(arg)
        return xpre, arg

    @staticmethod
    def backward(ctx, dxpre, _darg):
        if not ctx.training:
            raise RuntimeError("backward through the eval-mode (folded BN) encoder is not supported")
        lib = _lib.load_library()
        h, W, y, bnc, xpre, arg = ctx.saved_tensors
        ws = _Workspace.get(y.device)
        R, C = y.shape
        S = ctx.S
        M = R // S
        dxpre = dxpre.contiguous()
        sums = torch.empty((C, 2), dtype=torch.float64, device=y.device)
        _lib.check(lib.facl_segmax_bwd_stats(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(y), _lib.ptr(arg), M, S, C,
                                             _lib.ptr(bnc), _lib.ptr(sums), _lib.ptr(ws), _lib.stream()),
                   "facl_segmax_bwd_stats")
        sl = sums.float()
        dbeta, dgamma = sl[:, 0], sl[:, 1]
        if ctx.reduce_fn is not None:
            ctx.reduce_fn(sums)
            sl = sums.float()
        kk = (sl.t() * (1.0 / ctx.count)).contiguous()
        dy = torch.empty_like(y)
        _lib.check(lib.facl_segmax_bwd_apply(_lib.ptr(dxpre), _lib.ptr(xpre), _lib.ptr(y), _lib.ptr(arg), M, S, C,
                                             _lib.ptr(bnc), _lib.ptr(kk), _lib.ptr(dy), _lib.stream()),
                   "facl_segmax_bwd_apply")
        dW = dy.t() @ h
        dh = dy @ W
        return dh, dW, None, dgamma, dbeta, None, None, None, None


def linear_bn_relu(h, affine, bn, training, reduce_fn=None):
    """relu(bn(h W^T + b)) over rows; W is (Cout,Cin[,1,1])."""
    W = affine.weight.view(affine.weight.shape[0], -1)
    return _LinearBNReLU.apply(h, W, affine.bias, bn.weight, bn.bias, bn, training, reduce_fn)


def linear_bn_relu_segmax(h, affine, bn, training, S, reduce_fn=None):
    """(M*S,Cin) -> (M,Cout): the last per-centroid layer fused with the max over each cloud's S centroids."""
    W = affine.weight.view(affine.weight.shape[0], -1)
    xpre, _ = _LinearBNSegmax.apply(h, W, affine.bias, bn.weight, bn.bias, bn, training, reduce_fn, S)
    return xpre
