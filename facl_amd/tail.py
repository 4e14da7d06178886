"""Encoder tail (net3DV_3, netR_FC: cn3d_model_conbag.py:61-88) -- ROUND-1 INTERIM.

The per-centroid MLP and the FC head are plain dense GEMMs + train-mode BN over rows.  In this
round they run as rocBLAS GEMMs (torch.mm) with the BN reduce/apply written so that the
statistics are explicit (sum, sumsq) buffers that a SyncBN all-reduce can hook, exactly like the
HIP passes of sa_mlp.py; the fused MFMA GEMM (BN+ReLU prologue, statistics / segment-max
epilogue) that replaces them is the next kernel on the list ("what comes next").
"""
import torch

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class _BNRows(torch.autograd.Function):
    """Train-mode BatchNorm over the rows of (R,C) with explicit fp64 statistics; SyncBN-ready."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, reduce_fn):
        R = y.shape[0]
        st = torch.stack([y.sum(0, dtype=torch.float64), (y.double() * y.double()).sum(0),
                          torch.full((y.shape[1],), float(R), dtype=torch.float64, device=y.device)])
        if reduce_fn is not None:
            reduce_fn(st)
        n = st[2, 0]
        mean = st[0] / n
        var = (st[1] / n - mean * mean).clamp_min(0)
        invstd = torch.rsqrt(var + BN_EPS)
        with torch.no_grad():
            unb = var * (n / (n - 1)) if float(n) > 1 else var
            running_mean.mul_(1 - BN_MOMENTUM).add_(mean.float(), alpha=BN_MOMENTUM)
            running_var.mul_(1 - BN_MOMENTUM).add_(unb.float(), alpha=BN_MOMENTUM)
        scale = (gamma.double() * invstd).float()
        shift = (beta.double() - mean * gamma.double() * invstd).float()
        out = torch.addcmul(shift, y, scale)
        ctx.save_for_backward(y, gamma, mean.float(), invstd.float())
        ctx.reduce_fn, ctx.n = reduce_fn, float(n)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, mean, invstd = ctx.saved_tensors
        yhat = (y - mean) * invstd
        sums = torch.stack([dout.sum(0, dtype=torch.float64), (dout.double() * yhat.double()).sum(0)])
        dbeta_l, dgamma_l = sums[0].float(), sums[1].float()
        if ctx.reduce_fn is not None:
            ctx.reduce_fn(sums)
        k1 = (sums[0] / ctx.n).float()
        k2 = (sums[1] / ctx.n).float()
        dy = (dout - k1 - yhat * k2) * (gamma * invstd)
        return dy, dgamma_l, dbeta_l, None, None, None


def linear_bn_relu(h, affine, bn, training, reduce_fn=None):
    """relu(bn(h W^T + b)) over rows; W is (Cout,Cin[,1,1])."""
    W = affine.weight.view(affine.weight.shape[0], -1)
    y = torch.addmm(affine.bias, h, W.t())
    if training:
        y = _BNRows.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, reduce_fn)
        bn.num_batches_tracked += 1
    else:
        invstd = torch.rsqrt(bn.running_var + BN_EPS)
        y = (y - bn.running_mean) * (invstd * bn.weight) + bn.bias
    return torch.relu(y)
