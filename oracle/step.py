"""Oracle: one full contrastive training step on the CPU (the ``cpu_baseline`` workload).

Restates the body of /root/reference/training_code/cn3d_train_motion_GL.py:223-335:
view-major reshape (:226) -> grouping (:230) -> encoder (:234) -> global (:265-287) + circle
(:290-316) loss -> loss = circle + global (:329) -> backward -> Adam(lr 3e-4, betas (0.5,0.999),
eps 1e-6) (:180) with StepLR(4, 0.7) stepped with the explicit epoch (:181,:333).
"""
import numpy as np
import torch

from . import encoder as E
from . import grouping as Gp
from . import loss as L


def view_major(out_points):
    """(B,G,N,D) -> (G*B,N,D), cn3d_train_motion_GL.py:225-226."""
    B, G, N, D = out_points.shape
    return out_points.permute(1, 0, 2, 3).reshape(-1, N, D)


def lr_at(epoch, base_lr=3e-4, step_size=4, gamma=0.7):
    return base_lr * gamma ** (epoch // step_size)


def group_torch(points, S, K, r2):
    """torch-CPU version of Gp.group_points for the timed baseline (same op chain as
    utils_my.py:265-284, multi-threaded through ATen like the reference)."""
    M, N, D = points.shape
    diff = points[:, :, 0:3].transpose(1, 2).unsqueeze(1).expand(M, S, 3, N) \
        - points[:, 0:S, 0:3].unsqueeze(-1).expand(M, S, 3, N)
    d2 = (diff * diff).sum(2)
    dists, idx = torch.topk(d2, K, 2, largest=False, sorted=False)
    jj = torch.arange(S, device=points.device).view(1, S, 1).expand_as(idx)
    idx = torch.where(dists > r2, jj, idx)
    xt = points.gather(1, idx.reshape(M, S * K, 1).expand(M, S * K, D)).view(M, S, K, D)
    yt = points[:, 0:S, 0:3].contiguous()
    xt = torch.cat((xt[..., 0:3] - yt.unsqueeze(2), xt[..., 3:]), dim=-1)
    return idx, xt.permute(0, 3, 1, 2), yt.view(M, 1, S, 3).transpose(1, 3)


class AdamState:
    """Plain Adam (torch.optim.Adam semantics, no amsgrad / weight decay), :180."""

    def __init__(self, sd, betas=(0.5, 0.999), eps=1e-6):
        self.keys = E.param_keys(sd)
        self.m = {k: torch.zeros_like(sd[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.keys}
        self.t = 0
        self.betas, self.eps = betas, eps

    def step(self, sd, grads, lr):
        b1, b2 = self.betas
        self.t += 1
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        with torch.no_grad():
            for k in self.keys:
                g = grads[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / (bc2 ** 0.5)).add_(self.eps)
                sd[k].addcdiv_(self.m[k], denom, value=-lr / bc1)


def train_step(sd, opt, points_vm, B, G, S, K, r2, order, epoch=0, grouped=None):
    """points_vm (G*B,N,D) view-major fp32.  Mutates sd (params + BN buffers) and opt.
    Returns dict(loss, loss_c, loss_circle, grads, outputs)."""
    if grouped is None:
        _, xt, yt = group_torch(points_vm, S, K, r2)
    else:
        xt, yt = grouped
    for k in opt.keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    x, code, x_nor, x_global = E.encoder_forward(sd, xt, yt, G, training=True)
    loss_c = L.global_contrast(G, x_global, x, B)
    loss_circle = L.circle_contrast(G, x, B, order)
    loss = loss_circle + loss_c
    loss.backward()
    # mapping.weight never receives a gradient (``code`` does not feed the live loss); torch.optim.Adam
    # skips such parameters, which a zero gradient reproduces exactly (m = v = 0 -> update 0).
    grads = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else torch.zeros_like(sd[k]))
             for k in opt.keys}
    opt.step(sd, grads, lr_at(epoch))
    for k in opt.keys:
        sd[k].grad = None
    return dict(loss=float(loss), loss_c=float(loss_c), loss_circle=float(loss_circle), grads=grads,
                outputs=(x.detach(), code.detach(), x_nor.detach(), x_global.detach()))


def extract_step(sd, points_vm, B, G, S, K, r2):
    """The per-batch body of /root/reference/training_code/extract_motion_feature.py:171-182 on the CPU: grouping, the
    encoder under eval() (BatchNorm from its running statistics), cat((x, x_global), 0) -> (B, (G+1)*512) features
    (save_single_feature's layout, :217-221).  The ``cpu_baseline`` of `bench.py --config extract`."""
    with torch.no_grad():
        _, xt, yt = group_torch(points_vm, S, K, r2)
        x, _, _, x_global = E.encoder_forward(sd, xt, yt, G, training=False)
        feat = torch.cat((x, x_global), dim=0)
        return feat.reshape(G + 1, B, 512).permute(1, 0, 2).reshape(B, (G + 1) * 512)
