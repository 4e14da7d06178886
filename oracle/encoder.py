"""Oracle: the set-abstraction encoder forward (torch fp32 functional ops; autograd gives the
backward oracle).

Restates /root/reference/training_code/cn3d_model_conbag.py:213-234 (the 4-output forward;
identical to the commented-out ``PointNet_Plus.forward`` :116-137) over the modules built at
:43-91 / :162-210.  Parameters travel as a flat dict with the reference's 52 state_dict keys.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm default, never overridden in the reference
BN_MOMENTUM = 0.1


def _conv_bn_relu(sd, prefix, li, x, training):
    """1x1 Conv2d (:45/:49/:53) + BatchNorm2d (:46...) + ReLU (:47...).  In training mode
    F.batch_norm normalises with the biased batch variance and updates running_mean /
    running_var (unbiased) in place with momentum 0.1, and num_batches_tracked += 1."""
    x = F.conv2d(x, sd[f"{prefix}.{li}.weight"], sd[f"{prefix}.{li}.bias"])
    b = li + 1
    x = F.batch_norm(x, sd[f"{prefix}.{b}.running_mean"], sd[f"{prefix}.{b}.running_var"],
                     sd[f"{prefix}.{b}.weight"], sd[f"{prefix}.{b}.bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[f"{prefix}.{b}.num_batches_tracked"] += 1
    return F.relu(x, inplace=True)                                    # nn.ReLU(inplace=True), :47


def _fc_head(sd, x, training):
    """netR_FC :201-207: Linear -> BatchNorm1d -> ReLU -> Linear."""
    x = F.linear(x, sd["netR_FC.0.weight"], sd["netR_FC.0.bias"])
    x = F.batch_norm(x, sd["netR_FC.1.running_mean"], sd["netR_FC.1.running_var"],
                     sd["netR_FC.1.weight"], sd["netR_FC.1.bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        sd["netR_FC.1.num_batches_tracked"] += 1
    x = F.relu(x, inplace=True)                                       # :204
    return F.linear(x, sd["netR_FC.3.weight"], sd["netR_FC.3.bias"])


def encoder_forward(sd, xt, yt, gost, training=True, return_intermediates=False):
    """xt (M,D,S,K), yt (M,3,S,1) -> (x (M,512), code (M,64), x_nor (M,512), x_global (B,512)).

    ``sd`` holds parameters AND BN buffers; buffers are updated in place in training mode
    (netR_FC.1 twice per call, :228-229)."""
    M, D, S, K = xt.shape
    h = xt
    for li in (0, 3, 6):
        h = _conv_bn_relu(sd, "net3DV_1", li, h, training)
    pooled = F.max_pool2d(h, (1, K), stride=1)                        # :176
    h = torch.cat((yt, pooled), 1)                                    # :219
    for li in (0, 3, 6):
        h = _conv_bn_relu(sd, "net3DV_3", li, h, training)
    xt_local = h                                                      # (M,1024,S,1)
    x_pre = F.max_pool2d(xt_local, (S, 1), stride=1).squeeze(-1).squeeze(-1)   # :222
    xg = xt_local.reshape(gost, -1, 1024, S).permute(1, 2, 0, 3).reshape(-1, 1024, gost * S, 1)  # :225
    xg_pre = F.max_pool2d(xg, (S * gost, 1), stride=1).squeeze(-1).squeeze(-1)  # :226
    x = _fc_head(sd, x_pre, training)                                 # :228
    x_global = _fc_head(sd, xg_pre, training)                         # :229
    x_nor = F.normalize(x, p=2, dim=1)                                # :231
    code = F.linear(x_nor, sd["mapping.weight"])                      # :232
    if return_intermediates:
        return (x, code, x_nor, x_global), dict(pooled=pooled, x_pre=x_pre, xg_pre=xg_pre)
    return x, code, x_nor, x_global


def clone_state(sd_np, device="cpu", requires_grad=False):
    """numpy/torch state_dict -> fresh torch dict (float params optionally requiring grad)."""
    out = {}
    for k, v in sd_np.items():
        t = torch.as_tensor(v).clone().to(device)
        if requires_grad and t.is_floating_point() and not ("running_" in k):
            t.requires_grad_(True)
        out[k] = t
    return out


def param_keys(sd):
    return [k for k in sd if not ("running_" in k or "num_batches" in k)]
