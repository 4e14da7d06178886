"""ONE full contrastive training step at the headline size (BASELINE configs[1]: B=32, T=24, N=2048, D=3) and the same
through the appearance stream's entry (configs[2]: D=4, appearance-style clouds) against a plain torch-fp64 evaluation
of the same graph on the GPU (matmuls, train-mode BN with batch statistics, ReLU, max-pools, the reference's literal
loss construction) -- independent of the HIP kernels and of the size the oracle's goldens pin (C1)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import forward64, max_rel_rows, rel_err, routing_taps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4          # north_star: fp32 features / loss within 1e-4 relative


def _opt(D, B, N):
    return SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                           sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B,
                           pooling="concatenation", SAMPLE_NUM=N)


# prec "x3" / "x3b": the opt-in three-product arithmetic (facl_amd.tail.precision).  "x3b" (backward GEMMs only) must
# reproduce the default path's features and losses; "x3" is ~1e-5 per GEMM result (losses 4e-6, worst feature row ~1e-4)
@pytest.mark.parametrize("stream,D,prec", [("motion", 3, "f32"), ("appearance", 4, "f32"), ("motion", 3, "x3"), ("motion", 3, "x3b")])
def test_full_step_at_headline_size_vs_torch_fp64(stream, D, prec):
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from facl_amd.train_common import ContrastiveStep, appearance_batch, synthetic_batch
    from facl_amd.utils_my import knn_radius_group
    from oracle import loss as OL
    from oracle.weights import formula_state_dict
    B, G, N, S, K = 32, 24, 2048, 64, 64
    gen = torch.Generator(device=DEV)
    gen.manual_seed(11)
    clip = (appearance_batch if stream == "appearance" else synthetic_batch)(B, G, N, D, torch.device(DEV), gen)
    opt = _opt(D, B, N)
    sd = formula_state_dict(D)
    net = PointNet_Plus(opt, gost=G)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    net = net.to(DEV).train()
    net.precision = prec
    stat_tol = 1e-4 if prec == "x3" else 1e-5           # "x3b" changes the backward only
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06, fused=True)
    step = ContrastiveStep(net, optim, opt, G)
    order = np.random.RandomState(5).permutation(G)
    # outputs of the step's own forward: hook the model call
    taps = {}
    h = net.register_forward_hook(lambda m, i, o: taps.update(x=o[0].detach().clone(), xg=o[3].detach().clone()))
    with routing_taps() as routing:                      # argmax tensors of the three max-pools (tie-proof gradient check below)
        loss, loss_c, loss_circle = step(clip, epoch=0, order=order)
    h.remove()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().double().clone() for k, p in net.named_parameters() if p.grad is not None}

    # ---- fp64 truth on the same grouped input (the grouping itself is bit-exact integer work: tests/test_gpu_grouping.py)
    data1 = clip.permute(1, 0, 2, 3).reshape(-1, N, D).float()
    xt, yt = knn_radius_group(data1, S, K, 0.16)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D)
    centers = yt.permute(0, 2, 1, 3).reshape(-1, 3)
    with torch.no_grad():
        x64, xg64, stats, q, _ = forward64(x_rows, centers, sd, G, S, K, DEV)
        lc64 = float(OL.global_contrast(G, xg64, x64, B))                      # the reference's literal logits construction
        lo64 = float(OL.circle_contrast(G, x64, B, order))
    e_x = max_rel_rows(taps["x"].cpu().numpy(), x64.cpu().numpy())
    e_xg = max_rel_rows(taps["xg"].cpu().numpy(), xg64.cpu().numpy())
    e_lc, e_lo = abs(loss_c.item() - lc64) / abs(lc64), abs(loss_circle.item() - lo64) / abs(lo64)
    print(f"[{stream} {prec}] x {e_x:.2e}  x_global {e_xg:.2e}  loss_c {loss_c.item():.6f} vs {lc64:.6f} ({e_lc:.2e})  "
          f"loss_circle {loss_circle.item():.6f} vs {lo64:.6f} ({e_lo:.2e})")
    # "x3" everywhere sits AT the north_star bound on the worst feature row (measured 7e-5 / 9.9e-5): held to 2e-4 here and
    # documented as such; the default path and "x3b" are 20x inside it
    ftol = 2e-4 if prec == "x3" else TOL
    assert e_x < ftol and e_xg < ftol
    assert e_lc < TOL and e_lo < TOL
    assert abs(loss.item() - (lc64 + lo64)) < TOL * abs(lc64 + lo64)
    # ---- running statistics after the step (momentum 0.1; netR_FC.1 is updated twice: view rows, then clip rows)
    st = net.state_dict()
    for key in ("net3DV_1.1", "net3DV_1.4", "net3DV_1.7", "net3DV_3.1", "net3DV_3.4", "net3DV_3.7"):
        mean, uvar = stats[key]
        rm = 0.9 * q[f"{key}.running_mean"] + 0.1 * mean
        rv = 0.9 * q[f"{key}.running_var"] + 0.1 * uvar
        assert rel_err(st[f"{key}.running_mean"].cpu().numpy(), rm.cpu().numpy()) < stat_tol, key
        assert rel_err(st[f"{key}.running_var"].cpu().numpy(), rv.cpu().numpy()) < stat_tol, key
        assert int(st[f"{key}.num_batches_tracked"]) == int(sd[f"{key}.num_batches_tracked"]) + 1
    rm = 0.9 * (0.9 * q["netR_FC.1.running_mean"] + 0.1 * stats["fc_a"][0]) + 0.1 * stats["fc_b"][0]
    rv = 0.9 * (0.9 * q["netR_FC.1.running_var"] + 0.1 * stats["fc_a"][1]) + 0.1 * stats["fc_b"][1]
    assert rel_err(st["netR_FC.1.running_mean"].cpu().numpy(), rm.cpu().numpy()) < stat_tol
    assert rel_err(st["netR_FC.1.running_var"].cpu().numpy(), rv.cpu().numpy()) < stat_tol
    assert int(st["netR_FC.1.num_batches_tracked"]) == int(sd["netR_FC.1.num_batches_tracked"]) + 2
    # ---- the Adam step happened: every parameter that has a gradient moved by <= ~lr, the pre-BN biases did not move
    for k, p in net.named_parameters():
        d = (p.detach().cpu().double() - torch.as_tensor(sd[k]).double()).abs().max().item()
        assert d <= 2 * 3e-4, (k, d)
    # ---- end-to-end gradients at the headline size (default arithmetic), tie-proof: the fp64 graph routes its three max-pools
    # through the positions the HIP forward chose (each verified to be a numerical tie of the fp64 values)
    if prec != "f32":
        return
    del x64, xg64

    def routed(dtype):
        x_, xg_, _, q_, ties_ = forward64(x_rows, centers, sd, G, S, K, DEV, routing=routing, grad=True, dtype=dtype)
        (OL.global_contrast(G, xg_, x_, B) + OL.circle_contrast(G, x_, B, order)).backward()
        return {k: q_[k].grad.double() for k in grads if q_[k].grad is not None}, ties_
    g64, ties = routed(torch.float64)
    g32, _ = routed(torch.float32)                       # plain torch fp32, same routing: the conditioning yardstick
    print("ties", ties)
    assert max(v for k, v in ties.items() if not k.endswith("_flips")) < 1e-5, ties   # every differing decision was a tie
    pre_bn_bias = {"net3DV_1.0.bias", "net3DV_1.3.bias", "net3DV_1.6.bias", "net3DV_3.0.bias", "net3DV_3.3.bias",
                   "net3DV_3.6.bias", "netR_FC.0.bias", "net3DV_3.7.bias"}              # mathematically zero gradients
    gmax = max(float(g64[k].norm()) for k in g64)
    bad = []
    for k, mine in grads.items():
        if k in pre_bn_bias or k not in g64:
            continue
        r = g64[k].reshape(mine.shape)
        err, e32 = float((mine - r).norm()), float((g32[k].reshape(mine.shape) - r).norm())
        print(f"grad {k:20s} vs routed fp64: {err / float(r.norm()):.2e}   torch-fp32 (same routing) {e32 / float(r.norm()):.2e}  (|g| {float(r.norm()):.2e})")
        # every discrete decision pinned (max-pool argmax, ReLU signs): 1e-4 of the tensor's gradient norm
        if err > 1e-4 * max(float(r.norm()), 1e-2 * gmax):
            bad.append((k, err, e32, float(r.norm())))
    assert not bad, bad
