"""GPU parity of the whole encoder + losses + training step (through the reference-shaped host API)
vs the reference goldens and the fp64 oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import forward64, load_golden, max_rel_rows, rel_err, routing_taps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4          # north_star: fp32 features / loss within 1e-4 relative
# Gradients: the reference's own fp32 gradients sit 5e-3..1e-2 away from the fp64 evaluation of the same
# graph (near-ties in the 64- and G*64-wide max-pools resolve differently under rounding and re-route
# whole gradient rows; measured in scratch on C1: net3DV_1.0.weight 7e-3, net3DV_3.4.bias 1e-2).  So the
# golden (= reference fp32) pins us only to ~2e-2; the tight check is against the fp64 oracle below.
GTOL = 5e-3
PRE_BN_BIAS = {"net3DV_1.0.bias", "net3DV_1.3.bias", "net3DV_1.6.bias", "net3DV_3.0.bias",
               "net3DV_3.3.bias", "net3DV_3.6.bias", "netR_FC.0.bias"}


def _opt(D, B, N=512):
    return SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                           sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B,
                           pooling="concatenation", SAMPLE_NUM=N)


def _model(D, B, G, neg=False):
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from oracle.weights import formula_state_dict
    net = PointNet_Plus(_opt(D, B), gost=G)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D, neg_gamma=neg).items()})
    return net.to(DEV)


def _oracle_forward(g, D, neg, dtype):
    from oracle import encoder as E, grouping as OG, loss as OL
    from oracle.weights import formula_state_dict
    B, G, N, S, K, _ = [int(v) for v in g["meta"]]
    _, xt, yt = OG.group_points(g["points"], S, K, 0.06)
    M = G * B
    sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
          for k, v in formula_state_dict(D, neg_gamma=neg).items()}
    with torch.no_grad():
        x, code, xn, xg = E.encoder_forward(sd, torch.from_numpy(xt).permute(0, 3, 1, 2).to(dtype),
                                            torch.from_numpy(yt).view(M, 1, S, 3).transpose(1, 3).to(dtype), G, True)
        lc, lo = float(OL.global_contrast(G, xg, x, B)), float(OL.circle_contrast(G, x, B, g["order"]))
    return x, code, xn, xg, lc, lo, sd


def _oracle_grads(g, D, neg, dtype):
    from oracle import encoder as E, grouping as OG, loss as OL
    from oracle.weights import formula_state_dict
    B, G, N, S, K, _ = [int(v) for v in g["meta"]]
    _, xt, yt = OG.group_points(g["points"], S, K, 0.06)
    M = G * B
    sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
          for k, v in formula_state_dict(D, neg_gamma=neg).items()}
    keys = [k for k in sd if "running" not in k and "num_b" not in k]
    for k in keys:
        sd[k].requires_grad_(True)
    x, code, xn, xg = E.encoder_forward(sd, torch.from_numpy(xt).permute(0, 3, 1, 2).to(dtype),
                                        torch.from_numpy(yt).view(M, 1, S, 3).transpose(1, 3).to(dtype), G, True)
    (OL.global_contrast(G, xg, x, B) + OL.circle_contrast(G, x, B, g["order"])).backward()
    return {k: sd[k].grad.numpy() for k in keys if sd[k].grad is not None}


@pytest.mark.parametrize("tag,D,neg", [("d4", 4, False), ("d3", 3, False), ("d4_neg", 4, True)])
def test_c1_golden_forward_loss_backward_adam(tag, D, neg):
    from facl_amd.utils_my import circle_contrast, global_contrast, group_points_3DV
    g = load_golden(f"c1_{tag}.npz")
    B, G, N, S, K, _ = [int(v) for v in g["meta"]]
    opt = _opt(D, B)
    pts = torch.from_numpy(g["points"]).to(DEV)
    xt, yt = group_points_3DV(pts, opt)

    net = _model(D, B, G, neg).eval()
    with torch.no_grad():
        ev = net(xt, yt)
    for name, t in zip(("x", "code", "x_nor", "x_global"), ev):
        e = max_rel_rows(t.cpu().numpy(), g[f"eval_{name}"])
        print(f"eval {name}: {e:.2e}")
        assert e < TOL, name

    net = _model(D, B, G, neg).train()
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)
    losses = []
    for it in range(3):
        with routing_taps() as routing:                  # the argmax tensors of the three max-pools of this forward
            x, code, x_nor, x_global = net(xt, yt, 1)
        loss_c = global_contrast(G, x_global, x, opt)
        loss_circle = circle_contrast(G, x, B, order=g["order"])
        loss = loss_circle + loss_c
        optim.zero_grad()
        loss.backward()
        if it == 0:
            # Truth = the oracle evaluated in fp64.  The 1e-4 bar is asserted against THAT; the golden is the
            # reference's own fp32 run, which itself sits up to 8e-4 (c1_d3 x) / 1.5e-3 (x_global) from the
            # fp64 truth, so against the golden we assert "within the golden's own distance to truth + 1e-4".
            o64 = _oracle_forward(g, D, neg, torch.float64)
            for name, t, r64 in zip(("x", "code", "x_nor", "x_global"), (x, code, x_nor, x_global), o64[:4]):
                mine = t.detach().cpu().numpy()
                e_truth = max_rel_rows(mine, r64.numpy())
                e_gold = max_rel_rows(mine, g[f"train_{name}"])
                gold_noise = max_rel_rows(g[f"train_{name}"], r64.numpy())
                print(f"train {name}: vs fp64 {e_truth:.2e}  vs golden {e_gold:.2e}  (golden vs fp64 {gold_noise:.2e})")
                assert e_truth < TOL, name
                assert e_gold < gold_noise + TOL, name
            for mine_l, key, r64 in ((loss_c, "loss_c", o64[4]), (loss_circle, "loss_circle", o64[5])):
                assert abs(mine_l.item() - r64) <= TOL * abs(r64), key
                assert abs(mine_l.item() - float(g[key])) <= abs(float(g[key]) - r64) + TOL * abs(r64), key
            g64, g32 = _oracle_grads(g, D, neg, torch.float64), _oracle_grads(g, D, neg, torch.float32)
            gmax = max(float(g[k]) for k in g if k.startswith("gradnorm/"))
            for k, p in net.named_parameters():
                if f"gradnone/{k}" in g:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0
                    continue
                gn = float(g[f"gradnorm/{k}"])
                if k in PRE_BN_BIAS:
                    wn = float(g[f"gradnorm/{k[:-4]}weight"])
                    assert p.grad is None or np.linalg.norm(p.grad.cpu().numpy()) <= 1e-2 * wn, k
                    continue
                mine = p.grad.cpu().numpy()
                scale = max(gn, 1e-2 * gmax)
                # vs the golden (reference fp32): within the golden's own distance to the fp64 truth + margin
                if f"grad/{k}" in g:
                    gold_noise = np.linalg.norm(g[f"grad/{k}"] - g64[k])
                    assert np.linalg.norm(mine - g[f"grad/{k}"]) <= gold_noise + GTOL * scale + 1e-5, k
                else:
                    gold_noise = abs(gn - np.linalg.norm(g64[k]))
                    assert abs(np.linalg.norm(mine.astype(np.float64)) - gn) <= gold_noise + GTOL * scale + 1e-5, k
            # tight check: at least as close to the fp64 truth as the reference's fp32 arithmetic is
            for k, p in net.named_parameters():
                if k in PRE_BN_BIAS or k not in g64 or k == "net3DV_3.7.bias":   # mathematically ~0 gradients
                    continue
                e_mine = rel_err(p.grad.cpu().numpy(), g64[k])
                e_t32 = rel_err(g32[k], g64[k])
                print(f"grad {k:20s} mine-vs-fp64 {e_mine:.2e}   torch-fp32-vs-fp64 {e_t32:.2e}")
                # noise floor = max-pool near-tie flips (discrete), it moves with the host's reduction order:
                # 2e-4 on the GPU box's CPU, 7e-3 in the build container for the SAME oracle code.  The
                # kernel-level gradient check (same upstream gradient, no flips) is test_gpu_sa_mlp.py: 3e-7.
                assert e_mine <= max(3 * e_t32, 5e-3), k
            # THE gradient bound: fp64 evaluation of the same graph on the same grouped rows with every DISCRETE
            # decision pinned to the one the HIP forward took -- the three max-pools gather at its argmax, the ReLUs keep
            # the elements it kept (each differing decision verified to be a numerical tie of the fp64 values) -- so the
            # comparison is pure arithmetic: every parameter tensor within 1e-4 (of its norm, floored at 1 % of the largest)
            from oracle import loss as OL
            from oracle.weights import formula_state_dict
            xr, cr = xt.permute(0, 2, 3, 1).reshape(-1, D), yt.permute(0, 2, 1, 3).reshape(-1, 3)
            def routed(dtype):
                xr_, xg_, _, q_, ties_ = forward64(xr, cr, formula_state_dict(D, neg_gamma=neg), G, S, K, DEV, routing=routing,
                                                   grad=True, dtype=dtype)
                (OL.global_contrast(G, xg_, xr_, B) + OL.circle_contrast(G, xr_, B, g["order"])).backward()
                return q_, ties_
            q64, ties = routed(torch.float64)
            q32, _ = routed(torch.float32)               # plain torch fp32, same routing: the conditioning yardstick
            print("ties", ties)
            assert max(v for k, v in ties.items() if not k.endswith("_flips")) < 1e-5, ties   # every differing decision was a tie
            gmax64 = max(float(q64[k].grad.norm()) for k, _ in net.named_parameters() if q64[k].grad is not None)
            for k, p in net.named_parameters():
                if k in PRE_BN_BIAS or q64[k].grad is None or k == "net3DV_3.7.bias":   # mathematically ~0 gradients
                    continue
                r = q64[k].grad.reshape(p.shape)
                err = float((p.grad.double() - r).norm())
                e32 = float((q32[k].grad.reshape(p.shape).double() - r).norm())
                print(f"grad {k:20s} vs routed fp64: {err / float(r.norm()):.2e}   torch-fp32 (same routing) {e32 / float(r.norm()):.2e}")
                assert err <= 1e-4 * max(float(r.norm()), 1e-2 * gmax64), (k, err, e32, float(r.norm()))
            sd = net.state_dict()
            for k in sd:
                if "running_" in k:
                    r64 = o64[6][k].numpy()
                    assert rel_err(sd[k].cpu().numpy(), r64) < 1e-5, k
                    assert rel_err(sd[k].cpu().numpy(), g[f"buf1/{k}"]) < rel_err(g[f"buf1/{k}"], r64) + 1e-5, k
                if "num_batches" in k:
                    assert int(sd[k]) == int(g[f"buf1/{k}"]), k
        optim.step()
        losses.append(loss.item())
    print("losses", losses, g["losses3"])
    l64 = o64[4] + o64[5]
    assert abs(losses[0] - l64) <= TOL * abs(l64)
    assert abs(losses[0] - g["losses3"][0]) <= abs(g["losses3"][0] - l64) + TOL * abs(l64)
    # steps 2-3 inherit the gradient noise floor above through Adam's normalised update (the reference's
    # fp32 run and its fp64 evaluation diverge by the same amount)
    np.testing.assert_allclose(losses, g["losses3"], rtol=3e-2)
    # parameters after the 3 Adam steps vs the reference's (param3/*).  Adam divides by sqrt(v), so elements whose gradient
    # is comparable to the fp32 noise get a full-size update of either sign: the reference's OWN fp32 run sits 1e-2..0.37
    # (d3, net3DV_1.0.weight) from the fp64 evaluation of the same three steps.  Truth = the oracle's three steps in
    # fp64; we must be at least as close to it as the reference's fp32 run is, and within that distance of the golden.
    from test_oracle_golden import check_param3
    from oracle import grouping as OG, step as OS
    from oracle.weights import formula_state_dict
    p0 = formula_state_dict(D, neg_gamma=neg)
    check_param3(g, {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}, p0, tol=None)   # bounds only
    sd64 = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone()) for k, v in p0.items()}
    adam = OS.AdamState(sd64)
    _, xt_o, yt_o = OG.group_points(g["points"], S, K, 0.06)
    grouped = (torch.from_numpy(xt_o).permute(0, 3, 1, 2).double(), torch.from_numpy(yt_o).view(G * B, 1, S, 3).transpose(1, 3).double())
    for it in range(3):
        OS.train_step(sd64, adam, None, B, G, S, K, 0.06, g["order"], epoch=0, grouped=grouped)
    mine_p = dict(net.named_parameters())
    for key in [k for k in g if k.startswith("param3/")]:
        k = key[len("param3/"):]
        if k in PRE_BN_BIAS:
            continue
        base = np.asarray(p0[k], dtype=np.float64)
        d_truth = sd64[k].detach().numpy().reshape(base.shape) - base
        d_gold = g[key].astype(np.float64).reshape(base.shape) - base
        d_mine = mine_p[k].detach().cpu().double().numpy().reshape(base.shape) - base
        nt = np.linalg.norm(d_truth)
        e_mine, e_gold, e_mg = np.linalg.norm(d_mine - d_truth) / nt, np.linalg.norm(d_gold - d_truth) / nt, np.linalg.norm(d_mine - d_gold) / nt
        print(f"param3 {k:20s} update: mine-vs-fp64 {e_mine:.3f}  golden-vs-fp64 {e_gold:.3f}  mine-vs-golden {e_mg:.3f}")
        assert e_mine <= 1.5 * e_gold + 0.02, (k, e_mine, e_gold)
        assert e_mg <= 2.0 * e_gold + 0.02, (k, e_mg, e_gold)


def test_step_vs_fp64_oracle_headline_shapes_small_batch():
    """N=2048 clouds (the headline cloud size), small batch: the training step through
    facl_amd.train_common.ContrastiveStep vs the oracle evaluated in fp64 (truth), with the oracle's own
    fp32 evaluation beside it as the noise floor."""
    from facl_amd.train_common import ContrastiveStep
    from oracle import encoder as E, grouping as OG, loss as OL
    from oracle.weights import formula_state_dict
    D, B, G, N, S, K = 4, 3, 4, 2048, 64, 64
    torch.manual_seed(5)
    clip = torch.rand(B, G, N, D) - 0.5
    opt = _opt(D, B, N)
    net = _model(D, B, G).train()
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)
    step = ContrastiveStep(net, optim, opt, G)
    order = np.array([2, 0, 3, 1])
    loss, loss_c, loss_circle = step(clip.to(DEV), epoch=0, order=order)

    pts = clip.permute(1, 0, 2, 3).reshape(-1, N, D).numpy()
    _, xt, yt = OG.group_points(pts, S, K, 0.16)
    M = G * B

    def ref(dtype):
        sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
              for k, v in formula_state_dict(D).items()}
        with torch.no_grad():
            x, code, x_nor, xg = E.encoder_forward(sd, torch.from_numpy(xt).permute(0, 3, 1, 2).to(dtype),
                                                   torch.from_numpy(yt).view(M, 1, S, 3).transpose(1, 3).to(dtype),
                                                   G, training=True)
            return float(OL.global_contrast(G, xg, x, B) + OL.circle_contrast(G, x, B, order))

    l64, l32 = ref(torch.float64), ref(torch.float32)
    e_mine, e_t32 = abs(loss.item() - l64) / abs(l64), abs(l32 - l64) / abs(l64)
    print(f"loss: mine {loss.item():.6f} fp64 {l64:.6f}  rel {e_mine:.2e}   torch-fp32 rel {e_t32:.2e}")
    assert e_mine < TOL


@pytest.mark.parametrize("G,B,C,world", [(6, 5, 32, 1), (24, 32, 512, 1), (4, 3, 16, 2)])
def test_fused_loss_kernel_vs_closed_form_fp64(G, B, C, world):
    """facl_contrast (value + d/dsim) vs the device-agnostic closed form of utils_my.global/circle_contrast in fp64,
    including the data-parallel form (keys from a larger global batch, clip_offset)."""
    from facl_amd.utils_my import circle_contrast, contrastive_losses, global_contrast
    torch.manual_seed(G * B)
    Bk = B * world
    off = B * (world - 1)
    keys0 = (torch.randn(G, Bk, C, dtype=torch.float64) * 0.3).to(DEV)
    x0 = keys0[:, off:off + B].reshape(G * B, C).clone()
    xg0 = (torch.randn(B, C, dtype=torch.float64) * 0.3).to(DEV)
    order = np.random.RandomState(0).permutation(G)

    def keys_of(x, dtype):
        # the keys are the all-gathered embeddings: this rank's rows ARE x (gradient flows back into x)
        k = keys0.to(dtype).clone()
        k[:, off:off + B] = x.view(G, B, C)
        return k.reshape(G * Bk, C)

    x64, xg64 = x0.clone().requires_grad_(True), xg0.clone().requires_grad_(True)
    k64 = keys_of(x64, torch.float64)
    ref = global_contrast(G, xg64, x64, None, x_keys=k64, clip_offset=off) + \
        circle_contrast(G, x64, B, order=order, x_keys=k64, clip_offset=off)
    gr = torch.autograd.grad(ref, (xg64, x64))
    x32, xg32 = x0.float().requires_grad_(True), xg0.float().requires_grad_(True)
    lc, lo = contrastive_losses(G, xg32, x32, order, x_keys=keys_of(x32, torch.float32), clip_offset=off)
    gm = torch.autograd.grad(lc + lo, (xg32, x32))
    assert abs(float(lc + lo) - float(ref)) <= 2e-6 * abs(float(ref))
    for a, b, nm in zip(gm, gr, ("xg", "x")):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5, nm


def test_tiny_golden_K8_through_replication():
    """tiny.npz = the reference's PointNet_Plus_fine(opt, gost=3, sample_num_level1=16, knn_K=8) in train mode: K=8
    runs through the 64-position kernels by 8x replication inside each unit (facl_amd/sa_mlp.py)."""
    from facl_amd.cn3d_model_conbag import PointNet_Plus_fine
    from facl_amd.utils_my import knn_radius_group
    from oracle.weights import formula_state_dict
    g = load_golden("tiny.npz")
    opt = _opt(4, 2, 128)
    opt.sample_num_level1 = 16
    net = PointNet_Plus_fine(opt, gost=3, sample_num_level1=16, knn_K=8)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(4).items()})
    net = net.to(DEV).train()
    xt, yt = knn_radius_group(torch.from_numpy(g["points"]).to(DEV), 16, 8, 0.06)
    out = net(xt, yt, 1)
    for name, t in zip(("x", "code", "x_nor", "x_global"), out):
        tol = 1e-3 if name == "x_global" else TOL          # BatchNorm1d over B = 2 rows: see tests/test_oracle_golden.py
        assert max_rel_rows(t.detach().cpu().numpy(), g[name]) < tol, name


@pytest.mark.parametrize("S,K", [(32, 128), (16, 32)])
def test_other_K_forward_backward_vs_fp64_oracle(S, K):
    """PointNet_Plus_fine's default (S=32, K=128) and a K | 64 case: outputs, gradients and running statistics."""
    from facl_amd.cn3d_model_conbag import PointNet_Plus_fine
    from facl_amd.utils_my import knn_radius_group
    from oracle import encoder as E, grouping as OG
    from oracle.weights import formula_state_dict
    # B*G = 24 clouds: netR_FC's BatchNorm1d over only 6 rows (B = 2) amplified fp32 rounding to ~1e-2 by itself
    D, B, G, N = 4, 8, 3, 512
    torch.manual_seed(K)
    pts = torch.rand(G * B, N, D) - 0.5
    opt = _opt(D, B, N)
    net = PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
    sdn = formula_state_dict(D)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()})
    net = net.to(DEV).train()
    xt, yt = knn_radius_group(pts.to(DEV), S, K, 0.1)
    x, code, x_nor, xg = net(xt, yt, 1)
    w = torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    ((x * w).sum() + xg.sum()).backward()

    _, xt_o, yt_o = OG.group_points(pts.numpy(), S, K, 0.1)
    sd = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
          for k, v in sdn.items()}
    keys = [k for k in sd if "running" not in k and "num_b" not in k]
    for k in keys:
        sd[k].requires_grad_(True)
    M = G * B
    xo, _, _, xgo = E.encoder_forward(sd, torch.from_numpy(xt_o).permute(0, 3, 1, 2).double(),
                                      torch.from_numpy(yt_o).view(M, 1, S, 3).transpose(1, 3).double(), G, True)
    ((xo * w.cpu().double()).sum() + xgo.sum()).backward()
    assert max_rel_rows(x.detach().cpu().numpy(), xo.detach().numpy()) < TOL
    sa_keys = [k for k, _ in net.named_parameters()
               if k.startswith("net3DV_1") and not k.endswith(("0.bias", "3.bias", "6.bias"))]
    gmax = max(float(sd[k].grad.norm()) for k in sa_keys)
    for k, p in net.named_parameters():
        if k not in sa_keys:
            continue
        # End-to-end gradients are bimodal: ~5e-6 when every max-pool argmax agrees with the fp64 oracle, ~5e-3..1e-2
        # as soon as ONE near-tie (two neighbours within fp32 rounding) resolves the other way -- which happens for the
        # fp32-MFMA and the split-bf16 kernels alike, on different seeds.  The bound covers the flip floor; the tight
        # kernel-level check for these K is tests/test_gpu_sa_mlp.py::test_sa_other_K_kernel_level
        ref = sd[k].grad.numpy()
        err = float(np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - ref))
        assert err <= 3e-2 * max(float(np.linalg.norm(ref)), 1e-2 * gmax), (k, err)
    st = net.state_dict()
    for k in st:
        if "net3DV_1" in k and "running" in k:
            assert rel_err(st[k].cpu().numpy(), sd[k].numpy()) < 1e-5, k


@pytest.mark.parametrize("prec", ["x3b", "x3"])
def test_optin_precisions_vs_default_path(prec):
    """The opt-in arithmetic (facl_amd.tail.precision; never the default): "x3b" runs three bf16 products per multiply-add in
    the BACKWARD GEMMs only -- its forward, loss and BatchNorm buffers must equal the default path's to the bit -- and "x3"
    everywhere.  Gradients of both against the default path's (which tests above hold to the fp64 oracle): norm-wise
    relative difference per parameter tensor: x3b <= 1e-3 (measured 6e-5), x3 <= 2e-2 (measured 7e-3: max-pool re-routing)."""
    from facl_amd.train_common import ContrastiveStep
    D, B, G, N = 3, 4, 8, 2048
    torch.manual_seed(9)
    clip = (torch.rand(B, G, N, D) - 0.5).to(DEV)
    order = np.random.RandomState(3).permutation(G)

    def run(p):
        net = _model(D, B, G).train()
        net.precision = p
        step = ContrastiveStep(net, torch.optim.SGD(net.parameters(), lr=0.0), _opt(D, B, N), G)
        loss, lc, lo = step(clip, epoch=0, order=order)
        torch.cuda.synchronize()
        grads = {k: q.grad.detach().double().clone() for k, q in net.named_parameters() if q.grad is not None}
        bufs = {k: v.detach().clone() for k, v in net.state_dict().items() if "running" in k}
        return float(loss), float(lc), float(lo), grads, bufs

    l0, c0, o0, g0, b0 = run("f32")
    l1, c1, o1, g1, b1 = run(prec)
    if prec == "x3b":
        assert (l1, c1, o1) == (l0, c0, o0)
        assert all(torch.equal(b0[k], b1[k]) for k in b0)
    else:
        assert abs(l1 - l0) < TOL * abs(l0)
    assert set(g0) == set(g1)
    # net3DV_3.7.bias has a mathematically ZERO gradient (a per-channel shift in front of netR_FC's train-mode BatchNorm1d
    # cancels): both paths return cancellation noise ~1e-7 of the other tensors' norms for it -- not compared
    gmax = max(float(v.norm()) for v in g0.values())
    live = [k for k in g0 if float(g0[k].norm()) > 1e-5 * gmax]
    assert set(g0) - set(live) <= {"net3DV_3.7.bias"}
    worst = max(float((g1[k] - g0[k]).norm() / g0[k].norm()) for k in live)
    print(f"{prec}: loss {l1:.6f} vs {l0:.6f}; worst gradient tensor rel diff {worst:.2e}")
    # "x3" perturbs the FORWARD by ~1e-5, which re-routes near-ties of the max-pools (whole gradient rows move): the same
    # 5e-3..1e-2 the reference's own fp32 gradients sit away from fp64 (GTOL above); "x3b" leaves the routing untouched
    assert worst < (1e-3 if prec == "x3b" else 2e-2)
