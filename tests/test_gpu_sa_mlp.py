"""GPU parity of the set-abstraction point-MLP (HIP passes) vs the torch-fp32/fp64 oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, max_rel_rows, rel_err, routing_taps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _poisoned_scratch():
    """Every test of this module runs with NaN-poisoned scratch (facl_amd._lib.poisoned): the partial-sum workspace and
    every output / scratch tensor the host layer allocates are filled with NaN bytes before the launches, so a partial
    row or output element left unwritten at a ragged shape fails the comparison instead of reading recycled memory."""
    from facl_amd import _lib
    with _lib.poisoned():
        yield


def _params(sd, dev, dtype=torch.float32):
    m = {"W1": "net3DV_1.0.weight", "b1": "net3DV_1.0.bias", "g1": "net3DV_1.1.weight", "be1": "net3DV_1.1.bias",
         "rm1": "net3DV_1.1.running_mean", "rv1": "net3DV_1.1.running_var",
         "W2": "net3DV_1.3.weight", "b2": "net3DV_1.3.bias", "g2": "net3DV_1.4.weight", "be2": "net3DV_1.4.bias",
         "rm2": "net3DV_1.4.running_mean", "rv2": "net3DV_1.4.running_var",
         "W3": "net3DV_1.6.weight", "b3": "net3DV_1.6.bias", "g3": "net3DV_1.7.weight", "be3": "net3DV_1.7.bias",
         "rm3": "net3DV_1.7.running_mean", "rv3": "net3DV_1.7.running_var"}
    return {k: torch.as_tensor(sd[v]).to(dev).to(dtype).contiguous() for k, v in m.items()}


def _oracle_pooled(sd, xt_MDSK, training, dtype):
    """net3DV_1 of the oracle (oracle/encoder.py), in `dtype`, on CPU."""
    from oracle import encoder as E
    sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
          for k, v in sd.items()}
    h = xt_MDSK.to(dtype)
    with torch.no_grad():
        for li in (0, 3, 6):
            h = E._conv_bn_relu(sd, "net3DV_1", li, h, training)
        pooled = F.max_pool2d(h, (1, h.shape[-1]), stride=1)
    return pooled.squeeze(-1).permute(0, 2, 1).reshape(-1, 256), sd


@pytest.mark.parametrize("D,neg,training", [(4, False, True), (3, False, True), (4, True, True), (4, False, False),
                                            (3, True, False)])
def test_sa_forward_vs_oracle(D, neg, training):
    from facl_amd import sa_mlp, utils_my
    from oracle.weights import formula_state_dict
    torch.manual_seed(D)
    M, N, S, K = 12, 512, 64, 64
    pts = (torch.rand(M, N, D) - 0.5)
    xt, yt = utils_my.knn_radius_group(pts.to(DEV), S, K, 0.06)
    sd = formula_state_dict(D, neg_gamma=neg)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D)
    assert x_rows.is_contiguous()
    pooled, ctx = sa_mlp.sa_mlp_forward(x_rows, p, training)
    ref64, sd64 = _oracle_pooled(sd, xt.cpu(), training, torch.float64)
    ref32, sd32 = _oracle_pooled(sd, xt.cpu(), training, torch.float32)
    e_mine = max_rel_rows(pooled.cpu().numpy(), ref64.numpy())
    e_t32 = max_rel_rows(ref32.numpy(), ref64.numpy())
    print(f"pooled: mine-vs-fp64 {e_mine:.2e}   torch-fp32-vs-fp64 {e_t32:.2e}")
    assert e_mine < 2e-5                      # kernel-level bar, 5x inside the 1e-4 feature budget
    assert e_mine < 3 * e_t32 + 2e-6          # at least as accurate as the reference's own fp32 arithmetic
    if training:
        for i, li in ((1, 1), (2, 4), (3, 7)):
            for nm, key in (("rm", "running_mean"), ("rv", "running_var")):
                assert rel_err(p[f"{nm}{i}"].cpu().numpy(), sd64[f"net3DV_1.{li}.{key}"].numpy()) < 2e-6, (nm, i)
        # argmax is a valid neighbour slot
        assert int(ctx["arg"].max()) < 64


@pytest.mark.parametrize("D,neg,K", [(4, False, 64), (3, True, 64), (4, False, 128), (3, False, 16)])
def test_sa_eval_one_kernel_vs_training_kernels_and_fp64(D, neg, K):
    """eval mode (the extraction path) is ONE kernel, x -> pooled (csrc/sa_eval.hip), with a2 scaled by the power
    of two of each unit's own maximum.  Against the fp64 oracle (2e-5 kernel-level bar, as the training passes) and against the
    training passes run with folded constants (the rounds 1-3 eval path: same arithmetic up to the operand scales)."""
    from facl_amd import sa_mlp, utils_my
    from oracle.weights import formula_state_dict
    torch.manual_seed(D + K)
    M, N, S = 6, 512, 1024 // K if K > 64 else 64
    pts = (torch.rand(M, N, D) - 0.5)
    xt, yt = utils_my.knn_radius_group(pts.to(DEV), S, K, 0.1)
    sd = formula_state_dict(D, neg_gamma=neg)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
    assert sa_mlp._EVAL_FUSED
    fused, _ = sa_mlp.sa_mlp_forward(x_rows, p, False, K=K)
    sa_mlp._EVAL_FUSED = False
    try:
        unfused, _ = sa_mlp.sa_mlp_forward(x_rows, p, False, K=K)
    finally:
        sa_mlp._EVAL_FUSED = True
    ref64, _ = _oracle_pooled(sd, xt.cpu(), False, torch.float64)
    e_f, e_u = max_rel_rows(fused.cpu().numpy(), ref64.numpy()), max_rel_rows(unfused.cpu().numpy(), ref64.numpy())
    print(f"one kernel vs fp64 {e_f:.2e}   training passes with folded constants vs fp64 {e_u:.2e}")
    assert fused.shape == unfused.shape == (M * S, 256)
    assert e_f < 2e-5 and e_f < 3 * e_u + 2e-6
    # a NaN coordinate poisons exactly its own group (MaxPool2d propagates NaN; the reference's features would be NaN there)
    x_bad = x_rows.clone()
    x_bad[5 * K + 3, 1] = float("nan")
    bad, _ = sa_mlp.sa_mlp_forward(x_bad, p, False, K=K)
    assert not torch.isfinite(bad[5]).any()
    keep = [i for i in range(M * S) if i != 5]
    assert torch.equal(bad[keep], fused[keep])


@pytest.mark.parametrize("scale_x,gscale", [(1.0, 1.0), (40.0, 1.0), (1.0, 2.0 ** -11), (1e-3, 300.0)])
def test_fp16x3_activation_bounds_are_rigorous_and_scale_free(scale_x, gscale):
    """The device-side operand maxima of the fp16x3 passes (csrc/common.h): in training the bounds facl_bn_finalize writes
    (Samuelson: |gamma| sqrt(n-1) sigma invstd + |beta|) must lie ABOVE the true activation maxima -- never an fp16 overflow --
    and within 2^12 of them; max(pooled) is exact.  Inputs scaled by 40 or 1e-3 and BatchNorm weights of 2^-11 or 300 (outside
    the rounds 1-3 range contract |a| < 4094, |w| < 255 / the 2^-11 subnormal edge) leave the result at 2e-5 of fp64."""
    from facl_amd import sa_mlp, utils_my
    from oracle.weights import formula_state_dict
    torch.manual_seed(3)
    D, M, N, S, K = 4, 8, 512, 64, 64
    pts = (torch.rand(M, N, D) - 0.5) * scale_x
    xt, _ = utils_my.knn_radius_group(pts.to(DEV), S, K, 0.1 * scale_x * scale_x)
    sd = dict(formula_state_dict(D))
    for li in (1, 4, 7):                                               # BatchNorm weights / biases of the three layers
        sd[f"net3DV_1.{li}.weight"] = np.asarray(sd[f"net3DV_1.{li}.weight"]) * np.float32(gscale)
        sd[f"net3DV_1.{li}.bias"] = np.asarray(sd[f"net3DV_1.{li}.bias"]) * np.float32(gscale)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D).contiguous()
    pooled, ctx = sa_mlp.sa_mlp_forward(x_rows, p, True, update_running=False)
    am = ctx["amax"].view(torch.float32)[:, ::32]                     # the 64 slots of each buffer (the words between them are unused)
    bound_a1, bound_a2, max_pooled = float(am[1].max()), float(am[2].max()), float(am[3].max())
    # fp64 truth of the activations
    q = {k: p[k].double() for k in sa_mlp._PARAM_ORDER}
    h = x_rows.double()
    acts = []
    for Wk, bk, gk, bek in (("W1", "b1", "g1", "be1"), ("W2", "b2", "g2", "be2"), ("W3", "b3", "g3", "be3")):
        y = h @ q[Wk].reshape(q[Wk].shape[0], -1).t() + q[bk]
        mean, var = y.mean(0), y.var(0, unbiased=False)
        h = torch.relu((y - mean) / torch.sqrt(var + 1e-5) * q[gk] + q[bek])
        acts.append(h)
    ref = acts[2].view(M * S, K, 256).max(dim=1).values
    t1, t2 = float(acts[0].max()), float(acts[1].max())
    print(f"a1: bound {bound_a1:.3e} / true max {t1:.3e};  a2: bound {bound_a2:.3e} / true max {t2:.3e};  pooled max {max_pooled:.3e} / {float(ref.max()):.3e}")
    assert t1 <= bound_a1 <= 4096 * t1 and t2 <= bound_a2 <= 4096 * t2
    assert abs(max_pooled - float(ref.max())) <= 1e-5 * float(ref.max())
    assert torch.isfinite(pooled).all()
    assert max_rel_rows(pooled.cpu().numpy(), ref.cpu().numpy()) < 2e-5


def test_sa_forward_c1_golden():
    """Stage pin against the reference's own net3DV_1 output (forward hook tap in make_goldens)."""
    from facl_amd import sa_mlp, utils_my
    from oracle.weights import formula_state_dict
    g = load_golden("c1_d4.npz")
    pts = torch.from_numpy(g["points"]).to(DEV)
    xt, yt = utils_my.knn_radius_group(pts, 64, 64, 0.06)
    p = _params(formula_state_dict(4), DEV)
    pooled, _ = sa_mlp.sa_mlp_forward(xt.permute(0, 2, 3, 1).reshape(-1, 4), p, True)
    mine = pooled.view(32, 64, 256).cpu().numpy()[::4]
    assert max_rel_rows(mine, g["train_pooled"]) < 1e-4


@pytest.mark.parametrize("D,neg", [(4, False), (3, False), (4, True)])
def test_sa_backward_vs_oracle_fp64(D, neg):
    """Parameter gradients of net3DV_1 for a random upstream gradient vs fp64 autograd of the oracle."""
    from facl_amd import sa_mlp, utils_my
    from oracle import encoder as E
    from oracle.weights import formula_state_dict
    torch.manual_seed(10 + D)
    M, N, S, K = 6, 512, 64, 64
    pts = (torch.rand(M, N, D) - 0.5)
    xt, yt = utils_my.knn_radius_group(pts.to(DEV), S, K, 0.06)
    sd = formula_state_dict(D, neg_gamma=neg)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D)
    names = sa_mlp._PARAM_ORDER
    params = [p[k].clone().requires_grad_(True) for k in names]
    state = dict(training=True, buffers={k: p[k] for k in ("rm1", "rv1", "rm2", "rv2", "rm3", "rv3")})
    pooled = sa_mlp.SAMLPFunction.apply(x_rows, state, *params)
    up = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    up = up * (torch.rand(up.shape, generator=torch.Generator().manual_seed(2)).to(DEV) > 0.3)   # some exact zeros
    (pooled * up).sum().backward()

    arg = pooled.grad_fn.c["arg"].cpu().long()                  # (M*S, 256): the position the HIP forward's max-pool chose

    def ref(dtype, routed=False):
        sdr = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
               for k, v in sd.items()}
        keys = [k for k in sdr if k.startswith("net3DV_1") and "running" not in k and "num_batches" not in k]
        for k in keys:
            sdr[k].requires_grad_(True)
        h = xt.cpu().to(dtype)
        for li in (0, 3, 6):
            h = E._conv_bn_relu(sdr, "net3DV_1", li, h, True)
        mx = F.max_pool2d(h, (1, K), stride=1)                   # (M,256,S,1)
        tie = 0.0
        if routed:
            # tie-proof: route the gradient through the KERNEL's argmax (a gather) instead of this arithmetic's own; every
            # decision that differs must be a numerical tie -- the gathered value equals the fp64 maximum to 1e-5
            got = torch.gather(h, 3, arg.view(M, S, 256).permute(0, 2, 1).unsqueeze(-1))
            tie = float(((mx - got).abs() / torch.maximum(mx.abs(), h.abs().mean())).max())
            mx = got
        pl = mx.squeeze(-1).permute(0, 2, 1).reshape(-1, 256)
        (pl * up.cpu().to(dtype)).sum().backward()
        return {k: sdr[k].grad for k in keys}, tie

    (g64r, tie), (g64, _), (g32, _) = ref(torch.float64, True), ref(torch.float64), ref(torch.float32)
    print(f"largest distance of a kernel-chosen position from the fp64 maximum: {tie:.2e}")
    assert tie < 1e-5
    keymap = {"W1": "net3DV_1.0.weight", "b1": "net3DV_1.0.bias", "g1": "net3DV_1.1.weight", "be1": "net3DV_1.1.bias",
              "W2": "net3DV_1.3.weight", "b2": "net3DV_1.3.bias", "g2": "net3DV_1.4.weight", "be2": "net3DV_1.4.bias",
              "W3": "net3DV_1.6.weight", "b3": "net3DV_1.6.bias", "g3": "net3DV_1.7.weight", "be3": "net3DV_1.7.bias"}
    for k, prm in zip(names, params):
        r64 = g64[keymap[k]].numpy()
        mine = None if prm.grad is None else prm.grad.cpu().numpy().reshape(r64.shape)
        if k in ("b1", "b2", "b3"):                    # mathematically zero (bias before a train-mode BN)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0
            continue
        e_routed = rel_err(mine, g64r[keymap[k]].numpy())
        e_mine = rel_err(mine, r64)
        e_t32 = rel_err(g32[keymap[k]].numpy(), r64)
        print(f"{k}: vs routed fp64 {e_routed:.2e}   vs plain fp64 {e_mine:.2e}  torch-fp32-vs-fp64 {e_t32:.2e}")
        # THE bound: with the max-pool routing pinned to the kernel's own (verified-tie) decisions the
        # comparison is pure arithmetic -- 1e-4 per parameter, measured 2e-7 .. 3e-6
        assert e_routed < 1e-4, k
        # second, looser check against the un-routed fp64 graph: where torch's own fp32 run lands on the other side of a
        # near-tie than fp64 does it sits 3e-3..1e-2 away on every parameter, and an fp32-grade kernel may land on either side
        assert e_mine < max(1e-4, 1.5 * e_t32), k
        assert e_mine < 3 * e_t32 + 1e-5, k


@pytest.mark.parametrize("S,K", [(32, 128), (16, 32), (16, 8)])
def test_sa_other_K_kernel_level(S, K):
    """net3DV_1 alone for K = 64*R and K | 64, fixed upstream gradient: pooled output, all parameter gradients and the
    running statistics vs the fp64 oracle (no tie-flip noise: the upstream gradient does not depend on the forward)."""
    from facl_amd import sa_mlp, utils_my
    from oracle import encoder as E
    from oracle.weights import formula_state_dict
    D, M, N = 4, 5, 512
    torch.manual_seed(S + K)
    pts = (torch.rand(M, N, D) - 0.5)
    xt, yt = utils_my.knn_radius_group(pts.to(DEV), S, K, 0.1)
    sd = formula_state_dict(D)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D)
    names = sa_mlp._PARAM_ORDER
    params = [p[k].clone().requires_grad_(True) for k in names]
    state = dict(training=True, K=K, buffers={k: p[k] for k in ("rm1", "rv1", "rm2", "rv2", "rm3", "rv3")})
    pooled = sa_mlp.SAMLPFunction.apply(x_rows, state, *params)
    assert pooled.shape == (M * S, 256)
    up = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (pooled * up).sum().backward()
    sdr = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
           for k, v in sd.items()}
    keys = [k for k in sdr if k.startswith("net3DV_1") and "running" not in k and "num_batches" not in k]
    for k in keys:
        sdr[k].requires_grad_(True)
    h = xt.cpu().double()
    for li in (0, 3, 6):
        h = E._conv_bn_relu(sdr, "net3DV_1", li, h, True)
    pl = F.max_pool2d(h, (1, K), stride=1).squeeze(-1).permute(0, 2, 1).reshape(-1, 256)
    (pl * up.cpu().double()).sum().backward()
    assert max_rel_rows(pooled.detach().cpu().numpy(), pl.detach().numpy()) < 2e-5
    keymap = {"W1": "net3DV_1.0.weight", "g1": "net3DV_1.1.weight", "be1": "net3DV_1.1.bias",
              "W2": "net3DV_1.3.weight", "g2": "net3DV_1.4.weight", "be2": "net3DV_1.4.bias",
              "W3": "net3DV_1.6.weight", "g3": "net3DV_1.7.weight", "be3": "net3DV_1.7.bias"}
    for k, prm in zip(names, params):
        if k not in keymap:
            continue
        r64 = sdr[keymap[k]].grad.numpy()
        assert rel_err(prm.grad.cpu().numpy().reshape(r64.shape), r64) < 2e-5, k
    for i, li in ((1, 1), (2, 4), (3, 7)):
        for nm, key in (("rm", "running_mean"), ("rv", "running_var")):
            assert rel_err(p[f"{nm}{i}"].cpu().numpy(), sdr[f"net3DV_1.{li}.{key}"].numpy()) < 2e-6, (nm, i)


def test_sa_headline_size_forward_backward_vs_torch_fp64():
    """B=32, T=24, N=2048 (M=768 clouds, 3,145,728 grouped positions): the set-abstraction point-MLP against plain
    PyTorch fp64 ops on the GPU (1x1 convs as matmuls, train-mode BN with batch statistics, ReLU, max over K) -- an
    implementation independent of the oracle -- forward outputs, running statistics and all parameter gradients."""
    from facl_amd import sa_mlp, utils_my
    from oracle.weights import formula_state_dict
    D, M, N, S, K = 3, 768, 2048, 64, 64
    torch.manual_seed(0)
    pts = (torch.rand(M, N, D, device=DEV) - 0.5)
    xt, _ = utils_my.knn_radius_group(pts, S, K, 0.16)
    del pts
    sd = formula_state_dict(D)
    p = _params(sd, DEV)
    x_rows = xt.permute(0, 2, 3, 1).reshape(-1, D).contiguous()                   # (P, D), P = M*S*K
    del xt
    params = [p[k].clone().requires_grad_(True) for k in sa_mlp._PARAM_ORDER]
    state = {"buffers": {k: p[k] for k in ("rm1", "rv1", "rm2", "rv2", "rm3", "rv3")}, "training": True}
    with routing_taps() as taps:                         # the kernels' discrete decisions: max-pool argmax, ReLU signs
        pooled = sa_mlp.SAMLPFunction.apply(x_rows, state, *params)
    w = torch.randn(pooled.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (pooled * w).sum().backward()

    # ---- fp64 reference, layer by layer, plus its fp32 twin (plain torch ops, same graph): the conditioning yardstick
    my_arg = pooled.grad_fn.c["arg"].long()

    def ref(dtype):
        q = {k: p[k].detach().to(dtype).requires_grad_(True) for k in sa_mlp._PARAM_ORDER}
        h = x_rows.to(dtype)
        P = h.shape[0]
        stats, relu_flips, relu_tie = [], 0, 0.0
        for li, (Wk, bk, gk, bek) in enumerate((("W1", "b1", "g1", "be1"), ("W2", "b2", "g2", "be2"), ("W3", "b3", "g3", "be3"))):
            y = h @ q[Wk].reshape(q[Wk].shape[0], -1).t() + q[bk]
            mean, var = y.mean(0), y.var(0, unbiased=False)
            stats.append((mean.detach(), (var * P / (P - 1)).detach()))
            z = (y - mean) / torch.sqrt(var + 1e-5) * q[gk] + q[bek]
            del y
            if li < 2:
                # decision-pinned: the ReLU keeps exactly the elements the kernels kept; a decision that
                # differs from this arithmetic's own must be a numerical tie (|z| <= 1e-5 of the layer's mean |z|).  At this
                # size a SINGLE flipped element moves dbeta2 by 2e-4 (5 flips of 2e8 decisions: 8e-4; gpurun_out/r4c_dz2.log)
                mask = taps[f"relu_sa{li + 1}"]
                with torch.no_grad():
                    diff = mask != (z > 0)
                    relu_flips += int(diff.sum())
                    relu_tie = max(relu_tie, float((z.abs() * diff).max() / z.abs().mean()))
                h = z * mask
            else:
                h = z                                    # layer 3: ReLU behind the max-pool (monotone), pinned there
            del z
        h3 = h.view(M * S, K, 256)
        mx, mx_arg = h3.max(dim=1)
        # the gradient is routed through the KERNEL's argmax; every differing decision must be a numerical tie (gathered
        # value == the maximum to 1e-5 of the activation scale)
        got = torch.gather(h3, 1, my_arg.unsqueeze(1)).squeeze(1)
        tie = float(((mx - got).abs() / torch.maximum(mx.abs(), h3.detach().abs().mean())).max())
        flips = int((my_arg != mx_arg).sum())
        out = torch.relu(mx.detach())
        with torch.no_grad():
            diff = taps["relu_sa3"] != (got > 0)
            relu_flips += int(diff.sum())
            relu_tie = max(relu_tie, float((got.abs() * diff).max() / got.abs().mean()))
        got = got * taps["relu_sa3"]
        del h, h3, mx
        (got * w.to(dtype)).sum().backward()
        if dtype == torch.float64:
            print(f"ReLU decisions that differ from fp64's own: {relu_flips}, largest |z| among them {relu_tie:.2e} of the mean |z|")
            assert relu_tie < 1e-5
        return q, stats, out, tie, flips

    q, stats, ref64, tie, flips = ref(torch.float64)
    # the max-pool decisions themselves: bit-exact argmax except where two neighbours are within fp32 rounding
    print(f"argmax decisions {my_arg.numel()}, fp32-vs-fp64 flips {flips}; largest distance of a kernel-chosen position "
          f"from the fp64 maximum: {tie:.2e}")
    assert flips < 1e-4 * my_arg.numel()        # measured 436 of 12.6 M (y3 carries ~1e-6 relative fp32 noise)
    assert tie < 1e-5
    assert max_rel_rows(pooled.detach().cpu().numpy(), ref64.cpu().numpy()) < 2e-5
    for i, (mean, uvar) in enumerate(stats, 1):                                    # momentum 0.1 from (0, 1)
        sd0m, sd0v = torch.as_tensor(sd[f"net3DV_1.{3 * i - 2}.running_mean"]).double().to(DEV), \
            torch.as_tensor(sd[f"net3DV_1.{3 * i - 2}.running_var"]).double().to(DEV)
        assert rel_err(p[f"rm{i}"].cpu().numpy(), (0.9 * sd0m + 0.1 * mean).cpu().numpy()) < 5e-6, i
        assert rel_err(p[f"rv{i}"].cpu().numpy(), (0.9 * sd0v + 0.1 * uvar).cpu().numpy()) < 5e-6, i
    g64 = {k: q[k].grad.clone() for k in q}
    del q, stats, ref64
    q32 = ref(torch.float32)[0]
    gmax = max(float(g64[k].norm()) for k in g64)
    bad = []
    for k, mine in zip(sa_mlp._PARAM_ORDER, params):
        if k in ("b1", "b2", "b3"):
            # a conv bias in front of a train-mode BN has a mathematically zero gradient (returned as None)
            assert mine.grad is None and float(g64[k].norm()) < 1e-6 * gmax, k
            continue
        err = float((mine.grad.double() - g64[k]).norm())
        e32 = float((q32[k].grad.double() - g64[k]).norm())
        print(f"{k:4s} |g| {float(g64[k].norm()):.3e}  mine-vs-routed-fp64 {err / float(g64[k].norm()):.2e}   "
              f"torch-fp32 (same routing) {e32 / float(g64[k].norm()):.2e}")
        # With every discrete decision pinned to the kernel's own (verified ties) the comparison is pure arithmetic:
        # 1e-4 of the tensor's gradient norm (measured ~1e-6); the fp32 twin is printed as the conditioning yardstick
        if err > 1e-4 * max(float(g64[k].norm()), 1e-2 * gmax):
            bad.append((k, err, e32, float(g64[k].norm())))
    assert not bad, bad


@pytest.mark.parametrize("nunits", [1, 3, 5, 7, 1021, 1026])
def test_sa_ragged_unit_counts(nunits):
    """Unit counts that do not fill the persistent grids (bwd1 runs 4 producer/consumer pairs per workgroup and every
    wave of a workgroup takes the same number of rounds: partial rounds and idle pairs must be exact no-ops)."""
    from facl_amd import sa_mlp
    from oracle.weights import formula_state_dict
    D, K = 4, 64
    torch.manual_seed(nunits)
    x_rows = ((torch.rand(nunits * K, D, device=DEV) - 0.5) * 0.8).contiguous()
    sd = formula_state_dict(D)
    p = _params(sd, DEV)
    params = [p[k].clone().requires_grad_(True) for k in sa_mlp._PARAM_ORDER]
    state = {"buffers": {k: p[k] for k in ("rm1", "rv1", "rm2", "rv2", "rm3", "rv3")}, "training": True}
    pooled = sa_mlp.SAMLPFunction.apply(x_rows, state, *params)
    w = torch.randn(pooled.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (pooled * w).sum().backward()
    q = {k: p[k].detach().double().requires_grad_(True) for k in sa_mlp._PARAM_ORDER}
    h = x_rows.double()
    for Wk, bk, gk, bek in (("W1", "b1", "g1", "be1"), ("W2", "b2", "g2", "be2"), ("W3", "b3", "g3", "be3")):
        y = h @ q[Wk].reshape(q[Wk].shape[0], -1).t() + q[bk]
        mean, var = y.mean(0), y.var(0, unbiased=False)
        h = torch.relu((y - mean) / torch.sqrt(var + 1e-5) * q[gk] + q[bek])
    h3 = h.view(nunits, K, 256)
    ref, ref_arg = h3.max(dim=1)
    assert max_rel_rows(pooled.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 5e-5
    # a max-pool decision that differs from fp64 (two neighbours within fp32 rounding) re-routes a whole gradient row: the
    # fp64 graph is routed through the kernel's decisions (each verified to be a tie), so flips no longer loosen the bound
    my_arg = pooled.grad_fn.c["arg"].long()
    flips = int((my_arg != ref_arg).sum())
    assert flips <= max(2, 1e-4 * ref_arg.numel())
    got = torch.gather(h3, 1, my_arg.unsqueeze(1)).squeeze(1)
    assert float(((ref - got).abs() / torch.maximum(ref.abs(), h3.detach().abs().mean())).max()) < 1e-5
    (got * w.double()).sum().backward()
    tol = 2e-3
    gmax = max(float(q[k].grad.norm()) for k in q)
    for k, mine in zip(sa_mlp._PARAM_ORDER, params):
        if k in ("b1", "b2", "b3"):
            continue
        g64 = q[k].grad
        err = float((mine.grad.double() - g64).norm())
        # tiny batches make train-mode BN ill-conditioned (64 positions at nunits = 1): scaled like the golden tests
        assert err <= tol * max(float(g64.norm()), 1e-2 * gmax), (k, err, float(g64.norm()), flips)


@pytest.mark.parametrize("D", [3, 4])
def test_bn1_chain_equals_its_three_launches_bitwise(D):
    """facl_sa_bn1_chain (moments -> sums of y1 -> BatchNorm-1 constants, running statistics, activation bound -> folded layer-1
    table, one launch) against facl_bn1_sums_from_moments + facl_bn_finalize + facl_sa_l1tab: every output bit for bit."""
    from facl_amd import _lib
    lib = _lib.load_library()
    p = _lib.ptr
    g = torch.Generator(device=DEV).manual_seed(17 + D)
    P = 50000.0
    x = torch.randn(4096, D, device=DEV, generator=g, dtype=torch.float64)
    mom = torch.cat(((x.sum(0) * (P / 4096)), ((x.t() @ x) * (P / 4096)).reshape(-1))).contiguous()
    W1 = (torch.randn(64, D, device=DEV, generator=g) * 0.5).contiguous()
    b1 = torch.randn(64, device=DEV, generator=g)
    gam = torch.randn(64, device=DEV, generator=g)               # negative gammas included
    bet = torch.randn(64, device=DEV, generator=g)
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        sums = torch.full((64, 2), float("nan"), device=DEV, dtype=torch.float64)
        bnc = torch.full((5, 64), float("nan"), device=DEV)
        tab = torch.full((64, 8), float("nan"), device=DEV)
        amax = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=DEV)
        if fused:
            _lib.check(lib.facl_sa_bn1_chain(p(mom), P, D, p(W1), p(b1), p(gam), p(bet), 1e-5, 0.1, p(rm), p(rv), p(sums), p(bnc),
                                             p(amax), p(tab), _lib.stream()), "chain")
        else:
            _lib.check(lib.facl_bn1_sums_from_moments(p(mom), P, D, p(W1), p(b1), p(sums), _lib.stream()), "sums")
            _lib.check(lib.facl_bn_finalize(p(sums), 64, P, p(gam), p(bet), 1e-5, 0.1, p(rm), p(rv), p(bnc), p(amax), None,
                                            _lib.stream()), "finalize")
            _lib.check(lib.facl_sa_l1tab(p(W1), p(b1), D, p(bnc[2]), p(bnc[3]), p(tab), None, None, _lib.stream()), "l1tab")
        outs.append((sums, bnc, tab, rm, rv, amax))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[1][1]).all() and torch.isfinite(outs[1][2]).all()
