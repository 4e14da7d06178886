"""GPU tests of the small tail pieces: gobaol_max_pool tie rule, the linear probe against the reference fixture."""
import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("G,B,C", [(10, 4, 1024), (24, 32, 1024), (3, 5, 8), (1, 7, 16)])
def test_viewmax_first_view_wins_ties(G, B, C):
    """gobaol_max_pool (cn3d_model_conbag.py:225-226) = MaxPool2d over the concatenated (view, centroid) axis: on equal
    values the FIRST position wins, i.e. the lowest view index, and only that view's row receives the gradient
    (F.max_pool2d's backward).  Rows are view-major (g*B + b).  Inputs are built FROM ties: every value is drawn from
    3 levels, so most (clip, channel) columns have their maximum in several views."""
    from facl_amd import tail
    torch.manual_seed(G * 100 + B)
    x = torch.randint(0, 3, (G * B, C), device=DEV).float().requires_grad_(True)
    out = tail.view_max(x, G)
    ref, arg = x.detach().view(G, B, C).max(dim=0)
    # torch.max's argmax on ties is unspecified: derive the first-wins index explicitly
    first = (x.detach().view(G, B, C) == ref.unsqueeze(0)).float().argmax(dim=0)      # argmax of a 0/1 mask = first 1
    assert torch.equal(out, ref)
    w = torch.randn(B, C, device=DEV)
    (out * w).sum().backward()
    expect = torch.zeros(G, B, C, device=DEV)
    expect.scatter_(0, first.unsqueeze(0), w.unsqueeze(0))
    assert torch.equal(x.grad.view(G, B, C), expect)
    if G > 1:
        assert int((x.detach().view(G, B, C) == ref.unsqueeze(0)).sum(0).max()) > 1          # ties really occurred
    # the reference op itself on the same data (first-wins is F.max_pool2d's documented CPU/GPU behaviour on equal values)
    xr = x.detach().clone().requires_grad_(True)
    seq = xr.view(G, B, C).permute(1, 2, 0).reshape(B, C, G, 1)
    pooled = torch.nn.functional.max_pool2d(seq, (G, 1), stride=1).view(B, C)
    (pooled * w).sum().backward()
    assert torch.equal(pooled.detach(), out.detach()) and torch.equal(xr.grad, x.grad)


def test_final_fc_against_reference_fixture():
    """facl_amd.linear_classify.Final_FC (L2-normalise + the MFMA GEMM) vs the output of the REFERENCE class on the same
    features and weights (tests/golden/fc.npz, tools/make_goldens.py: make_fc): logits, CE loss, gradients."""
    from facl_amd.linear_classify import Final_FC
    from oracle.weights import _hash_uniform
    g = load_golden("fc.npz")
    net = Final_FC().to(DEV)
    with torch.no_grad():
        net.fc.weight.copy_(torch.as_tensor((0.02 * _hash_uniform(120 * 22 * 512, 555)).astype(np.float32)).view(120, -1))
        net.fc.bias.copy_(torch.as_tensor((0.02 * _hash_uniform(120, 556)).astype(np.float32)))
    assert list(net.state_dict().keys()) == ["fc.weight", "fc.bias"]
    pred = net(torch.from_numpy(g["x"]).to(DEV))
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], rtol=1e-4, atol=2e-6)
    loss = torch.nn.functional.cross_entropy(pred, torch.from_numpy(g["y"]).to(DEV))
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    np.testing.assert_allclose(net.fc.bias.grad.cpu().numpy(), g["grad_bias"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(net.fc.weight.grad.reshape(-1)[:4096].cpu().numpy(), g["gradhead_weight"], rtol=1e-3, atol=1e-7)
    gn = float(net.fc.weight.grad.double().norm())
    assert abs(gn - float(g["gradnorm_weight"])) < 1e-4 * float(g["gradnorm_weight"])


@pytest.mark.parametrize("M,C,K", [(800, 512, 64), (5, 512, 64), (33, 64, 7)])
def test_normalize_map_vs_torch(M, C, K):
    """facl_normalize_map == F.normalize(x, p=2, dim=1) + F.linear (cn3d_model_conbag.py:231-232), forward and backward
    (fp64 torch as truth), including an all-zero row (eps clamp)."""
    import torch.nn.functional as F
    from facl_amd import tail
    torch.manual_seed(M)
    x = torch.randn(M, C, device=DEV) * 3.0
    x[M // 2] = 0.0
    Wm = torch.randn(K, C, device=DEV) * 0.05
    xa, wa = x.clone().requires_grad_(True), Wm.clone().requires_grad_(True)
    xn, code = tail.normalize_map(xa, wa)
    xr, wr = x.double().requires_grad_(True), Wm.double().requires_grad_(True)
    xn_r = F.normalize(xr, p=2, dim=1)
    code_r = F.linear(xn_r, wr)
    assert float((xn.double() - xn_r).abs().max()) < 1e-6
    assert float((code.double() - code_r).abs().max()) < 1e-5
    g1, g2 = torch.randn_like(xn), torch.randn_like(code)
    ((xn * g1).sum() + (code * g2).sum()).backward()
    ((xn_r * g1.double()).sum() + (code_r * g2.double()).sum()).backward()
    keep = torch.ones(M, dtype=torch.bool, device=DEV)
    keep[M // 2] = False                                     # d/dx at x = 0 is not defined (torch returns 0/eps terms)
    assert float((xa.grad.double() - xr.grad)[keep].norm() / xr.grad[keep].norm()) < 1e-5
    assert float((wa.grad.double() - wr.grad).norm() / wr.grad.norm()) < 1e-5


@pytest.mark.parametrize("G,B,C,world", [(6, 5, 32, 1), (24, 32, 512, 1), (4, 3, 16, 2), (10, 4, 512, 1)])
def test_contrastive_pair_on_stacked_embeddings_vs_closed_form_fp64(G, B, C, world):
    """utils_my._ContrastivePair (one similarity GEMM on [x ; x_global], facl_contrast_pair, dgrad + wgrad) vs the
    device-agnostic closed form of global_contrast / circle_contrast in fp64: both values and the gradients wrt the
    stacked embeddings, single-process (keys = the view rows themselves) and data-parallel (gathered keys) forms."""
    from facl_amd.utils_my import circle_contrast, contrastive_losses_stacked, global_contrast
    torch.manual_seed(G * B + C)
    Bk, off = B * world, B * (world - 1)
    keys0 = (torch.randn(G, Bk, C, dtype=torch.float64) * 0.3).to(DEV)
    x0 = keys0[:, off:off + B].reshape(G * B, C).clone()
    xg0 = (torch.randn(B, C, dtype=torch.float64) * 0.3).to(DEV)
    order = np.random.RandomState(1).permutation(G)

    def keys_of(x, dtype):
        k = keys0.to(dtype).clone()
        k[:, off:off + B] = x.view(G, B, C)
        return k.reshape(G * Bk, C)

    x64, xg64 = x0.clone().requires_grad_(True), xg0.clone().requires_grad_(True)
    k64 = keys_of(x64, torch.float64) if world > 1 else None
    lc_r = global_contrast(G, xg64, x64, None, x_keys=k64, clip_offset=off)
    lo_r = circle_contrast(G, x64, B, order=order, x_keys=k64, clip_offset=off)
    gr = torch.autograd.grad(0.7 * lc_r + 1.3 * lo_r, (xg64, x64))
    st = torch.cat((x0, xg0), 0).float().requires_grad_(True)
    k32 = keys_of(st[:G * B], torch.float32) if world > 1 else None
    lc, lo = contrastive_losses_stacked(G, st, order, x_keys=k32, clip_offset=off)
    (0.7 * lc + 1.3 * lo).backward()                        # distinct upstream gradients exercise facl_scale_rows2
    assert abs(float(lc) - float(lc_r)) <= 2e-6 * abs(float(lc_r))
    assert abs(float(lo) - float(lo_r)) <= 2e-6 * abs(float(lo_r))
    g = st.grad.double()
    assert float((g[:G * B] - gr[1]).norm() / gr[1].norm()) < 2e-5
    assert float((g[G * B:] - gr[0]).norm() / gr[0].norm()) < 2e-5


def test_contrastive_pair_rejects_bad_order_and_clip_offset():
    """Error behaviour of the loss entry: a host-side `order` that is no permutation of range(G) raises before any launch;
    a clip window outside the key columns returns FACL_E_SHAPE; a corrupt DEVICE order is clamped in the kernel (finite
    values, no out-of-bounds access)."""
    from facl_amd.utils_my import contrastive_losses_stacked
    G, B, C = 4, 3, 16
    st = torch.randn((G + 1) * B, C, device=DEV)
    for bad in ([0, 1, 2, 4], [0, 1, 1, 2], [0, 1, 2]):
        with pytest.raises(ValueError):
            contrastive_losses_stacked(G, st, np.array(bad))
    keys = torch.randn(G * 2 * B, C, device=DEV)
    for off in (-1, B + 1):
        with pytest.raises(RuntimeError):
            contrastive_losses_stacked(G, st, np.arange(G), x_keys=keys, clip_offset=off)
    lc, lo = contrastive_losses_stacked(G, st, torch.tensor([0, 7, -3, 1], device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(lc) and torch.isfinite(lo)


def test_fused_adam_equals_torch_adam():
    """facl_amd.optim.FusedAdam (one launch over all tensors, device-resident step / lr) vs torch.optim.Adam with the
    reference's hyper-parameters (cn3d_train_motion_GL.py:180): 6 steps on tensors of ragged sizes, a parameter without a
    gradient, a learning-rate change in between."""
    from facl_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(1024, 1024), (64, 3, 1, 1), (5,), (2049,), (512, 1024), (7, 11)]
    pa = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam(pa, lr=3e-4, betas=(0.5, 0.999), eps=1e-6)
    ob = torch.optim.Adam(pb, lr=3e-4, betas=(0.5, 0.999), eps=1e-6)
    for it in range(6):
        if it == 3:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 3e-4 * 0.7
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 2:
                a.grad = b.grad = None                       # like mapping.weight: no gradient, no update
                continue
            g = torch.randn_like(a) * (10.0 ** (i - 3))
            a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max())
    sd = oa.state_dict()
    assert int(sd["state"][0]["step"]) == 6 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def _fc_head_reference_fp64(x_pre, G, W1, b1, gamma, beta, W2, b2, rm, rv, momentum=0.1, eps=1e-5):
    """netR_FC applied the reference's way (cn3d_model_conbag.py:225-229): once on the view rows, once on the per-clip view
    maximum, each with its own batch statistics; the running buffers are updated twice, view rows first.  fp64 torch."""
    M, C = x_pre.shape
    B = M // G
    xg = x_pre.view(G, B, C).max(dim=0).values
    outs = []
    for h in (x_pre, xg):
        y = h @ W1.t() + b1
        mean, var = y.mean(0), y.var(0, unbiased=False)
        n = y.shape[0]
        rm = (1 - momentum) * rm + momentum * mean.detach()
        rv = (1 - momentum) * rv + momentum * (var.detach() * n / max(n - 1, 1))
        a = torch.relu((y - mean) / torch.sqrt(var + eps) * gamma + beta)
        outs.append(a @ W2.t() + b2)
    return torch.cat(outs, 0), rm, rv


@pytest.mark.parametrize("G,B,Cin,C,dim", [(24, 32, 1024, 1024, 512), (8, 4, 64, 128, 32), (2, 16, 32, 64, 16)])
def test_fc_head_two_segment_batchnorm_vs_two_fp64_calls(G, B, Cin, C, dim):
    """tail.fc_head on the fused two-segment BatchNorm kernels (csrc/fchead.hip: slice statistics -> both finalisations ->
    one apply; backward: slice sums -> constants + gamma / beta gradients of both segments -> one apply; the view maximum
    written into the stacked input by the launch that reads it; the bias gradient as one column-sum launch) against the
    reference's TWO netR_FC calls evaluated in fp64: outputs, both running-statistics updates, every parameter gradient and
    the gradient of x_pre (view-max routing included).  M % 32 == 0 in all cases, so the fused path is the one that runs; the
    clip segment has 4 .. 32 rows, the case the GEMM-epilogue fp32 statistics could not serve.  The single-segment kernels
    (FACL_FC_FUSED=0 semantics, forced here through the module switch) must give the same numbers."""
    from facl_amd import tail
    from facl_amd.cn3d_model_conbag import _Affine, _BatchNormState
    torch.manual_seed(G * 1000 + B)
    M = G * B
    x_pre = (torch.randn(M, Cin, device=DEV) * 2.0).abs()              # post-ReLU-like features
    lin1, lin2, bn = _Affine((C, Cin), Cin).to(DEV), _Affine((dim, C), C).to(DEV), _BatchNormState(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, device=DEV) * 0.5 + 1.0)
        bn.weight[0] = -0.7                                            # a negative gamma
        bn.bias.copy_(torch.randn(C, device=DEV) * 0.2)
        bn.running_mean.copy_(torch.randn(C, device=DEV))
        bn.running_var.copy_(torch.rand(C, device=DEV) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    gout = torch.randn(M + B, dim, device=DEV)

    def run(fused):
        bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
        for p in (*lin1.parameters(), *lin2.parameters(), *bn.parameters()):
            p.grad = None
        xa = x_pre.clone().requires_grad_(True)
        prev, tail._FC_FUSED = tail._FC_FUSED, fused
        try:
            out = tail.fc_head(xa, G, lin1, bn, lin2, True)
            (out * gout).sum().backward()
        finally:
            tail._FC_FUSED = prev
        return (out.detach(), xa.grad, lin1.weight.grad, bn.weight.grad, bn.bias.grad, lin2.weight.grad, lin2.bias.grad,
                bn.running_mean.clone(), bn.running_var.clone())

    steps0 = bn.steps
    got = run(True)
    assert bn.steps == steps0 + 2                                      # two BatchNorm calls counted
    old = run(False)
    # fp64 truth
    d = lambda t: t.detach().double()
    xr = d(x_pre).requires_grad_(True)
    W1, b1, W2, b2 = (d(lin1.weight).requires_grad_(True), d(lin1.bias), d(lin2.weight).requires_grad_(True), d(lin2.bias).requires_grad_(True))
    ga, be = d(bn.weight).requires_grad_(True), d(bn.bias).requires_grad_(True)
    ref, rm, rv = _fc_head_reference_fp64(xr, G, W1, b1, ga, be, W2, b2, d(rm0), d(rv0))
    (ref * d(gout)).sum().backward()
    truth = (ref.detach(), xr.grad, W1.grad, ga.grad, be.grad, W2.grad, b2.grad, rm, rv)
    names = ("out", "dx_pre", "dW1", "dgamma", "dbeta", "dW2", "db2", "running_mean", "running_var")
    for name, g_, o_, t_ in zip(names, got, old, truth):
        scale = float(t_.abs().max()) + 1e-30
        e_new, e_old = float((g_.double() - t_).abs().max()) / scale, float((o_.double() - t_).abs().max()) / scale
        assert e_new < 2e-5, (name, e_new)                             # measured 1e-7 .. 4e-6 (bf16x6 GEMMs, fp64 statistics)
        assert e_new < 4 * e_old + 1e-6, (name, e_new, e_old)          # and no worse than the single-segment kernels


def test_col_sums_viewmax_stack_and_loss_sum_entries():
    """The three small entries of the FC head / loss glue through the C ABI: facl_col_sums == x.sum(0) (fp64 accumulation),
    facl_viewmax_stack == [x ; max over views] with the first view winning ties, facl_contrast_pair_sum's fp32 triple ==
    (float(loss_c), float(loss_circle), float(loss_circle) + float(loss_c)) of facl_contrast_pair on the same inputs."""
    from facl_amd import _lib
    from facl_amd.sa_mlp import _Workspace
    lib = _lib.load_library()
    torch.manual_seed(3)
    x = torch.randn(800, 512, device=DEV)
    out = torch.empty(512, device=DEV)
    _lib.check(lib.facl_col_sums(_lib.ptr(x), 800, 512, _lib.ptr(out), _lib.stream()), "facl_col_sums")
    assert float((out.double() - x.double().sum(0)).abs().max()) < 1e-5
    G, B, C = 6, 5, 16
    xv = torch.randint(0, 3, (G * B, C), device=DEV).float()           # ties on purpose
    h = torch.empty(G * B + B, C, device=DEV)
    arg = torch.empty(B, C, dtype=torch.int32, device=DEV)
    _lib.check(lib.facl_viewmax_stack(_lib.ptr(xv), G, B, C, _lib.ptr(h), _lib.ptr(arg), _lib.stream()), "facl_viewmax_stack")
    ref = xv.view(G, B, C).max(dim=0).values
    first = (xv.view(G, B, C) == ref.unsqueeze(0)).float().argmax(dim=0)
    assert torch.equal(h[:G * B], xv) and torch.equal(h[G * B:], ref) and torch.equal(arg.long(), first)
    # loss triple
    Bk = B
    J = G * Bk
    sim = torch.randn((G + 1) * B, J, device=DEV)
    order = torch.randperm(G, device=DEV)
    ws = _Workspace.get(torch.device(DEV))
    d0, d1 = torch.empty_like(sim), torch.empty_like(sim)
    l0, l1 = torch.empty(2, dtype=torch.float64, device=DEV), torch.empty(2, dtype=torch.float64, device=DEV)
    l32 = torch.empty(3, device=DEV)
    _lib.check(lib.facl_contrast_pair(_lib.ptr(sim), G, B, Bk, J, _lib.ptr(order), 0, _lib.ptr(d0), _lib.ptr(l0), _lib.ptr(ws), _lib.stream()), "pair")
    _lib.check(lib.facl_contrast_pair_sum(_lib.ptr(sim), G, B, Bk, J, _lib.ptr(order), 0, _lib.ptr(d1), _lib.ptr(l1), _lib.ptr(l32),
                                          _lib.ptr(ws), _lib.stream()), "pair_sum")
    assert torch.equal(d0, d1)
    assert float((l0 - l1).abs().max()) < 1e-12 * float(l0.abs().max())
    c, o = l1[0].float(), l1[1].float()
    assert torch.equal(l32, torch.stack((c, o, o + c)))
