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
