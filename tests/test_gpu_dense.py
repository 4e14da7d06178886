"""Dense configuration (BASELINE configs[4]): second-level groupers vs the reference goldens, the feature gather /
scatter kernels, and the 3-level encoder (forward, losses, gradients, running statistics) vs oracle/dense.py in fp64."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import canon_groups_np, load_golden, max_rel_rows, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_group_points_2_dropins_vs_reference_golden():
    """facl_amd.dense.group_points_2 / group_points_2_3DV (facl_group on the coordinates + facl_gather_rows) vs the outputs
    of the REFERENCE functions (utils_my.py:332-381, tests/golden/level2.npz): every gathered channel and the centred
    xyz bit-exact (K axis canonicalised: torch.topk(sorted=False) leaves the order unspecified), shapes as the reference's."""
    from facl_amd import dense
    g = load_golden("level2.npz")
    B, C, S1, S2 = [int(v) for v in g["meta"]]
    pts = torch.from_numpy(g["points"]).to(DEV)
    before = pts.clone()
    for tag, r2 in (("r005", 0.05), ("r030", 0.30)):
        xt, ct = dense.group_points_2(pts, S1, S2, 64, torch.tensor(r2))
        assert xt.shape == (B, 3 + C, S2, 64) and ct.shape == (B, 3, S2, 1)
        np.testing.assert_array_equal(canon_groups_np(xt.permute(0, 2, 3, 1).cpu().numpy()), g[f"gp2_{tag}"])
        np.testing.assert_array_equal(ct.cpu().numpy(), g[f"gp2_{tag}_center"])
    xt, ct = dense.group_points_2_3DV(pts, S1, S2, 0, None)
    np.testing.assert_array_equal(canon_groups_np(xt.permute(0, 2, 3, 1).cpu().numpy()), g["gp2_3dv"])
    np.testing.assert_array_equal(ct.cpu().numpy(), g["gp2_3dv_center"])
    assert torch.equal(pts, before)                                   # the input is not modified


@pytest.mark.parametrize("M,S1,C,S2,K", [(3, 128, 256, 32, 64), (2, 512, 256, 128, 64), (1, 40, 7, 5, 3), (2, 96, 320, 7, 9)])
def test_gather_scatter_rows_vs_torch(M, S1, C, S2, K):
    """facl_gather_rows == torch.gather on the rows; facl_scatter_rows == its autograd transpose (index_add), exactly
    the same sums up to fp32 addition order (the kernel adds a cloud's rows in index order)."""
    from facl_amd.dense import _GatherRows
    torch.manual_seed(M * S1 + C)
    feat = torch.randn(M * S1, C, device=DEV, requires_grad=True)
    idx = torch.randint(0, S1, (M, S2, K), device=DEV, dtype=torch.int32)
    idx[:, :, 0] = 0                                                  # heavy duplicates: every group hits row 0
    rows = _GatherRows.apply(feat, idx, S1)
    ref = feat.detach().view(M, S1, C).gather(1, idx.long().view(M, S2 * K, 1).expand(M, S2 * K, C)).reshape(-1, C)
    assert torch.equal(rows.detach(), ref)
    w = torch.randn_like(rows)
    (rows * w).sum().backward()
    exp = torch.zeros(M, S1, C, device=DEV, dtype=torch.float64)
    exp.index_put_((torch.arange(M, device=DEV).view(M, 1).expand(M, S2 * K).reshape(-1), idx.long().view(-1)),
                   w.double(), accumulate=True)
    assert rel_err(feat.grad.cpu().numpy(), exp.view(M * S1, C).cpu().numpy()) < 1e-6


def _dense_model(D, G, cfg, precision):
    from facl_amd.dense import PointNet_Plus_dense
    from oracle.dense import dense_formula_state_dict
    net = PointNet_Plus_dense(SimpleNamespace(INPUT_FEATURE_NUM=D), gost=G, precision=precision, **cfg)
    sd = dense_formula_state_dict(D)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return net.to(DEV).train(), sd


def _oracle(sd_np, pts, G, cfg, order, dtype=torch.float64):
    from oracle import dense as OD, loss as OL
    sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone()) for k, v in sd_np.items()}
    keys = [k for k in sd if "running" not in k and "num_b" not in k]
    for k in keys:
        sd[k].requires_grad_(True)
    x, code, xn, xg = OD.dense_encoder_forward(sd, pts.to(dtype), G, cfg["S1"], cfg["K1"], cfg["S2"], cfg["K2"], cfg["r1"], cfg["r2"])
    B = x.shape[0] // G
    loss = OL.global_contrast(G, xg, x, B) + OL.circle_contrast(G, x, B, order)
    loss.backward()
    return x.detach(), xg.detach(), float(loss), {k: sd[k].grad for k in keys if sd[k].grad is not None}, sd


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_dense_step_small_vs_oracle_fp64(precision):
    """3-level encoder + losses + backward at a small size vs oracle/dense.py evaluated in fp64.  "f32": the 1e-4 bar of
    the headline path.  "f16": level-2/3 GEMM inputs rounded to fp16 (11-bit significands, fp32 accumulate) through six
    GEMM + train-mode-BN layers: the tolerance is WIDENED to 3e-2 on the worst feature row, 1e-2 on the loss and 3e-1 on
    the gradients (measured 9e-3 / 1e-3 / 0.13 on the first layer's weight, whose gradient crosses every rounded GEMM and
    two max-pools; values are printed)."""
    from facl_amd.dense import DenseStep
    D, B, G, N = 4, 4, 3, 512
    cfg = dict(S1=128, K1=64, S2=64, K2=64, r1=0.16, r2=0.30)
    torch.manual_seed(7)
    clip = torch.rand(B, G, N, D) - 0.5
    net, sd_np = _dense_model(D, G, cfg, precision)
    optim = torch.optim.SGD(net.parameters(), lr=0.0)
    order = np.array([2, 0, 1])
    taps = {}
    hk = net.register_forward_hook(lambda m, i, o: taps.update(x=o[0].detach().clone(), xg=o[3].detach().clone()))
    loss, _, _ = DenseStep(net, optim, G)(clip.to(DEV), order=order)
    hk.remove()
    pts = clip.permute(1, 0, 2, 3).reshape(-1, N, D)
    x64, xg64, l64, g64, sd64 = _oracle(sd_np, pts, G, cfg, order)
    tol, ltol, gtol = (1e-4, 1e-4, 5e-3) if precision == "f32" else (3e-2, 1e-2, 3e-1)
    e_x, e_xg = max_rel_rows(taps["x"].cpu().numpy(), x64.numpy()), max_rel_rows(taps["xg"].cpu().numpy(), xg64.numpy())
    e_l = abs(loss.item() - l64) / abs(l64)
    print(f"[{precision}] x {e_x:.2e}  x_global {e_xg:.2e}  loss {loss.item():.6f} vs {l64:.6f} ({e_l:.2e})")
    assert e_x < tol and e_xg < tol and e_l < ltol
    gmax = max(float(v.norm()) for v in g64.values())
    worst = 0.0
    for k, p in net.named_parameters():
        if k not in g64 or p.grad is None:
            continue
        ref = g64[k].numpy()
        if np.linalg.norm(ref) < 1e-3 * gmax:                         # mathematically ~0 gradients (pre-BN biases etc.)
            continue
        e = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-2 * gmax)
        worst = max(worst, e)
        assert e < gtol, (k, e)
    print(f"[{precision}] worst parameter-gradient error {worst:.2e}")
    st = net.state_dict()
    for k in st:
        if "running" in k:
            assert rel_err(st[k].cpu().numpy(), sd64[k].detach().numpy()) < (1e-5 if precision == "f32" else 2e-3), k


_CONFIG_SIZE_ORACLE = {}


def _config_size_case():
    """Inputs + the fp64 oracle forward of the config-size case, evaluated once for both precisions."""
    if not _CONFIG_SIZE_ORACLE:
        D, B, G, N = 3, 2, 32, 4096
        cfg = dict(S1=512, K1=64, S2=128, K2=64, r1=0.16, r2=0.25)
        torch.manual_seed(3)
        clip = torch.rand(B, G, N, D) - 0.5
        order = np.random.RandomState(2).permutation(G)
        from oracle import dense as OD
        sd_np = OD.dense_formula_state_dict(D)
        pts = clip.permute(1, 0, 2, 3).reshape(-1, N, D)
        with torch.no_grad():
            from oracle import loss as OL
            sd = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone()) for k, v in sd_np.items()}
            x, _, _, xg = OD.dense_encoder_forward(sd, pts.double(), G, cfg["S1"], cfg["K1"], cfg["S2"], cfg["K2"], cfg["r1"], cfg["r2"])
            l64 = float(OL.global_contrast(G, xg, x, B) + OL.circle_contrast(G, x, B, order))
        _CONFIG_SIZE_ORACLE.update(D=D, B=B, G=G, N=N, cfg=cfg, clip=clip, order=order, x64=x, xg64=xg, l64=l64)
    return _CONFIG_SIZE_ORACLE


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_dense_forward_at_config_size_two_clips(precision):
    """N = 4096 points, T = 32 views (the dense configuration's cloud and view counts) for TWO clips (train-mode
    BatchNorm1d needs more than one clip row, in the reference's construction too), default level sizes (S1 = 512,
    S2 = 128, K = 64): features and loss vs the oracle in fp64.  "f32": fp32-grade arithmetic, the 1e-4 bar.  "f16": the
    configuration's own fp16-input MFMA point-MLP (BASELINE configs[4]): operands of the level-1 64->256 layer and of every
    level-2/3 GEMM rounded to fp16 (2^-11), fp32 accumulation, six such layers with train-mode BatchNorm between them --
    tolerance WIDENED to 3e-2 on the worst feature row and 1e-2 on the loss (the measured values are printed)."""
    from facl_amd.dense import DenseStep
    c = _config_size_case()
    D, B, G, N, cfg, clip, order = c["D"], c["B"], c["G"], c["N"], c["cfg"], c["clip"], c["order"]
    net, sd_np = _dense_model(D, G, cfg, precision)
    taps = {}
    hk = net.register_forward_hook(lambda m, i, o: taps.update(x=o[0].detach().clone(), xg=o[3].detach().clone()))
    loss, _, _ = DenseStep(net, torch.optim.SGD(net.parameters(), lr=0.0), G)(clip.to(DEV), order=order)
    hk.remove()
    x64, xg64, l64 = c["x64"], c["xg64"], c["l64"]
    e_x = max_rel_rows(taps["x"].cpu().numpy(), x64.numpy())
    e_xg = max_rel_rows(taps["xg"].cpu().numpy(), xg64.numpy())
    e_l = abs(loss.item() - l64) / abs(l64)
    print(f"[{precision}] x {e_x:.2e}  x_global {e_xg:.2e}  loss {loss.item():.6f} vs {l64:.6f} ({e_l:.2e})")
    if precision == "f32":
        assert e_x < 1e-4 and e_l < 1e-4
        # x_global of two clips goes through a BatchNorm1d over TWO rows: (a - b)/sqrt((a-b)^2/4 + eps) amplifies fp32
        # rounding (same conditioning note as tests/test_oracle_golden.py for the tiny golden): 1e-3 there
        assert e_xg < 1e-3
    else:
        assert e_x < 3e-2 and e_l < 1e-2
        assert e_xg < 3e-1                                            # the two-row BatchNorm1d amplifies the fp16 rounding likewise
