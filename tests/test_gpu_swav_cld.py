"""the optional SwAV (sinkhorn) and CLD (k-means) loss terms vs oracle/swav_cld.py and the reference fixture."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["plain", "queue", "inf"])
def test_sinkhorn_vs_reference_golden(tag):
    """facl_sinkhorn vs the output of the REFERENCE's distributed_sinkhorn (cn3d_model_conbag.py:391-406) on the same
    score matrices (tests/golden/swav.npz).  "inf": a matrix whose exp overflowed -- the reference's shoot_infs path
    (max of the finite entries ~1e38, the total overflows, the result is all NaN): reproduced, NaN for NaN."""
    from facl_amd.swav_cld import distributed_sinkhorn
    g = load_golden("swav.npz")
    out = distributed_sinkhorn(torch.from_numpy(g[f"{tag}_in"]).to(DEV), 3).cpu().numpy()
    ref = g[f"{tag}_out"]
    assert out.shape == ref.shape
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    if not np.isnan(ref).all():
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(out.sum(1), 1.0, rtol=1e-5)           # every sample's assignment sums to 1


def test_kmeans_and_cld_vs_reference_golden():
    """facl_kmeans / grouping / cld_loss vs the outputs of the REFERENCE training script's own KMeans / grouping
    (cn3d_train_motion_GL.py:36-70; tests/golden/cld.npz): labels exact incl. EMPTY clusters (count 1, zero centroid) and
    K > N, centroids to fp32 rounding, the CLD loop block's loss and its gradient w.r.t. the embeddings."""
    from facl_amd import swav_cld as P
    g = load_golden("cld.npz")
    x = torch.from_numpy(g["km_x"]).to(DEV)
    for K, it in ((20, 5), (12, 3), (60, 5)):
        cl, c = P.KMeans(x, K, it)
        np.testing.assert_array_equal(cl.cpu().numpy(), g[f"km_K{K}_it{it}_labels"])
        np.testing.assert_allclose(c.detach().cpu().numpy(), g[f"km_K{K}_it{it}_centroids"], rtol=1e-5, atol=1e-6)
        assert int((c.abs().sum(1) == 0).sum()) == int(g[f"km_K{K}_it{it}_nzero"])
    B, G, C = g["cld_meta"].tolist()
    for clusters, iters in ((10, 3), (60, 5)):
        xr = torch.from_numpy(g["cld_x"]).to(DEV).requires_grad_(True)
        l0, _ = P.KMeans(xr.detach()[:3 * B], clusters, iters)
        np.testing.assert_array_equal(l0.cpu().numpy(), g[f"cld_c{clusters}_it{iters}_labels0"])
        loss = P.cld_loss(xr, B, G, T=0.05, clusters=clusters, num_iters=iters)
        ref = float(g[f"cld_c{clusters}_it{iters}_loss"])
        assert abs(loss.item() - ref) < 1e-4 * abs(ref), (loss.item(), ref)
        loss.backward()
        gr = g[f"cld_c{clusters}_it{iters}_grad"].astype(np.float64)
        assert float(np.linalg.norm(xr.grad.cpu().numpy() - gr) / np.linalg.norm(gr)) < 1e-4


def test_kmeans_vs_oracle():
    """facl_kmeans vs oracle KMeans (pinned by cld.npz in tests/test_oracle_golden.py) on a second data set:
    labels exact on separated data, centroids to fp32 rounding, an EMPTY cluster (count 1, zero centroid) included."""
    from facl_amd.swav_cld import KMeans
    from oracle import swav_cld as O
    rng = np.random.RandomState(0)
    centres = rng.randn(12, 64).astype(np.float32) * 3
    x = np.concatenate([centres[i % 12] + 0.1 * rng.randn(64).astype(np.float32) for i in range(96)]).reshape(96, 64)
    x[:20] = x[0] + 0.01 * rng.randn(20, 64).astype(np.float32)           # the first K rows are near-duplicates: empty clusters
    for K, it in ((20, 5), (12, 3), (60, 5)):
        cl, c = KMeans(torch.from_numpy(x).to(DEV), K, it)
        cl_o, c_o = O.KMeans(torch.from_numpy(x), K, it)
        assert torch.equal(cl.cpu(), cl_o), (K, it)
        np.testing.assert_allclose(c.detach().cpu().numpy(), c_o.numpy(), rtol=1e-5, atol=1e-6)


def test_cld_and_swav_losses_vs_oracle_with_gradients():
    from facl_amd import swav_cld as P
    from oracle import swav_cld as O
    torch.manual_seed(4)
    B, G, C, K = 8, 6, 512, 64
    x = torch.nn.functional.normalize(torch.randn(G * B, C), dim=1)
    Wm = torch.randn(K, C) * 0.05
    # ---- CLD
    xa = x.clone().to(DEV).requires_grad_(True)
    xo = x.clone().double().requires_grad_(True)
    la = P.cld_loss(xa, B, G, clusters=10, num_iters=3)
    lo = O.cld_loss(xo, B, G, clusters=10, num_iters=3)
    assert abs(float(la) - float(lo)) < 1e-4 * abs(float(lo))
    la.backward(); lo.backward()
    assert float((xa.grad.cpu().double() - xo.grad).norm() / xo.grad.norm()) < 1e-4
    # ---- SwAV without and with a (pre-filled) queue
    for with_queue in (False, True):
        st = P.SwavState(B, G, C, queue_length=4 * B, epoch_queue_starts=0)
        q_o = None
        if with_queue:
            st.maybe_create(0, DEV)
            st.queue.copy_(torch.nn.functional.normalize(torch.randn(G - 1, 4 * B, C), dim=2))
            st.filled = 4 * B
            q_o = st.queue.cpu().clone()
        xa = x.clone().to(DEV).requires_grad_(True)
        wa = Wm.clone().to(DEV)
        code_a = xa @ wa.t()
        xo = x.clone().requires_grad_(True)
        code_o = xo @ Wm.t()
        la = P.swav_loss(code_a, xa.detach(), wa, st)
        lo, q_o2, _ = O.swav_loss(code_o, xo.detach(), Wm, B, G, queue=q_o, use_the_queue=with_queue)
        assert abs(float(la) - float(lo)) < 2e-4 * abs(float(lo)), (with_queue, float(la), float(lo))
        la.backward(); lo.backward()
        assert float((xa.grad.cpu() - xo.grad).norm() / xo.grad.norm()) < 1e-3
        if with_queue:
            assert torch.allclose(st.queue.cpu(), q_o2)                    # the queue rolled like the reference's


def test_training_step_with_both_terms_runs_and_changes_the_loss():
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from facl_amd.train_common import ContrastiveStep
    from oracle.weights import formula_state_dict
    D, B, G, N = 4, 4, 6, 512
    opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                          sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation",
                          SAMPLE_NUM=N)
    torch.manual_seed(1)
    clip = (torch.rand(B, G, N, D) - 0.5).to(DEV)
    losses = []
    for swa, cld in ((0, 0), (1, 1)):
        net = PointNet_Plus(opt, gost=G)
        net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()})
        net = net.to(DEV).train()
        step = ContrastiveStep(net, torch.optim.Adam(net.parameters(), lr=3e-4), opt, G, swa_if=swa, cld_if=cld)
        loss, _, _ = step(clip, epoch=0, order=np.arange(G))
        assert torch.isfinite(loss)
        assert net.mapping.weight.grad is not None if swa else net.mapping.weight.grad is None
        losses.append(float(loss))
    assert losses[1] != losses[0]
