"""Pins the CPU oracle against golden vectors captured from the reference's own functions
(tools/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import encoder as E
from oracle import fps as OF
from oracle import grouping as OG
from oracle import loss as OL
from oracle import step as OS
from oracle.weights import formula_state_dict, state_dict_shapes

from helpers import canon_groups_np, load_golden, max_rel_rows, rel_err

TOL = 1e-4     # north_star: fp32 features / loss within 1e-4 relative
# fp32 gradients of the early layers are sums over 1e5 positions with heavy cancellation: the
# reference itself moves by ~6e-4 when only the (unspecified) order inside a group changes.
GTOL = 2e-3
PRE_BN_BIAS = {"net3DV_1.0.bias", "net3DV_1.3.bias", "net3DV_1.6.bias", "net3DV_3.0.bias",
               "net3DV_3.3.bias", "net3DV_3.6.bias", "netR_FC.0.bias"}


@pytest.mark.parametrize("N", [512, 2048])
@pytest.mark.parametrize("dt", ["float64", "float32"])
def test_fps_matches_reference(N, dt):
    g = load_golden("fps.npz")
    for c in range(4):
        idx = OF.farthest_point_sampling_fast(g[f"pc_N{N}"][c].astype(dt), 64, int(g[f"start_N{N}"][c]))
        assert idx.shape == (64, 1) and idx.dtype == np.int32
        np.testing.assert_array_equal(idx.ravel(), g[f"idx_N{N}_{dt}"][c])


def test_fps_reorder_matches_reference():
    g = load_golden("fps.npz")
    out = OF.fps_sample_data_2level(g["reorder_in"], 64, 16, g["reorder_s1"], g["reorder_s2"])
    np.testing.assert_array_equal(out, g["reorder_out"])


@pytest.mark.parametrize("tag,r2", [("r016", 0.16), ("r006", 0.06)])
def test_grouping_tiny_bit_exact(tag, r2):
    g = load_golden("tiny.npz")
    idx, xt, yt = OG.group_points(g["points"], 16, 8, r2)
    np.testing.assert_array_equal(canon_groups_np(xt), g[f"xt_{tag}"])
    M = g["points"].shape[0]
    np.testing.assert_array_equal(yt.reshape(M, 1, 16, 3).transpose(0, 3, 2, 1), g[f"yt_{tag}"])
    if tag == "r006":
        assert (idx == np.arange(16)[None, :, None]).sum() > 16 * M      # radius branch exercised


def test_encoder_tiny_matches_reference():
    g = load_golden("tiny.npz")
    idx, xt, yt = OG.group_points(g["points"], 16, 8, 0.06)
    sd = E.clone_state(formula_state_dict(4))
    M = xt.shape[0]
    out = E.encoder_forward(sd, torch.from_numpy(xt).permute(0, 3, 1, 2),
                            torch.from_numpy(yt).view(M, 1, 16, 3).transpose(1, 3), gost=3, training=True)
    for name, t in zip(("x", "code", "x_nor", "x_global"), out):
        # x_global goes through BatchNorm1d over a batch of only B=2 rows here:
        # (a-b)/sqrt((a-b)^2/4+eps) amplifies 1e-7 summation-order noise (the golden's groups are
        # in torch.topk's order, the oracle's in ascending-index order) -> conditioned tolerance.
        tol = 1e-3 if name == "x_global" else TOL
        assert max_rel_rows(t.numpy(), g[name]) < tol, name


LR = 3e-4
PARAM3_TOL_CPU = 0.15     # measured 7e-4..8e-2: Adam divides by sqrt(v), which amplifies fp32 summation-order noise on small gradients


def check_param3(g, params, params0, tol, truth=None):
    """`param3/*` = reference parameters after 3 Adam steps (cn3d_train_motion_GL.py:180,329-333; five tensors kept
    by tools/make_goldens.py).  Adam's normalised update moves every element by <= ~lr per step whatever the gradient's
    size, so the comparison is on the UPDATE (param3 - param0), norm-wise.  net3DV_3.0.bias feeds a train-mode BN: its
    gradient is mathematically zero and the reference's update is a +-lr random walk on cancellation noise -- bounded
    here, not matched (the product leaves such biases untouched)."""
    for key in [k for k in g if k.startswith("param3/")]:
        k = key[len("param3/"):]
        p0 = np.asarray(params0[k], dtype=np.float64).reshape(g[key].shape)
        d_ref = g[key].astype(np.float64) - p0
        d_mine = np.asarray(params[k], dtype=np.float64).reshape(g[key].shape) - p0
        assert np.abs(d_ref).max() <= 6 * LR and np.abs(d_mine).max() <= 6 * LR, k     # 3 steps of O(lr) each
        if k in PRE_BN_BIAS or tol is None:
            continue
        err = np.linalg.norm(d_mine - d_ref) / np.linalg.norm(d_ref)
        floor = 0.0
        if truth is not None:                                   # the golden's own distance to the fp64 three-step update
            d_true = np.asarray(truth[k], dtype=np.float64).reshape(g[key].shape) - p0
            floor = np.linalg.norm(d_ref - d_true) / np.linalg.norm(d_true)
        print(f"param3 {k:22s} |update| {np.linalg.norm(d_ref):.3e}  rel err of the update {err:.2e}  (golden vs fp64 {floor:.2e})")
        assert err < 2 * floor + tol, (k, err, floor)


@pytest.mark.parametrize("tag,D,neg", [("d4", 4, False), ("d3", 3, False), ("d4_neg", 4, True)])
def test_c1_forward_loss_backward_adam(tag, D, neg):
    g = load_golden(f"c1_{tag}.npz")
    B, G, N, S, K, D_ = [int(v) for v in g["meta"]]
    assert D_ == D
    pts = g["points"]
    idx, xt, yt = OG.group_points(pts, S, K, 0.06)
    xt_c = canon_groups_np(xt)
    np.testing.assert_array_equal(xt_c[:8], g["xt_first8"])
    np.testing.assert_allclose(xt_c.sum(axis=2), g["xt_sum"], rtol=0, atol=1e-4)
    M = G * B
    xt_t = torch.from_numpy(xt).permute(0, 3, 1, 2)
    yt_t = torch.from_numpy(yt).view(M, 1, S, 3).transpose(1, 3)
    np.testing.assert_array_equal(yt_t.contiguous().numpy(), g["yt"])

    # Truth = the oracle evaluated in fp64.  The golden is the reference's fp32 run on the machine that made it; the
    # oracle's fp32 run on THIS machine (other core count / oneDNN blocking) rounds differently, and train-mode BN over few
    # rows plus max-pool near-ties amplify that (golden vs fp64: 4e-5 on d4 up to 1.5e-3 on d3's x_global).  So: the fp64
    # oracle must sit within fp32 noise of the golden (the PIN), and the fp32 oracle within twice that noise + TOL.
    def sd_as(dtype):
        return {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone())
                for k, v in formula_state_dict(D, neg_gamma=neg).items()}

    def check(mine32, ref64, gold, name, pin=3e-3):
        floor = max_rel_rows(gold, ref64)
        assert floor < pin, (name, floor)                               # fp64 oracle == reference up to its fp32 noise
        assert max_rel_rows(mine32, gold) < 2 * floor + TOL, name

    xt64, yt64 = xt_t.double(), yt_t.double()
    # eval
    sd = E.clone_state(formula_state_dict(D, neg_gamma=neg))
    with torch.no_grad():
        ev = E.encoder_forward(sd, xt_t, yt_t, G, training=False)
        ev64 = E.encoder_forward(sd_as(torch.float64), xt64, yt64, G, training=False)
    for name, t, t64 in zip(("x", "code", "x_nor", "x_global"), ev, ev64):
        check(t.numpy(), t64.numpy(), g[f"eval_{name}"], "eval_" + name)

    # fp64 truth of the first training step (outputs, taps, losses, gradients)
    sd64 = sd_as(torch.float64)
    pk = E.param_keys(sd64)
    for k in pk:
        sd64[k].requires_grad_(True)
    out64, inter64 = E.encoder_forward(sd64, xt64, yt64, G, training=True, return_intermediates=True)
    lc64, lo64 = OL.global_contrast(G, out64[3], out64[0], B), OL.circle_contrast(G, out64[0], B, g["order"])
    (lc64 + lo64).backward()
    g64 = {k: sd64[k].grad.numpy() for k in pk if sd64[k].grad is not None}

    # 3 training steps
    sd = E.clone_state(formula_state_dict(D, neg_gamma=neg))
    opt = OS.AdamState(sd)
    order = g["order"]
    losses = []
    for it in range(3):
        if it == 0:
            with torch.no_grad():      # stage-wise pins (taps captured by forward hooks in make_goldens)
                sd_probe = E.clone_state(formula_state_dict(D, neg_gamma=neg))
                _, inter = E.encoder_forward(sd_probe, xt_t, yt_t, G, training=True, return_intermediates=True)
            pooled = inter["pooled"].squeeze(-1).permute(0, 2, 1).numpy()[::4]
            pooled64 = inter64["pooled"].detach().squeeze(-1).permute(0, 2, 1).numpy()[::4]
            check(pooled, pooled64, g["train_pooled"], "pooled")
            check(inter["x_pre"].numpy(), inter64["x_pre"].detach().numpy(), g["train_x_pre"], "x_pre")
        r = OS.train_step(sd, opt, None, B, G, S, K, 0.06, order, epoch=0, grouped=(xt_t, yt_t))
        losses.append(r["loss"])
        if it == 0:
            for name, t, t64 in zip(("x", "code", "x_nor", "x_global"), r["outputs"], out64):
                check(t.numpy(), t64.detach().numpy(), g[f"train_{name}"], name)
            for mine_l, key, l64 in ((r["loss_c"], "loss_c", float(lc64)), (r["loss_circle"], "loss_circle", float(lo64))):
                gold = float(g[key])
                assert abs(gold - l64) <= 1e-3 * abs(l64), key
                assert abs(mine_l - gold) <= 2 * abs(gold - l64) + TOL * abs(gold), key
            gmax = max(float(g[k]) for k in g if k.startswith("gradnorm/"))
            for k, gr in r["grads"].items():
                if f"gradnone/{k}" in g:
                    continue
                gn = float(g[f"gradnorm/{k}"])
                mine = float(np.linalg.norm(gr.numpy().astype(np.float64)))
                if k in PRE_BN_BIAS:
                    # a bias feeding a train-mode BN has mathematically ZERO gradient; the reference's
                    # value is pure cancellation noise (norm ~1e-1 at loss ~1e2) -> only bound it.
                    wn = float(g[f"gradnorm/{k[:-4]}weight"])
                    assert mine <= 1e-2 * wn and gn <= 1e-2 * wn, (k, mine, gn, wn)
                    continue
                # atol: 1e-6 of the largest parameter-gradient norm (net3DV_3.7.bias is mathematically
                # ~0 too: a common shift of x_pre[:,c] is removed by netR_FC's BatchNorm1d).
                scale = max(gn, 1e-2 * gmax)
                floor_n = abs(gn - float(np.linalg.norm(g64[k])))        # the golden's own distance to the fp64 truth
                assert abs(mine - gn) <= 2 * floor_n + GTOL * scale + 1e-5, (k, mine, gn)
                if f"grad/{k}" in g:
                    floor_v = np.linalg.norm(g[f"grad/{k}"] - g64[k])
                    assert floor_v <= 6e-2 * scale + 1e-5, (k, floor_v)  # the PIN: fp64 oracle gradient == reference's (fp32 noise: up to 3e-2 on d3)
                    assert np.linalg.norm(gr.numpy() - g[f"grad/{k}"]) <= 2 * floor_v + GTOL * scale + 1e-5, k
            for k in sd:
                if "running_" in k:
                    floor_b = rel_err(g[f"buf1/{k}"], sd64[k].detach().numpy())      # golden (fp32) vs fp64 truth
                    assert floor_b < 1e-3, (k, floor_b)
                    assert rel_err(sd[k].numpy(), g[f"buf1/{k}"]) < 2 * floor_b + 1e-5, k
                if "num_batches" in k:
                    assert int(sd[k]) == int(g[f"buf1/{k}"]), k
    l64 = float(lc64 + lo64)
    assert abs(losses[0] - g["losses3"][0]) <= 2 * abs(g["losses3"][0] - l64) + TOL * abs(l64)
    # later steps amplify rounding differences through Adam's normalised update (another CPU's fp32 summation order moves
    # step 3 by up to 2 %): looser bound
    np.testing.assert_allclose(losses, g["losses3"], rtol=3e-2)
    for k in sd:
        if "running_" in k:
            # the pre-BN biases random-walk by +-lr per step on their pure-noise gradients (see
            # PRE_BN_BIAS) and running_mean follows them: 1e-4 absolute on values of ~5e-2.
            assert rel_err(sd[k].numpy(), g[f"buf3/{k}"]) < 1e-2, k
    # fp64 truth of the three Adam steps (the fp32 runs -- the reference's and this machine's -- scatter around it)
    sd64b = sd_as(torch.float64)
    opt64 = OS.AdamState(sd64b)
    for it in range(3):
        OS.train_step(sd64b, opt64, None, B, G, S, K, 0.06, order, epoch=0, grouped=(xt64, yt64))
    check_param3(g, {k: v.detach().numpy() for k, v in sd.items()}, formula_state_dict(D, neg_gamma=neg), tol=PARAM3_TOL_CPU,
                 truth={k: v.detach().numpy() for k, v in sd64b.items()})


def test_final_fc_golden():
    """linear_classify/fc_model.py:12-25 (Final_FC = F.normalize + Linear) restated with torch-CPU ops against the
    fixture captured from the reference class: forward, CrossEntropy value and gradients, and the seeded construction
    (nn.Linear's init draws, then normal_(0, 0.01)) that facl_amd.linear_classify.Final_FC must reproduce."""
    import torch.nn.functional as F
    from oracle.weights import _hash_uniform
    g = load_golden("fc.npz")
    W = torch.as_tensor((0.02 * _hash_uniform(120 * 22 * 512, 555)).astype(np.float32)).view(120, -1).requires_grad_(True)
    b = torch.as_tensor((0.02 * _hash_uniform(120, 556)).astype(np.float32)).requires_grad_(True)
    x = torch.from_numpy(g["x"])
    pred = F.linear(F.normalize(x, p=2, dim=1), W, b)
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], rtol=1e-5, atol=1e-6)
    loss = F.cross_entropy(pred, torch.from_numpy(g["y"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * float(g["loss"])
    loss.backward()
    np.testing.assert_allclose(b.grad.numpy(), g["grad_bias"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(W.grad.reshape(-1)[:4096].numpy(), g["gradhead_weight"], rtol=1e-4, atol=1e-8)
    assert abs(float(W.grad.double().norm()) - float(g["gradnorm_weight"])) < 1e-5 * float(g["gradnorm_weight"])
    # seeded construction of the product class (CPU: parameters only, no kernel runs)
    from facl_amd.linear_classify import Final_FC
    torch.manual_seed(3)
    w = Final_FC().fc.weight.detach().numpy().astype(np.float64).reshape(-1)
    np.testing.assert_allclose(np.concatenate(([w.size, w.sum(), np.abs(w).sum()], w[:8])), g["init_fingerprint"], rtol=1e-12)


def test_formula_state_dict_has_52_keys():
    for D in (3, 4):
        shapes = state_dict_shapes(D)
        assert len(shapes) == 52
        n = sum(int(np.prod(s)) for k, s in shapes if not ("running" in k or "num_batches" in k))
        assert n == {4: 2358144, 3: 2358144 - 64}[D]     # SURVEY §2 row 2 [probed] param count


def test_level2_groupers_vs_reference_golden():
    """oracle.dense.group_points_2 / group_points_2_3DV vs the outputs of the reference's second-level groupers
    (utils_my.py:332-381) stored by tools/make_goldens.py: make_level2 -- gathered channels exact, xyz centring exact,
    both radius regimes (r^2 = 0.05: most neighbours collapse to the centroid; 0.30: none do)."""
    from oracle import dense as OD
    g = load_golden("level2.npz")
    B, C, S1, S2 = [int(v) for v in g["meta"]]
    pts = g["points"]
    for tag, r2 in (("r005", 0.05), ("r030", 0.30)):
        xt, ct = OD.group_points_2(pts, S2, 64, r2)
        assert xt.shape == (B, 3 + C, S2, 64) and ct.shape == (B, 3, S2, 1)
        np.testing.assert_array_equal(canon_groups_np(xt.transpose(0, 2, 3, 1)), g[f"gp2_{tag}"])
        np.testing.assert_array_equal(ct, g[f"gp2_{tag}_center"])
    xt, ct = OD.group_points_2_3DV(pts, S2)
    np.testing.assert_array_equal(canon_groups_np(xt.transpose(0, 2, 3, 1)), g["gp2_3dv"])
    np.testing.assert_array_equal(ct, g["gp2_3dv_center"])
    # the collapse really happens in the small-radius regime
    zero_off = (np.abs(g["gp2_r005"][..., :3]).sum(-1) == 0).mean()
    assert zero_off > 0.3


def test_sinkhorn_oracle_vs_reference_golden():
    """oracle.swav_cld.distributed_sinkhorn (+ shoot_infs) vs the reference's outputs (cn3d_model_conbag.py:391-425,
    tests/golden/swav.npz); the overflow case reproduces the reference's all-NaN result."""
    from oracle import swav_cld as O
    g = load_golden("swav.npz")
    for tag in ("plain", "queue", "inf"):
        out = O.distributed_sinkhorn(torch.from_numpy(g[f"{tag}_in"]), 3).numpy()
        assert np.array_equal(np.isnan(out), np.isnan(g[f"{tag}_out"])), tag
        if not np.isnan(g[f"{tag}_out"]).all():
            np.testing.assert_allclose(out, g[f"{tag}_out"], rtol=1e-6, atol=0)
    assert int(g["inf_ninf"]) > 0 and np.isnan(g["inf_out"]).all()


@pytest.mark.parametrize("seed", [42, 7])
def test_views_oracle_vs_reference_golden(seed):
    """oracle/views.py vs the REFERENCE's NTU_RGBD_new.get_temporal_augment_data + get_data_train
    (cn3D_data_set.py:654-663, :285-350) run on the same clips under np.random.seed(seed) (tests/golden/views.npz):
    bit-exact float64 views, and the generator stream ends at the same position."""
    from helpers import golden_view_clips
    from oracle import views as OV
    g = load_golden("views.npz")
    rng = np.random.RandomState(seed)
    for tag, clip in golden_view_clips(g):
        got = OV.get_item(rng, *clip)
        ref = g[f"seed{seed}/{tag}"]
        assert got.dtype == ref.dtype == np.float64 and got.shape == (10, 512, 4)
        np.testing.assert_array_equal(got, ref, err_msg=tag)
    assert rng.rand() == float(g[f"seed{seed}/next_rand"])


def test_cld_oracle_vs_reference_golden():
    """oracle/swav_cld.py KMeans / grouping / cld_loss vs the REFERENCE training script's own functions
    (cn3d_train_motion_GL.py:36-70 = utils_my.py:164-197; tests/golden/cld.npz): labels exact (incl. the empty-cluster
    case and K > N), centroids / loss / d loss / d x to fp32 rounding."""
    from oracle import swav_cld as O
    g = load_golden("cld.npz")
    x = torch.from_numpy(g["km_x"])
    for K, it in ((20, 5), (12, 3), (60, 5)):
        cl, c = O.KMeans(x, K, it)
        np.testing.assert_array_equal(cl.numpy(), g[f"km_K{K}_it{it}_labels"])
        np.testing.assert_allclose(c.numpy(), g[f"km_K{K}_it{it}_centroids"], rtol=1e-6, atol=1e-7)
        assert int((c.abs().sum(1) == 0).sum()) == int(g[f"km_K{K}_it{it}_nzero"])
    assert int(g["km_K20_it5_nempty"]) > 0
    B, G, C = g["cld_meta"].tolist()
    for clusters, iters in ((10, 3), (60, 5)):
        xr = torch.from_numpy(g["cld_x"]).clone().requires_grad_(True)
        loss = O.cld_loss(xr, B, G, T=0.05, clusters=clusters, num_iters=iters)
        ref = float(g[f"cld_c{clusters}_it{iters}_loss"])
        assert abs(loss.item() - ref) < 1e-5 * abs(ref)
        loss.backward()
        assert rel_err(xr.grad.numpy(), g[f"cld_c{clusters}_it{iters}_grad"]) < 1e-5
        l0, _ = O.KMeans(torch.from_numpy(g["cld_x"])[:3 * B], clusters, iters)
        np.testing.assert_array_equal(l0.numpy(), g[f"cld_c{clusters}_it{iters}_labels0"])
    assert abs(float(g["cld_fn_loss"]) - float(g["cld_c60_it5_loss"])) < 1e-6 * abs(float(g["cld_fn_loss"]))
