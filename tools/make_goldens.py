#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own functions on CPU.

Runs only in the build container (needs /root/reference); the GPU box and the tests only read
the committed .npz files.  Nothing from the reference is copied: its modules are imported,
called on seeded synthetic inputs with formula weights (oracle/weights.py), and the outputs
are stored.  Harness-side monkeypatches: ``.cuda()`` -> identity (utils_my.py:58,64-65,...
hard-code it), ``np.random.shuffle`` / ``np.random.randint`` -> injected values, so that the
random choices of utils_my.py:97 and cn3d_data_load.py:305 are recorded in the fixture.

    python tools/make_goldens.py            # rewrites tests/golden/
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/training_code"
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self            # noqa: E731  (harness-side only)
torch.nn.Module.cuda = lambda self, *a, **k: self         # noqa: E731

import utils_my as R_utils                                  # noqa: E402  reference
import cn3d_model_conbag as R_model                         # noqa: E402  reference
import cn3d_data_load as R_load                             # noqa: E402  reference

# cn3D_data_set.py:6 imports imageio and cn3d_train_motion_GL.py:24 imports torchvision.transforms; neither module is
# installed here and neither name is used by any function of those files.  Harness-side EMPTY stand-in modules (same
# class of patch as the .cuda() identity above) let the two reference modules import, so that their own functions
# produce the views / CLD fixtures below.
import types                                                # noqa: E402
sys.modules.setdefault("imageio", types.ModuleType("imageio"))
if "torchvision" not in sys.modules:
    _tv = types.ModuleType("torchvision")
    _tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = _tv
    sys.modules["torchvision.transforms"] = _tv.transforms
_cvd = os.environ.get("CUDA_VISIBLE_DEVICES")
import cn3D_data_set as R_data                              # noqa: E402  reference
import cn3d_train_motion_GL as R_train                      # noqa: E402  reference (sets CUDA_VISIBLE_DEVICES at import)
if _cvd is None:
    os.environ.pop("CUDA_VISIBLE_DEVICES", None)
else:
    os.environ["CUDA_VISIBLE_DEVICES"] = _cvd

from oracle.weights import formula_state_dict               # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def canon_groups(xt_MDSK):
    """(M,D,S,K) reference tensor -> (M,S,K,D) numpy with the K axis sorted lexicographically
    (torch.topk(sorted=False) leaves the order inside a group unspecified)."""
    a = xt_MDSK.permute(0, 2, 3, 1).contiguous().numpy()
    M, S, K, D = a.shape
    flat = a.reshape(M * S, K, D)
    out = np.empty_like(flat)
    for i in range(flat.shape[0]):
        keys = tuple(flat[i, :, d] for d in reversed(range(D)))
        out[i] = flat[i][np.lexsort(keys)]
    return out.reshape(M, S, K, D)


def ref_opt(B, N, S, K, D):
    return SimpleNamespace(temperal_num=3, knn_K=K, ball_radius=0.16, ball_radius2=0.25,
                           sample_num_level1=S, sample_num_level2=S, INPUT_FEATURE_NUM=D,
                           Num_Class=512, batchSize=B, pooling="concatenation", SAMPLE_NUM=N)


def synth_points(B, G, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(B, G, N, D, generator=g) - 0.5
    return pts.permute(1, 0, 2, 3).reshape(-1, N, D).contiguous()     # cn3d_train_motion_GL.py:226


def make_fps():
    out = {}
    rng = np.random.RandomState(7)
    orig = np.random.randint
    for N in (512, 2048):
        pcs = (rng.rand(4, N, 3) - 0.5)
        starts = rng.randint(0, N, size=4)
        for dt in (np.float64, np.float32):
            idxs = []
            for c in range(4):
                np.random.randint = lambda lo, hi=None, size=None, _s=int(starts[c]): _s
                try:
                    idxs.append(R_load.farthest_point_sampling_fast(pcs[c].astype(dt), 64).ravel())
                finally:
                    np.random.randint = orig
            out[f"idx_N{N}_{np.dtype(dt).name}"] = np.stack(idxs).astype(np.int32)
        out[f"pc_N{N}"] = pcs.astype(np.float64)
        out[f"start_N{N}"] = starts.astype(np.int32)
    # reorder (fps_sample_data, cn3d_data_load.py:287-298 is 2-level; data_set.py:665-672 1-level is
    # restated by the oracle and checked through the 2-level golden with NUM_POINT patched)
    N = 512
    pts = (rng.rand(2, N, 4) - 0.5)
    s1 = rng.randint(0, N, size=2)
    s2 = rng.randint(0, 64, size=2)
    R_load.NUM_POINT = N
    seq = []
    for c in range(2):
        seq += [int(s1[c]), int(s2[c])]
    it = iter(seq)
    np.random.randint = lambda lo, hi=None, size=None: next(it)
    try:
        re = R_load.fps_sample_data(pts.copy(), 64, 16)
    finally:
        np.random.randint = orig
    out["reorder_in"] = pts
    out["reorder_s1"] = s1.astype(np.int32)
    out["reorder_s2"] = s2.astype(np.int32)
    out["reorder_out"] = re
    np.savez_compressed(os.path.join(OUT, "fps.npz"), **out)
    print("fps.npz", {k: v.shape for k, v in out.items()})


def make_tiny():
    B, G, N, S, K, D = 2, 3, 128, 16, 8, 4
    pts = synth_points(B, G, N, D, seed=11)
    out = {"points": pts.numpy()}
    xt, yt = R_utils.group_points_3DV_2048(pts.clone(), K, S, SAMPLE_NUM=N)       # r2 = 0.16
    out["xt_r016"] = canon_groups(xt)
    out["yt_r016"] = yt.contiguous().numpy()
    opt = ref_opt(B, N, S, K, D)
    xt, yt = R_utils.group_points_3DV_nums(pts.clone(), opt, S, K)               # r2 = 0.06
    out["xt_r006"] = canon_groups(xt)
    out["yt_r006"] = yt.contiguous().numpy()
    # encoder on the tiny grouping (train mode)
    sd = formula_state_dict(D)
    net = R_model.PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    net.train()
    x, code, x_nor, x_global = net(xt, yt, 1)
    out.update(x=x.detach().numpy(), code=code.detach().numpy(), x_nor=x_nor.detach().numpy(),
               x_global=x_global.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "tiny.npz"), **out)
    print("tiny.npz ok")


def run_c1(D, neg_gamma, tag):
    B, G, N, S, K = 4, 8, 512, 64, 64
    pts = synth_points(B, G, N, D, seed=100 + D)
    opt = ref_opt(B, N, S, K, D)
    out = {"points": pts.numpy(), "meta": np.array([B, G, N, S, K, D], dtype=np.int32)}
    xt, yt = R_utils.group_points_3DV(pts.clone(), opt)                          # r2 = 0.06
    assert opt.knn_K == 64 and abs(opt.ball_radius - 0.06) < 1e-12
    xt_c = canon_groups(xt)
    out["xt_sum"] = xt_c.sum(axis=2)                                             # (M,S,D) compact check
    out["xt_first8"] = xt_c[:8]
    out["yt"] = yt.contiguous().numpy()
    sd = formula_state_dict(D, neg_gamma=neg_gamma)
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}

    crit = torch.nn.CrossEntropyLoss()
    order = np.array([3, 0, 6, 1, 7, 5, 2, 4])
    out["order"] = order
    orig_shuffle = np.random.shuffle

    def patched(a):
        a[:] = order

    # --- eval-mode forward
    net = R_model.PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
    net.load_state_dict(tsd)
    net.eval()
    with torch.no_grad():
        ev = net(xt, yt, 0)
    for name, t in zip(("x", "code", "x_nor", "x_global"), ev):
        out[f"eval_{name}"] = t.numpy()

    # --- train-mode forward + losses + backward + 3 Adam steps (cn3d_train_motion_GL.py:180-181,329-333)
    net = R_model.PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
    net.load_state_dict(tsd)
    net.train()
    taps = {}
    h1 = net.net3DV_1.register_forward_hook(lambda m, i, o: taps.__setitem__("pooled", o.detach().clone()))
    h3 = net.net3DV_3.register_forward_hook(lambda m, i, o: taps.__setitem__("local", o.detach().clone()))
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)
    sched = torch.optim.lr_scheduler.StepLR(optim, step_size=4, gamma=0.7)
    losses = []
    for it in range(3):
        x, code, x_nor, x_global = net(xt, yt, 1)
        loss_c = R_utils.global_contrast(G, x_global, x, opt, crit)
        np.random.shuffle = patched
        try:
            loss_circle = R_utils.circle_contrast(G, x, B, crit)
        finally:
            np.random.shuffle = orig_shuffle
        loss = loss_circle + loss_c
        optim.zero_grad()
        loss.backward()
        if it == 0:
            h1.remove(); h3.remove()
            out["train_pooled"] = taps["pooled"].squeeze(-1).permute(0, 2, 1).contiguous().numpy()[::4]   # (M/4,S,256)
            out["train_x_pre"] = taps["local"].squeeze(-1).max(dim=2).values.numpy()                       # (M,1024)
            for name, t in zip(("x", "code", "x_nor", "x_global"), (x, code, x_nor, x_global)):
                out[f"train_{name}"] = t.detach().numpy()
            out["loss_c"] = np.float64(loss_c.item())
            out["loss_circle"] = np.float64(loss_circle.item())
            for k, p in net.named_parameters():
                if p.grad is None:          # mapping.weight: `code` does not feed the live loss
                    out[f"gradnone/{k}"] = np.int32(1)
                    continue
                g = p.grad.detach().numpy()
                out[f"gradnorm/{k}"] = np.float64(np.linalg.norm(g.astype(np.float64)))
                if g.size <= 70000:
                    out[f"grad/{k}"] = g.copy()
                else:
                    out[f"gradhead/{k}"] = g.reshape(-1)[:4096].copy()
            for k, b in net.named_buffers():
                out[f"buf1/{k}"] = b.detach().numpy().copy()
        optim.step()
        sched.step(0)
        losses.append(loss.item())
    out["losses3"] = np.array(losses, dtype=np.float64)
    for k, b in net.named_buffers():
        out[f"buf3/{k}"] = b.detach().numpy().copy()
    for k in ("net3DV_1.0.weight", "net3DV_1.6.weight", "net3DV_3.0.bias", "netR_FC.3.bias", "net3DV_1.4.weight"):
        out[f"param3/{k}"] = dict(net.named_parameters())[k].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"c1_{tag}.npz"), **out)
    print(f"c1_{tag}.npz losses", losses)


def make_init():
    """Default initialisation of the reference model under torch.manual_seed(1) (cn3d_train_motion_GL.py:142-144,173):
    per-key fingerprints, so the test can check that facl_amd's constructor draws the same weights."""
    out = {}
    for D in (3, 4):
        opt = ref_opt(4, 512, 64, 64, D)
        torch.manual_seed(1)
        net = R_model.PointNet_Plus(opt)
        for k, v in net.state_dict().items():
            a = v.detach().numpy().astype(np.float64).reshape(-1)
            out[f"D{D}/{k}"] = np.concatenate(([a.size, a.sum(), np.abs(a).sum()], a[:8]))
    np.savez_compressed(os.path.join(OUT, "init.npz"), **out)
    print("init.npz", len(out))


def make_level2():
    """Second-level groupers on channel-first features (utils_my.py:332-356 group_points_2: K = 64 literal, radius =
    the tensor argument; :358-381 group_points_2_3DV: K = 32 and r^2 = 0.11 literals).  No model of the reference
    consumes them; they are the only reference text the multi-level (dense, BASELINE configs[4]) set abstraction can
    follow, so the level-2 grouping kernel and its oracle restatement are pinned here."""
    out = {}
    g = torch.Generator().manual_seed(21)
    B, C, S1, S2 = 3, 5, 128, 32
    pts = torch.rand(B, 3 + C, S1, generator=g) - 0.5
    out["points"] = pts.numpy().copy()
    out["meta"] = np.array([B, C, S1, S2], dtype=np.int32)
    for tag, r in (("r005", 0.05), ("r030", 0.30)):
        xt, ct = R_utils.group_points_2(pts.clone(), S1, S2, 64, torch.tensor(r))
        out[f"gp2_{tag}"] = canon_groups(xt)                                   # (B,S2,64,3+C), K axis sorted
        out[f"gp2_{tag}_center"] = ct.contiguous().numpy()
    xt, ct = R_utils.group_points_2_3DV(pts.clone(), S1, S2, 0, None)
    out["gp2_3dv"] = canon_groups(xt)                                          # (B,S2,32,3+C)
    out["gp2_3dv_center"] = ct.contiguous().numpy()
    # the input must not be modified by the in-place centring (the gather makes a copy)
    assert np.array_equal(pts.numpy(), out["points"])
    np.savez_compressed(os.path.join(OUT, "level2.npz"), **out)
    print("level2.npz", {k: v.shape for k, v in out.items()})


def make_swav():
    """cn3d_model_conbag.py:391-425 distributed_sinkhorn / shoot_infs on SwAV-style score matrices exp(code / 0.03)^T:
    one ordinary, one whose exp overflows to +inf in a few entries (the shoot_infs branch), one with a queue-sized batch."""
    out = {}
    g = torch.Generator().manual_seed(31)
    for tag, (K, n, scale) in (("plain", (64, 32, 0.5)), ("inf", (64, 48, 4.0)), ("queue", (64, 32 * 9, 0.5))):
        code = torch.randn(n, K, generator=g) * scale
        po = torch.exp(code / 0.03).t().contiguous()
        out[f"{tag}_in"] = po.numpy().copy()
        out[f"{tag}_ninf"] = np.int32(torch.isinf(po).sum().item())
        out[f"{tag}_out"] = R_model.distributed_sinkhorn(po.clone(), 3).numpy()
    assert out["inf_ninf"] > 0 and out["plain_ninf"] == 0 and out["queue_ninf"] == 0
    np.savez_compressed(os.path.join(OUT, "swav.npz"), **out)
    print("swav.npz", {k: np.shape(v) for k, v in out.items()})


def make_fc():
    """linear_classify/fc_model.py:12-25 Final_FC: seeded construction (fingerprints of the N(0, 0.01) weights),
    forward on synthetic features, CrossEntropy loss and its gradients."""
    sys.path.insert(0, "/root/reference/linear_classify")
    import fc_model as R_fc                                                     # reference
    out = {}
    torch.manual_seed(3)
    net = R_fc.Final_FC()
    w = net.fc.weight.detach().numpy().astype(np.float64).reshape(-1)
    out["init_fingerprint"] = np.concatenate(([w.size, w.sum(), np.abs(w).sum()], w[:8]))
    g = torch.Generator().manual_seed(4)
    x = torch.randn(6, 22 * 512, generator=g) * 3.0 + 0.5
    y = torch.tensor([5, 0, 119, 7, 7, 64])
    wf = torch.as_tensor(_wave_fc(120 * 22 * 512)).view(120, 22 * 512)
    bf = torch.as_tensor(_wave_fc(120, key=1))
    with torch.no_grad():
        net.fc.weight.copy_(wf)
        net.fc.bias.copy_(bf)
    pred = net(x)
    loss = torch.nn.CrossEntropyLoss()(pred, y)
    loss.backward()
    out.update(x=x.numpy(), y=y.numpy(), pred=pred.detach().numpy(), loss=np.float64(loss.item()),
               grad_bias=net.fc.bias.grad.numpy().copy(), gradnorm_weight=np.float64(net.fc.weight.grad.double().norm().item()),
               gradhead_weight=net.fc.weight.grad.reshape(-1)[:4096].numpy().copy())
    np.savez_compressed(os.path.join(OUT, "fc.npz"), **out)
    print("fc.npz", {k: np.shape(v) for k, v in out.items()})


def synth_clip(seed, dt, P=900, Kp=300, R1=500, R2=200):
    """The four arrays `__getitem__` loads for one video (cn3D_data_set.py:105-116): (rows, 8) clouds; the temporal
    channels 4 and 7 carry exact zeros so that get_temporal_augment_data's non-zero filter bites."""
    r = np.random.RandomState(seed)
    pts = (r.rand(P, 8) - 0.5).astype(dt)
    pts[::3, 4] = 0
    pts[1::4, 7] = 0
    return pts, (r.rand(Kp, 8) - 0.5).astype(dt), (r.rand(R1, 8) - 0.5).astype(dt), (r.rand(R2, 8) - 0.5).astype(dt)


def make_views():
    """cn3D_data_set.py `__getitem__` body :117-120 = get_temporal_augment_data (:654-663) x2 + get_data_train (:285-350,
    with jitter_point_cloud :767-778, reverse_transform :708-713, rotate_trans :734-749) of the REFERENCE class on
    synthetic clouds under np.random.seed.  The instance is made with __new__ (the constructor lists the NTU directory);
    the methods called use no instance state.  Stored: the clip GENERATOR arguments (the inputs are rebuilt from them by
    tests/helpers.synth_clip, the same formula), the seed, and the (10,512,4) float64 outputs."""
    ds = R_data.NTU_RGBD_new.__new__(R_data.NTU_RGBD_new)
    out = {}
    cases = [("a", 1, np.float32, dict()), ("b", 2, np.float32, dict(P=777, Kp=513, R1=400, R2=64)),
             ("c", 3, np.float64, dict(P=2048, Kp=1024)), ("d", 4, np.float64, dict())]
    for seed_np in (42, 7):
        np.random.seed(seed_np)
        for tag, cseed, dt, kw in cases:                    # consecutive __getitem__ calls on ONE generator stream
            points, key_points, res1, res2 = synth_clip(cseed, dt, **kw)
            keep = [a.copy() for a in (points, key_points, res1, res2)]
            time_seg2 = ds.get_temporal_augment_data(points, 4)
            time_seg4 = ds.get_temporal_augment_data(points, 7)
            o = ds.get_data_train(points[:, :4], key_points[:, :4], time_seg2[:, :4], time_seg4[:, :4], res1[:, :4],
                                  res2[:, :4], num_crop=10)
            assert o.shape == (10, 512, 4) and o.dtype == np.float64
            assert all(np.array_equal(a, b) for a, b in zip(keep, (points, key_points, res1, res2)))   # inputs untouched
            out[f"seed{seed_np}/{tag}"] = o
        out[f"seed{seed_np}/next_rand"] = np.float64(np.random.rand())      # the stream position afterwards
    out["cases"] = np.array([[c[1], 0 if c[2] is np.float32 else 1, c[3].get("P", 900), c[3].get("Kp", 300),
                              c[3].get("R1", 500), c[3].get("R2", 200)] for c in cases], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "views.npz"), **out)
    print("views.npz", {k: np.shape(v) for k, v in out.items()})


def make_cld():
    """cn3d_train_motion_GL.py:36-70 `grouping` / `KMeans` of the REFERENCE training script (textually identical twins
    live in utils_my.py:164-197: asserted), and the CLD loop block :319-326 = utils_my.CLD_Loss :152-161.  Cases: separated
    clusters, near-duplicate leading rows (EMPTY clusters: count forced to 1, zero centroid), L2-normalised embeddings at
    the loop's own settings (clusters=60, 5 iterations, T=0.05) with the gradient of the loss w.r.t. the embeddings."""
    import inspect
    assert inspect.getsource(R_train.KMeans) == inspect.getsource(R_utils.KMeans)
    assert inspect.getsource(R_train.grouping) == inspect.getsource(R_utils.grouping)
    out = {}
    rng = np.random.RandomState(0)
    centres = rng.randn(12, 64).astype(np.float32) * 3
    x = np.concatenate([centres[i % 12] + 0.1 * rng.randn(64).astype(np.float32) for i in range(96)]).reshape(96, 64)
    x[:20] = x[0] + 0.01 * rng.randn(20, 64).astype(np.float32)
    x[1:4] = x[0]                          # exact duplicates among the initial centroids: argmin never picks 1..3 -> EMPTY clusters
    out["km_x"] = x
    for K, it in ((20, 5), (12, 3), (60, 5)):
        cl, c = R_train.KMeans(torch.from_numpy(x), K, it)
        out[f"km_K{K}_it{it}_labels"] = cl.numpy().astype(np.int64)
        out[f"km_K{K}_it{it}_centroids"] = c.numpy()
        out[f"km_K{K}_it{it}_nempty"] = np.int32(K - len(np.unique(cl.numpy())))
        out[f"km_K{K}_it{it}_nzero"] = np.int32((c.abs().sum(1) == 0).sum().item())
    assert out["km_K20_it5_nempty"] > 0 and out["km_K20_it5_nzero"] > 0
    # the loop block on normalised embeddings (x_nor rows are view-major g*B+b)
    g = torch.Generator().manual_seed(4)
    B, G, C = 8, 6, 512
    xn = torch.nn.functional.normalize(torch.randn(G * B, C, generator=g), dim=1)
    out["cld_x"] = xn.numpy().copy()
    out["cld_meta"] = np.array([B, G, C], dtype=np.int32)
    for clusters, iters in ((10, 3), (60, 5)):
        xr = xn.clone().requires_grad_(True)
        tot = 0
        for i in range(G - 4):                                                  # :322-325 with clusters / iters exposed
            tot = tot + R_train.grouping(xr[i * B:(i + 3) * B], xr[(i + 1) * B:(i + 4) * B], 0.05, 10, clusters, iters)
        tot.backward()
        out[f"cld_c{clusters}_it{iters}_loss"] = np.float64(tot.item())
        out[f"cld_c{clusters}_it{iters}_grad"] = xr.grad.numpy().copy()
        l1, c1 = R_train.KMeans(xn[:3 * B], clusters, iters)
        out[f"cld_c{clusters}_it{iters}_labels0"] = l1.numpy().astype(np.int64)
    # utils_my.CLD_Loss (the loop block as a function: literals 0.05, 10, 60, 5)
    xr = xn.clone().requires_grad_(True)
    loss = R_utils.CLD_Loss(0, G, xr, SimpleNamespace(batchSize=B))
    out["cld_fn_loss"] = np.float64(loss.item())
    assert abs(out["cld_fn_loss"] - out["cld_c60_it5_loss"]) < 1e-6 * abs(out["cld_fn_loss"])
    np.savez_compressed(os.path.join(OUT, "cld.npz"), **out)
    print("cld.npz", {k: np.shape(v) for k, v in out.items()})


def _wave_fc(n, key=0):
    from oracle.weights import _hash_uniform
    return (0.02 * _hash_uniform(n, 555 + key)).astype(np.float32)


def main():
    argparse.ArgumentParser(description=__doc__).parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    make_init()
    make_level2()
    make_swav()
    make_fc()
    make_views()
    make_cld()
    make_fps()
    make_tiny()
    run_c1(4, False, "d4")
    run_c1(3, False, "d3")
    run_c1(4, True, "d4_neg")


if __name__ == "__main__":
    main()
