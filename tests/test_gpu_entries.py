"""End-to-end entries on the GPU: training entry -> checkpoint -> extraction entry, the HIP-graph
replay of the step, and checkpoint compatibility with the reference's key names."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_train_entry_checkpoint_then_extract(tmp_path):
    from facl_amd import cn3d_train_motion_GL as train, extract_motion_feature as ext
    from oracle.weights import state_dict_shapes
    ck = str(tmp_path / "ck")
    net = train.main(["--batchSize", "4", "--nepoch", "1", "--steps_per_epoch", "2", "--num_crop", "4", "--SAMPLE_NUM", "512",
                      "--save_root_dir", ck, "--INPUT_FEATURE_NUM", "4"])
    path = os.path.join(ck, "corr_GL_0.pth")                           # reference checkpoint name (:341)
    assert os.path.exists(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert list(sd.keys()) == [k for k, _ in state_dict_shapes(4)]     # the reference's 52 keys, in order
    assert int(sd["net3DV_1.1.num_batches_tracked"]) == 2 and int(sd["netR_FC.1.num_batches_tracked"]) == 4
    feats = ext.main(["--checkpoint", path, "--batchSize", "3", "--num_crop", "4", "--SAMPLE_NUM", "512",
                      "--INPUT_FEATURE_NUM", "4", "--num_batches", "2", "--save_path", str(tmp_path / "f")])
    assert feats.shape == (6, 5 * 512) and np.isfinite(feats).all()
    one = np.load(str(tmp_path / "f" / "synthetic_0000_000.npy"))
    assert one.shape == (5 * 512,) and one.dtype == np.float32

    # the extraction forward equals the oracle's eval-mode forward on the same checkpoint
    from facl_amd.extract_common import extract_batch, save_single_feature
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from oracle import encoder as E, grouping as OG
    opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                          sample_num_level2=64, INPUT_FEATURE_NUM=4, Num_Class=512, batchSize=3, pooling="concatenation",
                          SAMPLE_NUM=512)
    m = PointNet_Plus(opt, gost=4)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    torch.manual_seed(0)
    clip = torch.rand(3, 4, 512, 4) - 0.5
    with torch.no_grad():
        f = extract_batch(m, clip.to(DEV), opt).cpu().numpy()
    pts = clip.permute(1, 0, 2, 3).reshape(-1, 512, 4).numpy()
    _, xt, yt = OG.group_points(pts, 64, 64, 0.06)
    sdo = {k: v.double() if v.is_floating_point() else v.clone() for k, v in sd.items()}
    with torch.no_grad():
        x, _, _, xg = E.encoder_forward(sdo, torch.from_numpy(xt).permute(0, 3, 1, 2).double(),
                                        torch.from_numpy(yt).view(12, 1, 64, 3).transpose(1, 3).double(), 4, training=False)
    ref = save_single_feature(torch.cat((x, xg), 0).numpy(), str(tmp_path), ["a", "b", "c"], num_crop=5)
    err = np.linalg.norm(f - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < 1e-4, err


def test_train_entry_graph_mode_follows_the_eager_trajectory(tmp_path):
    """--graph 1 (default): the entry captures the iteration on its first batch (three warm-up steps, state restored) and
    replays it; parameters, BatchNorm buffers and counters after 2 epochs x 3 steps equal the --graph 0 run's."""
    from facl_amd import cn3d_train_motion_GL as train
    args = ["--batchSize", "4", "--nepoch", "2", "--steps_per_epoch", "3", "--num_crop", "4", "--SAMPLE_NUM", "512",
            "--INPUT_FEATURE_NUM", "4"]
    sds = []
    for gflag in ("1", "0"):
        net = train.main(args + ["--graph", gflag, "--save_root_dir", str(tmp_path / ("ck" + gflag))])
        sds.append({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    for k, v in sds[0].items():
        if v.is_floating_point():
            assert torch.allclose(v, sds[1][k], rtol=1e-5, atol=1e-7), k
        else:
            assert torch.equal(v, sds[1][k]), k
    assert int(sds[0]["net3DV_1.1.num_batches_tracked"]) == 6


def test_train_entry_from_raw_clips_through_gpu_view_construction(tmp_path):
    """SURVEY 8 f-3 wired into the entry (--synthetic 2): raw clips -> facl_amd.views.build_views (the dataset class's 10 views,
    one HIP launch, draws in the reference's NumPy order) -> the loop body of cn3d_train_motion_GL.py:224-335 on the view-major
    rows.  The views the step consumed equal oracle/views.py for the same clips and generator; graph replay == eager."""
    from facl_amd import cn3d_train_motion_GL as train
    from facl_amd.views import build_views, synthetic_raw_clip
    from oracle import views as OV
    args = ["--batchSize", "4", "--nepoch", "1", "--steps_per_epoch", "3", "--num_crop", "10", "--SAMPLE_NUM", "512",
            "--INPUT_FEATURE_NUM", "4", "--synthetic", "2"]
    sds = []
    for gflag in ("1", "0"):
        net = train.main(args + ["--graph", gflag, "--save_root_dir", str(tmp_path / ("ck" + gflag))])
        sds.append({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    for k, v in sds[0].items():
        assert torch.isfinite(v.float()).all(), k
        if v.is_floating_point():
            assert torch.allclose(v, sds[1][k], rtol=1e-5, atol=1e-7), k
    assert int(sds[0]["net3DV_1.1.num_batches_tracked"]) == 3
    # the first batch the entry built: rank 0, iteration 0 -> clips seeded 0..3, generator RandomState(2000)
    clips = [synthetic_raw_clip(b) for b in range(4)]
    got = build_views(clips, np.random.RandomState(2000)).cpu().numpy()
    rng = np.random.RandomState(2000)
    want = OV.collate_view_major([OV.get_item(rng, *c) for c in clips])
    assert got.shape == (40, 512, 4)
    np.testing.assert_allclose(got, want, rtol=0, atol=np.spacing(np.float32(1.0)))
    with pytest.raises(RuntimeError, match="num_crop 10"):
        train.main(["--batchSize", "2", "--nepoch", "1", "--steps_per_epoch", "1", "--num_crop", "4", "--synthetic", "2",
                    "--save_root_dir", str(tmp_path / "bad")])


def test_graph_replay_equals_eager():
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from facl_amd.train_common import ContrastiveStep, GraphedStep
    from oracle.weights import formula_state_dict
    D, B, G, N = 4, 2, 3, 512
    opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                          sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation",
                          SAMPLE_NUM=N)
    torch.manual_seed(0)
    clips = [torch.rand(B, G, N, D, device=DEV) - 0.5 for _ in range(3)]
    orders = [np.array([1, 2, 0]), np.array([2, 0, 1]), np.array([0, 2, 1])]

    def make():
        net = PointNet_Plus(opt, gost=G)
        net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()})
        net = net.to(DEV).train()
        optim = torch.optim.Adam(net.parameters(), lr=3e-4, betas=(0.5, 0.999), eps=1e-6, capturable=True)
        return net, ContrastiveStep(net, optim, opt, G)

    net_e, step_e = make()
    eager = [float(step_e(c, order=o)[0]) for c, o in zip(clips, orders)]
    net_g, step_g = make()
    g = GraphedStep(step_g, clips[0], G)                     # capture runs warm-up steps: reset the state afterwards
    net_g.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()})
    for st in step_g.optimizer.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    graphed = [float(g(c, order=o)[0]) for c, o in zip(clips, orders)]
    print(eager, graphed)
    np.testing.assert_allclose(graphed, eager, rtol=2e-5)


def test_graph_replay_sees_learning_rate_changes_with_fused_adam():
    """A StepLR change made AFTER the capture must reach the replayed Adam (FusedAdam keeps lr on the device; the graph
    holds no fill, GraphedStep refreshes it before each replay).  Parameters after 3 steps with lr 3e-4, 1e-3, 1e-5:
    graph replay == eager FusedAdam (FusedAdam == torch.optim.Adam under an lr change: tests/test_gpu_tail.py)."""
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from facl_amd.optim import FusedAdam
    from facl_amd.train_common import ContrastiveStep, GraphedStep
    from oracle.weights import formula_state_dict
    D, B, G, N = 4, 2, 3, 512
    opt = SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                          sample_num_level2=64, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation",
                          SAMPLE_NUM=N)
    torch.manual_seed(0)
    clips = [torch.rand(B, G, N, D, device=DEV) - 0.5 for _ in range(3)]
    order = np.array([1, 2, 0])
    lrs = [3e-4, 1e-3, 1e-5]
    sd0 = {k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()}

    def make(kind):
        net = PointNet_Plus(opt, gost=G)
        net.load_state_dict(sd0)
        net = net.to(DEV).train()
        if kind == "torch":
            optim = torch.optim.Adam(net.parameters(), lr=lrs[0], betas=(0.5, 0.999), eps=1e-6)
        else:
            optim = FusedAdam(net.parameters(), lr=lrs[0], betas=(0.5, 0.999), eps=1e-6)
        return net, ContrastiveStep(net, optim, opt, G)

    def run(net, step, call):
        for c, lr in zip(clips, lrs):
            for gr in step.optimizer.param_groups:
                gr["lr"] = lr
            call(c)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in net.named_parameters()}

    net_e, step_e = make("fused")
    p_eager = run(net_e, step_e, lambda c: step_e(c, order=order))
    net_g, step_g = make("fused")
    g = GraphedStep(step_g, clips[0], G)                     # warm-up steps are real steps: reset the state afterwards
    net_g.load_state_dict(sd0)
    for st in step_g.optimizer.state.values():
        for v in st.values():
            v.zero_()
    step_g.optimizer._step.zero_()
    p_graph = run(net_g, step_g, lambda c: g(c, order=order))
    moved = 0
    for k in p_eager:
        ref = p_eager[k]
        assert torch.allclose(p_graph[k], ref, rtol=1e-5, atol=1e-7), k          # the graph followed the schedule
        moved += int((ref.cpu() - sd0[k]).abs().max() > 5e-4)
    assert moved > 10                                         # the 1e-3 step is visible: a stuck 3e-4 / 1e-5 would not pass


def test_linear_probe_consumes_extracted_feature_format():
    """(f)-2: the probe trains on vectors in the extraction format (motion ++ appearance, 22*512) and separates
    classes that differ in the features; state_dict keys match the reference's Final_FC."""
    from facl_amd import linear_classify as LC
    torch.manual_seed(0)
    n, C = 512, 8
    labels = torch.randint(0, C, (n,), device=DEV)
    protos = torch.randn(C, 22 * 512, device=DEV)
    feats = protos[labels] + 0.5 * torch.randn(n, 22 * 512, device=DEV)
    model, top1 = LC.fit(feats, labels, num_class=120, nepoch=6, batch=128, lr=3e-3)
    assert list(model.state_dict().keys()) == ["fc.weight", "fc.bias"]
    assert tuple(model.fc.weight.shape) == (120, 22 * 512)
    assert top1 > 90.0, top1
    ref = torch.nn.functional.linear(torch.nn.functional.normalize(feats[:7].double(), dim=1), model.fc.weight.double(),
                                     model.fc.bias.double())
    assert torch.allclose(model(feats[:7]).double(), ref, rtol=1e-4, atol=1e-5)


def test_appearance_entry_checkpoint_name_and_extract(tmp_path):
    """BASELINE configs[2]: the appearance stream = same model / kernels / loss through cn3d_train_apperance_GL's entry
    (branch '1', checkpoint corr_GL_appereance_<epoch>.pth, :341) and extract_apperance_feature."""
    from facl_amd import cn3d_train_apperance_GL as train, extract_apperance_feature as ext
    from oracle.weights import state_dict_shapes
    ck = str(tmp_path / "ck")
    train.main(["--batchSize", "3", "--nepoch", "1", "--steps_per_epoch", "1", "--num_crop", "4", "--SAMPLE_NUM", "512",
                "--save_root_dir", ck, "--INPUT_FEATURE_NUM", "4"])
    path = os.path.join(ck, "corr_GL_appereance_0.pth")
    assert os.path.exists(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert list(sd.keys()) == [k for k, _ in state_dict_shapes(4)]
    feats = ext.main(["--checkpoint", path, "--batchSize", "2", "--num_crop", "4", "--SAMPLE_NUM", "512",
                      "--INPUT_FEATURE_NUM", "4", "--num_batches", "1", "--save_path", str(tmp_path / "f")])
    assert feats.shape == (2, 5 * 512) and np.isfinite(feats).all()
