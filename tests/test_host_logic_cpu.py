"""CPU tests of the host-side mirror: model container (state_dict contract, initialisation), the loss
functions (pure tensor algebra, device-agnostic), LR schedule, and the world_size-2 gloo path."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(D=4, B=4):
    return SimpleNamespace(temperal_num=3, knn_K=64, ball_radius2=0.25, sample_num_level1=64, sample_num_level2=64,
                           INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation", SAMPLE_NUM=512)


@pytest.mark.parametrize("D", [3, 4])
def test_state_dict_contract_and_seeded_init(D):
    """52 reference keys/shapes, and the same initial weights as the reference under manual_seed(1)."""
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from oracle.weights import state_dict_shapes
    g = load_golden("init.npz")
    torch.manual_seed(1)
    net = PointNet_Plus(_opt(D))
    sd = net.state_dict()
    want = state_dict_shapes(D)
    assert list(sd.keys()) == [k for k, _ in want]
    for k, shape in want:
        assert tuple(sd[k].shape) == tuple(shape), k
        a = sd[k].numpy().astype(np.float64).reshape(-1)
        fp = g[f"D{D}/{k}"]
        assert a.size == int(fp[0])
        np.testing.assert_allclose([a.sum(), np.abs(a).sum()], fp[1:3], rtol=1e-12, atol=1e-12, err_msg=k)
        np.testing.assert_array_equal(a[:8], fp[3:3 + min(8, a.size)], err_msg=k)
    assert sum(p.numel() for p in net.parameters()) == 2358144 - (64 if D == 3 else 0)
    # round trip through the formula weights (= checkpoint compatibility of key names / shapes)
    from oracle.weights import formula_state_dict
    net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()}, strict=True)


def test_fine_variant_and_ctor_errors():
    from facl_amd.cn3d_model_conbag import PointNet_Plus, PointNet_Plus_fine
    net = PointNet_Plus_fine(_opt(), gost=8, sample_num_level1=64, knn_K=64)
    assert set(net.state_dict().keys()) == set(PointNet_Plus(_opt()).state_dict().keys())
    bad = _opt()
    bad.pooling = "vlad"
    with pytest.raises(AttributeError):          # reference: dim_out only exists for 'concatenation' (:40-41)
        PointNet_Plus(bad)
    with pytest.raises(RuntimeError):            # no CPU path
        PointNet_Plus(_opt())(torch.zeros(10, 4, 64, 64), torch.zeros(10, 3, 64, 1))


@pytest.mark.parametrize("tag", ["d4", "d3", "d4_neg"])
def test_losses_match_reference_golden(tag):
    from facl_amd.utils_my import circle_contrast, global_contrast
    g = load_golden(f"c1_{tag}.npz")
    B, G = int(g["meta"][0]), int(g["meta"][1])
    x = torch.from_numpy(g["train_x"])
    xg = torch.from_numpy(g["train_x_global"])
    lc = global_contrast(G, xg, x, SimpleNamespace(batchSize=B))
    lo = circle_contrast(G, x, B, order=g["order"])
    assert abs(float(lc) - float(g["loss_c"])) <= 1e-5 * abs(float(g["loss_c"]))
    assert abs(float(lo) - float(g["loss_circle"])) <= 1e-5 * abs(float(g["loss_circle"]))


def test_losses_and_grads_match_oracle_random():
    from facl_amd.utils_my import circle_contrast, global_contrast
    from oracle import loss as OL
    torch.manual_seed(0)
    G, B, C = 6, 5, 32
    x = torch.randn(G * B, C, dtype=torch.float64, requires_grad=True)
    xg = torch.randn(B, C, dtype=torch.float64, requires_grad=True)
    order = np.array([2, 0, 5, 1, 4, 3])
    mine = global_contrast(G, xg, x, None) + circle_contrast(G, x, B, order=order)
    gm = torch.autograd.grad(mine, (x, xg))
    ref = OL.global_contrast(G, xg, x, B) + OL.circle_contrast(G, x, B, order)
    gr = torch.autograd.grad(ref, (x, xg))
    assert abs(float(mine) - float(ref)) < 1e-10 * abs(float(ref))
    for a, b in zip(gm, gr):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_lr_schedule_matches_steplr_with_explicit_epoch():
    from facl_amd.train_common import lr_for_epoch
    from oracle.step import lr_at
    for e in range(0, 23):
        assert lr_for_epoch(3e-4, e) == pytest.approx(3e-4 * 0.7 ** (e // 4))
        assert lr_for_epoch(3e-4, e) == pytest.approx(lr_at(e))


def test_train_entry_flags_match_reference_defaults():
    from facl_amd.train_common import build_parser
    o = build_parser('0').parse_args([])
    assert (o.batchSize, o.nepoch, o.INPUT_FEATURE_NUM, o.SAMPLE_NUM, o.knn_K, o.sample_num_level1) == (64, 100, 4, 512, 64, 64)
    assert (o.learning_rate, o.ball_radius, o.ball_radius2, o.Num_Class, o.pooling) == (0.0003, 0.16, 0.25, 512, 'concatenation')
    assert build_parser('1').parse_args([]).branch_choose == '1'


# ---- world_size = 2 over gloo: sharded losses + all-gather == single process at the global batch ----
def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from facl_amd import dist as fdist
    from facl_amd.utils_my import circle_contrast, global_contrast
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    G, Bl, C = 5, 3, 16
    Bg = Bl * world
    x_full = torch.randn(G, Bg, C, dtype=torch.float64)
    xg_full = torch.randn(Bg, C, dtype=torch.float64)
    order = np.array([4, 1, 0, 3, 2])
    # single-process reference at the global batch
    xf = x_full.reshape(G * Bg, C).clone().requires_grad_(True)
    xgf = xg_full.clone().requires_grad_(True)
    ref = global_contrast(G, xgf, xf, None) + circle_contrast(G, xf, Bg, order=order)
    gx_ref, gxg_ref = torch.autograd.grad(ref, (xf, xgf))
    # sharded
    sl = slice(rank * Bl, (rank + 1) * Bl)
    xl = x_full[:, sl].reshape(G * Bl, C).clone().requires_grad_(True)
    xgl = xg_full[sl].clone().requires_grad_(True)
    keys = fdist.all_gather_view_major(xl, G)
    assert torch.equal(keys.detach(), x_full.reshape(G * Bg, C))             # re-layout to global view-major
    loss = global_contrast(G, xgl, xl, None, x_keys=keys, clip_offset=rank * Bl) + \
        circle_contrast(G, xl, Bl, order=order, x_keys=keys, clip_offset=rank * Bl)
    gx, gxg = torch.autograd.grad(loss, (xl, xgl))
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    ok = abs(float(tot) / world - float(ref)) < 1e-10 * abs(float(ref))
    # DDP averages gradients: (1/R) * sum_r dL_r  ==  d(global mean loss); per-row pieces are local
    ok &= torch.allclose(gx / world, gx_ref.view(G, Bg, C)[:, sl].reshape(G * Bl, C), rtol=1e-9, atol=1e-12)
    ok &= torch.allclose(gxg / world, gxg_ref[sl], rtol=1e-9, atol=1e-12)
    # SyncBN hook + flat gradient bucket
    red = fdist.make_bn_reduce_fn()
    t = torch.full((4,), float(rank + 1), dtype=torch.float64)
    red(t)
    ok &= bool((t == 3.0).all())
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.full((3,), float(rank))
    fdist.allreduce_gradients([p])
    ok &= bool(torch.allclose(p.grad, torch.full((3,), 0.5)))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_world_size_2_gloo_sharded_loss_equals_global():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---- failure protocol of the segmented graph capture (facl_amd/dist.py: GraphSegments votes), 2 gloo ranks, no GPU ----
def _worker_capture_protocol(rank, world, port, q, fail_rank, fail_cut, ncoll):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      FACL_DIST_BACKEND="gloo", FACL_DIST_TIMEOUT_S="30")
    if fail_rank is not None:
        os.environ["FACL_TEST_CAPTURE_FAIL"] = "%d:%d" % (fail_rank, fail_cut)
    import torch.distributed as dist
    from facl_amd import dist as fdist
    fdist.init_from_env()

    class FakeSegments(fdist.GraphSegments):             # the protocol without a device: "graphs" are plain markers
        def __init__(self):
            self.pool, self.items, self.cur, self.ncuts = None, [], None, 0
            self.voting = True
            self._inject = {}
            v = os.environ.get("FACL_TEST_CAPTURE_FAIL")
            if v and int(v.split(":")[0]) == dist.get_rank():
                self._inject["FACL_TEST_CAPTURE_FAIL"] = int(v.split(":")[1])

        def begin(self):
            self.cur = "graph"

        def _close(self):
            self.items.append(self.cur)
            self.cur = None

        def abort(self):
            self.cur = None

    t = torch.zeros(1, dtype=torch.float64)

    def body():
        for i in range(ncoll):
            fdist.collective(lambda: dist.all_reduce(t.fill_(1.0)))
        return "captured"

    rec = FakeSegments()
    try:
        out = fdist.run_capture(rec, body, rank)
    except fdist.CaptureFailed as e:
        out = "failed: " + str(e)
    # whatever happened, the ranks are at the SAME point of the collective sequence: an eager collective completes at once
    dist.all_reduce(t.fill_(float(rank + 1)))
    q.put((rank, out.split(":")[0], rec.ncuts, float(t)))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank,fail_cut,ncoll", [(None, 0, 5), (1, 3, 5), (0, 0, 5), (1, 4, 5)])
def test_capture_failure_on_one_rank_is_agreed_on_by_all_ranks(fail_rank, fail_cut, ncoll):
    """a capture that fails on ONE rank after k collectives must take EVERY rank out of the capture at the same
    collective (then all continue eagerly together) -- never one rank eager against others replaying."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 2000) + (0 if fail_rank is None else 7 * (fail_rank + 1) + fail_cut)
    procs = [ctx.Process(target=_worker_capture_protocol, args=(r, 2, port, q, fail_rank, fail_cut, ncoll)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    expect = "captured" if fail_rank is None else "failed"
    assert [r[1] for r in res] == [expect, expect], res
    assert res[0][2] == res[1][2] == (ncoll if fail_rank is None else fail_cut), res     # same number of collectives issued
    assert res[0][3] == res[1][3] == 3.0, res                                           # and the next eager one pairs up


# ---- the embeddings exchange with R simulated ranks (lists of tensors): layout math of BOTH backward branches ----
@pytest.mark.parametrize("R", [2, 4, 8])
def test_simulated_ranks_reduce_scatter_layout_equals_allreduce_slice_and_single_process(R):
    """RCCL's backward branch (`reduce_scatter_tensor` on rank-major chunks, dist.py) has never run with R > 1 on
    hardware; gloo takes all_reduce + slice.  Emulate the collectives exactly -- all_gather_into_tensor = cat of the
    rank buffers; reduce_scatter_tensor = rank r receives the sum over ranks of chunk r; all_reduce = sum over ranks --
    and hold the forward layout and both backward branches to the single-process gradient at the global batch."""
    from facl_amd import dist as fdist
    from facl_amd.utils_my import circle_contrast, global_contrast
    torch.manual_seed(R)
    G, Bl, C = 5, 3, 8
    Bg = R * Bl
    order = np.array([4, 1, 0, 3, 2])
    x_full = torch.randn(G, Bg, C, dtype=torch.float64)
    xg_full = torch.randn(Bg, C, dtype=torch.float64)
    # single process, global batch
    xf = x_full.reshape(G * Bg, C).clone().requires_grad_(True)
    xgf = xg_full.clone().requires_grad_(True)
    ref = global_contrast(G, xgf, xf, None) + circle_contrast(G, xf, Bg, order=order)
    gx_ref, = torch.autograd.grad(ref, xf)
    # forward: every rank's all_gather_into_tensor output is the cat of the rank-local view-major buffers
    local = [x_full[:, r * Bl:(r + 1) * Bl].reshape(G * Bl, C).clone() for r in range(R)]
    buf = torch.cat(local, 0)
    keys0 = fdist.gathered_to_view_major(buf, G, R, Bl, C)
    assert torch.equal(keys0, x_full.reshape(G * Bg, C))
    assert torch.equal(keys0, torch.stack(local, 0).view(R, G, Bl, C).transpose(0, 1).reshape(G * Bg, C))   # stack form
    # each rank: loss of its own anchors against the gathered keys; gradient w.r.t. the keys (the exchange's input)
    g_keys, g_direct = [], []
    for r in range(R):
        xl = local[r].clone().requires_grad_(True)
        keys = keys0.clone().requires_grad_(True)
        xgl = xg_full[r * Bl:(r + 1) * Bl]
        loss = global_contrast(G, xgl, xl, None, x_keys=keys, clip_offset=r * Bl) + \
            circle_contrast(G, xl, Bl, order=order, x_keys=keys, clip_offset=r * Bl)
        gk, gl = torch.autograd.grad(loss, (keys, xl))
        g_keys.append(gk)
        g_direct.append(gl)
    # branch A (RCCL): reduce_scatter_tensor over rank-major chunks
    chunks = [fdist.view_major_to_rank_chunks(g, G, R, Bl, C) for g in g_keys]
    assert all(c.is_contiguous() and c.shape == (R * G * Bl, C) for c in chunks)
    rs = [sum(c[r * G * Bl:(r + 1) * G * Bl] for c in chunks) for r in range(R)]
    # branch B (gloo): all_reduce + slice
    g_sum = sum(g_keys)
    ar = [fdist.local_rows_of(g_sum, G, R, r, Bl, C) for r in range(R)]
    for r in range(R):
        assert torch.allclose(rs[r], ar[r], rtol=1e-12, atol=1e-14), r
        total = (rs[r] + g_direct[r]) / R                               # + the direct path, then the DDP average
        want = gx_ref.view(G, Bg, C)[:, r * Bl:(r + 1) * Bl].reshape(G * Bl, C)
        assert torch.allclose(total, want, rtol=1e-9, atol=1e-12), r
    # the two layout functions are exact inverses
    t = torch.randn(G * Bg, C)
    assert torch.equal(fdist.gathered_to_view_major(fdist.view_major_to_rank_chunks(t, G, R, Bl, C), G, R, Bl, C), t)


def test_committed_bench_line_follows_the_contract():
    """profiles/r03_bench_line.json (the last bench.py line measured on an MI355X) carries every field of the driver's
    contract plus the roofline / cpu_baseline objects; `roofline` is the LONGEST measured kernel (round-1 defect: it was
    hard-wired); bench.py itself needs a GPU and is run by the driver."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r03_bench_line.json")))
    assert "fps-reorder off" in d["config"]["workload"] and "ms_per_step_median_fenced" in d and d["dtype"] == "f32"
    dense = json.load(open(os.path.join(root, "profiles", "r03_bench_dense.json")))
    assert dense["metric"] != d["metric"] and "cpu_baseline" in dense and dense["dtype"] == "f16"
    assert all(d["roofline"]["ms_per_launch"] >= r["ms_per_launch"] for r in d["roofline_more"])
    assert "3 timed steps" in d["cpu_baseline"]["sample"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(root, "BASELINE.json")))["metric"]
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
