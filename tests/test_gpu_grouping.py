"""GPU parity: FPS and kNN-then-radius grouping through the C ABI vs the CPU oracle and the
reference goldens (bit-exact indices)."""
import numpy as np
import pytest
import torch

from helpers import canon_groups_np, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


@pytest.mark.parametrize("N", [512, 2048])
@pytest.mark.parametrize("dt", ["float64", "float32"])
def test_fps_golden(dev, N, dt):
    from facl_amd import fps
    g = load_golden("fps.npz")
    pc = torch.from_numpy(g[f"pc_N{N}"].astype(dt)).to(dev)
    out = fps.farthest_point_sampling_batch(pc, 64, g[f"start_N{N}"])
    np.testing.assert_array_equal(out.cpu().numpy(), g[f"idx_N{N}_{dt}"])
    one = fps.farthest_point_sampling_fast(pc[0], 64, int(g[f"start_N{N}"][0]))
    assert one.shape == (64, 1) and one.dtype == torch.int32


@pytest.mark.parametrize("N,m,D", [(1, 1, 3), (63, 5, 3), (65, 65, 4), (1000, 64, 4), (4096, 128, 3), (300, 300, 3)])
def test_fps_vs_oracle_ragged(dev, N, m, D):
    from facl_amd import fps
    from oracle import fps as OF
    rng = np.random.RandomState(N + m)
    pts = (rng.rand(3, N, D) - 0.5).astype(np.float32)
    pts[1, N // 2:] = pts[1, : N - N // 2]            # duplicate points: exact ties -> lowest index wins
    starts = rng.randint(0, N, size=3)
    out = fps.farthest_point_sampling_batch(torch.from_numpy(pts).to(dev), m, starts).cpu().numpy()
    for c in range(3):
        ref = OF.farthest_point_sampling_fast(pts[c, :, :3], m, int(starts[c])).ravel()
        np.testing.assert_array_equal(out[c], ref)


def test_fps_reorder_vs_oracle(dev):
    from facl_amd import fps
    from oracle import fps as OF
    rng = np.random.RandomState(3)
    for N, S in [(512, 64), (2048, 64), (100, 100), (777, 1)]:
        pts = (rng.rand(5, N, 4) - 0.5).astype(np.float32)
        starts = rng.randint(0, N, size=5)
        out = fps.fps_sample_data(torch.from_numpy(pts).to(dev), S, start_idx=starts).cpu().numpy()
        np.testing.assert_array_equal(out, OF.fps_sample_data(pts, S, starts))
        # property: a permutation of the rows
        np.testing.assert_array_equal(np.sort(out.reshape(5, -1), axis=1), np.sort(pts.reshape(5, -1), axis=1))


def test_fps_reorder_two_level_vs_reference_golden(dev):
    """HIP 2-level fps_sample_data (cn3d_data_load.py:287-298) against `reorder_out`, the output of the REFERENCE function
    on float64 clouds with its np.random.randint draws recorded (tools/make_goldens.py: make_fps)."""
    from facl_amd import fps
    g = load_golden("fps.npz")
    pts64 = torch.from_numpy(g["reorder_in"]).to(dev)                      # (2,512,4) float64
    out = fps.fps_sample_data(pts64, 64, 16, start_idx=g["reorder_s1"], start_idx2=g["reorder_s2"],
                              xyz=pts64[:, :, :3])                          # sampling in the reference's precision
    np.testing.assert_array_equal(out.cpu().numpy(), g["reorder_out"].astype(np.float32))
    # float32 sampling picks the same rows on this (tie-free) cloud
    out32 = fps.fps_sample_data(pts64.float(), 64, 16, start_idx=g["reorder_s1"], start_idx2=g["reorder_s2"])
    np.testing.assert_array_equal(out32.cpu().numpy(), g["reorder_out"].astype(np.float32))


def test_fps_reorder_two_level_vs_oracle(dev):
    from facl_amd import fps
    from oracle import fps as OF
    rng = np.random.RandomState(5)
    for N, S1, S2 in [(512, 64, 16), (2048, 128, 32), (100, 100, 100), (300, 7, 1)]:
        pts = (rng.rand(4, N, 3) - 0.5).astype(np.float32)
        s1, s2 = rng.randint(0, N, size=4), rng.randint(0, S1, size=4)
        out = fps.fps_sample_data(torch.from_numpy(pts).to(dev), S1, S2, start_idx=s1, start_idx2=s2).cpu().numpy()
        np.testing.assert_array_equal(out, OF.fps_sample_data_2level(pts, S1, S2, s1, s2))


@pytest.mark.parametrize("B,G,N,D", [(3, 4, 512, 4), (2, 5, 2048, 3), (1, 1, 100, 3)])
def test_group_clip_major_equals_permuted_copy(dev, B, G, N, D):
    """facl_group_clips on the loader's (B,G,N,D) batch == facl_group on permute(1,0,2,3).reshape(G*B,N,D)
    (cn3d_train_motion_GL.py:226), bit for bit."""
    from facl_amd import utils_my
    torch.manual_seed(B * G + N)
    clips = (torch.rand(B, G, N, D, device=dev) - 0.5)
    S, K = min(64, N), min(64, N)
    a = utils_my.knn_radius_group(clips, S, K, 0.1, want_idx=True)
    b = utils_my.knn_radius_group(clips.permute(1, 0, 2, 3).reshape(-1, N, D).contiguous(), S, K, 0.1, want_idx=True)
    for ta, tb in zip(a, b):
        assert ta.shape == tb.shape and ta.stride() == tb.stride() and torch.equal(ta, tb)


@pytest.mark.parametrize("tag,r2", [("r016", 0.16), ("r006", 0.06)])
def test_group_tiny_golden(dev, tag, r2):
    from facl_amd import utils_my
    g = load_golden("tiny.npz")
    pts = torch.from_numpy(g["points"]).to(dev)
    before = pts.clone()
    xt, yt, idx = utils_my.knn_radius_group(pts, 16, 8, r2, want_idx=True)
    assert torch.equal(pts, before)                          # must not mutate its input [SURVEY 8b]
    assert xt.shape == (6, 4, 16, 8) and yt.shape == (6, 3, 16, 1)
    assert xt.stride() == (16 * 8 * 4, 1, 8 * 4, 4) and yt.stride() == (48, 1, 3, 48)   # reference views
    np.testing.assert_array_equal(canon_groups_np(xt.permute(0, 2, 3, 1).cpu().numpy()), g[f"xt_{tag}"])
    np.testing.assert_array_equal(yt.contiguous().cpu().numpy(), g[f"yt_{tag}"])


@pytest.mark.parametrize("M,N,D,S,K,r2", [
    (32, 512, 4, 64, 64, 0.06),      # C1
    (32, 512, 3, 64, 64, 0.06),
    (5, 2048, 4, 64, 64, 0.16),      # headline cloud size
    (3, 4096, 3, 128, 32, 0.01),     # config-5 cloud size
    (4, 100, 4, 7, 100, 0.05),       # ragged: N not a multiple of 64, K == N
    (2, 65, 3, 65, 1, 0.5),          # K == 1: every group is its own centroid
    (2, 64, 4, 1, 64, 0.0),          # r2 = 0: everything but the centroid collapses
])
def test_group_vs_oracle(dev, M, N, D, S, K, r2):
    from facl_amd import utils_my
    from oracle import grouping as OG
    rng = np.random.RandomState(M * N + K)
    pts = (rng.rand(M, N, D) - 0.5).astype(np.float32)
    xt, yt, idx = utils_my.knn_radius_group(torch.from_numpy(pts).to(dev), S, K, r2, want_idx=True)
    ridx, rxt, ryt = OG.group_points(pts, S, K, r2)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)          # bit-exact, same (ascending) order
    np.testing.assert_array_equal(xt.permute(0, 2, 3, 1).cpu().numpy(), rxt)
    np.testing.assert_array_equal(yt.squeeze(-1).permute(0, 2, 1).cpu().numpy(), ryt)


def test_group_random_sweep_vs_oracle(dev):
    """Seeded sweep over shapes the fixed cases do not hit: K in the slot range (<= 256) and above it (direct emission),
    K not a multiple of 64, N up to 4096, clouds with heavy exact duplicates (the tie-ranking emission path) next to
    continuous ones (the single-compare path), D = 3 / 4, radii from "everything collapses" to "nothing does"."""
    from facl_amd import utils_my
    from oracle import grouping as OG
    rng = np.random.RandomState(2024)
    for case in range(14):
        N = int(rng.choice([96, 130, 512, 777, 1024, 2048, 4096]))
        K = int(rng.choice([1, 5, 64, 100, 128, 200, 256, 300]))
        K = min(K, N)
        S = int(rng.choice([1, 16, 33, 64]))
        S = min(S, N)
        D = int(rng.choice([3, 4]))
        M = int(rng.choice([1, 3]))
        r2 = float(rng.choice([0.0, 0.01, 0.06, 0.16, 10.0]))
        pts = (rng.rand(M, N, D) - 0.5).astype(np.float32)
        if case % 2:                                                    # duplicates: a third of the points are copies
            src = rng.randint(0, N, size=N // 3)
            dst = rng.randint(0, N, size=N // 3)
            pts[:, dst] = pts[:, src]
        xt, yt, idx = utils_my.knn_radius_group(torch.from_numpy(pts).to(dev), S, K, r2, want_idx=True)
        ridx, rxt, ryt = OG.group_points(pts, S, K, r2)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx, err_msg=f"case {case}: N={N} K={K} S={S} D={D} r2={r2}")
        np.testing.assert_array_equal(xt.permute(0, 2, 3, 1).cpu().numpy(), rxt)


def test_group_ties_take_lowest_index(dev):
    """Exact distance ties (duplicated points): deterministic lowest-index choice, as the oracle."""
    from facl_amd import utils_my
    from oracle import grouping as OG
    rng = np.random.RandomState(0)
    base = (rng.rand(2, 64, 4) - 0.5).astype(np.float32)
    pts = np.concatenate([base, base, base], axis=1)                 # every point three times
    xt, yt, idx = utils_my.knn_radius_group(torch.from_numpy(pts).to(dev), 16, 32, 10.0, want_idx=True)
    ridx, _, _ = OG.group_points(pts, 16, 32, 10.0)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


def test_group_c1_golden_and_opt_side_effects(dev):
    from types import SimpleNamespace
    from facl_amd import utils_my
    g = load_golden("c1_d4.npz")
    opt = SimpleNamespace(SAMPLE_NUM=512, sample_num_level1=64, knn_K=1, ball_radius=0.16, INPUT_FEATURE_NUM=0)
    xt, yt = utils_my.group_points_3DV(torch.from_numpy(g["points"]).to(dev), opt)
    assert (opt.knn_K, opt.ball_radius, opt.INPUT_FEATURE_NUM) == (64, 0.06, 4)     # utils_my.py:259-261
    c = canon_groups_np(xt.permute(0, 2, 3, 1).cpu().numpy())
    np.testing.assert_array_equal(c[:8], g["xt_first8"])
    np.testing.assert_allclose(c.sum(axis=2), g["xt_sum"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(yt.contiguous().cpu().numpy(), g["yt"])


def test_group_headline_size_properties(dev):
    """B=32,T=24,N=2048 (M=768): size-independent properties + an independent torch.topk cross-check."""
    from facl_amd import utils_my
    torch.manual_seed(0)
    M, N, D, S, K, r2 = 768, 2048, 4, 64, 64, 0.16
    pts = (torch.rand(M, N, D) - 0.5).to(dev)
    xt, yt, idx = utils_my.knn_radius_group(pts, S, K, r2, want_idx=True)
    idx = idx.long()
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    # the centroid is always its own nearest neighbour (distance 0) [SURVEY 8c]
    assert bool((idx == torch.arange(S, device=dev).view(1, S, 1)).any(dim=2).all())
    # gather consistency: xt is points[idx] centred on the centroid
    g = torch.gather(pts, 1, idx.view(M, S * K, 1).expand(M, S * K, D)).view(M, S, K, D).clone()
    g[..., :3] -= pts[:, :S, None, :3]
    assert torch.equal(g, xt.permute(0, 2, 3, 1))
    # kNN property on a slice, against torch.topk (independent implementation; sets compared)
    sl = slice(0, 64)
    d = pts[sl, :, None, :3].transpose(1, 2) - pts[sl, :S, None, :3]          # (m,S,N,3)
    d2 = (d * d).sum(-1)
    dv, di = torch.topk(d2, K, dim=2, largest=False)
    jj = torch.arange(S, device=dev).view(1, S, 1).expand_as(di)
    di = torch.where(dv > r2, jj, di)
    assert torch.equal(torch.sort(di, dim=2).values, torch.sort(idx[sl], dim=2).values)
