"""GPU view construction (facl_amd/views.py + csrc/views.hip) against the reference fixture
tests/golden/views.npz (outputs of the reference's own NTU_RGBD_new.get_temporal_augment_data / get_data_train,
cn3D_data_set.py:654-663 / :285-350, under np.random.seed) and against oracle/views.py, which that fixture pins."""
import numpy as np
import pytest

from oracle import views as OV


from helpers import golden_view_clips, load_golden, synth_clip as _clip


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_host_draws_follow_the_reference_order(dt):
    """draw_clip must consume the generator exactly like the restated __getitem__ (same count, same order)."""
    from facl_amd.views import draw_clip
    clip = _clip(5, dt)
    r1, r2 = np.random.RandomState(11), np.random.RandomState(11)
    OV.get_item(r1, *clip)
    idx, noise, cs = draw_clip(r2, *clip, base=[0, 900, 1200, 1700])
    assert r1.rand() == r2.rand()                                   # streams in lock-step afterwards
    assert idx.shape == (10, 512) and idx.dtype == np.int32
    assert idx[:2].max() < 900 and idx[2:4].min() >= 900 and idx[2:4].max() < 1200
    assert (clip[0][idx[6], 4] != 0).all() and (clip[0][idx[7], 7] != 0).all()
    assert idx[8].min() >= 1200 and idx[8].max() < 1700 and idx[9].min() >= 1700 and idx[9].max() < 1900


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_build_views_matches_oracle(dt):
    import torch
    from facl_amd.views import build_views
    clips = [_clip(1, dt), _clip(2, dt, P=777, Kp=513, R1=400, R2=64), _clip(3, dt, P=2048, Kp=1024)]
    out = build_views(clips, np.random.RandomState(42)).cpu().numpy()
    rng = np.random.RandomState(42)
    want = OV.collate_view_major([OV.get_item(rng, *c) for c in clips])
    assert out.shape == want.shape == (30, 512, 4) and out.dtype == np.float32
    B = 3
    for g in range(10):
        a, w = out[g * B:(g + 1) * B], want[g * B:(g + 1) * B]
        if g in (4, 5):
            # rotated views: float32 xyz @ float64 Ry rounded to float32 -- NumPy's dgemm may fuse the second
            # multiply-add, the kernel does not: allow one float32 ulp
            np.testing.assert_allclose(a, w, rtol=0, atol=np.spacing(np.float32(1.0)))
            assert (a == w).mean() > 0.999
        else:
            np.testing.assert_array_equal(a, w)
    assert not np.array_equal(out[1 * B], out[0 * B])               # views differ


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [42, 7])
def test_build_views_matches_reference_golden(seed):
    """HIP views == float32(view-major collate of the REFERENCE's get_data_train outputs) for the same clips and the same
    generator seed.  build_views takes one source dtype per batch, so the float32 clips (a, b) and the float64 clips
    (c, d) go through two calls on ONE generator, in the fixture's stream order."""
    import torch
    from facl_amd.views import build_views
    g = load_golden("views.npz")
    clips = golden_view_clips(g)
    rng = np.random.RandomState(seed)
    for batch in (clips[:2], clips[2:]):
        out = build_views([c for _, c in batch], rng).cpu().numpy()
        B = len(batch)
        want = np.stack([g[f"seed{seed}/{t}"] for t, _ in batch], 0).transpose(1, 0, 2, 3).reshape(10 * B, 512, 4).astype(np.float32)
        for v in range(10):
            a, w = out[v * B:(v + 1) * B], want[v * B:(v + 1) * B]
            if v in (4, 5):                                           # rotated views: one float32 ulp (see above)
                np.testing.assert_allclose(a, w, rtol=0, atol=np.spacing(np.float32(1.0)))
                assert (a == w).mean() > 0.999
            else:
                np.testing.assert_array_equal(a, w)
    assert rng.rand() == float(g[f"seed{seed}/next_rand"])


@pytest.mark.gpu
def test_build_views_device_rng_has_the_reference_distributions():
    """device_rng mode draws on the GPU (another stream than NumPy's): every view must still be what get_data_train builds --
    rows of the right source cloud, jitter clipped at 0.05 with sigma 0.01, x mirrored, rotations norm-preserving about y,
    temporal views from rows with a non-zero temporal channel."""
    import torch
    from facl_amd.views import build_views
    B = 6
    clips = [_clip(10 + b, np.float32, P=700 + 13 * b, Kp=300 + b, R1=400, R2=100) for b in range(B)]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    out = build_views(clips, device_rng=gen).cpu().numpy().reshape(10, B, 512, 4)
    out2 = build_views(clips, device_rng=gen).cpu().numpy().reshape(10, B, 512, 4)
    assert not np.array_equal(out, out2)                                     # the generator advances
    for b, (pts, key, r1, r2) in enumerate(clips):
        def rows_of(a, src, cols=(0, 1, 2, 3)):
            s = {tuple(r) for r in src[:, list(cols)].astype(np.float32)}
            return all(tuple(r) in s for r in a)
        assert rows_of(out[0, b], pts) and rows_of(out[8, b], r1) and rows_of(out[9, b], r2)       # plain gathers
        assert rows_of(out[6, b], pts[pts[:, 4] != 0], (0, 1, 2, 4)) and rows_of(out[7, b], pts[pts[:, 7] != 0], (0, 1, 2, 7))
        assert rows_of(out[2, b][:, 3:], key, (3,))                                                  # 4th channel untouched
        # key-point view: xyz = a source row + clipped jitter
        d = np.abs(out[2, b][:, None, :3] - key[None, :, :3].astype(np.float32)).max(-1).min(1)
        assert d.max() <= 0.05 + 1e-6 and 0.004 < d.mean() < 0.03
        # rotated views: |(x, z)| and y of SOME jittered source row are preserved -> y within jitter of a source y
        dy = np.abs(out[4, b][:, None, 1] - pts[None, :, 1].astype(np.float32)).min(1)
        assert dy.max() <= 0.05 + 1e-6
    # reversed views: x mirrored -> the mean of x flips sign relative to the source's (sources are not centred per clip)
    assert np.isfinite(out).all()
