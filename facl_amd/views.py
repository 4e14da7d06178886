"""GPU view construction for a batch of clips.

Mirrors the per-sample work of the reference's dataset class (training_code/cn3D_data_set.py: `__getitem__` :99-140,
`get_temporal_augment_data` :654-663, `get_data_train` :285-350 with `jitter_point_cloud` :767-778,
`reverse_transform` :708-713, `rotate_trans` :734-749) plus the loop head of cn3d_train_motion_GL.py:225-228
(`permute(1,0,2,3).reshape(-1,N,D)`, `.type(FloatTensor)`): the 16-worker NumPy pipeline and the float64 H2D copy
collapse into one host pass that only DRAWS the random numbers -- in the reference's order, from the generator it is
given, so a seed reproduces the reference's views -- and one HIP launch (csrc/views.hip) that gathers, jitters,
mirrors, rotates, casts and writes the view-major (G*B, 512, 4) float32 tensor the grouping op consumes.
"""
import numpy as np
import torch

from . import _lib

NUM_POINT = 512        # cn3D_data_set.py:24
NUM_CROP = 10          # get_data_train(..., num_crop=10)


def draw_clip(rng, points, key_points, res_points_1, res_points_2, base):
    """All random draws of one `__getitem__`, in call order.  `base` = row offsets of the four source clouds inside
    the packed batch buffer.  Returns (idx (10,512) int32 absolute rows, noise (7,512,3) float64, cossin (2,2))."""
    def nonzero_rows(t):                                   # get_temporal_augment_data: rows with a non-zero channel t
        nz = np.flatnonzero(points[:, t] != 0)
        return nz[rng.randint(0, nz.shape[0], 512)]
    t2 = nonzero_rows(4)
    t4 = nonzero_rows(7)
    idx = np.empty((NUM_CROP, NUM_POINT), dtype=np.int64)
    noise = np.empty((7, NUM_POINT, 3), dtype=np.float64)
    cs = np.empty((2, 2), dtype=np.float64)
    P, Kp, R1, R2 = points.shape[0], key_points.shape[0], res_points_1.shape[0], res_points_2.shape[0]
    idx[0] = base[0] + rng.randint(0, P, NUM_POINT)                          # raw_p
    idx[1] = base[0] + rng.randint(0, P, NUM_POINT)                          # rev_p
    noise[0] = rng.randn(1, NUM_POINT, 3)[0]
    noise[1] = rng.randn(1, NUM_POINT, 3)[0]                                 # inside reverse_transform
    idx[2] = base[1] + rng.randint(0, Kp, NUM_POINT)                         # ke1_p
    noise[2] = rng.randn(1, NUM_POINT, 3)[0]
    idx[3] = base[1] + rng.randint(0, Kp, NUM_POINT)                         # ke2_p
    noise[3] = rng.randn(1, NUM_POINT, 3)[0]
    noise[4] = rng.randn(1, NUM_POINT, 3)[0]
    for k, (vi, ni) in enumerate(((4, 5), (5, 6))):                          # ro1_p, ro2_p
        idx[vi] = base[0] + rng.randint(0, P, NUM_POINT)
        noise[ni] = rng.randn(1, NUM_POINT, 3)[0]
        angle = (rng.rand() - 0.5) * np.pi * 0.8
        cs[k] = (np.cos(angle), np.sin(angle))
    idx[6] = base[0] + t2                                                    # ti1_p / ti2_p: no new draws
    idx[7] = base[0] + t4
    idx[8] = base[2] + rng.randint(0, R1, NUM_POINT)                         # rs1_p
    idx[9] = base[3] + rng.randint(0, R2, NUM_POINT)                         # rs2_p
    return idx.astype(np.int32), noise, cs


def build_views(clips, rng=None, device="cuda"):
    """clips: list of (points (P,>=8), key_points, res_points_1, res_points_2) NumPy arrays of one dtype (float32 or
    float64), the arrays `__getitem__` loads for a video.  Returns the (10*B, 512, 4) float32 CUDA tensor = the
    reference's `data1` (view-major rows g*B+b).  `rng`: np.random.RandomState (default: NumPy's global generator,
    like the reference)."""
    rng = np.random if rng is None else rng
    lib = _lib.load_library()
    B = len(clips)
    dt = clips[0][0].dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("source clouds must be float32 or float64")
    rows, off = [], 0
    idxs, noises, css = [], [], []
    for clip in clips:
        if any(a.dtype != dt or a.ndim != 2 or a.shape[1] < 8 for a in clip):
            raise ValueError("every source cloud must be (rows, >=8) of one dtype")
        base = []
        for a in clip:
            base.append(off)
            rows.append(np.ascontiguousarray(a[:, :8]))
            off += a.shape[0]
        i, n, c = draw_clip(rng, clip[0], clip[1], clip[2], clip[3], base)
        idxs.append(i); noises.append(n); css.append(c)
    dev = torch.device(device)
    src = torch.from_numpy(np.concatenate(rows, 0)).to(dev)
    idx = torch.from_numpy(np.stack(idxs)).to(dev)
    noise = torch.from_numpy(np.stack(noises)).to(dev)
    cs = torch.from_numpy(np.stack(css)).to(dev)
    out = _lib.empty((NUM_CROP * B, NUM_POINT, 4), dtype=torch.float32, device=dev)
    _lib.require_cuda(out)
    fn = lib.facl_build_views_f32 if dt == np.float32 else lib.facl_build_views_f64
    _lib.check(fn(_lib.ptr(src), src.shape[0], 8, _lib.ptr(idx), _lib.ptr(noise), _lib.ptr(cs), B, _lib.ptr(out),
                  _lib.stream()), "facl_build_views")
    return out
