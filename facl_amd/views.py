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


def _device_draws(clips, bases, gen, dev):
    """The random draws of a whole batch ON THE DEVICE (torch generator `gen`): the same distributions as draw_clip -- 512
    uniform row draws with replacement per view, rows of non-zero temporal channel for the two temporal views, standard-normal
    jitter, a uniform rotation angle in +-0.4 pi -- but NOT NumPy's stream (a seed does not reproduce the reference's views;
    the throughput mode: the host draws of draw_clip cost ~0.19 ms per clip, 6 ms per batch of 32)."""
    B = len(clips)
    src_of = (0, 0, 1, 1, 0, 0, 2, 3)                               # source cloud of views 0,1,2,3,4,5,8,9
    sizes = torch.tensor([[c[k].shape[0] for k in src_of] for c in clips], dtype=torch.float64).to(dev)
    base = torch.tensor([[bases[b][k] for k in src_of] for b in range(B)], dtype=torch.int64).to(dev)
    u = torch.rand((B, 8, NUM_POINT), generator=gen, device=dev, dtype=torch.float64)
    idx8 = torch.minimum((u * sizes.unsqueeze(-1)).long(), (sizes.long() - 1).unsqueeze(-1)) + base.unsqueeze(-1)
    pmax = max(c[0].shape[0] for c in clips)
    masks = np.zeros((2, B, pmax), dtype=np.float32)
    for b, c in enumerate(clips):
        masks[0, b, :c[0].shape[0]] = c[0][:, 4] != 0
        masks[1, b, :c[0].shape[0]] = c[0][:, 7] != 0
    m = torch.from_numpy(masks).to(dev)
    base0 = torch.tensor([bases[b][0] for b in range(B)], dtype=torch.int64, device=dev).unsqueeze(-1)
    t2 = torch.multinomial(m[0], NUM_POINT, replacement=True, generator=gen) + base0
    t4 = torch.multinomial(m[1], NUM_POINT, replacement=True, generator=gen) + base0
    idx = torch.stack((idx8[:, 0], idx8[:, 1], idx8[:, 2], idx8[:, 3], idx8[:, 4], idx8[:, 5], t2, t4, idx8[:, 6], idx8[:, 7]), dim=1)
    noise = torch.randn((B, 7, NUM_POINT, 3), generator=gen, device=dev, dtype=torch.float64)
    ang = (torch.rand((B, 2), generator=gen, device=dev, dtype=torch.float64) - 0.5) * (np.pi * 0.8)
    cs = torch.stack((torch.cos(ang), torch.sin(ang)), dim=-1)
    return idx.to(torch.int32).contiguous(), noise, cs.contiguous()


def build_views(clips, rng=None, device="cuda", device_rng=None):
    """clips: list of (points (P,>=8), key_points, res_points_1, res_points_2) NumPy arrays of one dtype (float32 or
    float64), the arrays `__getitem__` loads for a video.  Returns the (10*B, 512, 4) float32 CUDA tensor = the
    reference's `data1` (view-major rows g*B+b).  `rng`: np.random.RandomState (default: NumPy's global generator,
    like the reference): every draw happens on the host in the reference's order, a seed reproduces its views.
    `device_rng` (a torch.Generator on the device): draw on the device instead -- same distributions, another stream,
    no per-clip host work."""
    rng = np.random if rng is None else rng
    lib = _lib.load_library()
    B = len(clips)
    dt = clips[0][0].dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("source clouds must be float32 or float64")
    rows, off = [], 0
    idxs, noises, css, bases = [], [], [], []
    for clip in clips:
        if any(a.dtype != dt or a.ndim != 2 or a.shape[1] < 8 for a in clip):
            raise ValueError("every source cloud must be (rows, >=8) of one dtype")
        base = []
        for a in clip:
            base.append(off)
            rows.append(np.ascontiguousarray(a[:, :8]))
            off += a.shape[0]
        bases.append(base)
        if device_rng is None:
            i, n, c = draw_clip(rng, clip[0], clip[1], clip[2], clip[3], base)
            idxs.append(i); noises.append(n); css.append(c)
    dev = torch.device(device)
    src = torch.from_numpy(np.concatenate(rows, 0)).to(dev)
    if device_rng is None:
        idx = torch.from_numpy(np.stack(idxs)).to(dev)
        noise = torch.from_numpy(np.stack(noises)).to(dev)
        cs = torch.from_numpy(np.stack(css)).to(dev)
    else:
        idx, noise, cs = _device_draws(clips, bases, device_rng, dev)
    out = _lib.empty((NUM_CROP * B, NUM_POINT, 4), dtype=torch.float32, device=dev)
    _lib.require_cuda(out)
    fn = lib.facl_build_views_f32 if dt == np.float32 else lib.facl_build_views_f64
    _lib.check(fn(_lib.ptr(src), src.shape[0], 8, _lib.ptr(idx), _lib.ptr(noise), _lib.ptr(cs), B, _lib.ptr(out),
                  _lib.stream()), "facl_build_views")
    return out


def synthetic_raw_clip(seed, dtype=np.float32, P=900, Kp=300, R1=500, R2=200):
    """The four (rows, 8) source clouds `__getitem__` loads for one video (cn3D_data_set.py:105-116), synthetic: uniform
    coordinates / channels in [-0.5, 0.5) with zeros sprinkled into the two temporal channels (the rows the temporal views
    must skip).  `--synthetic 2` of the training entries feeds these through build_views, so the loop body runs from the
    loader's output on (cn3d_train_motion_GL.py:224-228)."""
    r = np.random.RandomState(seed)
    pts = (r.rand(P, 8) - 0.5).astype(dtype)
    pts[::3, 4] = 0
    pts[1::4, 7] = 0
    return pts, (r.rand(Kp, 8) - 0.5).astype(dtype), (r.rand(R1, 8) - 0.5).astype(dtype), (r.rand(R2, 8) - 0.5).astype(dtype)
