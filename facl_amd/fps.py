"""Farthest point sampling on the GPU -- mirrors the reference's NumPy helpers
``farthest_point_sampling_fast`` / ``fps_sample_data`` (cn3d_data_load.py:287-320,
cn3D_data_set.py:665-694) with the random start index made explicit."""
import torch

from . import _lib


def farthest_point_sampling_batch(xyz, sample_num, start_idx):
    """xyz (M,N,>=3) float32|float64 CUDA tensor, start_idx (M,) int -> (M,sample_num) int32.
    np.argmax tie-break (lowest index), dist^2 = (dx*dx+dy*dy)+dz*dz in xyz's dtype."""
    _lib.require_cuda(xyz)
    if xyz.dim() != 3 or xyz.shape[-1] < 3:
        raise ValueError("xyz must be (M,N,>=3)")
    xyz = xyz.contiguous()
    M, N, ld = xyz.shape
    start = torch.as_tensor(start_idx, device=xyz.device).to(torch.int32).contiguous()
    if start.numel() != M:
        raise ValueError("start_idx must have one entry per cloud")
    out = _lib.empty((M, sample_num), dtype=torch.int32, device=xyz.device)
    lib = _lib.load_library()
    fn = {torch.float32: lib.facl_fps_f32, torch.float64: lib.facl_fps_f64}.get(xyz.dtype)
    if fn is None:
        raise TypeError("xyz must be float32 or float64")
    with _lib.timed("facl_fps"):
        _lib.check(fn(_lib.ptr(xyz), M, N, ld, sample_num, _lib.ptr(start), _lib.ptr(out), _lib.stream()), "facl_fps")
    return out


def farthest_point_sampling_fast(pc, sample_num, start_idx=None):
    """Reference signature (cn3d_data_load.py:301): pc (N,3) -> (sample_num,1) int32.
    ``start_idx`` replaces the reference's ``np.random.randint(0, pc_num)`` draw; when omitted it
    is drawn from torch's RNG."""
    if start_idx is None:
        start_idx = int(torch.randint(0, pc.shape[0], (1,)).item())
    return farthest_point_sampling_batch(pc.unsqueeze(0), sample_num, [start_idx]).view(sample_num, 1)


def _reorder(pts, picks, m):
    out = _lib.empty_like(pts)
    b, N, D = pts.shape
    lib = _lib.load_library()
    _lib.check(lib.facl_fps_reorder(_lib.ptr(pts), b, N, D, _lib.ptr(picks), m, _lib.ptr(out), _lib.stream()),
               "facl_fps_reorder")
    return out


def fps_sample_data(points_xyzc, sample_num_level1, sample_num_level2=None, start_idx=None, start_idx2=None, xyz=None):
    """Reorder every cloud so its FPS picks come first, the remaining rows following in ascending order
    (np.setdiff1d).  points (b,N,D) CUDA -> new float32 tensor (the reference permutes in place).

    ``sample_num_level2=None``: the 1-level form of cn3D_data_set.py:665-672 (which receives a second count and ignores
    it).  With ``sample_num_level2`` the 2-level form of cn3d_data_load.py:287-298: the first ``sample_num_level1`` rows
    are FPS-reordered AGAIN among themselves so that their first ``sample_num_level2`` rows are an FPS subset.
    ``start_idx`` / ``start_idx2`` (b,) replace the reference's np.random.randint draws (torch's RNG when omitted).
    ``xyz``: optional (b,N,3) coordinates in the precision the sampling should run in (the reference's clouds are
    float64; FPS in float32 picks the same rows on tie-free data)."""
    _lib.require_cuda(points_xyzc)
    pts = points_xyzc.contiguous().float()
    b, N, D = pts.shape
    S1 = int(sample_num_level1)
    if start_idx is None:
        start_idx = torch.randint(0, N, (b,))
    src = pts if xyz is None else xyz.contiguous()
    picks = farthest_point_sampling_batch(src, S1, start_idx)
    out = _reorder(pts, picks, S1)
    if sample_num_level2 is None:
        return out
    S2 = int(sample_num_level2)
    if start_idx2 is None:
        start_idx2 = torch.randint(0, S1, (b,))
    if xyz is not None:                                   # the level-1 rows in the sampling precision, same permutation
        src = torch.gather(src, 1, picks.long().unsqueeze(-1).expand(b, S1, src.shape[-1]))
    else:
        src = out[:, :S1]
    picks2 = farthest_point_sampling_batch(src, S2, start_idx2)
    out[:, :S1] = _reorder(out[:, :S1].contiguous(), picks2, S2)
    return out
