"""Oracle: kNN-then-radius grouping (NumPy, fp32).

Restates /root/reference/training_code/utils_my.py:255-291 (``group_points_3DV``), its
N-parametrised twin :7-42 (``group_points_3DV_2048``) and :293-328 / :217-253, which differ
only in where K / r^2 / N / S come from.
"""
import numpy as np


def knn_radius_indices(points, S, K, r2):
    """points (M,N,D) float32 -> (idx int32 (M,S,K) ascending along K, dist2 float32 (M,S,N)).

    Centroids are rows 0..S-1 (utils_my.py:266).  dist^2 = (dx*dx + dy*dy) + dz*dz in fp32
    (:265-268, ``sum(2)`` of a 3-wide axis).  The K smallest are kept (:269; torch.topk with
    sorted=False leaves their order unspecified -> the oracle canonicalises to ascending
    index, ties broken toward the lower index); any kept neighbour with dist^2 > r2
    (strict, :272) has its index replaced by the centroid's own row jj (:274-275)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    M, N, D = points.shape
    xyz = points[:, :, 0:3]
    cen = xyz[:, 0:S, :]
    d = xyz[:, None, :, :] - cen[:, :, None, :]                 # (M,S,N,3) p - c
    d = d * d
    dist2 = (d[..., 0] + d[..., 1]) + d[..., 2]                 # fp32, (M,S,N)
    order = np.argsort(dist2, axis=2, kind="stable")[:, :, :K]  # K smallest, lower idx on ties
    order = np.sort(order, axis=2)
    kept = np.take_along_axis(dist2, order, axis=2)
    jj = np.arange(S, dtype=np.int64)[None, :, None]
    idx = np.where(kept > np.float32(r2), jj, order)
    return idx.astype(np.int32), dist2


def gather_center(points, idx):
    """idx (M,S,K) -> xt (M,S,K,D) with xyz centred on the centroid, yt (M,S,3).
    utils_my.py:277-284.  The reference returns transposed VIEWS of exactly these buffers:
    inputs_level1 = xt.permute(0,3,1,2)  (M,D,S,K);  center = yt.view(M,1,S,3).transpose(1,3)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    M, N, D = points.shape
    S, K = idx.shape[1], idx.shape[2]
    flat = idx.reshape(M, S * K).astype(np.int64)
    xt = np.take_along_axis(points, flat[:, :, None], axis=1).reshape(M, S, K, D).copy()
    yt = points[:, 0:S, 0:3].copy()
    xt[..., 0:3] = xt[..., 0:3] - yt[:, :, None, :]
    return xt, yt


def group_points(points, S, K, r2):
    idx, _ = knn_radius_indices(points, S, K, r2)
    xt, yt = gather_center(points, idx)
    return idx, xt, yt
