"""Grouping ops -- drop-in for the reference's ``training_code/utils_my.py`` grouping functions.

Same names, arguments, return shapes/strides and ``opt`` side effects as the reference
(utils_my.py:7-42, :217-253, :255-291, :293-328); the body is one fused HIP kernel
(facl_group) instead of the expand/sub/mul/sum/topk/masked-assign/gather chain.
"""
import torch

from . import _lib


def knn_radius_group(points, sample_num_level1, knn_K, ball_radius, want_idx=False):
    """points (M,N,D) float32 CUDA -> (inputs_level1 (M,D,S,K) view, center (M,3,S,1) view[, idx]).

    The returned tensors are transposed VIEWS of contiguous (M,S,K,D) / (M,S,3) buffers, exactly
    like the reference's (utils_my.py:283-284).  ``points`` is not modified."""
    _lib.require_cuda(points)
    if points.dtype != torch.float32:
        raise TypeError("points must be float32 (the reference casts with .type(torch.FloatTensor))")
    pts = points.contiguous()
    M, N, D = pts.shape
    S, K = int(sample_num_level1), int(knn_K)
    xt = torch.empty((M, S, K, D), dtype=torch.float32, device=pts.device)
    yt = torch.empty((M, S, 3), dtype=torch.float32, device=pts.device)
    idx = torch.empty((M, S, K), dtype=torch.int32, device=pts.device) if want_idx else None
    lib = _lib.load_library()
    _lib.check(lib.facl_group(_lib.ptr(pts), M, N, D, S, K, float(ball_radius), _lib.ptr(idx), _lib.ptr(xt),
                              _lib.ptr(yt), _lib.stream()), "facl_group")
    inputs_level1 = xt.permute(0, 3, 1, 2)                       # (M,D,S,K), utils_my.py:283
    inputs_level1_center = yt.view(M, 1, S, 3).transpose(1, 3)   # (M,3,S,1), utils_my.py:284
    if want_idx:
        return inputs_level1, inputs_level1_center, idx
    return inputs_level1, inputs_level1_center


def group_points_3DV(points, opt):
    """utils_my.py:255-291.  Like the reference it overwrites ``opt.INPUT_FEATURE_NUM`` (from
    the data), ``opt.knn_K = 64`` and ``opt.ball_radius = 0.06`` (:259-261)."""
    cur_train_size = points.shape[0]
    opt.INPUT_FEATURE_NUM = points.shape[-1]
    opt.knn_K = 64
    opt.ball_radius = 0.06
    points = points.view(cur_train_size, opt.SAMPLE_NUM, -1)
    return knn_radius_group(points, opt.sample_num_level1, opt.knn_K, opt.ball_radius)


def group_points_3DV_2048(points, knn_K, sample_num_level1, SAMPLE_NUM=2048):
    """utils_my.py:7-42: N-parametrised twin, r^2 = 0.16 (:13)."""
    cur_train_size = points.shape[0]
    points = points.view(cur_train_size, -1, points.shape[-1])
    if points.shape[1] != SAMPLE_NUM:
        raise RuntimeError("points has %d rows per cloud, SAMPLE_NUM=%d" % (points.shape[1], SAMPLE_NUM))
    return knn_radius_group(points, sample_num_level1, knn_K, 0.16)


def group_points_3DV_nums(points, opt, sample_num_level1, knn_K):
    """utils_my.py:293-328: explicit S and K, r^2 = 0.06 (:299)."""
    cur_train_size = points.shape[0]
    opt.INPUT_FEATURE_NUM = points.shape[-1]
    opt.ball_radius = 0.06
    points = points.view(cur_train_size, opt.SAMPLE_NUM, -1)
    return knn_radius_group(points, sample_num_level1, knn_K, opt.ball_radius)


def group_points(points, opt):
    """utils_my.py:217-253: K = 64, r^2 = 0.14 (:220-221)."""
    opt.knn_K = 64
    opt.ball_radius = 0.14
    cur_train_size = points.shape[0]
    opt.INPUT_FEATURE_NUM = points.shape[-1]
    points = points.view(cur_train_size, opt.SAMPLE_NUM, -1)
    return knn_radius_group(points, opt.sample_num_level1, opt.knn_K, opt.ball_radius)
