"""Grouping ops -- drop-in for the reference's ``training_code/utils_my.py`` grouping functions.

Same names, arguments, return shapes/strides and ``opt`` side effects as the reference
(utils_my.py:7-42, :217-253, :255-291, :293-328); the body is one fused HIP kernel
(facl_group) instead of the expand/sub/mul/sum/topk/masked-assign/gather chain.
"""
import numpy as np
import torch

from . import _lib


def knn_radius_group(points, sample_num_level1, knn_K, ball_radius, want_idx=False):
    """points (M,N,D) float32 CUDA -> (inputs_level1 (M,D,S,K) view, center (M,3,S,1) view[, idx]).

    The returned tensors are transposed VIEWS of contiguous (M,S,K,D) / (M,S,3) buffers, exactly
    like the reference's (utils_my.py:283-284).  ``points`` is not modified.
    A 4-D ``points`` (B,G,N,D) is the loader's clip-major batch: the outputs are those of
    ``points.permute(1,0,2,3).reshape(G*B,N,D)`` (cn3d_train_motion_GL.py:226) without materialising that copy."""
    _lib.require_cuda(points)
    if points.dtype != torch.float32:
        raise TypeError("points must be float32 (the reference casts with .type(torch.FloatTensor))")
    pts = points.contiguous()
    if pts.dim() == 4:
        return _knn_radius_group_clips(pts, sample_num_level1, knn_K, ball_radius, want_idx)
    M, N, D = pts.shape
    S, K = int(sample_num_level1), int(knn_K)
    xt = _lib.empty((M, S, K, D), dtype=torch.float32, device=pts.device)
    yt = _lib.empty((M, S, 3), dtype=torch.float32, device=pts.device)
    idx = _lib.empty((M, S, K), dtype=torch.int32, device=pts.device) if want_idx else None
    lib = _lib.load_library()
    with _lib.timed("facl_group"):
        _lib.check(lib.facl_group(_lib.ptr(pts), M, N, D, S, K, float(ball_radius), _lib.ptr(idx), _lib.ptr(xt),
                                  _lib.ptr(yt), _lib.stream()), "facl_group")
    inputs_level1 = xt.permute(0, 3, 1, 2)                       # (M,D,S,K), utils_my.py:283
    inputs_level1_center = yt.view(M, 1, S, 3).transpose(1, 3)   # (M,3,S,1), utils_my.py:284
    if want_idx:
        return inputs_level1, inputs_level1_center, idx
    return inputs_level1, inputs_level1_center


def _knn_radius_group_clips(clips, sample_num_level1, knn_K, ball_radius, want_idx):
    B, G, N, D = clips.shape
    M, S, K = B * G, int(sample_num_level1), int(knn_K)
    xt = _lib.empty((M, S, K, D), dtype=torch.float32, device=clips.device)
    yt = _lib.empty((M, S, 3), dtype=torch.float32, device=clips.device)
    idx = _lib.empty((M, S, K), dtype=torch.int32, device=clips.device) if want_idx else None
    lib = _lib.load_library()
    with _lib.timed("facl_group"):
        _lib.check(lib.facl_group_clips(_lib.ptr(clips), B, G, N, D, S, K, float(ball_radius), _lib.ptr(idx), _lib.ptr(xt),
                                        _lib.ptr(yt), _lib.stream()), "facl_group_clips")
    out = (xt.permute(0, 3, 1, 2), yt.view(M, 1, S, 3).transpose(1, 3))
    return out + (idx,) if want_idx else out


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


# ---- contrastive losses (utils_my.py:53-116 = cn3d_train_motion_GL.py:265-316) -------------------
def _masked_sim(anchors, keys, anchor_clip, key_clip):
    """l_neg = (anchors @ keys^T) * mask, mask = 0 where the key belongs to the anchor's own clip
    (utils_my.py:55-56,71-72).  Same-clip columns are multiplied by 0 -- NOT removed -- so each of them
    still contributes exp(0) = 1 to the softmax denominator, exactly like the reference."""
    sim = anchors @ keys.t()
    same = anchor_clip[:, None] == key_clip[None, :]
    return torch.where(same, torch.zeros((), dtype=sim.dtype, device=sim.device), sim)


def _clip_ids(G, B, device, offset=0):
    return (torch.arange(B, device=device) + offset).repeat(G)     # row g*B+b -> clip id b


def global_contrast(num_crop, x_global, x, opt, criterion=None, x_keys=None, clip_offset=0):
    """utils_my.py:53-83.  loss_c = sum_g CE([<xg_n, x_{gB+n}> | (xg @ x^T)*mask], 0), CE = mean over B.
    The (G,B,1+GB) logits tensor and its ``repeat`` are never built: every g shares the negatives, so
    CE_g[n] = logaddexp(pos[g,n], LSE_n) - pos[g,n].

    Data-parallel form: ``x_keys`` = the all-gathered view-major embeddings (G*B_global rows) and
    ``clip_offset`` = rank*B_local; anchors stay local (mean over local B, gradients averaged by DDP)."""
    B, G = x_global.shape[0], num_crop
    keys = x if x_keys is None else x_keys
    Bk = keys.shape[0] // G
    a_clip = torch.arange(B, device=x.device) + clip_offset
    neg = _masked_sim(x_global, keys, a_clip, _clip_ids(G, Bk, x.device))      # (B, G*Bk)
    lse = torch.logsumexp(neg, dim=1)
    pos = (x_global.unsqueeze(0) * x.view(G, B, -1)).sum(-1)                   # (G,B)
    return (torch.logaddexp(pos, lse.unsqueeze(0)) - pos).mean(dim=1).sum()


def circle_contrast(num_crop, x, batchSize, criterion=None, order=None, x_keys=None, clip_offset=0):
    """utils_my.py:85-116.  ``order`` replaces the reference's np.random.shuffle(arange(num_crop)) (:96-97);
    when omitted it is drawn from NumPy's global RNG like the reference."""
    import numpy as np
    G, B = num_crop, batchSize
    if order is None:
        order = np.arange(0, G, 1)
        np.random.shuffle(order)
    order = torch.as_tensor(np.asarray(order), device=x.device, dtype=torch.long)
    keys = x if x_keys is None else x_keys
    Bk = keys.shape[0] // G
    xv = x.view(G, B, -1)
    anchors = xv[order[:-1]]                                                   # (G-1,B,C)
    pos = (anchors * xv[order[1:]]).sum(-1)                                    # (G-1,B)   :100
    a_clip = (torch.arange(B, device=x.device) + clip_offset).repeat(G - 1)
    neg = _masked_sim(anchors.reshape((G - 1) * B, -1), keys, a_clip, _clip_ids(G, Bk, x.device))
    neg = neg.view(G - 1, B, G * Bk).permute(1, 0, 2).reshape(B, -1)           # all anchors' negatives, shared (:105-109)
    lse = torch.logsumexp(neg, dim=1)
    return (torch.logaddexp(pos, lse.unsqueeze(0)) - pos).mean(dim=1).sum()


# ---- fused HIP path for both losses (csrc/loss.hip): 2 library GEMMs + one kernel per loss ------------
_POSCOL_CACHE = {}


def _positive_columns(G, B, Bk, clip_offset, dev):
    """(n_idx (B,), pos_g (G*B,)) int32: this rank's clip columns and the positive column of every (view, clip) slot of
    the global loss.  They only depend on the shapes, so they are built once (6 tiny launches per step otherwise)."""
    key = (G, B, Bk, clip_offset, dev.type, dev.index)
    if key not in _POSCOL_CACHE:
        n_idx = torch.arange(B, device=dev, dtype=torch.int32) + clip_offset
        g_idx = torch.arange(G, device=dev, dtype=torch.int32)
        _POSCOL_CACHE[key] = (n_idx, (g_idx[:, None] * Bk + n_idx[None, :]).reshape(-1).contiguous())
    return _POSCOL_CACHE[key]


class _ContrastiveLosses(torch.autograd.Function):
    """(loss_c, loss_circle) with the value and d/dsim computed by facl_contrast; the similarity GEMMs and
    their transposes in the backward are plain library GEMMs."""

    @staticmethod
    def forward(ctx, x_global, x, x_keys, order, G, clip_offset):
        from .sa_mlp import _Workspace
        lib = _lib.load_library()
        _lib.require_cuda(x, x_global, x_keys)
        dev = x.device
        ws = _Workspace.get(dev)
        B, C = x_global.shape
        Bk = x_keys.shape[0] // G
        J = G * Bk
        xg = x_global.contiguous()
        keys = x_keys.contiguous()
        xv = x.view(G, B, C)
        anchors = xv[order[:-1]].reshape((G - 1) * B, C).contiguous()              # circle anchors (:100,:103)
        n_idx, pos_g = _positive_columns(G, B, Bk, clip_offset, dev)                 # global: sim_g is (B,J); slots = views
        pos_c = (order[1:].to(torch.int32)[:, None] * Bk + n_idx[None, :]).reshape(-1).contiguous()
        sim_c = anchors @ keys.t()                                                 # ((G-1)B, J)  :103
        sim_g = xg @ keys.t()                                                      # (B, J)       :71
        dsim_g = _lib.empty_like(sim_g)
        dsim_c = _lib.empty_like(sim_c)
        out = _lib.empty(2, dtype=torch.float64, device=dev)
        st = _lib.stream()
        _lib.check(lib.facl_contrast(_lib.ptr(sim_g), B, J, B, Bk, 1, G, 0, _lib.ptr(pos_g), clip_offset,
                                     _lib.ptr(dsim_g), out[0:1].data_ptr(), _lib.ptr(ws), st), "facl_contrast(global)")
        _lib.check(lib.facl_contrast(_lib.ptr(sim_c), (G - 1) * B, J, B, Bk, G - 1, G - 1, 1, _lib.ptr(pos_c), clip_offset,
                                     _lib.ptr(dsim_c), out[1:2].data_ptr(), _lib.ptr(ws), st), "facl_contrast(circle)")
        ctx.save_for_backward(xg, keys, anchors, dsim_g, dsim_c, order)
        ctx.dims = (G, B, C, J)
        return out[0].float(), out[1].float()

    @staticmethod
    def backward(ctx, g_c, g_o):
        xg, keys, anchors, dsim_g, dsim_c, order = ctx.saved_tensors
        G, B, C, J = ctx.dims
        dg = dsim_g * g_c
        dc = dsim_c * g_o
        d_xg = dg @ keys
        d_keys = dg.t() @ xg + dc.t() @ anchors
        d_anchors = dc @ keys
        d_x = torch.zeros(G, B, C, dtype=keys.dtype, device=keys.device)
        d_x.index_add_(0, order[:-1], d_anchors.view(G - 1, B, C))
        return d_xg, d_x.view(G * B, C), d_keys, None, None, None


class _ContrastivePair(torch.autograd.Function):
    """(loss_c, loss_circle) from the stacked embeddings [x ; x_global] ((G+1)*B rows): ONE similarity GEMM
    sim = stacked @ keys^T on the hand-written MFMA GEMM, ONE loss launch (facl_contrast_pair: both values and
    d/dsim, the circle anchors addressed through ``order`` instead of gathered), and in the backward one dgrad
    (d stacked = dsim @ keys) plus one wgrad (d keys = dsim^T @ stacked).  No library GEMM, no anchors gather, no
    positive-column index tensors, no index_add (utils_my.py:63-71,100-103 and their autograd)."""

    @staticmethod
    def forward(ctx, stacked, keys, order, G, clip_offset):
        from . import tail as _tail
        from .sa_mlp import _Workspace
        lib = _lib.load_library()
        _lib.require_cuda(stacked, keys)
        dev = stacked.device
        ws = _Workspace.get(dev)
        stacked = stacked.contiguous()
        R, C = stacked.shape
        B = R // (G + 1)
        ctx.own_keys = keys is None                # single process: the keys ARE the view rows of `stacked`
        keys = stacked[:G * B] if keys is None else keys.contiguous()
        J = keys.shape[0]
        Bk = J // G
        ctx.mfma = (J % 4 == 0 and C % 4 == 0)     # the MFMA GEMMs contract over multiples of 4; odd toy shapes: library GEMM
        ctx.prec = _tail.current_precision()       # the backward GEMMs run in the forward's arithmetic
        sim = _tail.gemm_fwd(stacked, keys, None, prec=ctx.prec)[0] if ctx.mfma else stacked @ keys.t()     # ((G+1)B, J)  :71 and :103
        dsim = _lib.empty_like(sim)
        out = _lib.empty(2, dtype=torch.float64, device=dev)
        # [loss_c, loss_circle, loss_circle + loss_c] in fp32 from the loss launch's own finishing kernel (no cast / add launches)
        out32 = _lib.empty(3, dtype=torch.float32, device=dev)
        _lib.check(lib.facl_contrast_pair_sum(_lib.ptr(sim), G, B, Bk, J, _lib.ptr(order), clip_offset, _lib.ptr(dsim),
                                              _lib.ptr(out), _lib.ptr(out32), _lib.ptr(ws), _lib.stream()), "facl_contrast_pair_sum")
        ctx.save_for_backward(stacked, keys, dsim)
        ctx.GB = G * B
        ctx.set_materialize_grads(False)           # an unused loss output arrives as None, not as a freshly filled zero tensor
        return out32[0], out32[1], out32[2]

    @staticmethod
    def backward(ctx, g_c, g_o, g_s):
        from . import tail as _tail
        bp = _tail.backward_precision(ctx.prec)
        stacked, keys, dsim = ctx.saved_tensors
        # upstream gradients of (loss_c, loss_circle, their sum): the training step only uses the sum (no launch here); any
        # other combination is tensor algebra on device scalars
        def tot(g):
            if g is None:
                return g_s
            g = g.contiguous().float()
            return g if g_s is None else g + g_s.float()
        g_c, g_o = tot(g_c), tot(g_o)
        if g_c is None and g_o is None:
            return None, None, None, None, None
        zero = None
        if g_c is None or g_o is None:
            zero = torch.zeros((), dtype=torch.float32, device=dsim.device)
        g_c = zero if g_c is None else g_c.contiguous().float()
        g_o = zero if g_o is None else g_o.contiguous().float()
        # rows [0, G*B) carry the circle loss, rows [G*B, (G+1)*B) the global loss: one scaling launch for both
        lib = _lib.load_library()
        ds = _lib.empty_like(dsim)
        R, J = dsim.shape
        _lib.check(lib.facl_scale_rows2(_lib.ptr(dsim), _lib.ptr(ds), ctx.GB, R, J, _lib.ptr(g_o), _lib.ptr(g_c), _lib.stream()),
                   "facl_scale_rows2")
        d_stacked = _tail.gemm_dgrad(ds, keys, prec=bp) if ctx.mfma else ds @ keys
        if ctx.own_keys and ctx.mfma:
            # d keys = dsim^T @ stacked accumulated straight into the view rows of d stacked (the keys ARE those rows)
            M_, N_, K_ = ds.shape[0], ds.shape[1], stacked.shape[1]
            pc = {"f32": 0, "x3b": 0, "f16": 1, "x3": 2}[bp]
            with _lib.timed("facl_gemm_wgrad %dx%dx%d%s" % (M_, N_, K_, _tail._LABEL[bp])):
                rc = lib.facl_gemm_wgrad_acc(_lib.ptr(ds), _lib.ptr(stacked), M_, N_, K_, stacked.stride(0), _lib.ptr(d_stacked), pc,
                                             _lib.stream())
            if rc == 0:
                return d_stacked, None, None, None, None
            if rc != -4:
                _lib.check(rc, "facl_gemm_wgrad_acc")
        d_keys = _tail.gemm_wgrad(ds, stacked, prec=bp) if ctx.mfma else ds.t() @ stacked
        if ctx.own_keys:
            d_stacked[:ctx.GB] += d_keys
            d_keys = None
        return d_stacked, d_keys, None, None, None


def _check_order(order, G):
    """A host-side `order` (list / ndarray / CPU tensor) must be a permutation of range(G) (np.random.shuffle of
    arange(G), cn3d_train_motion_GL.py:297-298).  A device tensor is not read back (no sync on the step); the kernel
    clamps its entries so that a corrupt one cannot address outside the similarity matrix."""
    o = np.asarray(order.cpu() if torch.is_tensor(order) else order).reshape(-1)
    if o.shape[0] != G or not np.array_equal(np.sort(o), np.arange(G)):
        raise ValueError("order must be a permutation of range(%d), got %r" % (G, o.tolist()))


def contrastive_losses_stacked(num_crop, stacked, order, x_keys=None, clip_offset=0, with_sum=False):
    """(loss_c, loss_circle) from the model's stacked output [x ; x_global] (facl_amd.cn3d_model_conbag: ``_stacked``);
    with_sum: also `loss_circle + loss_c` (fp32, cn3d_train_motion_GL.py:329) as a third output of the same launch."""
    G = num_crop
    if not (torch.is_tensor(order) and order.device == stacked.device and order.dtype == torch.long):
        _check_order(order, G)
        order = torch.as_tensor(order, device=stacked.device, dtype=torch.long)
    out = _ContrastivePair.apply(stacked, x_keys, order.contiguous(), G, clip_offset)
    return out if with_sum else out[:2]


def contrastive_losses(num_crop, x_global, x, order, x_keys=None, clip_offset=0):
    """(loss_c, loss_circle) = (global_contrast(...), circle_contrast(...)) through the HIP loss kernel."""
    keys = x if x_keys is None else x_keys
    if not (torch.is_tensor(order) and order.device == x.device and order.dtype == torch.long):
        order = torch.as_tensor(order, device=x.device, dtype=torch.long)
    if keys is x:
        keys = x.view_as(x)            # distinct autograd input so that d_keys and d_x accumulate separately
    return _ContrastiveLosses.apply(x_global, x, keys, order, num_crop, clip_offset)
