"""Dense configuration (BASELINE.json configs[4]: N=4096 points, T=32 views, 3-level set abstraction).

What the reference holds for it: the second-level groupers ``group_points_2`` / ``group_points_2_3DV``
(training_code/utils_my.py:332-381) and the second-level widths ``nstates_plus_2 = [128, 128, 256]``
(cn3d_model_conbag.py:16); no model of the reference consumes them.  This module provides

  * drop-ins for the two groupers (same names, arguments, return shapes; HIP kernels: facl_group on the level-1 centroid
    coordinates + facl_gather_rows for the features) -- pinned against the reference by tests/golden/level2.npz;
  * ``PointNet_Plus_dense``: the live encoder with a second set-abstraction level inserted (SURVEY 8d "C5", a
    build-side composition: parity is against oracle/dense.py only):
        level 1  kNN/radius grouping of the N points around the first S1 rows, net3DV_1 (D->64->64->256) + max over K1
        level 2  group_points_2 semantics on cat(centre_1, feat_1) around the first S2 level-1 centroids,
                 net3DV_2 (259->128->128->256) + max over K2
        level 3  net3DV_3 (259->256->512->1024) on cat(centre_2, feat_2), max over S2 / over gost*S2, netR_FC twice,
                 normalize, mapping -- exactly cn3d_model_conbag.py:61-88,:218-232 with level-2 inputs.
    Level 1 runs on the fused set-abstraction kernels (facl_amd.sa_mlp), levels 2-3 on the row GEMMs of
    facl_amd.tail with the grouped rows as GEMM rows (the max over the K2 = 64 neighbours is the GEMM's fused segment
    max); ``precision="f16"`` switches the level-1 64->256 layer and the level-2/3 GEMMs to fp16-input MFMA with fp32
    accumulation;
  * ``DenseStep`` (grouping -> forward -> global + circle loss -> backward -> optimizer) and the bench hook.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import dist as fdist
from . import sa_mlp
from . import tail as _tail
from .cn3d_model_conbag import _Affine, _BatchNormState, _Slots, _conv_stack, nstates_plus_1, nstates_plus_2, nstates_plus_3
from .utils_my import contrastive_losses_stacked, knn_radius_group


# ---- second-level grouping --------------------------------------------------------------------------------------------
class _GatherRows(torch.autograd.Function):
    """rows[(m,s2,k)] = feat[m][idx[m,s2,k]]  (feat (M*S1, C) row-major); backward = deterministic scatter-add."""

    @staticmethod
    def forward(ctx, feat, idx, S1):
        lib = _lib.load_library()
        _lib.require_cuda(feat, idx)
        feat = feat.contiguous()
        M, S2, K = idx.shape
        C = feat.shape[1]
        out = _lib.empty((M * S2 * K, C), dtype=torch.float32, device=feat.device)
        _lib.check(lib.facl_gather_rows(_lib.ptr(feat), C, M, S1, C, _lib.ptr(idx), S2 * K, None, _lib.ptr(out), C, 0,
                                        _lib.stream()), "facl_gather_rows")
        ctx.save_for_backward(idx)
        ctx.dims = (M, S1, C, S2 * K)
        return out

    @staticmethod
    def backward(ctx, drows):
        lib = _lib.load_library()
        idx, = ctx.saved_tensors
        M, S1, C, rpc = ctx.dims
        drows = drows.contiguous()
        dfeat = _lib.empty((M * S1, C), dtype=torch.float32, device=drows.device)
        _lib.check(lib.facl_scatter_rows(_lib.ptr(drows), C, 0, M, S1, C, _lib.ptr(idx), rpc, _lib.ptr(dfeat), _lib.stream()),
                   "facl_scatter_rows")
        return dfeat, None, None


def group_level2_rows(xyz_rows, feat_rows, S2, K, r2):
    """Row-major fast path of the second-level grouper.  xyz_rows (M,S1,3), feat_rows (M*S1,C) ->
    (feat rows (M*S2*K, C) [differentiable], centred xyz rows (M*S2*K,3), centres (M*S2,3), idx (M,S2,K) int32)."""
    M, S1, _ = xyz_rows.shape
    xt, yt, idx = knn_radius_group(xyz_rows, S2, K, r2, want_idx=True)             # same select / tie rule as level 1
    xyz_g = xt.permute(0, 2, 3, 1).reshape(M * S2 * K, 3)                          # the contiguous (M,S2,K,3) buffer
    centres = yt.permute(0, 2, 1, 3).reshape(M * S2, 3)
    return _GatherRows.apply(feat_rows, idx, S1), xyz_g, centres, idx


def _group_points_2(points, sample_num_level2, K, r2):
    _lib.require_cuda(points)
    if points.dim() != 3 or points.shape[1] < 3:
        raise ValueError("points must be (B, 3+C, S1) channel-first")
    lib = _lib.load_library()
    B, C3, S1 = points.shape
    S2 = int(sample_num_level2)
    rows = points.detach().float().transpose(1, 2).contiguous()                    # (B,S1,3+C); a no-copy if it was a view of this
    xyz = rows[:, :, :3].contiguous()
    xt, yt, idx = knn_radius_group(xyz, S2, K, float(r2), want_idx=True)
    out = _lib.empty((B, S2, K, C3), dtype=torch.float32, device=points.device)
    xyz_g = xt.permute(0, 2, 3, 1).contiguous()
    if C3 > 3:
        _lib.check(lib.facl_gather_rows(rows[:, :, 3:].data_ptr(), C3, B, S1, C3 - 3, _lib.ptr(idx), S2 * K, _lib.ptr(xyz_g),
                                        _lib.ptr(out), C3, 3, _lib.stream()), "facl_gather_rows")
    else:
        out.copy_(xyz_g)
    return out.permute(0, 3, 1, 2), points[:, 0:3, 0:S2].unsqueeze(3)              # (B,3+C,S2,K) view, (B,3,S2,1): :352


def group_points_2(points, sample_num_level1, sample_num_level2, knn_K, ball_radius):
    """utils_my.py:332-356.  points (B,3+C,S1) channel-first.  Like the reference, ``knn_K`` is overridden with the
    literal 64 (:335) and ``ball_radius`` (a tensor there) is compared with the SQUARED distance (:344)."""
    return _group_points_2(points, sample_num_level2, 64, float(ball_radius))


def group_points_2_3DV(points, sample_num_level1, sample_num_level2, knn_K=None, ball_radius=None):
    """utils_my.py:358-381: K = 32 and r^2 = 0.11 literals (:361-362)."""
    return _group_points_2(points, sample_num_level2, 32, 0.11)


# ---- model ---------------------------------------------------------------------------------------------------------------
class PointNet_Plus_dense(nn.Module):
    """3-level set-abstraction encoder (see the module docstring).  Same constructor conventions, state_dict naming
    (net3DV_1 / net3DV_2 / net3DV_3 / netR_FC / mapping) and 4-output forward contract as PointNet_Plus; ``forward``
    takes the view-major clouds (M,N,D) -- or the loader's clip-major (B,G,N,D) batch -- because the second grouping
    level sits between network layers."""

    def __init__(self, opt, num_clusters=64, gost=10, dim=512, S1=512, K1=64, S2=128, K2=64, r1=0.16, r2=0.25,
                 precision="f32"):
        super().__init__()
        self.INPUT_FEATURE_NUM = opt.INPUT_FEATURE_NUM
        self.gost, self.dim, self.num_clusters = gost, dim, num_clusters
        self.S1, self.K1, self.S2, self.K2, self.r1, self.r2 = S1, K1, S2, K2, r1, r2
        self.precision = precision
        self.net3DV_1 = _conv_stack(self.INPUT_FEATURE_NUM, nstates_plus_1)
        self.net3DV_2 = _conv_stack(3 + nstates_plus_1[2], nstates_plus_2)
        self.net3DV_3 = _conv_stack(3 + nstates_plus_2[2], nstates_plus_3[:3])
        self.netR_FC = _Slots([(0, _Affine((nstates_plus_3[4], 1024), 1024)), (1, _BatchNormState(nstates_plus_3[4])),
                               (3, _Affine((dim, nstates_plus_3[4]), nstates_plus_3[4]))])
        self.mapping = _Affine((num_clusters, dim), dim, bias=False)
        self.bn_reduce_fn = None

    def forward(self, points, loss_mode=0):
        training, red = self.training, self.bn_reduce_fn
        if points.dim() == 4:
            B_, G_, N, D = points.shape
            M = B_ * G_
        else:
            M, N, D = points.shape
        if D != self.INPUT_FEATURE_NUM:
            raise RuntimeError("input has %d channels, model was built for %d" % (D, self.INPUT_FEATURE_NUM))
        if M % self.gost:
            raise RuntimeError("first dim (%d) must be gost*batch with gost=%d" % (M, self.gost))
        S1, K1, S2, K2 = self.S1, self.K1, self.S2, self.K2
        # ---- level 1: grouping + fused set-abstraction kernels
        xt, yt = knn_radius_group(points, S1, K1, self.r1)
        x_rows = xt.permute(0, 2, 3, 1).reshape(M * S1 * K1, D)
        n = self.net3DV_1
        params = [n[0].weight, n[0].bias, n[1].weight, n[1].bias, n[3].weight, n[3].bias, n[4].weight, n[4].bias,
                  n[6].weight, n[6].bias, n[7].weight, n[7].bias]
        buffers = {"rm1": n[1].running_mean, "rv1": n[1].running_var, "rm2": n[4].running_mean,
                   "rv2": n[4].running_var, "rm3": n[7].running_mean, "rv3": n[7].running_var}
        feat1 = sa_mlp.SAMLPFunction.apply(x_rows, dict(training=training, buffers=buffers, reduce_fn=red, K=K1, precision=self.precision),
                                         *params)
        if training:
            for i in (1, 4, 7):
                n[i].count_batch()
        # ---- level 2: group_points_2 on cat(centre_1, feat_1), point-MLP as row GEMMs, max over the K2 neighbours
        xyz1 = yt.permute(0, 2, 1, 3).reshape(M, S1, 3)
        rows2, xyz_g, centres2, _ = group_level2_rows(xyz1, feat1, S2, K2, self.r2)
        with _tail.precision(self.precision):
            n = self.net3DV_2
            h = _tail.linear_bn_relu(rows2, n[0], n[1], training, red, centers=xyz_g)
            h = _tail.linear_bn_relu(h, n[3], n[4], training, red)
            feat2 = _tail.linear_bn_relu_segmax(h, n[6], n[7], training, K2, red)      # (M*S2, 256)
            # ---- level 3 (= the live model's tail on the level-2 centroids)
            n = self.net3DV_3
            h = _tail.linear_bn_relu(feat2, n[0], n[1], training, red, centers=centres2)
            h = _tail.linear_bn_relu(h, n[3], n[4], training, red)
            x_pre = _tail.linear_bn_relu_segmax(h, n[6], n[7], training, S2, red)
            fc = self.netR_FC
            stacked = _tail.fc_head(x_pre, self.gost, fc[0], fc[1], fc[3], training, red)
        self._stacked = stacked
        x, x_global = stacked[:M], stacked[M:]
        x_nor, code = _tail.normalize_map(x, self.mapping.weight)
        return x, code, x_nor, x_global


class DenseStep:
    """One training iteration of the dense configuration: the loop body of cn3d_train_motion_GL.py:224-335 with the
    3-level encoder (grouping happens inside the model)."""

    def __init__(self, netR, optimizer, num_crop):
        self.netR, self.optimizer, self.G = netR, optimizer, num_crop
        self.rank = torch.distributed.get_rank() if fdist.is_distributed() else 0
        self.grad_sync = fdist.GradSync(list(netR.named_parameters()), early_prefixes=("net3DV_3.", "netR_FC.")) \
            if fdist.is_distributed() else None

    def __call__(self, out_points, epoch=0, order=None):
        if order is None:
            order = np.arange(0, self.G, 1)
            np.random.shuffle(order)
        if not torch.is_tensor(order):
            order = torch.as_tensor(np.asarray(order), dtype=torch.long).to(out_points.device)
        return self.run(out_points, order)

    def run(self, out_points, order):
        netR, G = self.netR, self.G
        B = out_points.shape[0]
        x, code, x_nor, x_global = netR(out_points if out_points.dtype == torch.float32 else out_points.float(), 1)
        x_keys = fdist.all_gather_view_major(x, G)
        loss_c, loss_circle = contrastive_losses_stacked(G, netR._stacked, order, x_keys=None if x_keys is x else x_keys,
                                                         clip_offset=self.rank * B)
        loss = loss_circle + loss_c
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer.step()
        return loss, loss_c, loss_circle


def make_bench_step(a, dev, rank, world):
    """bench.py --config dense: (step, eager_step, batches, launch mode, workload string, dtype, dtype note)."""
    from types import SimpleNamespace
    from .train_common import synthetic_batch
    opt = SimpleNamespace(INPUT_FEATURE_NUM=a.D)
    net = PointNet_Plus_dense(opt, gost=a.T, precision="f16").to(dev).train()
    net.bn_reduce_fn = fdist.make_bn_reduce_fn()
    from .optim import FusedAdam
    optim = FusedAdam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)
    step = DenseStep(net, optim, a.T)
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    batches = [synthetic_batch(a.B, a.T, a.N, a.D, dev, gen) for _ in range(2)]
    workload = (f"dense: motion stream, B={a.B}/GPU T={a.T} N={a.N} D={a.D}, 3-level set abstraction (S1={net.S1} K1={net.K1}, "
                f"S2={net.S2} K2={net.K2}), global+circle loss, backward, Adam")
    note = ("point-MLP contractions with fp16 inputs on v_mfma_f32_32x32x16_f16 (level-1 64->256 layer, level-2/3 GEMMs, their "
            "dgrad/wgrad), fp32 accumulation and storage; level-1 first layers and level-1 backward fp32-grade")
    mode, run_step = "eager", step
    if world == 1 and getattr(a, "graph", 1):
        from .train_common import GraphedStep
        try:
            run_step = GraphedStep(step, batches[0], a.T)           # the whole iteration replayed as one HIP graph
            mode = "hipgraph"
        except Exception as e:                                      # never lose the measurement to a capture problem
            import sys
            print("graph capture failed (%s: %s); running eager" % (type(e).__name__, e), file=sys.stderr)
            net.zero_grad(set_to_none=True)
    return run_step, step, batches, mode, workload, "f16", note
