"""Encoder model -- drop-in for the reference's ``training_code/cn3d_model_conbag.py`` classes
``PointNet_Plus`` / ``PointNet_Plus_fine`` (:22-234): same constructor arguments, the same 52
state_dict keys and shapes (so checkpoints written by either side load in the other), the
4-output forward contract ``forward(xt, yt, loss_mode=0) -> (x, code, x_nor, x_global)`` (:213-234,
the form the shipped training loop unpacks at cn3d_train_motion_GL.py:234), ``train()/eval()``.

The modules below only HOLD parameters under the reference's key names; compute happens in the
HIP passes (facl_amd.sa_mlp for net3DV_1) -- there is no nn.Conv2d / nn.BatchNorm2d execution.
"""
import math

import torch
import torch.nn as nn

from . import sa_mlp
from . import tail as _tail

nstates_plus_1 = [64, 64, 256]
nstates_plus_2 = [128, 128, 256]
nstates_plus_3 = [256, 512, 1024, 1024, 1024]


class _Affine(nn.Module):
    """weight/bias holder initialised exactly like nn.Conv2d / nn.Linear.reset_parameters (same RNG
    draws in the same order, so a seeded construction reproduces the reference's initial weights)."""

    def __init__(self, shape, fan_in, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(shape))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.empty(shape[0]))
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)


class _BatchNormState(nn.Module):
    """gamma/beta + running buffers under nn.BatchNorm's names.

    ``num_batches_tracked`` is counted on the HOST (``steps``) and written into the buffer only when a
    state_dict is taken: incrementing a device int64 costs one kernel launch per BN layer per call (8 per
    training step) for a number nothing on the device ever reads."""

    def __init__(self, C):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(C))
        self.bias = nn.Parameter(torch.zeros(C))
        self.register_buffer("running_mean", torch.zeros(C))
        self.register_buffer("running_var", torch.ones(C))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.steps = 0
        self.register_state_dict_pre_hook(_BatchNormState._sync_counter)

    @staticmethod
    def _sync_counter(module, prefix, keep_vars):
        module.num_batches_tracked.fill_(module.steps)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        key = prefix + "num_batches_tracked"
        if key in state_dict:
            self.steps = int(state_dict[key])

    def count_batch(self):
        self.steps += 1


class _Slots(nn.Module):
    """Children registered under the integer names nn.Sequential would have used (0,1,3,4,6,7)."""

    def __init__(self, mods):
        super().__init__()
        for idx, m in mods:
            self.add_module(str(idx), m)

    def __getitem__(self, i):
        return self._modules[str(i)]


def _conv_stack(cin, widths):
    mods = []
    for li, cout in zip((0, 3, 6), widths):
        mods += [(li, _Affine((cout, cin, 1, 1), cin)), (li + 1, _BatchNormState(cout))]
        cin = cout
    return _Slots(mods)


class PointNet_Plus(nn.Module):
    """cn3d_model_conbag.py:22-137.  The reference class hard-codes the K=64 pooling window (:57) and
    pools over ``opt.sample_num_level2`` centroids (:80); ``PointNet_Plus_fine`` takes them as
    arguments.  Both are served by this implementation."""

    def __init__(self, opt, num_clusters=64, gost=10, dim=512, normalize_input=True,
                 _sample_num_level1=None, _knn_K=64):
        super().__init__()
        self.temperal_num = opt.temperal_num
        self.knn_K = opt.knn_K if _sample_num_level1 is None else _knn_K
        self.ball_radius2 = opt.ball_radius2
        self.sample_num_level1 = opt.sample_num_level1 if _sample_num_level1 is None else _sample_num_level1
        self.sample_num_level2 = opt.sample_num_level2
        self.INPUT_FEATURE_NUM = opt.INPUT_FEATURE_NUM
        self.num_outputs = opt.Num_Class
        self.batch = opt.batchSize
        self.dim = dim
        self.num_clusters = num_clusters
        self.gost = gost
        self.normalize_input = normalize_input
        self.pooling = opt.pooling
        if self.pooling == "concatenation":
            self.dim_out = 1024                      # else: attribute missing, like the reference (:40-41)
        self._pool_K = _knn_K                        # :57 hard-codes 64
        self._pool_S = self.sample_num_level2 if _sample_num_level1 is None else _sample_num_level1   # :80 / :199

        self.net3DV_1 = _conv_stack(self.INPUT_FEATURE_NUM, nstates_plus_1)
        self.net3DV_3 = _conv_stack(3 + nstates_plus_2[2], nstates_plus_3[:3])
        self.netR_FC = _Slots([(0, _Affine((nstates_plus_3[4], self.dim_out), self.dim_out)),
                               (1, _BatchNormState(nstates_plus_3[4])),
                               (3, _Affine((self.dim, nstates_plus_3[4]), nstates_plus_3[4]))])
        self.mapping = _Affine((self.num_clusters, self.dim), self.dim, bias=False)
        # SyncBN hook: callable(fp64 tensor) that all-reduces in place across the data-parallel group
        self.bn_reduce_fn = None

    # ---- helpers --------------------------------------------------------------------------
    def _sa_args(self):
        n = self.net3DV_1
        params = [n[0].weight, n[0].bias, n[1].weight, n[1].bias, n[3].weight, n[3].bias, n[4].weight, n[4].bias,
                  n[6].weight, n[6].bias, n[7].weight, n[7].bias]
        buffers = {"rm1": n[1].running_mean, "rv1": n[1].running_var, "rm2": n[4].running_mean,
                   "rv2": n[4].running_var, "rm3": n[7].running_mean, "rv3": n[7].running_var}
        return params, buffers

    def forward(self, xt, yt, loss_mode=0):
        # xt: M x INPUT_FEATURE_NUM x S x K (the transposed view group_points_3DV returns), yt: M x 3 x S x 1
        if xt.dim() != 4 or yt.dim() != 4:
            raise ValueError("expected xt (M,D,S,K) and yt (M,3,S,1)")
        M, D, S, K = xt.shape
        if D != self.INPUT_FEATURE_NUM:
            raise RuntimeError("input has %d channels, model was built for %d" % (D, self.INPUT_FEATURE_NUM))
        if K != self._pool_K:
            raise RuntimeError("pooling window is %d neighbours, got K=%d" % (self._pool_K, K))
        if S != self._pool_S:
            raise RuntimeError("model pools over %d centroids, got S=%d" % (self._pool_S, S))
        if M % self.gost:
            raise RuntimeError("first dim (%d) must be gost*batch with gost=%d" % (M, self.gost))
        x_rows = xt.permute(0, 2, 3, 1)                       # (M,S,K,D): the buffer behind the view
        x_rows = x_rows.contiguous().view(M * S * K, D).float()
        centers = yt.permute(0, 2, 1, 3).reshape(M * S, 3)    # (M*S,3) rows
        training = self.training

        params, buffers = self._sa_args()
        prec = getattr(self, "precision", "f32")              # "f32" (default, fp32-grade) or the opt-in "x3" (tail.precision)
        state = dict(training=training, buffers=buffers, reduce_fn=self.bn_reduce_fn, K=K, precision=prec)
        pooled = sa_mlp.SAMLPFunction.apply(x_rows, state, *params)           # (M*S,256)   net3DV_1 (:218)
        if training:
            for i in (1, 4, 7):
                self.net3DV_1[i].count_batch()
        with _tail.precision(prec):
            return self._tail_forward(pooled, centers, M, S, training, state.get("pooled_amax"))

    def _tail_forward(self, pooled, centers, M, S, training, pooled_amax=None):

        # net3DV_3 (:220).  torch.cat((yt, xt), 1) (:219) is never built: the first GEMM takes the centroid xyz as a
        # rank-3 term in its epilogue
        n3 = self.net3DV_3
        widths = (n3[0].weight.shape[1] - 3, n3[0].weight.shape[0], n3[3].weight.shape[0], n3[6].weight.shape[0])
        if _tail.net3dv3_supported(pooled.shape[0], widths, S, _tail.current_precision()):
            # the 49,152-row case: row-streamed GEMMs, BN + ReLU of a layer applied in the next GEMM's prologue
            x_pre = _tail.net3dv3(pooled, centers, n3, training, S, self.bn_reduce_fn, pooled_amax)
        else:
            h = _tail.linear_bn_relu(pooled, n3[0], n3[1], training, self.bn_reduce_fn, centers=centers)
            h = _tail.linear_bn_relu(h, n3[3], n3[4], training, self.bn_reduce_fn)
            # last layer fused with my_max_pool (:222-223): xt_local (M,1024,S,1) is never materialised post-BN
            x_pre = _tail.linear_bn_relu_segmax(h, n3[6], n3[7], training, S, self.bn_reduce_fn)
        # gobaol_max_pool over all gost*S local features of a clip (:225-226) = max over the gost views of the per-view
        # maxima (rows are view-major: g*B+b; first view wins ties, like the reference's max-pool over the sequence), then
        # :228 x = netR_FC(x_pre), :229 x_global = netR_FC(x_global_pre): one pass over the two Linear layers for both,
        # BatchNorm statistics (and the two running-statistics updates) per call like the reference.  The stacked
        # (M + B, dim) tensor is kept: the training step's loss runs one similarity GEMM on it.
        fc = self.netR_FC
        stacked = _tail.fc_head(x_pre, self.gost, fc[0], fc[1], fc[3], training, self.bn_reduce_fn)
        self._stacked = stacked
        x, x_global = stacked[:M], stacked[M:]
        # :231-232.  `lazy_code` (set by a caller that does not read x_nor / code before it calls _lib.join_pending(): the
        # training step without the SwAV / CLD terms): the kernel runs on the side stream beside the loss block
        x_nor, code = _tail.normalize_map(x, self.mapping.weight, lazy=bool(getattr(self, "lazy_code", False)))
        return x, code, x_nor, x_global


class PointNet_Plus_fine(PointNet_Plus):
    """cn3d_model_conbag.py:141-234: explicit sample_num_level1 / knn_K."""

    def __init__(self, opt, num_clusters=64, gost=10, dim=512, sample_num_level1=32, knn_K=128,
                 normalize_input=True):
        super().__init__(opt, num_clusters=num_clusters, gost=gost, dim=dim, normalize_input=normalize_input,
                         _sample_num_level1=sample_num_level1, _knn_K=knn_K)
