"""The two optional loss terms of the reference's training loop, switched off in the shipped loop
(``swa_if = 0`` cn3d_train_motion_GL.py:238, ``cld_if = 0`` :319); here they are real flags of the training entry.

Same names as the reference: ``distributed_sinkhorn`` / ``shoot_infs`` (cn3d_model_conbag.py:391-425), ``KMeans`` /
``grouping`` (cn3d_train_motion_GL.py:36-70).  The iterative parts run as single HIP launches (csrc/swav.hip); the
differentiable remainder (cross-entropies on (B x 64) / (3B x 60) logits) is tensor algebra.  Under data parallelism the
terms are evaluated on the local clips (the reference's commented-out ``dist.all_reduce`` lines of the sinkhorn are the
places where a global assignment would add collectives)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import _lib


def shoot_infs(inp_tensor):
    """cn3d_model_conbag.py:409-425 (in place, like the reference), without the reference's host round trip."""
    mask = torch.isinf(inp_tensor)
    z = torch.where(mask, torch.zeros((), dtype=inp_tensor.dtype, device=inp_tensor.device), inp_tensor)
    inp_tensor.copy_(torch.where(mask, z.max(), z))
    return inp_tensor


def distributed_sinkhorn(Q, nmb_iters):
    """cn3d_model_conbag.py:391-406: Q (K prototypes, n samples) -> (n, K) float32; one HIP launch."""
    _lib.require_cuda(Q)
    lib = _lib.load_library()
    with torch.no_grad():
        Qc = Q.detach().float().contiguous()
        R, C = Qc.shape
        scratch = _lib.empty(R * C + R, dtype=torch.float32, device=Q.device)
        out = _lib.empty((C, R), dtype=torch.float32, device=Q.device)
        _lib.check(lib.facl_sinkhorn(_lib.ptr(Qc), R, C, int(nmb_iters), _lib.ptr(scratch), _lib.ptr(out), _lib.stream()),
                   "facl_sinkhorn")
    return out


class _SegmentMean(torch.autograd.Function):
    """centroids[k] = sum_{i: label_i = k} x_i / count_k -- the last line of the reference's KMeans loop
    (cn3d_train_motion_GL.py:67-68), through which its autograd graph reaches x."""

    @staticmethod
    def forward(ctx, x, labels, cent, counts):
        ctx.save_for_backward(labels, counts)
        return cent

    @staticmethod
    def backward(ctx, dc):
        labels, counts = ctx.saved_tensors
        return (dc / counts.to(dc.dtype).unsqueeze(1))[labels.long()], None, None, None


def KMeans(x, K=10, Niters=10, verbose=False):
    """cn3d_train_motion_GL.py:54-70.  Returns (labels (N,) int64, centroids (K, D)); the centroids are differentiable
    in x through the final scatter-mean, like the reference's."""
    _lib.require_cuda(x)
    lib = _lib.load_library()
    xc = x.detach().float().contiguous()
    N, D = xc.shape
    labels = _lib.empty(N, dtype=torch.int32, device=x.device)
    cent = _lib.empty((K, D), dtype=torch.float32, device=x.device)
    counts = _lib.empty(K, dtype=torch.int32, device=x.device)
    _lib.check(lib.facl_kmeans(_lib.ptr(xc), N, D, int(K), int(Niters), _lib.ptr(labels), _lib.ptr(cent), _lib.ptr(counts),
                               _lib.stream()), "facl_kmeans")
    return labels.long(), _SegmentMean.apply(x, labels, cent, counts)


def grouping(features_groupDis1, features_groupDis2, T, k_eigen, clusters, num_iters):
    """cn3d_train_motion_GL.py:36-52."""
    l1, c1 = KMeans(features_groupDis1, clusters, num_iters)
    l2, c2 = KMeans(features_groupDis2, clusters, num_iters)
    loss = F.cross_entropy(torch.mm(features_groupDis1, c2.t()) / T, l2)
    return (loss + F.cross_entropy(torch.mm(features_groupDis2, c1.t()) / T, l1)) / 2


def cld_loss(x_nor, batchSize, num_crop, T=0.05, clusters=60, num_iters=5):
    """cn3d_train_motion_GL.py:319-326 (k_eigen = 10, clusters = 60, num_iters = 5, T = 0.05 literals at :325)."""
    B = batchSize
    total = 0
    for i in range(num_crop - 4):
        total = total + grouping(x_nor[i * B:(i + 3) * B], x_nor[(i + 1) * B:(i + 4) * B], T, 10, clusters, num_iters)
    return total


class SwavState:
    """The queue bookkeeping of the loop (cn3d_train_motion_GL.py:186-190,:215-220): a (num_crop-1, 32*B, 512) queue of
    normalised embeddings created at epoch >= 10.  ``use_the_queue`` latches once the queue has filled up (the
    reference tests ``torch.all(queue[crop, -1] == 0)`` on the host every step; the fill level is known from the step
    count, so no device read is needed here)."""

    def __init__(self, batchSize, num_crop, dim=512, queue_length=None, epoch_queue_starts=10):
        self.B, self.G, self.dim = batchSize, num_crop, dim
        self.queue_length = 32 * batchSize if queue_length is None else queue_length
        self.epoch_queue_starts = epoch_queue_starts
        self.queue = None
        self.filled = 0                                  # rows of every crop's queue that hold real embeddings

    def maybe_create(self, epoch, device):
        if self.queue_length > 0 and epoch >= self.epoch_queue_starts and self.queue is None:
            self.queue = torch.zeros(self.G - 1, self.queue_length, self.dim, device=device)

    @property
    def use_the_queue(self):
        return self.queue is not None and self.filled >= self.queue_length


def swav_loss(code, x_nor, mapping_weight, state):
    """cn3d_train_motion_GL.py:239-263.  code (G*B, K) = mapping(x_nor), x_nor (G*B, 512)."""
    B, G = state.B, state.G
    loss_swa = 0
    use = state.use_the_queue
    for crop_id in range(G - 1):
        with torch.no_grad():
            po = code[B * crop_id:B * (crop_id + 1), :]
            if state.queue is not None:
                if use:
                    po = torch.cat((torch.mm(state.queue[crop_id], mapping_weight.t()), po))
                state.queue[crop_id, B:, :] = state.queue[crop_id, :-B, :].clone()
                state.queue[crop_id, 0:B, :] = x_nor[crop_id * B:(crop_id + 1) * B, :]
            q = distributed_sinkhorn(torch.exp(po / 0.03).t(), 3)[-B:]
        subloss = 0
        for v in np.delete(np.arange(G - 1), crop_id):
            logp = F.log_softmax(code[B * v:B * (v + 1)] / 0.1, dim=1)        # torch.log(softmax(.)) at :257-258
            subloss = subloss - torch.mean(torch.sum(q * logp, dim=1))
        loss_swa = loss_swa + subloss
    if state.queue is not None:
        state.filled = min(state.filled + B, state.queue_length)
    return loss_swa / (G - 1)
