"""Oracle for the two optional loss terms of the training loop, torch-CPU.

TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

  * SwAV term: ``distributed_sinkhorn`` / ``shoot_infs`` (training_code/cn3d_model_conbag.py:391-425) and the loop block
    cn3d_train_motion_GL.py:239-263.  The two functions import on CPU here: tests/golden/swav.npz holds their outputs
    (tools/make_goldens.py: make_swav) and pins the restatement below.
  * CLD term: ``KMeans`` / ``grouping`` (cn3d_train_motion_GL.py:36-70; textually identical twins in utils_my.py:164-197)
    and the loop block :319-326 (= utils_my.CLD_Loss :152-161).  tests/golden/cld.npz holds the outputs of the training
    script's own functions (tools/make_goldens.py: make_cld; the script imports with empty harness-side stand-ins for
    its unused imageio / torchvision imports) and pins the restatement below: labels, centroids, loss, gradient.
Both terms are switched off in the shipped loop (``swa_if = 0`` :238, ``cld_if = 0`` :319).
"""
import numpy as np
import torch
import torch.nn.functional as F


def shoot_infs(t):
    """cn3d_model_conbag.py:409-425: every +-inf entry is replaced by the maximum of the tensor with those entries set
    to 0 (in place in the reference; a new tensor here)."""
    mask = torch.isinf(t)
    if not bool(mask.any()):
        return t
    z = torch.where(mask, torch.zeros((), dtype=t.dtype), t)
    return torch.where(mask, z.max(), z)


def distributed_sinkhorn(Q, nmb_iters):
    """cn3d_model_conbag.py:391-406 (single process: the all_reduce lines are commented out there).  Q (K, n) -> (n, K)."""
    with torch.no_grad():
        Q = shoot_infs(Q.clone())
        Q = Q / torch.sum(Q)
        r = torch.ones(Q.shape[0], dtype=Q.dtype) / Q.shape[0]
        c = torch.ones(Q.shape[1], dtype=Q.dtype) / Q.shape[1]
        for _ in range(nmb_iters):
            u = shoot_infs(r / torch.sum(Q, dim=1))
            Q = Q * u.unsqueeze(1)
            Q = Q * (c / torch.sum(Q, dim=0)).unsqueeze(0)
        return (Q / torch.sum(Q, dim=0, keepdim=True)).t().float()


def swav_loss(code, x_nor, mapping_weight, B, num_crop, queue=None, use_the_queue=False):
    """cn3d_train_motion_GL.py:239-263.  code (G*B, K), x_nor (G*B, 512), mapping_weight (K, 512); ``queue``
    (num_crop-1, L, 512) or None.  Returns (loss_swa, queue, use_the_queue) -- the queue is updated like the reference."""
    loss_swa = 0
    for crop_id in range(num_crop - 1):
        with torch.no_grad():
            po = code[B * crop_id:B * (crop_id + 1), :]
            if queue is not None:
                if use_the_queue or not torch.all(queue[crop_id, -1, :] == 0):
                    use_the_queue = True
                    po = torch.cat((torch.mm(queue[crop_id], mapping_weight.t()), po))
                queue[crop_id, B:, :] = queue[crop_id, :-B, :].clone()
                queue[crop_id, 0:B, :] = x_nor[crop_id * B:(crop_id + 1) * B, :]
            po = po / 0.03
            po = torch.exp(po).t()
            q = distributed_sinkhorn(po, 3)[-B:]
        subloss = 0
        for v in np.delete(np.arange(num_crop - 1), crop_id):
            p = F.softmax(code[B * v:B * (v + 1)] / 0.1, dim=1)
            subloss = subloss - torch.mean(torch.sum(q * torch.log(p), dim=1))
        loss_swa = loss_swa + subloss
    return loss_swa / (num_crop - 1), queue, use_the_queue


def KMeans(x, K=10, Niters=10):
    """cn3d_train_motion_GL.py:54-70: Lloyd iterations from the first K rows; an empty cluster keeps count 1 (its
    centroid becomes 0).  Returns (labels (N,), centroids (K, D)); the centroids stay differentiable in x through the
    last scatter-mean, exactly like the reference's autograd graph."""
    N, D = x.shape
    c = x[:K, :].clone()
    x_i = x[:, None, :]
    cl = None
    for _ in range(Niters):
        D_ij = ((x_i - c[None, :, :]) ** 2).sum(-1)
        cl = D_ij.argmin(dim=1).long().view(-1)
        Ncl = cl.view(N, 1).expand(-1, D)
        counts = torch.ones(K, dtype=torch.long)
        uniq, cnt = cl.unique(return_counts=True)
        counts[uniq] = cnt
        c = torch.zeros(K, D, dtype=x.dtype).scatter_add_(0, Ncl, x) / counts.to(x.dtype).unsqueeze(1)
    return cl, c


def grouping(f1, f2, T, k_eigen, clusters, num_iters):
    """cn3d_train_motion_GL.py:36-52: cross-level discrimination between two groups of embeddings."""
    l1, c1 = KMeans(f1, clusters, num_iters)
    l2, c2 = KMeans(f2, clusters, num_iters)
    loss = F.cross_entropy(torch.mm(f1, c2.t()) / T, l2)
    return (loss + F.cross_entropy(torch.mm(f2, c1.t()) / T, l1)) / 2


def cld_loss(x_nor, B, num_crop, T=0.05, clusters=60, num_iters=5):
    """cn3d_train_motion_GL.py:319-326: sum over num_crop-4 windows of three consecutive views vs the next three."""
    total = 0
    for i in range(num_crop - 4):
        total = total + grouping(x_nor[i * B:(i + 3) * B], x_nor[(i + 1) * B:(i + 4) * B], T, 10, clusters, num_iters)
    return total
