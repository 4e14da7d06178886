"""Oracle: global + circle InfoNCE-style losses (torch fp32).

Restates /root/reference/training_code/utils_my.py:53-83 (``global_contrast``) and :85-116
(``circle_contrast``), which equal the inline code of cn3d_train_motion_GL.py:265-287 and
:290-316 line for line.  The np.random.shuffle of :97 / :298 is an explicit ``order`` argument.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _mask(B, reps, device):
    m = np.tile(np.ones((B, B)) - np.identity(B), (1, reps))          # utils_my.py:55-56
    return torch.from_numpy(m).float().to(device)


def global_contrast(num_crop, x_global, x, B):
    """sum_g CE([<xg_n, x_{gB+n}> | (xg @ x^T) * mask], label 0); CE = mean over the B rows."""
    mask = _mask(B, num_crop, x.device)
    l_pos = torch.stack([(x_global * x[g * B:(g + 1) * B]).sum(1, keepdim=True) for g in range(num_crop)])
    l_neg = (x_global @ x.t()) * mask                                  # :71-72
    logits = torch.cat([l_pos, l_neg.unsqueeze(0).expand(num_crop, -1, -1)], dim=2)  # :74-75
    labels = torch.zeros(B, dtype=torch.long, device=x.device)
    return sum(F.cross_entropy(logits[g], labels) for g in range(num_crop))          # :79-82


def circle_contrast(num_crop, x, B, order):
    """order: a permutation of 0..num_crop-1.  Chain positives <x[o_i], x[o_{i+1}]>; the
    negatives of ALL anchors are concatenated and shared by every i (:105-109)."""
    G = num_crop
    mask = _mask(B, G * (G - 1), x.device)
    blk = lambda g: x[int(g) * B:(int(g) + 1) * B]
    l_pos = torch.stack([(blk(order[i]) * blk(order[i + 1])).sum(1, keepdim=True) for i in range(G - 1)])
    l_neg_all = torch.stack([blk(order[i]) @ x.t() for i in range(G - 1)])           # (G-1,B,GB) :103
    l_neg = l_neg_all.permute(1, 0, 2).reshape(B, -1) * mask                          # :105-106
    logits = torch.cat([l_pos, l_neg.unsqueeze(0).expand(G - 1, -1, -1)], dim=2)
    labels = torch.zeros(B, dtype=torch.long, device=x.device)
    return sum(F.cross_entropy(logits[i], labels) for i in range(G - 1))
