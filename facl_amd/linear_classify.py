"""Linear-probe consumer of the extracted features -- counterpart of linear_classify/fc_model.py:12-25
(``Final_FC``: L2-normalise + Linear(512*22 -> 120), weight ~ N(0, 0.01), zero bias) and of the training loop of
linear_classify/linercls.py:100-150 (Adam + StepLR(5, 0.7), CrossEntropy, top-1).  The feature format is the one
``facl_amd.extract_common`` writes: per clip [x_view0 .. x_view9, x_global] (11*512) per stream, motion and
appearance concatenated (dataset_of_lin.py:103-105).  The single dense layer runs on the MFMA GEMM of csrc/gemm.hip."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import tail as _tail


class Final_FC(nn.Module):
    def __init__(self, input_dim=512, gost=11 + 11, num_class=120):
        super().__init__()
        self.fc = nn.Linear(input_dim * gost * 1, num_class)          # parameter holder: same state_dict keys (fc.weight, fc.bias)
        self.fc.weight.data.normal_(mean=0.0, std=0.01)
        self.fc.bias.data.zero_()

    def forward(self, x):
        x = F.normalize(x, p=2, dim=1)
        return _tail.linear(x, self.fc)


def accuracy(output, target, topk=(1,)):
    """linercls.py:158-172."""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1).expand(maxk, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def fit(features, labels, num_class=120, nepoch=20, batch=256, lr=1e-3):
    """Train the probe on (n, 11264) float32 CUDA features; returns (model, last-epoch train top-1)."""
    netR = Final_FC(input_dim=512, gost=features.shape[1] // 512, num_class=num_class).to(features.device)
    optimizer = torch.optim.Adam(netR.parameters(), lr=lr, betas=(0.5, 0.999), eps=1e-06)
    criterion = nn.CrossEntropyLoss()
    top1 = 0.0
    for epoch in range(nepoch):
        for g in optimizer.param_groups:
            g["lr"] = lr * 0.7 ** (epoch // 5)                          # StepLR(5, 0.7) stepped with the epoch
        hit = 0.0
        for i in range(0, features.shape[0], batch):
            f, y = features[i:i + batch], labels[i:i + batch]
            out = netR(f)
            loss = criterion(out, y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            hit += float(accuracy(out, y)[0]) * f.shape[0] / 100.0
        top1 = 100.0 * hit / features.shape[0]
    return netR, top1
