import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from helpers import rel_err, max_rel_rows
import test_gpu_encoder as T
from facl_amd.cn3d_model_conbag import PointNet_Plus_fine
from facl_amd.utils_my import knn_radius_group
from oracle import encoder as E, grouping as OG
from oracle.weights import formula_state_dict
DEV = T.DEV
S, K = int(sys.argv[1]), int(sys.argv[2])
D, B, G, N = 4, int(sys.argv[3]), 3, 512
torch.manual_seed(K)
pts = torch.rand(G * B, N, D) - 0.5
opt = T._opt(D, B, N)
net = PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
sdn = formula_state_dict(D)
net.load_state_dict({k: torch.as_tensor(v) for k, v in sdn.items()})
net = net.to(DEV).train()
xt, yt = knn_radius_group(pts.to(DEV), S, K, 0.1)
x, code, x_nor, xg = net(xt, yt, 1)
w = torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
((x * w).sum() + xg.sum()).backward()
_, xt_o, yt_o = OG.group_points(pts.numpy(), S, K, 0.1)
sd = {k: (torch.as_tensor(v).double() if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v).clone()) for k, v in sdn.items()}
keys = [k for k in sd if "running" not in k and "num_b" not in k]
for k in keys: sd[k].requires_grad_(True)
M = G * B
xo, _, _, xgo = E.encoder_forward(sd, torch.from_numpy(xt_o).permute(0, 3, 1, 2).double(), torch.from_numpy(yt_o).view(M, 1, S, 3).transpose(1, 3).double(), G, True)
((xo * w.cpu().double()).sum() + xgo.sum()).backward()
print("x", max_rel_rows(x.detach().cpu().numpy(), xo.detach().numpy()), "xg", max_rel_rows(xg.detach().cpu().numpy(), xgo.detach().numpy()))
for k, p in net.named_parameters():
    if sd[k].grad is None or p.grad is None: continue
    print(f"{k:28s} rel {rel_err(p.grad.cpu().numpy(), sd[k].grad.numpy()):.3e}  |true| {float(sd[k].grad.norm()):.3e}")
# GEMM accuracy vs fp64
from facl_amd import tail
torch.manual_seed(0)
for (Mr, Kc, Nc) in [(4096, 512, 1024), (4096, 256, 256)]:
    a = torch.randn(Mr, Kc, device=DEV); W = torch.randn(Nc, Kc, device=DEV) / Kc ** 0.5; b = torch.zeros(Nc, device=DEV)
    y, _ = tail.gemm_fwd(a, W, b)
    ref = a.double() @ W.double().t()
    ytorch = a @ W.t()
    print("gemm", Mr, Kc, Nc, "facl rel", float((y.double() - ref).norm() / ref.norm()), "max", float((y.double() - ref).abs().max() / ref.abs().max()),
          "| rocBLAS rel", float((ytorch.double() - ref).norm() / ref.norm()))
