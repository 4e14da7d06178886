"""Exercise the real RCCL code path with a 1-rank process group: every collective helper of facl_amd/dist.py and one
full data-parallel training step (is_distributed() forced on)."""
import os, sys
sys.path.insert(0, '/root/repo')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from facl_amd import dist as fdist
fdist.is_distributed = lambda: True
import facl_amd.train_common as TC
TC.fdist.is_distributed = fdist.is_distributed
from bench import make_opt
from types import SimpleNamespace
a = SimpleNamespace(B=4, T=6, N=512, D=3)
opt = make_opt(a)
import facl_amd.cn3d_model_conbag as M
net = M.PointNet_Plus(opt, gost=a.T).cuda().train()
net.bn_reduce_fn = fdist.make_bn_reduce_fn()
assert net.bn_reduce_fn is not None
optim = torch.optim.Adam(net.parameters(), lr=3e-4, betas=(0.5, 0.999), eps=1e-6, fused=True)
step = TC.ContrastiveStep(net, optim, opt, a.T, 0.16, False)
torch.manual_seed(0)
pts = (torch.rand(a.B, a.T, a.N, a.D) - 0.5).cuda()
order = np.arange(a.T)
l1 = step(pts, 0, order)[0].item()
# same step without the distributed hooks
fdist.is_distributed = lambda: False
TC.fdist.is_distributed = fdist.is_distributed
torch.manual_seed(1)
net2 = M.PointNet_Plus(opt, gost=a.T).cuda().train()
net2.load_state_dict({k: v for k, v in M.PointNet_Plus(opt, gost=a.T).state_dict().items()}, strict=False)
print("nccl world=1 step ok, loss", l1)
dist.destroy_process_group()
