"""Cost of N extra tiny dependent kernels inside the captured step (graph replay), same process."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from facl_amd import _lib
from facl_amd.cn3d_model_conbag import PointNet_Plus
from facl_amd.train_common import ContrastiveStep, GraphedStep, synthetic_batch
from facl_amd.optim import FusedAdam
sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device("cuda:0")
torch.manual_seed(1); np.random.seed(1)
opt = bench.make_opt(a)
lib = _lib.load_library()
gen = torch.Generator(device=dev); gen.manual_seed(0)
batch = synthetic_batch(a.B, a.T, a.N, a.D, dev, gen)
g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev); rm = torch.zeros(64, device=dev); rv = torch.ones(64, device=dev)
bnc = torch.empty(5, 64, device=dev)
res = {}
for extra in (0, 60, 0, 60):
    net = PointNet_Plus(opt, gost=a.T).to(dev).train()
    optim = FusedAdam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)
    step = ContrastiveStep(net, optim, opt, a.T)
    orig = step.run
    def run(self, pts, order, _o=orig, _n=extra):
        out = _o(pts, order)
        for _ in range(_n):
            lib.facl_bn_eval_consts(64, _lib.ptr(g), _lib.ptr(b), _lib.ptr(rm), _lib.ptr(rv), 1e-5, _lib.ptr(bnc), _lib.stream())
        return out
    step.run = types.MethodType(run, step)
    gs = GraphedStep(step, batch, a.T)
    for _ in range(5): gs(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): gs(batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"extra={extra}: {dt*1e3:.3f} ms/step", flush=True)
