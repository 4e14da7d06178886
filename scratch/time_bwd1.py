import sys, os; sys.path.insert(0,'/root/repo')
import torch
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
lib=_lib.load_library(); dev=torch.device('cuda:0')
nunits=768*64
y2f=torch.randn(nunits*4096, device=dev); bnc2=torch.rand(5,64,device=dev)+0.5; G3=torch.randn(64,64,device=dev)*0.01; h3=torch.randn(64,device=dev)
W3=torch.randn(256,64,device=dev)*0.1; coef=torch.randn(nunits,256,device=dev); arg=torch.randint(0,64,(nunits,256),dtype=torch.uint8,device=dev)
dz=torch.empty_like(y2f); sums=torch.empty(64,2,dtype=torch.float64,device=dev); ws=_Workspace.get(dev)
def run():
    _lib.check(lib.facl_sa_bwd1(_lib.ptr(y2f), nunits, _lib.ptr(bnc2), _lib.ptr(G3), _lib.ptr(h3), _lib.ptr(W3), _lib.ptr(coef), _lib.ptr(arg), _lib.ptr(dz), _lib.ptr(sums), _lib.ptr(ws), _lib.stream()),"b1")
for _ in range(2): run()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record(); 
for _ in range(5): run()
e1.record(); torch.cuda.synchronize(); print("FACL_DBG", os.environ.get("FACL_DBG"), "bwd1 ms", e0.elapsed_time(e1)/5)
