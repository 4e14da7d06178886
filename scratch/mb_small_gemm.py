import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
from tools.microbench_sa import timeit
lib = _lib.load_library(); dev = torch.device("cuda:0"); ws = _Workspace.get(dev); p = _lib.ptr
st = _lib.stream()
# heat the chip like the step does
big = torch.randn(8192, 8192, device=dev)
for M in (800, 80):
  for N in (1024, 512):
    for K in (32, 128, 512, 1024, 2048):
        a = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        t = timeit(lambda: lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), None, p(ws), st))
        t0 = timeit(lambda: torch.addmm(b, a, W.t(), out=y))
        print(f"fwd {M}x{K}x{N}: facl {t*1e3:.1f} us | rocBLAS {t0*1e3:.1f} us", flush=True)
