#!/usr/bin/env python
"""facl GEMMs vs rocBLAS (torch.mm) at the tail's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib                      # noqa: E402
from facl_amd.sa_mlp import _Workspace         # noqa: E402
from tools.microbench_sa import timeit         # noqa: E402

lib = _lib.load_library()
dev = torch.device("cuda:0")
ws = _Workspace.get(dev)
p = _lib.ptr
M = 49152
NZT = int(os.environ.get("NZT", "512"))
for K, N in [(256, 256), (256, 512), (512, 1024)]:
    a = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    da = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev)
    sums = torch.empty(N, 2, dtype=torch.float64, device=dev)
    nz = max(1, min(M // 256, NZT // ((N // 128) * (K // 128))))
    slices = torch.empty(nz * N * K, device=dev)
    st = _lib.stream()
    fl = 2.0 * M * N * K
    t = timeit(lambda: lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), p(sums), p(ws), st))
    t0 = timeit(lambda: torch.addmm(b, a, W.t(), out=y))
    print(f"fwd   {M}x{K}x{N}: facl {t:.3f} ms {fl/t/1e9:6.1f} TF | rocBLAS {t0:.3f} ms {fl/t0/1e9:6.1f} TF")
    t = timeit(lambda: lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da), st))
    t0 = timeit(lambda: torch.mm(dy, W, out=da))
    print(f"dgrad {M}x{N}x{K}: facl {t:.3f} ms {fl/t/1e9:6.1f} TF | rocBLAS {t0:.3f} ms {fl/t0/1e9:6.1f} TF")
    t = timeit(lambda: lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW), p(slices), nz, st))
    t0 = timeit(lambda: torch.mm(dy.t(), a, out=dW))
    print(f"wgrad {N}x{M}x{K} (nz={nz}): facl {t:.3f} ms {fl/t/1e9:6.1f} TF | rocBLAS {t0:.3f} ms {fl/t0/1e9:6.1f} TF")
