import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
from tools.microbench_sa import timeit
lib = _lib.load_library(); dev = torch.device("cuda:0"); ws = _Workspace.get(dev); p = _lib.ptr; st = _lib.stream()
for (M, K, N) in [(49152, 512, 1024), (49152, 256, 512)]:
    a = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev); da = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev)
    sums = torch.empty(N, 2, dtype=torch.float64, device=dev)
    tiles = ((N + 127) // 128) * ((K + 127) // 128); nz = max(1, min((M + 255) // 256, 512 // tiles)); sl = torch.empty(nz * N * K, device=dev)
    for rep in range(2):
        t1 = timeit(lambda: lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), p(sums), p(ws), st))
        t2 = timeit(lambda: lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da), st))
        t3 = timeit(lambda: lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW), p(sl), nz, st))
        print(f"{M}x{K}x{N}: fwd {t1*1e3:.1f} dgrad {t2*1e3:.1f} wgrad {t3*1e3:.1f} us", flush=True)
