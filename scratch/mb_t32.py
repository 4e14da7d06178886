"""A/B: k_gemm_t32 vs k_gemm_sbk on the FC-head / loss shapes (same process; FACL_GEMM_NOT32 read once per process, so two runs)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from facl_amd import _lib
lib = _lib.load_library(); DEV = "cuda:0"; p = _lib.ptr; st = _lib.stream()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tag = "sbk" if os.environ.get("FACL_GEMM_NOT32") else "t32"
for (M, K, N) in ((800, 1024, 1024), (800, 1024, 512), (800, 512, 768), (80, 1024, 1024)):
    a = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    t = timeit(lambda: lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y), None, None, st))
    ref = a.double() @ W.double().t() + b.double()
    err = float((y.double() - ref).norm() / ref.norm())
    dy = torch.randn(M, N, device=DEV); da = torch.empty(M, K, device=DEV)
    t2 = timeit(lambda: lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da), st))
    err2 = float((da.double() - dy.double() @ W.double()).norm() / (dy.double() @ W.double()).norm())
    dW = torch.empty(N, K, device=DEV); sl = torch.empty(4 * N * K, device=DEV)
    t3 = timeit(lambda: lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW), p(sl), 4, st))
    r3 = dy.double().t() @ a.double()
    err3 = float((dW.double() - r3).norm() / r3.norm())
    print(f"{tag} {M}x{K}x{N}: fwd {t:.1f} us ({err:.1e})  dgrad {t2:.1f} us ({err2:.1e})  wgrad {t3:.1f} us ({err3:.1e})", flush=True)
