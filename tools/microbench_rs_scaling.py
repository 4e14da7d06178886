import sys, torch
sys.path.insert(0, "/root/repo")
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
lib = _lib.load_library(); DEV = "cuda:0"; p = _lib.ptr
ws = _Workspace.get(torch.device(DEV)); st = _lib.stream()
def planes_of(W, tr):
    N, K = W.shape
    nb = lib.facl_gemm_rs_planes_bytes(K if tr else N, N if tr else K, 0)
    pl = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.check(lib.facl_gemm_rs_planes(p(W), W.stride(0), N, K, int(tr), None, 0, p(pl), st), "planes")
    return pl
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, N in ((256, 256), (256, 512), (512, 256)):
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    pl = planes_of(W, False)
    for M in (6144, 12288, 24576, 49152, 98304, 196608):
        a = torch.randn(M, K, device=DEV); y = torch.empty(M, N, device=DEV)
        t = timeit(lambda: lib.facl_gemm_rs_fwd(p(a), M, K, p(pl), N, None, None, None, None, p(y), None, None, None, None, p(ws), st))
        fl = 2.0 * M * K * N * 6
        print(f"rs plain {M}x{K}x{N}: {t:.1f} us  frac {fl / t / 1e6 / 2500e3:.3f}  wgs {M // 128 * (N // 256)}", flush=True)
print("---- fixed cost: K sweep at 48 workgroups (one per CU, no sharing)")
for K in (64, 128, 256, 512, 1024):
    N, M = 256, 6144
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    pl = planes_of(W, False)
    a = torch.randn(M, K, device=DEV); y = torch.empty(M, N, device=DEV)
    t = timeit(lambda: lib.facl_gemm_rs_fwd(p(a), M, K, p(pl), N, None, None, None, None, p(y), None, None, None, None, p(ws), st))
    print(f"K={K}: {t:.1f} us", flush=True)
empty = torch.empty(1, device=DEV)
t = timeit(lambda: empty.fill_(1.0))
print(f"tiny fill kernel: {t:.1f} us")
