"""A/B timing: row-streamed GEMM vs k_gemm_sb at the headline shapes (one process, interleaved rounds)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
lib = _lib.load_library()
DEV = "cuda:0"
p = _lib.ptr
ws = _Workspace.get(torch.device(DEV))
st = _lib.stream()
def planes_of(W, tr, Wc=None):
    N, K = W.shape
    nb = lib.facl_gemm_rs_planes_bytes(K if tr else N, N if tr else K, 1 if Wc is not None else 0)
    pl = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.check(lib.facl_gemm_rs_planes(p(W), W.stride(0), N, K, int(tr), p(Wc), 3, p(pl), st), "planes")
    return pl
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 49152
for K, N in ((256, 256), (256, 512), (512, 1024)):
    a = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    ps = torch.rand(K, device=DEV) + 0.5; pt = torch.randn(K, device=DEV) * 0.3
    y = torch.empty(M, N, device=DEV); sums = torch.empty(N, 2, dtype=torch.float64, device=DEV)
    pl = planes_of(W, False)
    f_rs = lambda: _lib.check(lib.facl_gemm_rs_fwd(p(a), M, K, p(pl), N, p(b), p(ps), p(pt), None, p(y), p(sums), None, None, None, p(ws), st), "rs")
    f_sb = lambda: _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), p(ps), p(pt), None, None, 0, p(y), p(sums), p(ws), st), "sb")
    f_pl = lambda: planes_of(W, False)
    for r in range(2):
        t1, t2 = timeit(f_rs), timeit(f_sb)
        fl = 2.0 * M * K * N * 6
        print("fwd %dx%dx%d  rs %.4f ms (%.3f of bf16 peak)   sb %.4f ms (%.3f)   planes %.4f ms" % (M, K, N, t1, fl / t1 / 1e9 / 2500, t2, fl / t2 / 1e9 / 2500, timeit(f_pl)), flush=True)
    # dgrad of this layer: dy (M,N) W (N,K) -> (M,K)
    dy = torch.randn(M, N, device=DEV); da = torch.empty(M, K, device=DEV)
    plt = planes_of(W, True)
    g_rs = lambda: _lib.check(lib.facl_gemm_rs_dgrad(p(dy), M, N, p(plt), 0, None, K, p(da), st), "rsd")
    g_sb = lambda: _lib.check(lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da), st), "sbd")
    for r in range(2):
        t1, t2 = timeit(g_rs), timeit(g_sb)
        fl = 2.0 * M * K * N * 6
        print("dgrad %dx%dx%d  rs %.4f ms (%.3f)   sb %.4f ms (%.3f)" % (M, N, K, t1, fl / t1 / 1e9 / 2500, t2, fl / t2 / 1e9 / 2500), flush=True)
