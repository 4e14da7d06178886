import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace
from tools.microbench_sa import timeit
lib = _lib.load_library(); dev = torch.device("cuda:0"); ws = _Workspace.get(dev); p = _lib.ptr; st = _lib.stream()

def planes(x, transposed=False):
    R, C = x.shape
    pl = torch.empty(3, R, C, dtype=torch.bfloat16, device=dev)
    plt = torch.empty(3, C, R, dtype=torch.bfloat16, device=dev) if transposed else None
    _lib.check(lib.facl_split_planes(p(x), R, C, x.stride(0), p(pl), p(plt), st), "split")
    return pl, plt

def ref_planes(x):
    hi = x.bfloat16(); r = x - hi.float(); mi = r.bfloat16(); lo = (r - mi.float()).bfloat16()
    return torch.stack((hi, mi, lo))

def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())

for (M, K, N) in [(49152, 512, 1024), (49152, 256, 512), (49152, 256, 256), (1000, 104, 200)]:
    a = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    apl, _ = planes(a); wpl, wtpl = planes(W, True); dypl, _ = planes(dy)
    assert torch.equal(apl, ref_planes(a)) and torch.equal(wpl, ref_planes(W)) and torch.equal(wtpl, ref_planes(W.t().contiguous())), "split mismatch"
    y = torch.empty(M, N, device=dev); y0 = torch.empty(M, N, device=dev)
    sums = torch.empty(N, 2, dtype=torch.float64, device=dev); sums0 = torch.empty_like(sums)
    f_new = lambda: _lib.check(lib.facl_gemm_pl_fwd(p(apl), M, K, p(wpl), N, p(b), None, None, 0, p(y), p(sums), None, None, None, p(ws), st), "plfwd")
    f_old = lambda: _lib.check(lib.facl_gemm_fwd(p(a), M, K, p(W), K, N, p(b), None, None, None, None, 0, p(y0), p(sums0), p(ws), st), "fwd")
    f_new(); f_old()
    ref = a.double() @ W.double().t() + b.double()
    print(f"fwd {M}x{K}x{N}: err new {rel(y, ref):.2e} old {rel(y0, ref):.2e} sums {rel(sums[:,1], (ref*ref).sum(0)):.2e}", flush=True)
    tn, to = timeit(f_new), timeit(f_old)
    print(f"   time new {tn*1e3:.1f} us old {to*1e3:.1f} us", flush=True)
    # dgrad
    da = torch.empty(M, K, device=dev); da0 = torch.empty(M, K, device=dev)
    d_new = lambda: _lib.check(lib.facl_gemm_pl_fwd(p(dypl), M, N, p(wtpl), K, None, None, None, 0, p(da), None, None, None, None, p(ws), st), "pldgrad")
    d_old = lambda: _lib.check(lib.facl_gemm_dgrad(p(dy), M, N, p(W), K, K, p(da0), st), "dgrad")
    d_new(); d_old()
    ref = dy.double() @ W.double()
    print(f"dgrad: err new {rel(da, ref):.2e} old {rel(da0, ref):.2e}", flush=True)
    tn, to = timeit(d_new), timeit(d_old)
    print(f"   time new {tn*1e3:.1f} us old {to*1e3:.1f} us", flush=True)
    # wgrad
    tiles = ((N + 255) // 256) * ((K + 127) // 128)
    nz = max(1, min(M // 256, 256 // tiles))
    tiles0 = ((N + 127) // 128) * ((K + 127) // 128)
    nz0 = max(1, min((M + 255) // 256, 512 // tiles0))
    dW = torch.empty(N, K, device=dev); dW0 = torch.empty(N, K, device=dev)
    sl = torch.empty(max(nz, nz0) * N * K, device=dev)
    w_new = lambda: _lib.check(lib.facl_gemm_pl_wgrad(p(dypl), p(apl), M, N, K, p(dW), p(sl), nz, st), "plwgrad")
    w_old = lambda: _lib.check(lib.facl_gemm_wgrad(p(dy), p(a), M, N, K, K, p(dW0), p(sl), nz0, st), "wgrad")
    w_new(); w_old()
    ref = dy.double().t() @ a.double()
    print(f"wgrad nz={nz}: err new {rel(dW, ref):.2e} old {rel(dW0, ref):.2e}", flush=True)
    tn, to = timeit(w_new), timeit(w_old)
    print(f"   time new {tn*1e3:.1f} us old {to*1e3:.1f} us", flush=True)
