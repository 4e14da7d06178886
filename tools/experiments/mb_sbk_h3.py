"""few-row GEMMs: bf16x6 entries vs the fp16x3 (h3) entries -- accuracy vs fp64 and time (stand-alone loops)"""
import sys, torch
sys.path.insert(0, ".")
from facl_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda")
torch.manual_seed(0)
def amax_of(t):
    b = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=dev)
    _lib.check(lib.facl_absmax(_lib.ptr(t.contiguous()), t.numel(), _lib.ptr(b), _lib.stream()), "absmax")
    return b
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def err(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())
M = 800
for (K, N) in ((1024, 1024), (1024, 512), (512, 768)):
    a = torch.randn(M, K, device=dev) * 3.0
    W = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    ref = a.double() @ W.double().t() + bias.double()
    y0 = torch.empty(M, N, device=dev); y1 = torch.empty(M, N, device=dev)
    ws = torch.empty(lib.facl_ws_bytes(), dtype=torch.uint8, device=dev)
    am_a, am_w = amax_of(a), amax_of(W)
    out_amax = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device=dev)
    f0 = lambda: _lib.check(lib.facl_gemm_fwd(_lib.ptr(a), M, K, _lib.ptr(W), K, N, _lib.ptr(bias), None, None, None, None, 0, _lib.ptr(y0), None, _lib.ptr(ws), _lib.stream()), "fwd")
    f1 = lambda: _lib.check(lib.facl_gemm_fwd_h3(_lib.ptr(a), M, K, _lib.ptr(W), K, N, _lib.ptr(bias), _lib.ptr(am_a), _lib.ptr(am_w), _lib.ptr(y1), _lib.ptr(out_amax), _lib.stream()), "fwd_h3")
    f0(); f1()
    got = out_amax.view(torch.float32).max().item()
    print(f"fwd {M}x{K}x{N}: err bf16x6 {err(y0, ref):.2e} h3 {err(y1, ref):.2e}  amax_out {got:.4f} vs {y1.abs().max().item():.4f}  us {timeit(f0):.1f} -> {timeit(f1):.1f}")
    # dgrad: da = dy W
    dy = torch.randn(M, N, device=dev) * 1e-3
    refd = dy.double() @ W.double()
    d0 = torch.empty(M, K, device=dev); d1 = torch.empty(M, K, device=dev)
    am_dy = amax_of(dy)
    g0 = lambda: _lib.check(lib.facl_gemm_dgrad(_lib.ptr(dy), M, N, _lib.ptr(W), K, K, _lib.ptr(d0), _lib.stream()), "dgrad")
    g1 = lambda: _lib.check(lib.facl_gemm_dgrad_h3(_lib.ptr(dy), M, N, _lib.ptr(W), K, K, _lib.ptr(am_dy), _lib.ptr(am_w), _lib.ptr(d1), None, _lib.stream()), "dgrad_h3")
    g0(); g1()
    print(f"dgrad      : err bf16x6 {err(d0, refd):.2e} h3 {err(d1, refd):.2e}  us {timeit(g0):.1f} -> {timeit(g1):.1f}")
    # wgrad: dW = dy^T a
    refw = dy.double().t() @ a.double()
    w0 = torch.empty(N, K, device=dev); w1 = torch.empty(N, K, device=dev); w2 = torch.ones(N, K, device=dev)
    sl = torch.empty(N * K, device=dev)
    h0 = lambda: _lib.check(lib.facl_gemm_wgrad(_lib.ptr(dy), _lib.ptr(a), M, N, K, K, _lib.ptr(w0), _lib.ptr(sl), 1, _lib.stream()), "wgrad")
    h1 = lambda: _lib.check(lib.facl_gemm_wgrad_sk_h3(_lib.ptr(dy), _lib.ptr(a), M, N, K, K, _lib.ptr(am_dy), _lib.ptr(am_a), _lib.ptr(w1), 0, None, _lib.stream()), "wgrad_h3")
    h0(); h1()
    _lib.check(lib.facl_gemm_wgrad_sk_h3(_lib.ptr(dy), _lib.ptr(a), M, N, K, K, _lib.ptr(am_dy), _lib.ptr(am_a), _lib.ptr(w2), 1, None, _lib.stream()), "wgrad_h3 acc")
    print(f"wgrad      : err bf16x6 {err(w0, refw):.2e} h3 {err(w1, refw):.2e} acc {err(w2 - 1.0, refw):.2e} us {timeit(h0):.1f} -> {timeit(h1):.1f}")
