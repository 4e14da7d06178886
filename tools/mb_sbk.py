"""few-row GEMM facl_gemm_fwd: stand-alone time, hot (back-to-back) and with the caches flushed between calls"""
import os, sys, torch
sys.path.insert(0, ".")
from facl_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda")
torch.manual_seed(0)
big = torch.empty(96 * 1024 * 1024, device=dev)      # 384 MB > the 256 MB memory-side cache
def run(f, n, flush):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        if flush: big.add_(1.0)
        e0.record(); f(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
M = 800
for (K, N) in ((1024, 1024), (1024, 512), (512, 768)):
    ws = torch.empty(lib.facl_ws_bytes(), dtype=torch.uint8, device=dev)
    # four different operand sets per timed group, so that `flush` really means cold operands for each call
    sets = []
    for i in range(4):
        a = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        sets.append((a, W, b, y))
    it = [0]
    def f():
        a, W, b, y = sets[it[0] % 4]; it[0] += 1
        _lib.check(lib.facl_gemm_fwd(_lib.ptr(a), M, K, _lib.ptr(W), K, N, _lib.ptr(b), None, None, None, None, 0, _lib.ptr(y), None, _lib.ptr(ws), _lib.stream()), "fwd")
    for _ in range(8): f()
    torch.cuda.synchronize()
    hot = run(f, 20, False) / 4
    cold = run(f, 10, True) / 4
    print(f"{os.environ.get('FACL_LIB','current')[-20:]} fwd {M}x{K}x{N}: per call hot {hot:.1f} us  cold {cold:.1f} us")
