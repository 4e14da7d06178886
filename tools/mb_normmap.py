import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from facl_amd import _lib
lib = _lib.load_library(); p = _lib.ptr
dev = torch.device("cuda:0")
M, C, K = 768, 512, 64
x = torch.randn(M, C, device=dev); W = torch.randn(K, C, device=dev) * 0.05
xn = torch.empty(M, C, device=dev); code = torch.empty(M, K, device=dev)
big = torch.empty(64 * 1024 * 1024, device=dev)   # 256 MB: flush caches between calls
def run(n, flush):
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        if flush: big.fill_(1.0)
        ev0.record()
        _lib.check(lib.facl_normalize_map(p(x), M, C, p(W), K, p(xn), p(code), _lib.stream()), "nm")
        ev1.record(); torch.cuda.synchronize(); tot += ev0.elapsed_time(ev1)
    return tot / n * 1e3
run(5, False)
print("FACL_NORMMAP4", os.environ.get("FACL_NORMMAP4"), "hot %.1f us  flushed %.1f us" % (run(50, False), run(30, True)))
ref = torch.nn.functional.normalize(x.double(), dim=1)
print("xn err", float((xn.double() - ref).abs().max()), "code err", float((code.double() - ref @ W.double().t()).abs().max()))
