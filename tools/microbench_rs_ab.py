#!/usr/bin/env python
"""A/B of the row-streamed forward GEMM (fp16x3, prologue + statistics + segment max as in the step) between the built
library and a variant (--ab path): interleaved rounds in one process."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib
from facl_amd.sa_mlp import _Workspace

ap = argparse.ArgumentParser()
ap.add_argument("--ab", default="")
a = ap.parse_args()
lib = _lib.load_library()
libs = [("base", lib)]
if a.ab:
    lb = ctypes.CDLL(os.path.abspath(a.ab))
    for name, argtypes in _lib.SIGNATURES.items():
        fn = getattr(lb, name); fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name in _lib.RESTYPE_I64 else ctypes.c_int
    libs.append(("exp", lb))
DEV = "cuda:0"; p = _lib.ptr; st = _lib.stream(); ws = _Workspace.get(torch.device(DEV))

def timeit(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 49152
for K, N, seg in ((512, 1024, True), (256, 512, False), (256, 256, False)):
    act = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    ps = torch.rand(K, device=DEV) + 0.5; pt = torch.randn(K, device=DEV) * 0.3
    y = torch.empty(M, N, device=DEV); sums = torch.empty(N, 2, dtype=torch.float64, device=DEV)
    sgn = torch.ones(N, device=DEV) if seg else None
    ymax = torch.empty(M // 64, N, device=DEV) if seg else None
    arg = torch.empty(M // 64, N, dtype=torch.int32, device=DEV) if seg else None
    fns = []
    for tag, L in libs:
        nb = L.facl_gemm_rs_planes_bytes(N, K, 0)
        pl = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _lib.check(L.facl_gemm_rs_planes(p(W), K, N, K, 0, None, 0, 1, p(pl), st), "planes")
        fns.append((tag, (lambda L=L, pl=pl: L.facl_gemm_rs_fwd(p(act), M, K, p(pl), 1, N, p(b), p(ps), p(pt), None, p(y), p(sums),
                                                                p(sgn), p(ymax), p(arg), p(ws), st))))
    res = {t: [] for t, _ in fns}
    for _ in range(5):
        for t, f in fns:
            res[t].append(timeit(f))
    out = "fwd h3 %dx%dx%d  " % (M, K, N)
    for t in res:
        v = sorted(res[t]); out += "%s median %.4f min %.4f ms   " % (t, v[2], v[0])
    print(out, flush=True)
