#!/usr/bin/env python
"""Per-kernel timing of the SA point-MLP passes at the headline shape (HIP events on the launch stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facl_amd import _lib                      # noqa: E402
from facl_amd.sa_mlp import _Workspace         # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=768)
    ap.add_argument("--D", type=int, default=3)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--ab", type=str, default="", help="second libfacl_hip build: A/B both in THIS process, interleaved")
    a = ap.parse_args()
    lib = _lib.load_library()
    lib_b = None
    if a.ab:
        import ctypes
        lib_b = ctypes.CDLL(os.path.abspath(a.ab))
        for name, argtypes in _lib.SIGNATURES.items():
            fn = getattr(lib_b, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name in _lib.RESTYPE_I64 else ctypes.c_int
    dev = torch.device("cuda:0")
    nunits, D = a.M * 64, a.D
    P = nunits * 64
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)          # noqa: E731
    x = R(P, D) * 0.3
    y2f, dz2f = R(nunits * 4096), torch.empty(nunits * 4096, device=dev)
    bnc2 = torch.rand(5, 64, device=dev, generator=g) + 0.5
    bnc3 = torch.rand(5, 256, device=dev, generator=g) + 0.5
    W1, b1 = R(64, D) * 0.3, R(64) * 0.1
    W2, b2, W3, b3 = R(64, 64) * 0.1, R(64) * 0.1, R(256, 64) * 0.1, R(256) * 0.1
    l1tab = torch.empty(64, 8, device=dev)
    _lib.check(lib.facl_sa_l1tab(_lib.ptr(W1), _lib.ptr(b1), D, None, None, _lib.ptr(l1tab), None, None, _lib.stream()), "l1tab")
    # fp16x3 activation bounds (csrc/common.h) for these random inputs: every slot of the two buffers holds the bits of 8.0
    am = torch.full((2, _lib.AMAX_WORDS), 8.0, device=dev).view(torch.int32)
    sgn = torch.ones(256, device=dev)
    ymax, arg = torch.empty(nunits, 256, device=dev), torch.randint(0, 64, (nunits, 256), dtype=torch.uint8, device=dev)
    coef = R(nunits, 256)
    G3, h3 = R(64, 64) * 0.01, R(64) * 0.01
    bw2 = torch.rand(4, 64, device=dev, generator=g)
    ws = _Workspace.get(dev)
    s64 = torch.empty(64, 2, dtype=torch.float64, device=dev)
    s256 = torch.empty(256, 2, dtype=torch.float64, device=dev)
    o3 = torch.empty(20544, dtype=torch.float64, device=dev)
    o2 = torch.empty(4608, dtype=torch.float64, device=dev)
    st = _lib.stream()
    p = _lib.ptr
    F = 2.0 * 64 * 64 * P          # flops of one 64x64 layer over all positions
    def make(lib):
      return {
        "fwd2": (lambda: lib.facl_sa_fwd2(p(x), nunits, D, p(l1tab), p(W2), p(b2), p(dz2f), p(s64), p(ws), p(am[0]), st), F),
        "fwd3": (lambda: lib.facl_sa_fwd3(p(y2f), nunits, p(bnc2[2]), p(bnc2[3]), p(W3), p(b3), p(sgn), p(ymax), p(arg), p(s256), p(ws), st), 4 * F),
        "fwd3h": (lambda: lib.facl_sa_fwd3_h3(p(y2f), nunits, p(bnc2[2]), p(bnc2[3]), p(W3), p(b3), p(sgn), p(ymax), p(arg), p(s256), p(ws), p(am[1]), st), 4 * F),
        "bwd0": (lambda: lib.facl_sa_bwd0(p(coef), p(ymax), nunits, p(bnc3), p(coef), p(s256), p(ws), st), 0),
        "bwd1": (lambda: lib.facl_sa_bwd1(p(y2f), nunits, p(bnc2), p(G3), p(h3), p(W3), p(coef), p(arg), p(dz2f), p(s64), p(ws), p(am[1]), st), F),
        "bwd_w3": (lambda: lib.facl_sa_bwd_w3(p(y2f), nunits, p(bnc2), p(coef), p(arg), p(o3), p(ws), p(am[1]), st), 0.75 * F),
        "bwd2": (lambda: lib.facl_sa_bwd2(p(dz2f), p(y2f), p(x), nunits, D, p(bw2), p(W2), p(l1tab), p(o2), p(ws), p(am[0]), st), 2.5 * F),
      }
    kernels = make(lib)
    kernels_b = make(lib_b) if lib_b is not None else None
    for name, (fn, fl) in kernels.items():
        if a.only and name not in a.only.split(","):
            continue
        rc = fn()
        assert rc == 0, (name, rc)
        if kernels_b is None:
            ms = timeit(fn)
            print(f"{name:7s} {ms:8.3f} ms   {fl / ms / 1e9:7.1f} TFLOP/s (MFMA work only)")
        else:
            fb = kernels_b[name][0]
            assert fb() == 0
            ta, tb = [], []
            for _ in range(5):                      # interleaved rounds in one process (guide rule 24)
                ta.append(timeit(fn, iters=6, warmup=1))
                tb.append(timeit(fb, iters=6, warmup=1))
            ta.sort(); tb.sort()
            print(f"{name:7s} A(base) median {ta[2]:.3f} min {ta[0]:.3f} ms | B(exp) median {tb[2]:.3f} min {tb[0]:.3f} ms | B/A {tb[2]/ta[2]:.3f}")


if __name__ == "__main__":
    main()
