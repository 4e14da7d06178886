#!/usr/bin/env python
"""Per-kernel timing at the headline shapes (HIP events on the launch stream)."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters=20, warmup=3):
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
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--G", type=int, default=24)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--D", type=int, default=4)
    a = ap.parse_args()
    from facl_amd import fps, utils_my
    dev = torch.device("cuda:0")
    M, N, D, S, K = a.B * a.G, a.N, a.D, 64, 64
    pts = (torch.rand(M, N, D, device=dev) - 0.5)
    res = {}
    t = timeit(lambda: utils_my.knn_radius_group(pts, S, K, 0.16, want_idx=True))
    byt = M * N * D * 4 + M * S * K * (4 + 4 * D) + M * S * 12
    res["group"] = dict(ms=t, GBps=byt / t / 1e6)
    start = torch.zeros(M, dtype=torch.int32, device=dev)
    t = timeit(lambda: fps.farthest_point_sampling_batch(pts, S, start))
    res["fps_f32"] = dict(ms=t, GBps=(M * N * D * 4) / t / 1e6)
    t = timeit(lambda: fps.fps_sample_data(pts, S, start))
    res["fps+reorder"] = dict(ms=t)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
