#!/usr/bin/env python
"""clips/s of the GPU view construction (facl_amd.views.build_views: host draws in the reference's NumPy order +
one HIP launch per batch) next to the NumPy restatement of the dataset class's per-sample pipeline (oracle/views.py =
cn3D_data_set.py:285-350, what the reference runs in its 16 DataLoader workers) on the host cores.  Prints one JSON line.

    python tools/time_views.py [--B 32] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from facl_amd.views import build_views, synthetic_raw_clip
    from oracle import views as OV
    clips = [synthetic_raw_clip(b, np.float32, P=4000, Kp=1500, R1=2000, R2=1000) for b in range(a.B)]
    rng = np.random.RandomState(0)
    for _ in range(3):
        build_views(clips, rng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = build_views(clips, rng)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / a.iters
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    for _ in range(3):
        build_views(clips, device_rng=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = build_views(clips, device_rng=gen)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / a.iters
    t0 = time.perf_counter()
    for _ in range(3):
        want = OV.collate_view_major([OV.get_item(rng, *c) for c in clips])
    t_cpu = (time.perf_counter() - t0) / 3
    print(json.dumps({"what": "view construction, 10 views x 512 points x 4 channels per clip (SURVEY 8 f-3)", "B": a.B,
                      "build_views_ms_per_batch": round(1e3 * t_gpu, 3), "build_views_clips_per_s": round(a.B / t_gpu, 1),
                      "build_views_device_rng_ms_per_batch": round(1e3 * t_dev, 3), "build_views_device_rng_clips_per_s": round(a.B / t_dev, 1),
                      "oracle_numpy_ms_per_batch_1_core": round(1e3 * t_cpu, 3), "oracle_numpy_clips_per_s_1_core": round(a.B / t_cpu, 1),
                      "host_cores": len(os.sched_getaffinity(0)),
                      "note": "build_views = host draws (reference NumPy order, one core) + H2D of sources / indices / noise + one "
                              "HIP launch; the reference spreads the NumPy pipeline over 16 DataLoader workers"}))


if __name__ == "__main__":
    main()
