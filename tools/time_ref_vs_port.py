#!/usr/bin/env python
"""Time the REFERENCE's own CPU training step against oracle.step.train_step (the "port" that bench.py's
cpu_baseline leg runs on the GPU box, where /root/reference does not exist), side by side on the same inputs.

Build container only (imports /root/reference/training_code with `.cuda()` neutralised, like tools/make_goldens.py).
BASELINE.md section 4 asks for agreement within +-10 %; the numbers this prints are recorded in BASELINE.md section 2.

    python tools/time_ref_vs_port.py [--threads 8]
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/training_code")
torch.Tensor.cuda = lambda self, *a, **k: self            # noqa: E731  (harness-side only)
torch.nn.Module.cuda = lambda self, *a, **k: self         # noqa: E731

import utils_my as R_utils                                  # noqa: E402  reference
import cn3d_model_conbag as R_model                         # noqa: E402  reference
from oracle import encoder as E, step as OS                 # noqa: E402
from oracle.weights import formula_state_dict               # noqa: E402


def ref_step_fn(B, G, N, D, S=64, K=64):
    opt = SimpleNamespace(temperal_num=3, knn_K=K, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=S,
                          sample_num_level2=S, INPUT_FEATURE_NUM=D, Num_Class=512, batchSize=B, pooling="concatenation",
                          SAMPLE_NUM=N)
    net = R_model.PointNet_Plus_fine(opt, gost=G, sample_num_level1=S, knn_K=K)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in formula_state_dict(D).items()})
    net.train()
    crit = torch.nn.CrossEntropyLoss()
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06)

    def step(pts):
        if N == 512:
            xt, yt = R_utils.group_points_3DV(pts, opt)
        else:
            xt, yt = R_utils.group_points_3DV_2048(pts, K, S, SAMPLE_NUM=N)
        x, code, x_nor, x_global = net(xt, yt, 1)
        loss = R_utils.circle_contrast(G, x, B, crit) + R_utils.global_contrast(G, x_global, x, opt, crit)
        optim.zero_grad()
        loss.backward()
        optim.step()
        return float(loss)
    return step


def port_step_fn(B, G, N, D, S=64, K=64):
    sd = E.clone_state(formula_state_dict(D))
    opt = OS.AdamState(sd)
    order = np.arange(G)
    r2 = 0.06 if N == 512 else 0.16

    def step(pts):
        return OS.train_step(sd, opt, pts, B, G, S, K, r2, order)["loss"]
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    for name, (B, G, N, D) in (("C1 B4 T8 N512 D4", (4, 8, 512, 4)), ("B8 T24 N2048 D3", (8, 24, 2048, 3))):
        # interleaved in ONE process (ref step, port step, ref step, ...): back-to-back blocks of one kind pick up
        # allocator / page-cache state from whatever ran before and differ by 30 % between runs on this box
        fns = {"reference": ref_step_fn(B, G, N, D), "port": port_step_fn(B, G, N, D)}
        g = torch.Generator().manual_seed(0)
        ts = {"reference": [], "port": []}
        for it in range(a.steps + 1):
            pts = OS.view_major(torch.rand(B, G, N, D, generator=g) - 0.5).contiguous()
            for kind in ("reference", "port"):
                t0 = time.time()
                fns[kind](pts.clone())
                ts[kind].append(time.time() - t0)
        res = {k: float(np.median(v[1:])) for k, v in ts.items()}
        print("%-18s reference %.3f s/step  port %.3f s/step  port/reference x%.2f  (%d threads, median of %d after 1 warm-up)"
              % (name, res["reference"], res["port"], res["port"] / res["reference"], a.threads, a.steps))


if __name__ == "__main__":
    main()
