#!/usr/bin/env python
"""bench.py -- contrastive-step clips/sec on synthetic clouds (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of the reference loop body (cn3d_train_motion_GL.py:224-335) on a
batch already resident in HBM: view-major reshape -> kNN/radius grouping -> encoder forward -> global + circle
loss -> backward -> Adam.  Workload at N=1 = BASELINE.json configs[1]: motion stream, B=32, T=24 views,
N=2048 points.  Weak scaling: every rank processes its own B=32 clips; the embeddings all-gather, SyncBN
all-reduces and the gradient all-reduce are the exchange steps (facl_amd/dist.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_MFMA_BF16_TFLOPS = 2500.0   # same guide, BF16 dense (the split-bf16 kernels run on this pipe)
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--B", type=int, default=32, help="clips per GPU")
    ap.add_argument("--T", type=int, default=24, help="views per clip (reference: gost / num_crop)")
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--D", type=int, default=3, help="input channels (north_star: 3-ch; checkpoints: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fps", type=int, default=0, help="1: FPS-reorder the views on the GPU inside the timed step")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step as one HIP graph (single GPU only)")
    ap.add_argument("--cpu-clips", type=int, default=8, help="clips in the bounded CPU-baseline sample")
    return ap.parse_args()


def make_opt(a):
    from types import SimpleNamespace
    return SimpleNamespace(temperal_num=3, knn_K=64, ball_radius=0.16, ball_radius2=0.25, sample_num_level1=64,
                           sample_num_level2=64, INPUT_FEATURE_NUM=a.D, Num_Class=512, batchSize=a.B,
                           pooling="concatenation", SAMPLE_NUM=a.N)


def _baseline_metric():
    """BASELINE.json's metric string, verbatim (the file ships with the repository)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "contrastive-step clips/sec (B=32,T=24,N=2048) at 1/2/4/8 MI355X"


def _pmc_traffic(a, kernel):
    # HBM bytes per launch: PMC counters are collected offline (rocprofv3 --pmc, profiles/pmc_traffic.json) at this
    # exact shape; reported only when the shape matches, else null.
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pm["shape"] == {"B": a.B, "T": a.T, "N": a.N, "D": a.D}:
            return pm.get(kernel)
    except Exception:
        pass
    return None


def dominant_kernel_roofline(a, eager_step, batches, nsteps=8):
    """Average duration of the three heaviest kernels measured INSIDE the training step: `nsteps` extra eager steps
    (after the timed region) with HIP events recorded on the launch stream around the three launches
    (facl_amd/_lib.py: timed).  `roofline` is the single longest kernel, k_sa_bwd1 (HBM-side: it streams y2 in and dz2
    out, its MFMA part is small); `roofline_more` carries the two MFMA kernels (k_sa_fwd3_sb and the largest
    k_gemm_sb), priced against the bf16 MFMA peak with the bf16 FLOPs they EXECUTE (6 per fp32 multiply-add, see
    DESIGN.md).  The brackets include the two ~5 us partial-sum reduction launches that follow each of these kernels
    inside its C entry point."""
    from facl_amd import _lib
    nunits = a.B * a.T * 64
    M, K, N = nunits, 512, 1024
    glabel = "facl_gemm_fwd %dx%dx%d" % (M, K, N)
    _lib.TIMING = {"facl_sa_bwd1": [], "facl_sa_fwd3": [], glabel: []}
    try:
        for i in range(nsteps):
            eager_step(batches[i % 2], epoch=0)
        ms1, ms3, msg = _lib.timing_ms("facl_sa_bwd1"), _lib.timing_ms("facl_sa_fwd3"), _lib.timing_ms(glabel)
    finally:
        _lib.TIMING = None
    # ---- k_sa_bwd1: y2 (16 KiB/unit) + coef (1 KiB) + arg (256 B) in, dz2 (16 KiB) out
    bytes1 = float(nunits) * (2 * 16384 + 1024 + 256)
    ach1 = bytes1 / (ms1 * 1e-3) / 1e9
    main = {"kernel": "k_sa_bwd1", "bound": "hbm", "achieved": round(ach1, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": round(ach1 / PEAK_HBM_GBPS, 4), "traffic": _pmc_traffic(a, "k_sa_bwd1"),
            "ms_per_launch": round(ms1, 4), "algorithmic_bytes_per_launch": bytes1, "measured": "in-step, %d steps" % nsteps}
    # ---- k_sa_fwd3_sb: 64 -> 256 layer of the SA-MLP, 2*64*256 FLOP per position, x6 bf16 MFMA FLOPs
    fl3 = 2.0 * 64 * 256 * nunits * 64
    more = [{"kernel": "k_sa_fwd3_sb", "bound": "mfma", "achieved": round(6 * fl3 / (ms3 * 1e-3) / 1e12, 1),
             "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(6 * fl3 / (ms3 * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS, 4),
             "traffic": _pmc_traffic(a, "k_sa_fwd3_sb"), "ms_per_launch": round(ms3, 4), "algorithmic_flops_per_launch": fl3,
             "algorithmic_tflops": round(fl3 / (ms3 * 1e-3) / 1e12, 1), "executed_bf16_flops_per_launch": 6 * fl3}]
    # ---- largest tail GEMM (net3DV_3.6: 512 -> 1024 over the M*S centroid rows), forward
    if msg is not None:
        flg = 2.0 * M * K * N
        more.append({"kernel": "k_gemm_sb<KC,KC> %dx%dx%d (+ fused max over the 64 centroids)" % (M, K, N), "bound": "mfma",
                     "achieved": round(6 * flg / (msg * 1e-3) / 1e12, 1), "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(6 * flg / (msg * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS, 4), "traffic": None,
                     "ms_per_launch": round(msg, 4), "algorithmic_flops_per_launch": flg,
                     "algorithmic_tflops": round(flg / (msg * 1e-3) / 1e12, 1), "executed_bf16_flops_per_launch": 6 * flg})
    return main, more


def cpu_baseline(a):
    """The oracle's training step (a port of the reference's op sequence onto torch-CPU) timed on the host
    cores, on a bounded sample: `cpu_clips` clips of the same (T, N, D) shape."""
    from oracle import step as OS
    from oracle import encoder as E
    from oracle.weights import formula_state_dict
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    Bc, G = a.cpu_clips, a.T
    sd = E.clone_state(formula_state_dict(a.D))
    opt = OS.AdamState(sd)
    g = torch.Generator().manual_seed(0)
    order = np.arange(G)
    times = []
    for it in range(3):
        pts = OS.view_major(torch.rand(Bc, G, a.N, a.D, generator=g) - 0.5)
        t0 = time.time()
        OS.train_step(sd, opt, pts, Bc, G, 64, 64, 0.16 if a.N != 512 else 0.06, order)
        times.append(time.time() - t0)
    t = min(times[1:])
    return {"value": round(Bc / t, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"B={Bc} clips x T={G} views x N={a.N} pts, D={a.D}: 1 warm-up + 2 timed steps "
                      f"(best {t:.2f} s/step) of oracle.step.train_step"}


def main():
    a = parse()
    from facl_amd import dist as fdist
    rank, world = fdist.init_from_env()
    # one rank per GPU; `% device_count` only matters when rehearsing N>1 on a single-GPU box
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from facl_amd.cn3d_model_conbag import PointNet_Plus
    from facl_amd.train_common import ContrastiveStep, GraphedStep, synthetic_batch

    torch.manual_seed(1)                       # opt.manualSeed = 1 (cn3d_train_motion_GL.py:142-144)
    np.random.seed(1)
    opt = make_opt(a)
    net = PointNet_Plus(opt, gost=a.T).to(dev).train()
    net.bn_reduce_fn = fdist.make_bn_reduce_fn()
    use_graph = bool(a.graph) and world == 1
    optim = torch.optim.Adam(net.parameters(), lr=0.0003, betas=(0.5, 0.999), eps=1e-06, capturable=use_graph, fused=True)
    step = ContrastiveStep(net, optim, opt, a.T, fps_reorder=bool(a.fps))
    eager_step = step
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    batches = [synthetic_batch(a.B, a.T, a.N, a.D, dev, gen) for _ in range(2)]   # resident in HBM
    mode = "eager"
    if use_graph:
        try:
            step = GraphedStep(step, batches[0], a.T)
            mode = "hipgraph"
        except Exception as e:                          # never lose the measurement to a capture problem
            print("graph capture failed (%s: %s); running eager" % (type(e).__name__, e), file=sys.stderr)
            net.zero_grad(set_to_none=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(batches[i % 2], epoch=0)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss, _, _ = step(batches[i % 2], epoch=0)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # in-step kernel timing: extra eager steps on EVERY rank (they contain the collectives), reported by rank 0
    rl_main, rl_more = dominant_kernel_roofline(a, eager_step, batches)
    out = None
    if rank == 0:
        clips = a.B * world * a.steps / dt
        out = {"metric": _baseline_metric(), "value": round(clips, 2), "unit": "clips/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "dtype_note": "fp32 storage and accumulation; dense contractions as exact 3-way bf16 splits on the bf16 MFMA (6 products per multiply-add, fp32-grade accuracy)",
               "config": {"workload": f"motion stream, B={a.B}/GPU T={a.T} N={a.N} D={a.D}, S=64 K=64, full "
                                      f"cn3d_model_conbag encoder, global+circle loss, backward, Adam",
                          "global_batch": a.B * world, "parallelism": f"dp{world}", "launch": mode},
               "final_loss": final_loss}
        out["roofline"], out["roofline_more"] = rl_main, rl_more
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
