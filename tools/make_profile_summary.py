#!/usr/bin/env python
"""Assemble profiles/rNN_bench_summary.md from the files a GPU run left under gpurun_out/ (bench lines of the four
commands, prof_rNN kernel stats + summary).  usage: make_profile_summary.py r02"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def d_steps(d):
    """executed steps of a default run: 3 graph warm-up + W + K timed + K fenced + 8 eager (the capture pass executes nothing)"""
    return d["warmup"] + 2 * d["steps"] + 8


def main():
    tag = sys.argv[1]
    go, pr = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    names = ["bench_line", "bench_appearance", "bench_dense", "bench_2rank_gloo_rehearsal"]
    if os.path.exists(os.path.join(go, f"{tag}_bench_fps.json")):
        names.insert(1, "bench_fps")
    if os.path.exists(os.path.join(go, f"{tag}_bench_rehearse_dp.json")):
        names.append("bench_rehearse_dp")
    optin = [n for n in ("bench_optin_x3b", "bench_optin_x3") if os.path.exists(os.path.join(go, f"{tag}_{n}.json"))]
    names += optin
    extra = [n for n in ("bench_rehearse_dp_fullgraph", "bench_extract", "bench_extract_unfused") if os.path.exists(os.path.join(go, f"{tag}_{n}.json"))]
    names += extra
    if os.path.exists(os.path.join(go, f"{tag}_views.json")):
        shutil.copy(os.path.join(go, f"{tag}_views.json"), os.path.join(pr, f"{tag}_views.json"))
    for n in names:
        shutil.copy(os.path.join(go, f"{tag}_{n}.json"), os.path.join(pr, f"{tag}_{n}.json"))
    shutil.copy(os.path.join(go, f"prof_{tag}", "r_kernel_stats.csv"), os.path.join(pr, f"{tag}_default_cmd_kernel_stats.csv"))
    L = {n: json.load(open(os.path.join(pr, f"{tag}_{n}.json"))) for n in names}
    for extra in ("traffic.md", "mfma_util.md", "mfma_util.json", "wave_cycles.md"):          # PMC summaries of the same run
        src = os.path.join(go, f"pmc_{tag}_{extra}")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(pr, f"{tag}_pmc_{extra}"))
    # profiles/pmc_traffic.json + pmc_mfma_util.json: what bench.py attaches to its roofline records (same shape only)
    tmd = os.path.join(go, f"pmc_{tag}_traffic.md")
    if os.path.exists(tmd):
        rows = {}
        for line in open(tmd):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            if len(c) == 6 and c[0].startswith("`"):
                rows[c[0].strip("`")] = float(c[5]) * 1e6
        pick = lambda pre: next((v for k, v in rows.items() if k.startswith(pre)), None)
        table = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, `bench.py --steps 2 "
                             "--warmup 1 --no-cpu-baseline --graph 0`), FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                             f"requests at 64 B); see {tag}_pmc_traffic.md.",
                 "shape": {"B": 32, "T": 24, "N": 2048, "D": 3}, "build": tag,
                 "k_sa_bwd1": pick("k_sa_bwd1"), "k_sa_bwd2_sb": pick("k_sa_bwd2_sb"), "k_sa_bwd_w3": pick("k_sa_bwd_w3("), "k_sa_bwd_w3p": pick("k_sa_bwd_w3p"),
                 "k_sa_fwd3_sb": pick("k_sa_fwd3_sb"), "k_sa_fwd2_sb": pick("k_sa_fwd2_sb"), "k_group": pick("k_group"),
                 "k_gemm_rs fwd 49152x512x1024": pick("k_gemm_rs<true, true, false"),
                 "k_gemm_rs fwd 49152x256x512": pick("k_gemm_rs<true, false, false"),
                 "k_wgrad_rs wgrad 49152x1024x512": pick("k_wgrad_rs")}
        json.dump({k: v for k, v in table.items() if v is not None}, open(os.path.join(pr, "pmc_traffic.json"), "w"), indent=1)
    mj = os.path.join(go, f"pmc_{tag}_mfma_util.json")
    if os.path.exists(mj):
        mu = json.load(open(mj))
        pick = lambda pre: next((v for k, v in mu.items() if k.startswith(pre)), None)
        out = {"_comment": f"MFMA-pipe utilisation per launch from rocprofv3 --pmc passes of the {tag} build (see {tag}_pmc_mfma_util.md)",
               "shape": {"B": 32, "T": 24, "N": 2048, "D": 3}, "build": tag}
        for name, pre in (("k_sa_fwd3_sb", "k_sa_fwd3_sb"), ("k_sa_bwd1", "k_sa_bwd1"), ("k_sa_bwd_w3", "k_sa_bwd_w3"),
                          ("k_sa_bwd2_sb", "k_sa_bwd2_sb"), ("k_sa_fwd2_sb", "k_sa_fwd2_sb"),
                          ("k_gemm_rs fwd 49152x512x1024", "k_gemm_rs<true, true, false"),
                          ("k_gemm_rs fwd 49152x256x512", "k_gemm_rs<true, false, false"),
                          ("k_wgrad_rs wgrad 49152x1024x512", "k_wgrad_rs")):
            v = pick(pre)
            if v:
                out[name] = {"busy_frac": round(v["busy_frac"], 4), "MfmaUtil": None if v["MfmaUtil"] is None else round(v["MfmaUtil"], 2),
                             "clock_ghz": round(v["clock_ghz"], 3)}
        json.dump(out, open(os.path.join(pr, "pmc_mfma_util.json"), "w"), indent=1)
    fps_row = ""
    if "bench_fps" in L:
        fps_row = (f'| `python bench.py --fps 1` (the same step with the FPS reorder of every view inside the timed region: north_star lists FPS, '
                   f'the reference loop never calls it) | {L["bench_fps"]["ms_per_step"]} | {L["bench_fps"]["value"]} | `{tag}_bench_fps.json` |\n')
    steps = 3 + d_steps(L["bench_line"])
    d = L["bench_line"]
    summ = open(os.path.join(go, f"prof_{tag}_summary.txt")).read()
    tbl = ("| kernel | ms/launch (in-step HIP events) | arithmetic | bound | achieved (executed) | peak | frac | algorithmic TFLOP/s | hbm_frac | PMC traffic / algorithmic bytes |\n"
           "|---|---|---|---|---|---|---|---|---|---|\n")
    for r in ([d["roofline"]] + d["roofline_more"])[:16]:
        tr = ("%.2f" % (r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "-"
        tbl += "| `%s` | %.4f | %s | %s | %.1f %s | %.0f | %.3f | %.1f | %.3f | %s |\n" % (
            r["kernel"], r["ms_per_launch"], r.get("pipe", "-"), r["bound"], r["achieved"], r["unit"], r["peak"], r["frac"],
            r.get("algorithmic_tflops", 0.0), r["hbm_frac"], tr)
    dp_row = ""
    if "bench_rehearse_dp" in L:
        dp_row = (f'| `python bench.py --rehearse-dp 1` (the DATA-PARALLEL code path on one GPU: every collective of the N > 1 step executes '
                  f'on a 1-rank RCCL group, the step replays as graph segments cut at the collectives; not a scaling number) | '
                  f'{L["bench_rehearse_dp"]["ms_per_step"]} | {L["bench_rehearse_dp"]["value"]} | `{tag}_bench_rehearse_dp.json` |\n')
    if "bench_rehearse_dp_fullgraph" in L:
        dp_row += (f'| `FACL_DP_GRAPH=full python bench.py --rehearse-dp 1` (OPT-IN: the same data-parallel step with its collectives captured INSIDE '
                   f'one graph -- no cuts; 1-rank RCCL rehearsal, never run with N > 1) | {L["bench_rehearse_dp_fullgraph"]["ms_per_step"]} | '
                   f'{L["bench_rehearse_dp_fullgraph"]["value"]} | `{tag}_bench_rehearse_dp_fullgraph.json` |\n')
    if "bench_extract" in L:
        e = L["bench_extract"]
        dp_row += (f'| `python bench.py --config extract` (SURVEY 8 f-1: feature extraction, eval-mode encoder, net3DV_1 as ONE kernel; own metric '
                   f'string) | {e["ms_per_step"]} | {e["value"]} | `{tag}_bench_extract.json` (k_sa_eval {e["roofline"]["ms_per_launch"]} ms = '
                   f'{e["roofline"]["frac"]} of the fp16 MFMA peak; cpu_baseline {e.get("cpu_baseline", {}).get("value", "-")} clips/s) |\n')
    if "bench_extract_unfused" in L:
        dp_row += (f'| `FACL_EVAL_FUSED=0 python bench.py --config extract` (the rounds 1-3 eval path: training kernels with folded constants) | '
                   f'{L["bench_extract_unfused"]["ms_per_step"]} | {L["bench_extract_unfused"]["value"]} | `{tag}_bench_extract_unfused.json` |\n')
    optin_rows = "".join(
        f'| `python bench.py --precision {n.split("_")[-1]}` (OPT-IN arithmetic, not the headline: DESIGN 3.0) | {L[n]["ms_per_step"]} | '
        f'{L[n]["value"]} | `{tag}_{n}.json` |\n' for n in optin)
    open(os.path.join(pr, f"{tag}_bench_summary.md"), "w").write(f'''# Round {int(tag[1:])}: bench lines and rocprofv3 --kernel-trace --stats of the driver's command `python3 bench.py`

MI355X, B=32 T=24 N=2048 D=3.  One profiled run = 1 graph-capture step + 3 graph warm-up steps + 5 warm-up + 20 timed steps
+ 20 fenced steps (HIP-graph replay) + 8 eager steps for the in-step kernel timing of the roofline section + the cpu_baseline leg (CPU only).
Command (on the GPU box, from /tmp with TMPDIR=/tmp):
`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_{tag} -o r -- python3 bench.py`
Raw per-kernel table: `profiles/{tag}_default_cmd_kernel_stats.csv` (the per-step figures below divide by {steps} executed steps).

## bench.py lines of this build, un-profiled (`profiles/{tag}_bench_*.json`; re-run AFTER the PMC tables below were regenerated so that
## their `traffic` / `mfma_util_pmc` fields come from this build -- possibly on another box of the pool than the profiled run: box to box the headline varies by ~7 %)

| command | ms/step | clips/s | file |
|---|---|---|---|
| `python bench.py` (BASELINE configs[1], headline) | {d["ms_per_step"]} | {d["value"]} | `{tag}_bench_line.json` (cpu_baseline {d["cpu_baseline"]["value"]} clips/s on {d["cpu_baseline"]["cores"]} cores) |
{fps_row}| `python bench.py --config appearance --D 4` (configs[2]) | {L["bench_appearance"]["ms_per_step"]} | {L["bench_appearance"]["value"]} | `{tag}_bench_appearance.json` |
| `python bench.py --config dense` (configs[4]: B=8 T=32 N=4096, 3-level SA, fp16-input MFMA) | {L["bench_dense"]["ms_per_step"]} | {L["bench_dense"]["value"]} | `{tag}_bench_dense.json` (own metric string; cpu_baseline {L["bench_dense"].get("cpu_baseline", {}).get("value", "-")} clips/s) |
| `FACL_DIST_BACKEND=gloo python bench.py --gpus 2 --B 16` (2 ranks REHEARSED on one GPU with CPU collectives: launcher, sharded step, SyncBN, all-gather; not a scaling number) | {L["bench_2rank_gloo_rehearsal"]["ms_per_step"]} | {L["bench_2rank_gloo_rehearsal"]["value"]} | `{tag}_bench_2rank_gloo_rehearsal.json` |
{dp_row}{optin_rows}
## Roofline section of `{tag}_bench_line.json` (every heavy entry, timed inside the step)

{tbl}
`roofline` = the first row (the longest kernel).  bf16x6 / fp16x3 kernels are priced with the bf16 / fp16 FLOPs they execute
(6 / 3 per algorithmic multiply-add) against 2,500 TFLOP/s -- a kernel that moved from bf16x6 to fp16x3 executes half the FLOPs,
so its `frac` falls while its time and its algorithmic (fp32-equivalent) TFLOP/s improve: compare that column across rounds; fp32-MFMA kernels (`k_sa_bwd1`, `k_sa_bwd_w3`) with algorithmic FLOPs against
157.3 TFLOP/s; PMC traffic from `profiles/pmc_traffic.json` (offline `--pmc FETCH_SIZE` / `WRITE_SIZE` passes of this build:
`{tag}_pmc_traffic.md`); MFMA-pipe utilisation and the waves' cycle breakdown from the same call: `{tag}_pmc_mfma_util.md`,
`{tag}_pmc_wave_cycles.md`.

## rocprofv3 kernel stats of the profiled run

```
{summ}```
No `Cijk_*` (rocBLAS / hipBLASLt) kernel is left in the trace.
''')
    print("wrote", os.path.join(pr, f"{tag}_bench_summary.md"))


if __name__ == "__main__":
    main()
