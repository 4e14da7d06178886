#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel ms/step and category totals."""
import collections
import csv
import glob
import sys


def main():
    d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    cat, calls = collections.Counter(), collections.Counter()
    tot = 0.0
    for r in rows:
        n, t, c = r["Name"], float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps
        tot += t
        k = ("rocBLAS GEMM" if "Cijk" in n else "torch small ops" if ("at::" in n or "rocclr" in n) else
             "SA-MLP HIP" if ("k_sa_" in n or "k_x_mom" in n) else "tail rows HIP" if ("k_rows" in n or "k_segmax" in n)
             else "group HIP" if ("k_group" in n or "k_fps" in n) else "loss HIP" if "k_contrast" in n else
             "finalize HIP" if ("k_reduce" in n or "k_bn" in n or "k_l1tab" in n or "k_fc_finalize" in n or "k_fc_bwd_consts" in n
                                or "k_loss_finish" in n or "k_fc_reduce" in n) else
             "tail rows HIP" if ("k_fc_" in n or "k_col_sums" in n or "k_viewmax" in n or "k_normalize" in n or "k_scale_rows" in n) else
             "GEMM HIP (row-streamed)" if ("k_gemm_rs" in n or "k_wgrad_rs" in n or "k_rs_planes" in n or "k_wg_sum" in n) else
             "GEMM HIP (LDS-staged)" if ("k_gemm" in n or "k_sum_slices" in n) else "other")
        cat[k] += t
        calls[k] += c
    for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
        print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step calls/step={int(r['Calls'])/steps:6.1f} "
              f"avg={float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:84]}")
    print("-- categories")
    for k, v in cat.most_common():
        print(f"{v:8.3f} ms/step {calls[k]:7.1f} launches/step  {k}")
    print(f"{tot:8.3f} ms/step total GPU kernel time")


if __name__ == "__main__":
    main()
