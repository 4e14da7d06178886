#!/usr/bin/env python
"""MFMA utilisation per kernel from two rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0`:
  pass A: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace
  pass B: --pmc MfmaUtil --kernel-trace        (rocprofv3's derived metric; gfx94x formula on gfx950)
usage: pmc_mfma_summary.py <dirA> <dirB> [out.json]

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all MFMA pipes (32 per v_mfma_f32_32x32x16_bf16, 64 per
v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GUI_ACTIVE / 8 / duration is the average shader
clock of the launch and BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs) the fraction of MFMA-pipe cycles that were busy.
"""
import collections
import csv
import glob
import json
import statistics as st
import sys


def name(x):
    return x.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    da, db = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(glob.glob(da + "/**/*counter_collection.csv", recursive=True)[0])):
        agg[name(x["Kernel_Name"])][x["Counter_Name"]].append(float(x["Counter_Value"]))
    dur = collections.defaultdict(list)
    for x in csv.DictReader(open(glob.glob(da + "/**/*kernel_trace.csv", recursive=True)[0])):
        dur[name(x["Kernel_Name"])].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
    util = collections.defaultdict(list)
    for x in csv.DictReader(open(glob.glob(db + "/**/*counter_collection.csv", recursive=True)[0])):
        if x["Counter_Name"] == "MfmaUtil":
            util[name(x["Kernel_Name"])].append(float(x["Counter_Value"]))
    rows = []
    for k, v in agg.items():
        mb, ga = st.mean(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), st.mean(v.get("GRBM_GUI_ACTIVE", [0]))
        if mb < 1e6 or not ga:
            continue
        us = st.mean(dur[k])
        clk = ga / 8 / us / 1e3                                   # GHz
        frac = mb / (ga / 8 * 1024)
        rows.append(dict(kernel=k, launches=len(dur[k]), us=us, mfma_busy_cycles=mb, clock_ghz=clk, busy_frac=frac,
                         MfmaUtil=st.mean(util[k]) if util.get(k) else None, of_nominal_peak=frac * clk / 2.4))
    rows.sort(key=lambda r: -r["us"] * r["launches"])
    print("| kernel | launches | us / launch (profiled) | SQ_VALU_MFMA_BUSY_CYCLES | avg shader clock (GHz) | MFMA pipes busy | "
          "rocprofv3 MfmaUtil (%) | busy x clock / 2.4 GHz (share of the nominal MFMA peak) |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %d | %.1f | %.4g | %.2f | %.3f | %s | %.3f |" % (
            r["kernel"][:48], r["launches"], r["us"], r["mfma_busy_cycles"], r["clock_ghz"], r["busy_frac"],
            ("%.1f" % r["MfmaUtil"]) if r["MfmaUtil"] is not None else "-", r["of_nominal_peak"]))
    if len(sys.argv) > 3:
        json.dump({r["kernel"]: {k: r[k] for k in ("us", "mfma_busy_cycles", "clock_ghz", "busy_frac", "MfmaUtil", "of_nominal_peak")}
                   for r in rows}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
