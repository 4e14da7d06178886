#!/usr/bin/env python
"""Per-kernel wave-cycle breakdown from two rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0`:
  pass A: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace
  pass C: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
                SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace
usage: pmc_wave_summary.py <dirA> <dirC> [substring filter]"""
import collections, csv, glob, statistics as st, sys


def name(x):
    return x.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for x in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
        agg[name(x["Kernel_Name"])][x["Counter_Name"]].append(float(x["Counter_Value"]))
    dur = collections.defaultdict(list)
    for x in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
        dur[name(x["Kernel_Name"])].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
    return agg, dur


def main():
    A, durA = load(sys.argv[1])
    C, _ = load(sys.argv[2])
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    print("| kernel | us | clock GHz | MFMA busy | of nominal | COEXEC/BUSY | VALU/MFMA | issuing | issue-stalled | parked | LDS stall |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for k, v in A.items():
        if flt not in k:
            continue
        mb, ga = st.mean(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), st.mean(v.get("GRBM_GUI_ACTIVE", [0]))
        if mb < 1e6 or not ga or k not in C:
            continue
        us = st.mean(durA[k])
        clk = ga / 8 / us / 1e3
        busy = mb / (ga / 8 * 1024)
        c = {n: st.mean(x) for n, x in C[k].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        rows.append((us * len(durA[k]), "| `%s` | %.1f | %.2f | %.3f | %.3f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
            k[:60], us, clk, busy, busy * clk / 2.4, c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1),
            4 * c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1), c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc)))
    for _, r in sorted(rows, reverse=True):
        print(r)


if __name__ == "__main__":
    main()
