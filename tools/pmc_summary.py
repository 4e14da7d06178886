#!/usr/bin/env python
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes).

usage: pmc_summary.py <dir_fetch> <dir_write>
FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): the read side is doubled here.
"""
import collections
import csv
import glob
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fe:
        rd = 2.0 * fe[k][0] * 1024
        w = wr.get(k, (0.0, 0))[0] * 1024
        rows.append((rd + w, k, fe[k][1], fe[k][0], rd, w))
    print("| kernel | launches | FETCH_SIZE KiB (raw) | read MB (x2) | write MB | traffic MB |")
    print("|---|---|---|---|---|---|")
    for t, k, n, raw, rd, w in sorted(rows, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
        print(f"| `{k[:60]}` | {n} | {raw:.0f} | {rd/1e6:.1f} | {w/1e6:.1f} | {t/1e6:.1f} |")


if __name__ == "__main__":
    main()
