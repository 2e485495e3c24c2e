#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv output per kernel name.
Usage: python tools/pmc_summary.py <dir> m1 m2 m3 > profiles/xxx.md   (m1: MFMA busy, m2: FETCH_SIZE, m3: WRITE_SIZE)"""
import collections
import csv
import re
import sys


def short(n):
    return re.sub(r"\(.*", "", n)[:70]


def load(d, tag):
    rows = list(csv.DictReader(open(f"{d}/{tag}_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in rows:
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in cnt.items()}


def durations(d, tag):
    rows = list(csv.DictReader(open(f"{d}/{tag}_kernel_trace.csv")))
    dur = collections.defaultdict(float)
    for r in rows:
        dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return dur


def main():
    d, t1, t2, t3 = sys.argv[1:5]
    a1, n1 = load(d, t1)
    a2, _ = load(d, t2)
    a3, _ = load(d, t3)
    dur = durations(d, t1)
    keys = [k for k in sorted(dur, key=lambda k: -dur[k]) if not k.startswith("void at::") and "rocclr" not in k][:24]
    print("| kernel | launches | total us | MFMA busy % of 1024 SIMD-cycles | FETCH_SIZE x2 (MB) | WRITE_SIZE (MB) | HBM GB/s |")
    print("|---|---|---|---|---|---|---|")
    for k in keys:
        us = dur[k]
        mf = a1[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = a1[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0        # summed over 8 XCDs
        util = 100.0 * mf / (1024.0 * gui) if gui else 0.0
        fetch = 2.0 * a2[k].get("FETCH_SIZE", 0.0) / 1024.0   # KB -> MB, x2 gfx950 correction (MI355X_MICROARCH.md, HBM)
        write = a3[k].get("WRITE_SIZE", 0.0) / 1024.0
        bw = (fetch + write) * 1e6 / (us * 1e-6) / 1e9 if us else 0.0
        print(f"| `{k}` | {n1.get(k, 0)} | {us:.0f} | {util:.1f} | {fetch:.0f} | {write:.0f} | {bw:.0f} |")


if __name__ == "__main__":
    main()
