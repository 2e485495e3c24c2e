#!/usr/bin/env python3
"""Recompute bench.py's `roofline` figure for the dominant kernel from the committed profiles, so the JSON line and the
rocprofv3 evidence cannot drift apart:

  flops  = sum over the kernel's rows of profiles/<round>_step_gemm_shapes.csv of launches x 2 M N K
           (written by bench.py's launch profiler, BENCH_SHAPES=...: one whole step, ViT inline)
  time   = the kernel's total in the "profiled step" table of profiles/<round>_step_breakdown.md
           (tools/rocpd_step.py over a rocprofv3 --kernel-trace run of the same command: the region between the two
           mh_prof_marker_kernel launches)
  frac   = flops / time / 2.5 PFLOP/s  (dense bf16 MFMA peak, MI355X_MICROARCH.md)

and compare with profiles/<round>_bench_n1.json (the bench line of the same box).  Also prints the same ratio for the
un-instrumented batch-8 step window (look-ahead ViT on its side stream: the kernels share the chip, so each runs longer).
Usage: python tools/roofline_from_profiles.py [round, default r03]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles")
PEAK = 2500.0
KERNEL = "gemm_x8_kernel" if rnd >= "r04" else "gemm_256_kernel"     # plan kernel 2 (round 4: the hand-scheduled 64-deep loop)

flops, launches = 0.0, 0
for r in csv.DictReader(open(os.path.join(P, f"{rnd}_step_gemm_shapes.csv"))):
    if r["kernel"] == KERNEL:
        flops += int(r["launches"]) * 2.0 * int(r["M"]) * int(r["N"]) * int(r["K"])
        launches += int(r["launches"])

sections, cur = {}, None
for ln in open(os.path.join(P, f"{rnd}_step_breakdown.md")):
    if ln.startswith("## "):
        cur = ln[3:].strip()
    m = re.match(r"\| `(?:void )?" + KERNEL + r"[^|]*\| (\d+) \| ([\d.]+) \|", ln)
    if m and cur and cur not in sections:
        sections[cur] = (int(m.group(1)), float(m.group(2)))

out = {}
for title, (calls, ms) in sections.items():
    key = "profiled_step" if title.startswith("profiled") else ("timed_step_window" if title.startswith("heaviest") else None)
    if key is None:
        continue
    tf = flops / (ms * 1e-3) / 1e12
    out[key] = dict(rocprof_calls=calls, csv_launches=launches, rocprof_total_ms=ms, tflops=round(tf, 1), frac=round(tf / PEAK, 4))
    print(f"{key:18s}: {calls} launches in the trace ({launches} in the csv), {ms:.3f} ms, {flops / 1e12:.2f} TFLOP "
          f"-> {tf:.1f} TFLOP/s = {tf / PEAK:.4f} of the dense bf16 peak")
bj = os.path.join(P, f"{rnd}_bench_n1.json")
if os.path.exists(bj):
    roof = json.load(open(bj)).get("roofline", {})
    print(f"bench line        : {roof.get('launches_per_step')} launches, {roof.get('kernel_ms_per_step')} ms, "
          f"{roof.get('achieved')} TFLOP/s = {roof.get('frac')}")
    if "profiled_step" in out and roof.get("achieved"):
        d = out["profiled_step"]["tflops"] / roof["achieved"] - 1.0
        print(f"rocprof / bench   : {100 * d:+.1f} %  (launch counts {'match' if out['profiled_step']['rocprof_calls'] == launches else 'DIFFER'})")
