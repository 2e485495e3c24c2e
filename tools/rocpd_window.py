#!/usr/bin/env python3
"""Dump one batch-8 step (the heaviest window between two clamp_ce launches) of a rocprofv3 rocpd trace as a CSV timeline:
kernel, stream/queue, start offset (us), duration (us), grid -- for offline critical-path analysis (tools/timeline_report.py).
Usage: python tools/rocpd_window.py <results.db> <out.csv>"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols, file=sys.stderr)
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]
best, bt = None, -1
for i in range(len(ce) - 1):
    s = db.execute("select sum(end-start) from kernels where start>=? and start<?", (ce[i], ce[i + 1])).fetchone()[0] or 0
    if s > bt:
        best, bt = (ce[i], ce[i + 1]), s
t0, t1 = best
pick = [c for c in ("stream_id", "queue_id", "stream", "queue", "tid") if c in cols]
gx = [c for c in ("grid_size_x", "grid_x") if c in cols]
gy = [c for c in ("grid_size_y", "grid_y") if c in cols]
wx = [c for c in ("workgroup_size_x", "workgroup_x") if c in cols]
wy = [c for c in ("workgroup_size_y", "workgroup_y") if c in cols]
sel = ["name", "start", "end"] + pick + gx[:1] + gy[:1] + wx[:1] + wy[:1]
rows = db.execute(f"select {', '.join(sel)} from kernels where start>=? and start<? order by start", (t0, t1)).fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "start_us", "dur_us"] + sel[3:])
    for r in rows:
        w.writerow([re.sub(r"\(.*", "", r[0])[:60], f"{(r[1] - t0) / 1e3:.2f}", f"{(r[2] - r[1]) / 1e3:.2f}"] + list(r[3:]))
print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches", file=sys.stderr)
