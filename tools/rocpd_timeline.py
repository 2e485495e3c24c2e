#!/usr/bin/env python3
"""Time-bucketed occupancy of one batch-8 step of a rocprofv3 (rocpd sqlite) trace of bench.py: per bucket, the fraction of
the bucket each stream has at least one kernel running, the number of launches and the kernel that owns most of the bucket.
Answers "what does the main stream do while the look-ahead ViT runs" without guessing from gap statistics (kernels of a
replayed hipGraph overlap inside one stream, so start-to-end gaps say little there).  The step is the median-length window
between two clamp_ce launches outside the instrumented step.
Usage: python tools/rocpd_timeline.py <results.db> [bucket_us] > profiles/<name>.md"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
bucket = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 and sys.argv[2][0].isdigit() else 250e3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols)
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]
marks = [r[0] for r in db.execute("select start from kernels where name like '%mh_prof_marker_kernel%' order by start")]
wins = [(ce[i], ce[i + 1]) for i in range(len(ce) - 1)
        if not (len(marks) >= 2 and not (ce[i + 1] < marks[0] or ce[i] > marks[1]))]
wins.sort(key=lambda w: w[1] - w[0])
t0, t1 = wins[len(wins) // 2]                       # the median-length window: a steady-state step, no host stall inside
rows = db.execute(f"select name, start, end, {qcol} from kernels where end>? and start<? order by start", (t0, t1)).fetchall()
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:34]
streams = [q for q, _ in collections.Counter(r[3] for r in rows).most_common()]
nb = int((t1 - t0) / bucket) + 1
busy = {q: [[] for _ in range(nb)] for q in streams}
for n, s, e, q in rows:
    s, e = max(s, t0), min(e, t1)
    b0, b1 = int((s - t0) / bucket), int((e - t0 - 1) / bucket)
    for b in range(b0, min(b1, nb - 1) + 1):
        lo, hi = max(s, t0 + b * bucket), min(e, t0 + (b + 1) * bucket)
        if hi > lo:
            busy[q][b].append((lo, hi, short(n)))


def union(iv):
    tot, cur_lo, cur_hi = 0.0, None, None
    for lo, hi, _ in sorted(iv):
        if cur_hi is None or lo > cur_hi:
            if cur_hi is not None:
                tot += cur_hi - cur_lo
            cur_lo, cur_hi = lo, hi
        else:
            cur_hi = max(cur_hi, hi)
    if cur_hi is not None:
        tot += cur_hi - cur_lo
    return tot


print(f"# One step (clamp_ce to clamp_ce, {(t1 - t0) / 1e6:.2f} ms) in {bucket / 1e3:.0f}-us buckets; streams by launch count: {streams}\n")
print("| t ms | " + " | ".join(f"s{q} busy | s{q} n | s{q} top" for q in streams) + " |")
print("|---|" + "---|---|---|" * len(streams))
for b in range(nb):
    cells, any_row = [], False
    for q in streams:
        iv = busy[q][b]
        if not iv:
            cells += ["", "", ""]
            continue
        any_row = True
        top = collections.Counter()
        for lo, hi, n in iv:
            top[n] += hi - lo
        cells += [f"{union(iv) / bucket:.2f}", str(len(iv)), top.most_common(1)[0][0]]
    if any_row:
        print(f"| {b * bucket / 1e6:.2f} | " + " | ".join(cells) + " |")

# optional: `--seq <from_ms> <to_ms>` lists the busiest stream's launches in that part of the window, in order
if "--seq" in sys.argv:
    i = sys.argv.index("--seq")
    a, b = t0 + float(sys.argv[i + 1]) * 1e6, t0 + float(sys.argv[i + 2]) * 1e6
    main = [(n, s, e) for n, s, e, q in rows if q == streams[0] and a <= s < b]
    print(f"\n## stream {streams[0]} launches between {sys.argv[i + 1]} and {sys.argv[i + 2]} ms: {len(main)}, kernel time "
          f"{sum(e - s for _, s, e in main) / 1e6:.2f} ms\n\n| t ms | kernel | us | gap before us |\n|---|---|---|---|")
    prev = None
    for n, s, e in main:
        print(f"| {(s - t0) / 1e6:.3f} | `{short(n)}` | {(e - s) / 1e3:.1f} | {'' if prev is None else f'{(s - prev) / 1e3:.1f}'} |")
        prev = e
