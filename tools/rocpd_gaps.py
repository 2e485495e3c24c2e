#!/usr/bin/env python3
"""Where one batch-8 step's wall time goes on the MAIN stream of a rocprofv3 (rocpd sqlite) trace of bench.py: kernel time,
idle gaps between consecutive launches, and which kernel transitions own the gaps.  The step is the heaviest un-instrumented
window between two clamp_ce launches (see tools/rocpd_step.py).
Usage: python tools/rocpd_gaps.py <results.db> > profiles/<name>.md"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols)
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]
marks = [r[0] for r in db.execute("select start from kernels where name like '%mh_prof_marker_kernel%' order by start")]
# the step window: among the un-instrumented clamp_ce-to-clamp_ce windows whose kernel time is within 15 % of the heaviest (the
# batch-8 steps), the one with the SHORTEST wall -- a window that straddles the end of the timed region carries host pauses
cands = []
for i in range(len(ce) - 1):
    if len(marks) >= 2 and not (ce[i + 1] < marks[0] or ce[i] > marks[1]):
        continue
    s = db.execute("select sum(end-start) from kernels where start>=? and start<?", (ce[i], ce[i + 1])).fetchone()[0] or 0
    cands.append((s, ce[i + 1] - ce[i], (ce[i], ce[i + 1])))
top = max(c[0] for c in cands)
best = min((c for c in cands if c[0] >= 0.85 * top), key=lambda c: c[1])[2]
t0, t1 = best
rows = db.execute(f"select name, start, end, {qcol} from kernels where start>=? and start<? order by start", (t0, t1)).fetchall()
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:48]
byq = collections.defaultdict(list)
for n, s, e, q in rows:
    byq[q].append((short(n), s, e))
print(f"# One batch-8 step on the main stream: kernel time and idle gaps\n\nwindow {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches "
      f"on {len(byq)} streams\n")
print("| stream | launches | kernel ms | first start (ms) | last end (ms) |\n|---|---|---|---|---|")
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"| {q} | {len(ks)} | {sum(e - s for _, s, e in ks) / 1e6:.2f} | {(ks[0][1] - t0) / 1e6:.2f} | {(max(e for _, _, e in ks) - t0) / 1e6:.2f} |")
main = max(byq.values(), key=len)
busy = sum(e - s for _, s, e in main)
gaps = [(main[i + 1][1] - main[i][2], main[i][0], main[i + 1][0]) for i in range(len(main) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f"\nmain stream: {len(main)} launches, kernel time {busy / 1e6:.2f} ms, idle between launches {sum(g[0] for g in pos) / 1e6:.2f} ms "
      f"({len(pos)} gaps, median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.2f} us)\n")
hist = collections.Counter()
for g, _, _ in pos:
    b = "<1" if g < 1e3 else "1-2" if g < 2e3 else "2-4" if g < 4e3 else "4-8" if g < 8e3 else "8-16" if g < 16e3 else "16-64" if g < 64e3 else ">=64"
    hist[b] += g
print("| gap size (us) | total ms |\n|---|---|")
for b in ("<1", "1-2", "2-4", "4-8", "8-16", "16-64", ">=64"):
    print(f"| {b} | {hist[b] / 1e6:.3f} |")
trans = collections.defaultdict(lambda: [0, 0])
for g, a, b in pos:
    trans[(a, b)][0] += 1
    trans[(a, b)][1] += g
print("\n| after kernel | before kernel | count | idle ms | avg us |\n|---|---|---|---|---|")
for (a, b), (c, g) in sorted(trans.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"| `{a}` | `{b}` | {c} | {g / 1e6:.3f} | {g / c / 1e3:.2f} |")
