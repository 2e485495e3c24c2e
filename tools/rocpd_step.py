#!/usr/bin/env python3
"""Kernel breakdown of ONE step out of a rocprofv3 (rocpd sqlite) trace of bench.py: the window between the last two
clamp_ce launches (loss of step i .. loss of step i+1 = one full backward + optimiser + next forward).
Usage: python tools/rocpd_step.py <results.db> > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]
t0, t1 = ce[-2], ce[-1]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where start>=? and start<? "
                  "group by name order by 3 desc", (t0, t1)).fetchall()
tot = sum(r[2] for r in rows)
print(f"One step (window between the last two clamp_ce launches): wall {(t1 - t0) / 1e6:.2f} ms, summed kernel time {tot / 1e6:.2f} ms "
      f"(side-stream optimiser kernels overlap the main stream)\n")
print("| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|")
for n, c, s, a in rows[:48]:
    nm = re.sub(r"\(.*", "", n)[:110]
    print(f"| `{nm}` | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {100 * s / tot:.1f} |")
