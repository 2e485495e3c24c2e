#!/usr/bin/env python3
"""Average duration of the kernels whose names match the given substrings, over the heaviest clamp_ce-to-clamp_ce window
(= one batch-8 step) of a rocprofv3 (rocpd sqlite) trace of bench.py.  Usage: rocpd_kavg.py <results.db> name [name ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]
marks = [r[0] for r in db.execute("select start from kernels where name like '%mh_prof_marker_kernel%' order by start")]
wins = [(ce[i], ce[i + 1]) for i in range(len(ce) - 1)]
if len(marks) >= 2:
    wins = [w for w in wins if w[1] < marks[0] or w[0] > marks[1]]


def ktime(w):
    return db.execute("select sum(end-start) from kernels where start>=? and start<?", w).fetchone()[0] or 0


_kt = {x: ktime(x) for x in wins}
_top = max(_kt.values())
w = min((x for x in wins if _kt[x] >= 0.85 * _top), key=lambda x: x[1] - x[0])     # a batch-8 step without host pauses in it
out = [f"window {(w[1] - w[0]) / 1e6:.2f} ms"]
for pat in sys.argv[2:]:
    c, s, a = db.execute("select count(*), sum(end-start), avg(end-start) from kernels where start>=? and start<? and name like ?",
                         (w[0], w[1], f"%{pat}%")).fetchone()
    out.append(f"{pat}: {c} x {(a or 0) / 1e3:.1f} us = {(s or 0) / 1e6:.3f} ms")
print(" | ".join(out))
