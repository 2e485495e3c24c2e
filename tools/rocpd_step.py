#!/usr/bin/env python3
"""(With bench.py's launch profiler on, the region between its two marker kernels is printed first.)
Kernel breakdown of ONE step out of a rocprofv3 (rocpd sqlite) trace of bench.py: a window between two consecutive
clamp_ce launches (loss of step i .. loss of step i+1 = one full backward + optimiser + next forward).  bench.py runs
batch-8 steps and then a few batch-1 steps (config1_b1): the heaviest window is a batch-8 step, the lightest complete
one a batch-1 step; both are printed.
Usage: python tools/rocpd_step.py <results.db> > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ce = [r[0] for r in db.execute("select start from kernels where name like '%clamp_ce%' order by start")]


def window(t0, t1):
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where start>=? and start<? "
                      "group by name order by 3 desc", (t0, t1)).fetchall()
    return rows, sum(r[2] for r in rows)


marks = [r[0] for r in db.execute("select start from kernels where name like '%mh_prof_marker_kernel%' order by start")]
wins = [(window(ce[i], ce[i + 1]), ce[i], ce[i + 1]) for i in range(len(ce) - 1)]
if len(marks) >= 2:
    # bench.py's profiled step (ViT forward inline, every GEMM launch bracketed by HIP events): exactly the launches of
    # roofline.* in the bench line -- tools/roofline_from_profiles.py recomputes that figure from this table
    rows, tot = window(marks[0], marks[1])
    print(f"## profiled step = between the two mh_prof_marker_kernel launches (bench.py LaunchProfile; ViT inline)\n\n"
          f"wall {(marks[1] - marks[0]) / 1e6:.2f} ms, summed kernel time {tot / 1e6:.2f} ms\n")
    print("| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|")
    for n, c, s_, a_ in rows[:40]:
        nm = re.sub(r"\(.*", "", n)[:110]
        print(f"| `{nm}` | {c} | {s_ / 1e6:.3f} | {a_ / 1e3:.2f} | {100 * s_ / tot:.1f} |")
    print()
    wins = [w for w in wins if w[2] < marks[0] or w[1] > marks[1]]      # the other sections: un-instrumented steps only
_top = max(w[0][1] for w in wins)
heavy = min((w for w in wins if w[0][1] >= 0.85 * _top), key=lambda w: w[2] - w[1])   # a batch-8 step without host pauses in it
light = min((w for w in wins if w[0][1] > 0.2 * heavy[0][1]), key=lambda w: w[0][1])
for title, ((rows, tot), t0, t1) in (("heaviest window = one batch-8 step", heavy), ("lightest window = one batch-1 step (config1_b1)", light)):
    if title.startswith("lightest") and light is heavy:
        break
    print(f"## {title}\n\nwall {(t1 - t0) / 1e6:.2f} ms, summed kernel time {tot / 1e6:.2f} ms "
          f"(side-stream optimiser kernels overlap the main stream)\n")
    print("| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|")
    for n, c, s_, a_ in rows[:40]:
        nm = re.sub(r"\(.*", "", n)[:110]
        print(f"| `{nm}` | {c} | {s_ / 1e6:.3f} | {a_ / 1e3:.2f} | {100 * s_ / tot:.1f} |")
    print()
