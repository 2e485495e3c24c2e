#!/usr/bin/env python3
"""The LLaMA forward and backward chains of one timed batch-8 step in a rocprofv3 (rocpd sqlite) trace of bench.py: every
launch on the stream that runs clamp_ce, from the first lora_down / qkv GEMM of the forward to the loss kernel, and from the
loss kernel to the last rmsnorm_bwd of the backward -- kernel time, idle time between consecutive launches, and per
kernel -> next-kernel transition the mean gap.  Usage: python tools/rocpd_llama_chain.py <results.db>"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols)
marks = [r[0] for r in db.execute("select start from kernels where name like '%mh_prof_marker_kernel%' order by start")]
ce = db.execute(f"select start, end, {qcol} from kernels where name like '%clamp_ce%' order by start").fetchall()
ce = [c for c in ce if not (len(marks) >= 2 and marks[0] - 60e6 < c[0] < marks[1] + 60e6)]
# batch-8 steps come first in bench.py; take the middle one of the first half
c0 = ce[len(ce) // 4]
q = c0[2]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:40]


def chain(t0, t1, title):
    rows = db.execute(f"select name, start, end from kernels where {qcol}=? and start>=? and start<=? order by start", (q, t0, t1)).fetchall()
    rows = [(short(n), s, e) for n, s, e in rows]
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = [(rows[i + 1][1] - rows[i][2], rows[i][0], rows[i + 1][0]) for i in range(len(rows) - 1)]
    print(f"## {title}: {len(rows)} launches, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {sum(max(g[0], 0) for g in gaps) / 1e6:.2f} ms\n")
    per = collections.defaultdict(lambda: [0, 0.0])
    for _, s, e in [(n, s, e) for n, s, e in rows]:
        pass
    byk = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        byk[n][0] += 1; byk[n][1] += e - s
    print("| kernel | launches | total ms | avg us |\n|---|---|---|---|")
    for n, (c, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{n}` | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} |")
    tr = collections.defaultdict(lambda: [0, 0.0])
    for g, a, b in gaps:
        tr[(a, b)][0] += 1; tr[(a, b)][1] += max(g, 0)
    print("\n| after | before | count | idle ms | avg gap us |\n|---|---|---|---|---|")
    for (a, b), (c, g) in sorted(tr.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{a}` | `{b}` | {c} | {g / 1e6:.3f} | {g / c / 1e3:.2f} |")
    print()


first_fwd = db.execute(f"select min(start) from kernels where {qcol}=? and start>? and start<? and (name like '%lora_down%' or name like '%rmsnorm_fwd%')",
                       (q, c0[0] - 30e6, c0[0])).fetchone()[0]
chain(first_fwd, c0[0], "LLaMA forward (first norm / lora_down .. clamp_ce)")
last_bwd = db.execute(f"select max(start) from kernels where {qcol}=? and start>? and start<? and name like '%rmsnorm_bwd%'",
                      (q, c0[0], c0[0] + 30e6)).fetchone()[0]
chain(c0[0], last_bwd, "LLaMA backward (clamp_ce .. last rmsnorm_bwd)")
