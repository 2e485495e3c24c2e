#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg duration, like `--stats`.
Usage: python tools/rocpd_stats.py gpurun_out/prof/r01_results.db [n_steps] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {namecol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx in rows[:40]:
        print(f"| `{short(n)}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
    # GEMM kernels by K-split count (grid y; 1 = plain launch, the population bench.py's roofline block times)
    gy = next((c for c in cols if c in ("grid_size_y", "grid_y", "grid_size.y")), None)
    wy = next((c for c in cols if c in ("workgroup_size_y", "workgroup_y")), None)
    if gy:
        expr = f"{gy}/{wy}" if wy else gy
        print(f"\n| GEMM kernel | K splits | calls | total ms | avg us |\n|---|---|---|---|---|")
        for n, y, c, s_, a in db.execute(f"select {namecol}, {expr}, count(*), sum(end-start), avg(end-start) from kernels "
                                          f"where {namecol} like '%gemm_%' group by {namecol}, {expr} order by 4 desc"):
            print(f"| `{short(n)}` | {int(y)} | {c} | {s_/1e6:.3f} | {a/1e3:.2f} |")
    print(f"\nTotal kernel time {tot/1e6:.2f} ms over the trace ({tot/1e6/steps:.2f} ms per step for {steps:g} steps incl. warm-up).")


if __name__ == "__main__":
    main()
