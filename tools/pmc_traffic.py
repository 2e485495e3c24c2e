#!/usr/bin/env python3
"""Per-kernel HBM-side bytes per launch from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B read requests as 64 B for wide coalesced
streams, so read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is taken as reported (KiB).  Writes profiles-ready markdown and,
with --json PATH, the dominant GEMM kernel's per-launch traffic for bench.py's roofline.traffic field."""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
by_grid = {}                                     # counter -> {workgroups of an unsplit gemm_256 launch: [launches, sum]}


def load(counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    gy_of, gx_of = {}, {}                        # Dispatch_Id -> grid y (K splits) / workgroups in x, from the kernel trace of the same pass
    for f in glob.glob(f"{d}/{counter}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            gy_of[r["Dispatch_Id"]] = int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))
            gx_of[r["Dispatch_Id"]] = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    for f in glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
            gy = gy_of.get(r["Dispatch_Id"], 1)
            key = (name, gy if "gemm" in name else 0)
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
            if "gemm_256" in name or "gemm_x8" in name or "gemm_x4" in name:                    # keyed "tiles x K splits": the launch grids bench.py's profiler reports
                e = by_grid.setdefault(counter, {}).setdefault(f"{gx_of.get(r['Dispatch_Id'], 0)}x{gy}", [0, 0.0])
                e[0] += 1
                e[1] += float(r["Counter_Value"])
    return agg
fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
rows = []
for k in fe:
    n = fe[k][0]
    rd = 2.0 * fe[k][1] * 1024 / n
    w = wr.get(k, [1, 0.0])
    wb = w[1] * 1024 / max(1, w[0])
    rows.append((k, n, rd, wb))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
print("| kernel | K splits | launches | read MB/launch (2 x FETCH_SIZE) | write MB/launch | total GB over the run |")
print("|---|---|---|---|---|---|")
for (name, gy), n, rd, wb in rows[:24]:
    print(f"| `{name}` | {gy or ''} | {n} | {rd / 1e6:.1f} | {wb / 1e6:.1f} | {(rd + wb) * n / 1e9:.2f} |")
if "--json" in sys.argv:
    dom = [r for r in rows if "gemm_x8" in r[0][0]] or [r for r in rows if "gemm_256" in r[0][0]]
    if dom:
        (name, gy), n, rd, wb = dom[0]
        grids = {}
        for gx, (cnt, tot) in by_grid.get("FETCH_SIZE", {}).items():
            wcnt, wtot = by_grid.get("WRITE_SIZE", {}).get(gx, [1, 0.0])
            grids[gx] = {"launches": cnt, "read_bytes_per_launch": 2.0 * tot * 1024 / cnt,
                         "write_bytes_per_launch": wtot * 1024 / max(1, wcnt)}
        json.dump({"kernel": name, "launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wb,
                   "traffic_bytes_per_launch": rd + wb, "by_grid": grids,
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); read = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB; plan-kernel-2 GEMM (gemm_x8_kernel; gemm_256_kernel before round 4) launches of python bench.py --steps 2 --warmup 1 --no-probe, by launch grid (tiles x K splits)"},
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
