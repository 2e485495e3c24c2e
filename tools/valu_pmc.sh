R=$(pwd); O=$R/gpurun_out/valu; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p -o v -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-b1 > $O/v.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/valu/p/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES": cnt[k] += 1
rows = []
for k, c in agg.items():
    busy = c.get("SQ_BUSY_CU_CYCLES", 0); wc = c.get("SQ_WAVE_CYCLES", 0)
    if busy <= 0: continue
    rows.append((busy, k, cnt[k], c.get("SQ_ACTIVE_INST_VALU", 0) / busy, c.get("SQ_ACTIVE_INST_ANY", 0) / busy, wc / busy))
rows.sort(reverse=True)
print("| kernel | launches | CU-busy Mcycles | VALU-active / CU-busy | any-inst-active / CU-busy | wave-cycles / CU-busy |\n|---|---|---|---|---|---|")
for busy, k, n, v, a, w in rows[:32]:
    print(f"| `{k}` | {n} | {busy / 1e6:.1f} | {v:.2f} | {a:.2f} | {w:.2f} |")
PY
rm -rf $O/p
