# PMC comparison of the two 256-column main loops on one unsplit LLaMA shape (warm operands)
REPO=$(pwd); mkdir -p $REPO/gpurun_out/r2/pmc
cd /tmp && export TMPDIR=/tmp
for I in 0 1; do
  i=0
  for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    i=$((i+1))
    MYRIAD_G256I=$I MYRIAD_G256_MB=8 timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $REPO/gpurun_out/r2/pmc/i${I}_p$i -o p -- python $REPO/tools/gemm_one.py 1184 12288 4096 0 12 > $REPO/gpurun_out/r2/pmc/i${I}_p$i.log 2>&1
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for I in (0, 1):
    agg = collections.defaultdict(float); n = collections.defaultdict(set); dur=[]
    for f in glob.glob(f"gpurun_out/r2/pmc/i{I}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_256" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
    for f in glob.glob(f"gpurun_out/r2/pmc/i{I}_p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_256" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"## MYRIAD_G256I={I}: kernel us (profiled) median {sorted(dur)[len(dur)//2]:.1f} min {min(dur):.1f} n={len(dur)}")
    for k in sorted(agg): print(f"| {k} | {agg[k] / max(1, len(n[k])):.5g} |")
PY
rm -rf gpurun_out/r2/pmc/*/
