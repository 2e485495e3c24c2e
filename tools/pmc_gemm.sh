#!/bin/bash
# PMC passes for one GEMM variant: tools/pmc_gemm.sh M N K variant tag   (writes gpurun_out/pmc_<tag>/)
M=$1; N=$2; K=$3; V=$4; TAG=$5
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $REPO/tools/gemm_one.py $M $N $K $V 10 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg): print(f"| {k} | {agg[k] / max(1, len(n[k])):.4g} |")
PY
