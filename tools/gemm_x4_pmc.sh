#!/bin/bash
# PMC counters of the four-wave GEMM on one big shape (each group its own pass, --kernel-trace only)
R=$(pwd); O=$R/gpurun_out/x4pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from myriad_amd import ops, _lib
L = _lib.load(); dev = torch.device("cuda:0"); ops.ensure_workspace(dev)
M, N, K = 8192, 8192, 8192
a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for impl in (0, 1):
    L.mh_set_option(b"gemm256_impl", impl)
    for _ in range(3): ops.gemm(a, b, out=out, variant=12)
torch.cuda.synchronize()
PY
for G in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC"; do
  T=$(echo $G | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/$T -o p -- python /tmp/one.py > $O/$T.log 2>&1
  F=$(find $O/$T -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open("$F")):
    k = r["Kernel_Name"][:40]
    if "gemm" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
done
