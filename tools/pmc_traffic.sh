#!/bin/bash
# HBM-side traffic of the step's kernels: two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass),
# --kernel-trace only, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/traffic/{fetch,write}/*.csv
R=$(pwd); O=$R/gpurun_out/traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o $C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $O/$C.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O > $O/summary.md; cat $O/summary.md
