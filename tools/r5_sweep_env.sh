#!/bin/bash
# A/B of one environment variable over the bench: for each VALUE, the un-traced ms/step and the traced per-kernel averages.
# usage: r5_sweep_env.sh VAR "v1 v2 .." kernel-name-substring [more substrings]   (MYRIAD_HIP_DEBUG_LIB=1 is set: debug-only knobs work)
R=$(pwd); VAR=$1; VALS=$2; shift 2
export MYRIAD_HIP_DEBUG_LIB=1
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  ms=$(python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  rm -rf /tmp/kt_$v; timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$v -o kt -- python $R/bench.py --no-cpu-baseline --no-probe --no-b1 --steps 3 --warmup 2 > /tmp/kt_$v.log 2>&1
  echo "$VAR=$v: $ms ms/step | $(python $R/tools/rocpd_kavg.py $(find /tmp/kt_$v -name '*.db' | head -1) "$@")"
done
