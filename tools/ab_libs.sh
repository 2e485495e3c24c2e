#!/bin/bash
# A/B of two builds of the library on one box (copies of libmyriad_hip.so placed under gpurun_libs/, not tracked): ab_libs.sh libA libB [rounds] [steps] ["extra bench.py arguments"]
R=$(pwd)
for i in $(seq 1 ${3:-3}); do
  for l in $1 $2; do
    cp $R/gpurun_libs/$l $R/myriad_amd/libmyriad_hip.so
    ms=$(python $R/bench.py --steps ${4:-40} --warmup 4 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 ${5:-} 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$l: $ms ms/step"
  done
done
