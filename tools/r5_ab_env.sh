#!/bin/bash
# quick A/B of environment settings over the un-traced bench (no rocprof): r5_ab_env.sh "VAR=a VAR=b ..." [steps]
R=$(pwd); STEPS=${2:-30}
for kv in $1; do
  ms=$(env $kv python $R/bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$kv: $ms ms/step"
done
