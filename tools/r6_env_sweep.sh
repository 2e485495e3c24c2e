#!/bin/bash
# r6_env_sweep.sh "A=1,B=2 A=0,B=2 ..." [steps] [reps]: un-traced bench under each comma-separated environment set, alternating
R=$(pwd); STEPS=${2:-30}; REPS=${3:-2}
for i in $(seq $REPS); do for kv in $1; do
  ms=$(env $(echo $kv | tr ',' ' ') python $R/bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$kv: $ms ms/step"
done; done
