#!/bin/bash
# round-6 A/B of one environment switch over the un-traced bench, alternating: r6_ab.sh VAR [steps] [reps]
R=$(pwd); V=$1; STEPS=${2:-30}; REPS=${3:-3}
for i in $(seq $REPS); do for val in 1 0; do
  ms=$(env $V=$val python $R/bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$V=$val: $ms ms/step"
done; done
