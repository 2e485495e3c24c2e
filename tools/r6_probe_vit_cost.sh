#!/bin/bash
# What the look-ahead ViT costs the step: the un-traced bench at full depth against a 1-block ViT (its work ~ free), alternating
R=$(pwd)
for i in 1 2; do for d in 39 1; do
  ms=$(python $R/bench.py --vit-depth $d --steps 30 --warmup 4 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "vit-depth $d: $ms ms/step"
done; done
