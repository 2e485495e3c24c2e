#!/bin/bash
# Batch-1 step (BASELINE configs[1]) under a rocprofv3 kernel trace: per-kernel table, stream gaps, phases.  Outputs in gpurun_out/<tag>/.
R=$(pwd); O=$R/gpurun_out/${1:-r5b1}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --batch 1 --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 --steps 4 --warmup 2 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_gaps.py $DB > $O/step_gaps.md 2>&1
python tools/rocpd_llama_chain.py $DB > $O/llama_chain.md 2>&1
rm -rf $O/kt
tail -1 $O/kt.log
head -60 $O/step_breakdown.md
