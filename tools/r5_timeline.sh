#!/bin/bash
# kernel trace of a short bench -> time-bucketed stream occupancy of one step (tools/rocpd_timeline.py) + main-stream gaps
R=$(pwd); O=$R/gpurun_out/${1:-r5tl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-probe --no-b1 --no-minigpt4 --steps 6 --warmup 3 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB ${2:-250} > $O/timeline.md 2>&1
python tools/rocpd_gaps.py $DB > $O/step_gaps.md 2>&1
rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-200
