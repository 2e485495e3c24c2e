#!/bin/bash
# rocprofv3 kernel trace of bench.py -> step breakdown + main-stream gaps (outputs in gpurun_out/<tag>/)
R=$(pwd); O=$R/gpurun_out/${1:-r4kt}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH_SHAPES=$O/step_gemm_shapes_profiled.csv timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline ${BENCH_ARGS} > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
grep '^{' $O/kt.log | tail -1 > $O/bench_under_rocprof.json
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_gaps.py $DB > $O/step_gaps.md 2>&1
python tools/rocpd_llama_chain.py $DB > $O/llama_chain.md 2>&1
python tools/rocpd_timeline.py $DB 250 ${TIMELINE_ARGS} > $O/step_timeline.md 2>&1
rm -rf $O/kt
cat $O/llama_chain.md
