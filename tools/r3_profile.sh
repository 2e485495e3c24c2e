#!/bin/bash
# Round-3 measured artifacts on a GPU box (through gpurun): bench line + per-shape launch profile, rocprofv3 kernel trace of
# the same command -> step breakdown (profiled region between the marker kernels first), main-stream gap analysis.
R=$(pwd); O=$R/gpurun_out/${1:-r3p}; mkdir -p $O
BENCH_SHAPES=$O/step_gemm_shapes.csv python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
BENCH_SHAPES=$O/step_gemm_shapes_profiled.csv timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
tail -1 $O/kt.log > $O/bench_under_rocprof.json
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_gaps.py $DB > $O/step_gaps.md 2>&1
rm -rf $O/kt
head -c 1500 $O/bench_n1.json; echo; head -30 $O/step_gaps.md
