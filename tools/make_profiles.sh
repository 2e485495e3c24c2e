#!/bin/bash
# Regenerate the round's measured artifacts on a GPU box (run through gpurun; outputs land in gpurun_out/final/).
R=$(pwd); O=$R/gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
BENCH_SHAPES=$O/step_gemm_shapes.csv python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/ktdec -o dec -- python $R/tools/decode_bench.py --new 96 > $O/ktdec.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_stats.py $(find $O/ktdec -name "*.db" | head -1) > $O/decode_trace.md 2>&1
python tools/gemv_bench.py 2>&1 | grep "M=" > $O/gemv.log
python tools/decode_bench.py --new 96 2>&1 | tail -1 > $O/decode.log
python tools/decode_bench.py --new 96 --batch 8 2>&1 | tail -1 >> $O/decode.log
rm -rf $O/kt $O/ktdec          # the sqlite traces are large; the summaries are what gets committed
tail -3 $O/smoke.log; cat $O/bench_n1.json; cat $O/decode.log
