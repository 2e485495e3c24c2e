R=$(pwd); O=$R/gpurun_out/r2/prof1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
rm -rf $O/kt
head -60 $O/step_breakdown.md; tail -2 $O/kt.log | cut -c1-400
