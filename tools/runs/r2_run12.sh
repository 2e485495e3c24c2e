R=$(pwd); O=$R/gpurun_out/r2/tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 2 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_window.py $DB $O/step_timeline.csv 2> $O/window.log; cat $O/window.log
rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-300
