timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
R=$(pwd); O=$R/gpurun_out/r2/prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-probe --no-b1 --steps 3 --warmup 2 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
rm -rf $O/kt
grep -E "attn_seq|summed" $O/step_breakdown.md | head -4
