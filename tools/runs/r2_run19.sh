mkdir -p gpurun_out/r2
( time python bench.py ) > gpurun_out/r2/bench_full2.log 2>&1; tail -5 gpurun_out/r2/bench_full2.log | cut -c1-3500
