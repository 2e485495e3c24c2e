mkdir -p gpurun_out/r2
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r2/t_all.log 2>&1; tail -40 gpurun_out/r2/t_all.log
