mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_self_sup.py -x -q -m gpu > gpurun_out/r2/t_ss.log 2>&1; tail -25 gpurun_out/r2/t_ss.log
