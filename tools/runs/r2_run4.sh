mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rope or attention" > gpurun_out/r2/t_attn.log 2>&1; tail -30 gpurun_out/r2/t_attn.log
python tools/attn_bench.py 2>&1 | tail -4
