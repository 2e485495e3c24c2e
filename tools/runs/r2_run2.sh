mkdir -p gpurun_out/r2
for I in 0 1; do for MB in 8 0; do
  MYRIAD_G256I=$I MYRIAD_G256_MB=$MB python tools/gemm_mb_bench.py > gpurun_out/r2/g_i${I}_mb${MB}.log 2>&1
  grep -E "check|TF" gpurun_out/r2/g_i${I}_mb${MB}.log | grep -v "^| M" 
done; done
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r2/t_kernels.log 2>&1; tail -3 gpurun_out/r2/t_kernels.log
