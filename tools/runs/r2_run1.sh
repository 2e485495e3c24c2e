mkdir -p gpurun_out/r2
MYRIAD_G256_MB=8 python tools/gemm_mb_bench.py > gpurun_out/r2/mb8.log 2>&1
python tools/gemm_mb_bench.py > gpurun_out/r2/mbauto.log 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r2/t_kernels.log 2>&1; tail -3 gpurun_out/r2/t_kernels.log
MYRIAD_G256_MB=8 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_mb8.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_mbauto.log 2>&1
tail -12 gpurun_out/r2/mb8.log; tail -12 gpurun_out/r2/mbauto.log
python - <<'PY'
import json
for n in ("mb8","mbauto"):
    try:
        l=[x for x in open(f"gpurun_out/r2/bench_{n}.log") if x.startswith("{")][-1]
        d=json.loads(l); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("all_gemm"))
    except Exception as e: print(n, "ERR", e)
PY
