mkdir -p gpurun_out/r2
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/r2/t_all.log 2>&1; tail -15 gpurun_out/r2/t_all.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_a.log 2>&1
MYRIAD_SWIGLU_FUSED=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b1 > gpurun_out/r2/bench_b.log 2>&1
python - <<'PY'
import json
for n in ("a","b"):
    try:
        l=[x for x in open(f"gpurun_out/r2/bench_{n}.log") if x.startswith("{")][-1]
        d=json.loads(l); print(n, d["value"], d["ms_per_step"], d["loss"], d["roofline"]["frac"], d["roofline"].get("all_gemm"), d.get("config1_b1"))
    except Exception as e: print(n, "ERR", e); print(open(f"gpurun_out/r2/bench_{n}.log").read()[-2000:])
PY
