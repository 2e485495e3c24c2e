mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rope or attention" > gpurun_out/r2/t_attn.log 2>&1; tail -5 gpurun_out/r2/t_attn.log
python tools/attn_bench.py 2>&1 | tail -3
B=1 python tools/attn_bench.py 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_a.log 2>&1
python - <<'PY'
import json
for n in ("a",):
    try:
        l=[x for x in open(f"gpurun_out/r2/bench_{n}.log") if x.startswith("{")][-1]
        d=json.loads(l); print(n, d["value"], d["ms_per_step"], d["loss"], d.get("config1_b1"))
    except Exception as e: print(n, "ERR", e); print(open(f"gpurun_out/r2/bench_{n}.log").read()[-2000:])
PY
