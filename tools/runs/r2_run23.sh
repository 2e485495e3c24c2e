timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -k "lora" 2>&1 | tail -3
python tools/lora_bench.py 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
