timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/lora_bench.py 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
