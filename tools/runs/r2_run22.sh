python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --lora 0 2>&1 | tail -1 | cut -c1-200
python tools/lora_bench.py 2>&1 | tail -8
