timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | cut -c100-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | cut -c100-200
