timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/step_phases.py --batch 8 2>&1 | tail -13
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-300
