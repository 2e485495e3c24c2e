timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5
python tools/step_phases.py --batch 8 2>&1 | tail -13
python tools/step_phases.py --batch 1 2>&1 | tail -13
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
