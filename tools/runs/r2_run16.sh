MYRIAD_HP_STREAM=1 python tools/step_phases.py --batch 8 2>&1 | tail -13
MYRIAD_HP_STREAM=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-300
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-300
