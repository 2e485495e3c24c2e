mkdir -p gpurun_out/r2
python tools/step_phases.py --batch 8 2>&1 | tail -14
python tools/step_phases.py --batch 8 --no-prefetch 2>&1 | tail -13
python tools/step_phases.py --batch 1 2>&1 | tail -14
