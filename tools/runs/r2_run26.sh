timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_entrypoints_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-200
