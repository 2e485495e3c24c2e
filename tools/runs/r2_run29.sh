timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_entrypoints_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -15
python tools/decode_bench.py --new 96 2>&1 | tail -1
python tools/decode_bench.py --new 96 --batch 8 2>&1 | tail -1
python tools/generate_phases.py 2>&1 | grep -v amdgpu | tail -8
