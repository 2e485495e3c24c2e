timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3
for S in 39 17 12 8 22 0; do
echo "split $S: $(MYRIAD_VIT_SPLIT=$S python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | cut -c100-190)"
done
MYRIAD_VIT_SPLIT=12 python tools/step_phases.py --batch 8 2>&1 | tail -11
