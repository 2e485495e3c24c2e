for ARGS in "--arch mini_gpt4 --stage 0" "--stage 0" "--stage 2" "--batch 16" "--batch 4" "--lora 0" "--no-prefetch-vit"; do
echo "$ARGS: $(python bench.py $ARGS --steps 6 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["seq_len"])')"
done
