for A in 1 0; do
MYRIAD_ATTN_SEQ=$A python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | cut -c100-200
done
python tools/attn_bench.py 2>&1 | tail -12
