for S in 20 24 26 28 30 33 36 24 28; do
echo "split $S: $(MYRIAD_VIT_SPLIT=$S python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-b1 2>&1 | tail -1 | cut -c160-186)"
done
