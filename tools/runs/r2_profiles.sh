# Round-2 profile refresh at HEAD: bench, kernel trace, three PMC passes (separate, --kernel-trace only), decode
R=$(pwd); O=$R/gpurun_out/r2/final; mkdir -p $O
BENCH_SHAPES=$O/step_gemm_shapes.csv python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json; cut -c1-600 $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/kt.log 2>&1
T=$R/gpurun_out/traffic; rm -rf $T; mkdir -p $T
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $T/$C -o $C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $T/$C.log 2>&1
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $T/MFMA -o MFMA -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $T/MFMA.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_window.py $DB $O/step_timeline.csv 2> $O/window.log
python tools/pmc_traffic.py $T --json $O/gemm256_traffic.json > $O/step_traffic.md 2>&1
mkdir -p $O/pmcflat
for C in FETCH_SIZE WRITE_SIZE MFMA; do
  for K in counter_collection kernel_trace; do F=$(find $T/$C -name "*${K}.csv" | head -1); [ -n "$F" ] && cp $F $O/pmcflat/${C}_${K}.csv; done
done
python tools/pmc_summary.py $O/pmcflat MFMA FETCH_SIZE WRITE_SIZE > $O/step_pmc.md 2>&1
rm -rf $O/kt $O/pmcflat $T/*/
python tools/decode_bench.py --new 96 2>&1 | tail -1 > $O/decode.log
python tools/decode_bench.py --new 96 --batch 8 2>&1 | tail -1 >> $O/decode.log
cat $O/decode.log
head -34 $O/step_pmc.md
