#!/bin/bash
# Round-6 measured artifacts on a GPU box (through gpurun; outputs in gpurun_out/<tag>/, copy the summaries to profiles/):
#   bench line + per-shape launch profile; rocprofv3 kernel trace of the same command -> step breakdown (the profiled step
#   between the two marker kernels first) + main-stream gaps; three PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE; each with
#   --kernel-trace only) -> per-kernel table + per-launch-grid traffic of the dominant kernel; decode bench + trace.
R=$(pwd); O=$R/gpurun_out/${1:-r6p}; mkdir -p $O
# clocks and socket power WHILE the bench runs (one rocm-smi sample per 0.5 s, in the background; the DVFS ceiling under matrix load)
( for i in $(seq 1 90); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > $O/smi_during_bench.jsonl &
SMI=$!
BENCH_SHAPES=$O/step_gemm_shapes.csv python bench.py --steps 60 --warmup 3 > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json
kill $SMI 2>/dev/null
python - <<PY > $O/smi_during_bench.md
import json
sc, pw = [], []
for ln in open("$O/smi_during_bench.jsonl"):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    for card in d.values():
        for k, v in card.items():
            if "sclk" in k and "MHz" in str(v).replace("Mhz", "MHz"):
                sc.append(int("".join(c for c in str(v).split("(")[-1] if c.isdigit())))
            if "Power" in k and "W" in k:
                try: pw.append(float(v))
                except Exception: pass
print("rocm-smi sampled every 0.5 s while python bench.py --steps 60 ran (model build, warm-up, 60 timed steps, probe, batch-1 line, cpu baseline)")
if sc: print(f"sclk MHz: samples {len(sc)}, max {max(sc)}, sorted tail {sorted(sc)[-8:]}")
if pw: print(f"socket power W: samples {len(pw)}, max {max(pw):.0f}, sorted tail {[round(x) for x in sorted(pw)[-8:]]}")
PY
cd /tmp && export TMPDIR=/tmp
BENCH_SHAPES=$O/step_gemm_shapes_profiled.csv timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-minigpt4 --side-steps 3 > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/ktdec -o dec -- python $R/tools/decode_bench.py --new 96 > $O/ktdec.log 2>&1
mkdir -p $O/pmc
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/m1 -o m1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-minigpt4 --side-steps 3 > $O/pmc/m1.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$C -o $C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-minigpt4 --side-steps 3 > $O/pmc/$C.log 2>&1
done
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
grep '^{' $O/kt.log | tail -1 > $O/bench_under_rocprof.json
python tools/rocpd_stats.py $DB 8 > $O/kernel_trace.md 2>&1
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_gaps.py $DB > $O/step_gaps.md 2>&1
python tools/rocpd_llama_chain.py $DB > $O/llama_chain.md 2>&1
python tools/rocpd_stats.py $(find $O/ktdec -name "*.db" | head -1) > $O/decode_trace.md 2>&1
python tools/pmc_traffic.py $O/pmc --json $O/gemm256_traffic.json > $O/step_traffic.md 2>&1
mkdir -p $O/pmc/flat; for t in m1 FETCH_SIZE WRITE_SIZE; do for f in $(find $O/pmc/$t -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do cp $f $O/pmc/flat/; done; done
python tools/pmc_summary.py $O/pmc/flat m1 FETCH_SIZE WRITE_SIZE > $O/step_pmc.md 2>&1
rm -rf $O/pmc                     # raw per-dispatch csv files are large; the summaries are what gets committed
python tools/decode_bench.py --new 96 2>&1 | tail -1 > $O/decode.log
python tools/decode_bench.py --new 96 --batch 8 2>&1 | tail -1 >> $O/decode.log
MYRIAD_DECODE_FUSED=0 python tools/decode_bench.py --new 96 2>&1 | tail -1 >> $O/decode.log
python tools/gemm_vendor_calib.py > $O/vendor_calib.md 2>&1
python tools/step_phases.py > $O/step_phases.md 2>&1
python tools/lora_bench.py > $O/lora_kernels.log 2>&1
python tools/vit_alone.py > $O/vit_alone.log 2>&1
bash tools/r6_probe_vit_cost.sh > $O/vit_cost.log 2>&1
python tools/gemm_split_xcd_ab.py > $O/gemm_split_xcd_ab.log 2>&1
nproc > $O/host.txt; rocm-smi --showclocks --showpower 2>/dev/null | head -30 >> $O/host.txt
rm -rf $O/kt $O/ktdec
head -c 1200 $O/bench_n1.json; echo; cat $O/decode.log; head -12 $O/step_traffic.md
