#!/bin/bash
# Round-5 quick look (through gpurun; outputs in gpurun_out/<tag>/): bench line, rocprofv3 kernel trace of a short bench ->
# step breakdown + LLaMA chain tables, phase timeline.  ~4 min of box time.
R=$(pwd); O=$R/gpurun_out/${1:-r5q}; mkdir -p $O
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-minigpt4 --side-steps 3 --steps 4 --warmup 2 > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_step.py $DB > $O/step_breakdown.md 2>&1
python tools/rocpd_llama_chain.py $DB > $O/llama_chain.md 2>&1
python tools/step_phases.py > $O/step_phases.md 2>&1
rm -rf $O/kt
python - <<PY
import json
d = json.loads(open("$O/bench_n1.json").read())
print("ms/step", d["ms_per_step"], "img/s", d["value"], "step_frac", d.get("step_frac_of_peak"), "roofline", d["roofline"]["frac"], "b1", d.get("config1_b1", {}).get("ms_per_step"))
PY
head -30 $O/llama_chain.md; grep -A16 "LLaMA backward" $O/llama_chain.md | head -24; tail -12 $O/step_phases.md
