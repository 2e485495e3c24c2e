#!/bin/bash
# Two data-parallel ranks on ONE GPU through the mh_ctx verbs over the stand-in RCCL (tests/fake_rccl), rank 0 under rocprofv3
# (kernel + memory-copy trace): where in the step the tokenizer segment's collective starts (tools/rocpd_dp_order.py).
R=$(pwd); O=$R/gpurun_out/${1:-r6dp}; mkdir -p $O
rm -f $O/dp_segment_order.md; hipcc -O2 -shared -fPIC -o $O/libfake_rccl.so tests/fake_rccl/fake_rccl.cpp || exit 1
cd /tmp && export TMPDIR=/tmp
for EARLY in 1 0; do
  PORT=$((29733 + EARLY))
  MYRIAD_DP_EARLY=$EARLY PYTHONPATH=$R python $R/tests/dp_worker.py 1 2 $PORT allreduce $O/rank1.pt ctx f32 $O/libfake_rccl.so > $O/rank1_$EARLY.log 2>&1 &
  MYRIAD_DP_EARLY=$EARLY PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt$EARLY -o r0 -- python $R/tests/dp_worker.py 0 2 $PORT allreduce $O/rank0.pt ctx f32 $O/libfake_rccl.so > $O/rank0_$EARLY.log 2>&1
  wait
  DB=$(find $O/kt$EARLY -name "*.db" | head -1)
  { echo "## MYRIAD_DP_EARLY=$EARLY ($([ $EARLY = 1 ] && echo 'default: the tokenizer segment starts inside the backward' || echo 'round-5 schedule: one exchange after the whole backward'))"; echo; python $R/tools/rocpd_dp_order.py $DB; echo; } >> $O/dp_segment_order.md 2>&1
  rm -rf $O/kt$EARLY
done
cd $R
rm -rf $O/libfake_rccl.so $O/rank0.pt $O/rank1.pt
cat $O/dp_segment_order.md; tail -3 $O/rank0_1.log
