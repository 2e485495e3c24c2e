#!/bin/bash
# Two data-parallel ranks on ONE GPU (gloo between the processes, both on cuda:0): exercises the N > 1 control flow of
# bench.py / runner.DataParallel -- both exchange modes, both wire dtypes -- on a single-GPU box.  It is a control-flow
# harness, not an RCCL measurement (RCCL refuses two ranks per device); run it through gpurun.  (Round 4: the same control flow
# is a test now -- tests/test_dp_gpu.py, bit-compared with a one-process run; this script remains the full-width bench variant.)
O=gpurun_out/dp2; mkdir -p $O
export MYRIAD_DIST_BACKEND=gloo MYRIAD_SINGLE_DEVICE=1
for MODE in allreduce rs_ag; do for GD in f32 bf16; do
  MYRIAD_DP_MODE=$MODE MYRIAD_DP_GRAD_DTYPE=$GD timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-probe --llm-layers 4 --vit-depth 4 \
    --qf-layers 2 > $O/dp2_${MODE}_${GD}.log 2>&1
  echo "$MODE $GD: $(grep '^{' $O/dp2_${MODE}_${GD}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["loss"], d["config"]["dp_exchange"], d["config"]["rccl_ranks"])' 2>&1 | tail -1)"
done; done
