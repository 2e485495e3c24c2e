mkdir -p gpurun_out/r2
export MYRIAD_DIST_BACKEND=gloo MYRIAD_SINGLE_DEVICE=1
for MODE in allreduce rs_ag; do
MYRIAD_DP_MODE=$MODE timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no-probe --llm-layers 4 --vit-depth 4 --qf-layers 2 > gpurun_out/r2/dp2g_${MODE}.log 2>&1
echo "$MODE: $(grep '^{' gpurun_out/r2/dp2g_${MODE}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["loss"], d["config"]["dp_exchange"], d["config"]["rccl_ranks"])' 2>&1 | tail -1)"
done
tail -3 gpurun_out/r2/dp2g_rs_ag.log | cut -c1-300
