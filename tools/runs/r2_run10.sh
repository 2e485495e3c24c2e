mkdir -p gpurun_out/r2
python bench.py --steps 10 --warmup 3 > gpurun_out/r2/bench_full.log 2>&1; tail -1 gpurun_out/r2/bench_full.log | cut -c1-3000
export MYRIAD_DIST_BACKEND=gloo MYRIAD_SINGLE_DEVICE=1
for MODE in allreduce rs_ag; do for GD in f32 bf16; do
MYRIAD_DP_MODE=$MODE MYRIAD_DP_GRAD_DTYPE=$GD timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-probe --llm-layers 4 --vit-depth 4 --qf-layers 2 > gpurun_out/r2/dp2_${MODE}_${GD}.log 2>&1
echo "$MODE $GD: $(grep '^{' gpurun_out/r2/dp2_${MODE}_${GD}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["loss"], d["config"]["dp_exchange"])' 2>&1 | tail -1)"
done; done
tail -5 gpurun_out/r2/dp2_rs_ag_f32.log | cut -c1-300
