timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c100-200
python tools/step_phases.py --batch 8 2>&1 | tail -11
python tools/step_phases.py --batch 1 2>&1 | tail -11
export MYRIAD_DIST_BACKEND=gloo MYRIAD_SINGLE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 2 --no-probe --llm-layers 4 --vit-depth 4 --qf-layers 2 2>&1 | grep '^{' | cut -c100-200
