R=$(pwd); O=$R/gpurun_out/r2/final; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_n1.json; cut -c1-400 $O/bench_n1.json
export MYRIAD_DIST_BACKEND=gloo MYRIAD_SINGLE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-probe --llm-layers 4 --vit-depth 4 --qf-layers 2 > $O/dp2.log 2>&1
grep '^{' $O/dp2.log | cut -c1-200
