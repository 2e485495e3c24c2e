"""How much of a fine-tune step is the host: wall time of the train_step call itself (launch enqueue, Python, allocator) against
the synchronised step time, at batch 8 and batch 1 (BASELINE configs[1]).  A step whose enqueue time equals its wall time is
host-bound: the GPU waits for launches.   python tools/host_enqueue_time.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from myriad_amd import _lib  # noqa: E402
from myriad_amd.myriad import MyriadHIP  # noqa: E402
from myriad_amd.runner import LinearWarmupCosineLRScheduler, setup_seeds  # noqa: E402
from myriad_amd.synthetic import SyntheticWeights, full_config  # noqa: E402


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    setup_seeds(42, 0)
    cfg = full_config()
    model = MyriadHIP(SyntheticWeights(cfg, dev, seed=0, arch="myriad"),
                      dict(fixed_stage=1, fixed_taskstage=0, vit_heads=cfg["vit_heads"], qf_heads=cfg["qf_heads"],
                           llm_heads=cfg["llm_heads"], use_lora=True), device=dev)
    sched = LinearWarmupCosineLRScheduler(None, max_epoch=10, iters_per_epoch=1600, min_lr=0.0, init_lr=1e-4, warmup_steps=0,
                                          warmup_start_lr=1e-6)
    for B in [int(x) for x in sys.argv[1:]] or [8, 1]:
        smp = bench.make_samples(B, cfg["vocab"], 42, dev)
        model.prepare_vit_graph(smp)
        for i in range(3):
            model.train_step(smp, sched.step(0, i), 0.05, next_samples=smp)
        torch.cuda.synchronize()
        n, enq = 10, 0.0
        t0 = time.perf_counter()
        for i in range(n):
            t1 = time.perf_counter()
            model.train_step(smp, sched.step(0, 3 + i), 0.05, next_samples=smp)
            enq += time.perf_counter() - t1
        t_enq_all = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"batch {B}: step {1e3 * wall / n:.2f} ms, inside train_step calls {1e3 * enq / n:.2f} ms, "
              f"loop before the final synchronize {1e3 * t_enq_all / n:.2f} ms")


if __name__ == "__main__":
    main()
