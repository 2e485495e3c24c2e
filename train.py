#!/usr/bin/env python3
"""Training entry point: counterpart of the reference's train.py (:44-112) for the MI355X hot path.

    python train.py --cfg-path train_configs/loraadapter_simple_myriad_finetune.yaml [--options run.max_epoch=1 ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --cfg-path ...

Same sequence as the reference's main(): job id -> Config(yaml + model defaults + --options) -> distributed init
(env:// from torchrun, RCCL) -> seeds (seed + rank) -> datasets -> registry.get_model_class(arch).from_config(model_cfg)
-> RunnerBase(...).train().  The reference's YAML files are read unchanged.
"""
import argparse
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Training")
    parser.add_argument("--cfg-path", required=True, help="path to configuration file.")
    parser.add_argument("--options", nargs="+",
                        help="override some settings in the used config, the key-value pair in xxx=yyy format will be merged "
                             "into config file (deprecate), change to --cfg-options instead.")
    return parser.parse_args(argv)


def now() -> str:
    return datetime.datetime.now().strftime("%Y%m%d%H%M")[:-1]          # common/utils.py:35-38


def main(argv=None):
    import torch
    from myriad_amd import myriad  # noqa: F401  (registers the model classes)
    from myriad_amd.config import Config
    from myriad_amd.datasets import build_datasets
    from myriad_amd.registry import registry
    from myriad_amd.runner import RunnerBase, init_distributed, setup_seeds

    job_id = now()                                   # before the distributed init: shared by every rank (train.py:88-89)
    cfg = Config(parse_args(argv))
    rank, world, local = init_distributed()
    setup_seeds(int(cfg.run_cfg.get("seed", 42)), rank)
    if rank == 0:
        cfg.pretty_print()
    datasets = build_datasets(cfg.datasets_cfg, split="train")
    model_cfg = cfg.model_cfg
    model_cfg["device"] = f"cuda:{local}"
    if model_cfg.get("use_lora") and model_cfg.get("lora_seed") is None:
        model_cfg["lora_seed"] = int(cfg.run_cfg.get("seed", 42))
    model_cls = registry.get_model_class(model_cfg.arch)
    assert model_cls is not None, f"Model '{model_cfg.arch}' has not been registered."
    model = model_cls.from_config(model_cfg)
    if getattr(model, "use_lora", False):
        model.lora.base_seed = int(cfg.run_cfg.get("seed", 42)) + rank
    runner = RunnerBase(cfg=cfg, job_id=job_id, model=model, datasets=datasets, rank=rank, world=world,
                        device=torch.device(f"cuda:{local}"))
    runner.train()
    return runner


if __name__ == "__main__":
    main()
