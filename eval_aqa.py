#!/usr/bin/env python3
"""AQA evaluation entry point: counterpart of the reference's evaluation_aqa_dataset.py (:40-62, 233-390) for the
task types that script can actually run (`1cls`, `shot`: the other dataset classes are never imported there).

    python eval_aqa.py --cfg-path eval_configs/myriad.yaml --task_type 1cls --split mvtec --bs 4 [--ckpt 9] [--k_shot 0]

Same flow: Config -> model_cls.from_config(model_cfg) on cuda:{gpu-id} -> DataLoader(bs) -> model.generate(samples,
max_new_tokens=90, stopping_criteria=[###], do_sample=True, top_p=0.01, temperature=1.0, min_length=1, use_cache=True)
-> ids clamped to [1, 40000] -> batch_decode -> one jsonl record per sample (image_id, image_path, is_anomaly, error,
output, anomaly_score) -> memory / mean latency.  `--world/--rank` shard the test set by index for multi-GPU runs
(replicas only: no collective on the data path; merge with myriad_amd.eval_protocol.merge_shards).
"""
import argparse
import os
import sys
import time
from datetime import datetime

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ANNO_FILES = {"1cls": {"visa": "DC_VISA_test_normal.jsonl", "mvtec": "DC_MVTEC_test_normal.jsonl"},
              "shot": {"visa": "DC_VISA_test_normal.jsonl", "mvtec": "DC_MVTEC_test_normal.jsonl"}}
ROOTS = {"visa": "./data/EvalADDataset", "mvtec": "./data/EvalADDataset"}


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Demo")
    parser.add_argument("--cfg-path", required=True, help="path to configuration file.")
    parser.add_argument("--gpu-id", type=int, default=0, help="specify the gpu to load the model.")
    parser.add_argument("--task_type", type=str, default="1cls", choices=["1cls", "shot"])
    parser.add_argument("--split", type=str, default="mvtec", choices=["visa", "mvtec"])
    parser.add_argument("--ckpt", type=int, default=-1)
    parser.add_argument("--bs", type=int, default=1)
    parser.add_argument("--round_index", type=int, default=14)
    parser.add_argument("--k_shot", type=int, default=0)
    parser.add_argument("--start", type=int, default=0)
    parser.add_argument("--options", nargs="+")
    # additions of this build (the reference derives the output path from the checkpoint path and runs one process)
    parser.add_argument("--out", type=str, default=None, help="result jsonl (default: next to the checkpoint, reference naming)")
    parser.add_argument("--dataset", type=str, default="anomaly_detection", help="dataset builder (tests use 'synthetic')")
    parser.add_argument("--world", type=int, default=1)
    parser.add_argument("--rank", type=int, default=0)
    parser.add_argument("--limit", type=int, default=0, help="evaluate only the first N batches (smoke runs)")
    return parser.parse_args(argv)


def main(argv=None):
    import torch
    from torch.utils.data import DataLoader, Subset
    from myriad_amd import myriad  # noqa: F401
    from myriad_amd import datasets as D
    from myriad_amd import eval_protocol as EP
    from myriad_amd.config import Config
    from myriad_amd.myriad import StoppingCriteriaSub
    from myriad_amd.registry import registry
    from myriad_amd.runner import setup_seeds

    args = parse_args(argv)
    cfg = Config(args)
    setup_seeds(int(cfg.run_cfg.get("seed", 42)), 0)
    model_config = cfg.model_cfg
    model_config["round_index"] = args.round_index
    model_config["k_shot"] = args.k_shot
    if args.ckpt != -1:                                  # evaluation_aqa_dataset.py:245-249
        parts = model_config["ckpt"].split("/")
        parts[-1] = f"checkpoint_{args.ckpt}.pth"
        model_config["ckpt"] = "/".join(parts)
    model_config["device_8bit"] = args.gpu_id
    model_config["device"] = f"cuda:{args.gpu_id}"
    model_config["need_backward"] = False                # no dgrad copies of the frozen weights for an evaluation run
    model_cls = registry.get_model_class(model_config.arch)
    model = model_cls.from_config(model_config).to("cuda:{}".format(args.gpu_id))

    stop_words_ids = [torch.tensor([835]).to(model.device), torch.tensor([2277, 29937]).to(model.device)]   # '###', two encodings
    stopping_criteria = [StoppingCriteriaSub(stops=stop_words_ids)]
    if args.dataset == "synthetic":
        ds = D.build_datasets({"synthetic": cfg.datasets_cfg.get("synthetic", {})}, split="test")["synthetic"]
    else:
        dcfg = dict(cfg.datasets_cfg.get("anomaly_detection", {}) or {})
        dcfg["build_info"] = {"vis_root": ROOTS[args.split], "ann_paths": [ANNO_FILES[args.task_type][args.split]]}
        ds = D.build_datasets({"anomaly_detection": dcfg}, split="test")["anomaly_detection"]
    if args.world > 1:
        ds = Subset(ds, EP.shard_indices(len(ds), args.rank, args.world))
    loader = DataLoader(ds, batch_size=args.bs, num_workers=0, collate_fn=D.collate)

    ckpt_name = os.path.splitext(os.path.basename(model_config.get("ckpt", "") or "checkpoint_0.pth"))[0]
    num_ckpt = ckpt_name.split("_")[-1]
    prefix = (f"results_ckpt{num_ckpt}_training={args.task_type}_split={args.split}_kshot={args.k_shot}_roundindex={args.round_index}_"
              f"{datetime.now().strftime('%Y%m%d_%H%M')}")
    save_path = args.out or os.path.join(os.path.dirname(model_config.get("ckpt", "") or "."), f"{prefix}.jsonl")
    if args.world > 1 and args.out is None:
        save_path = save_path[:-6] + f".rank{args.rank}.jsonl"
    print(f"Results will be saved to {save_path}")
    generate_kwargs = {"max_new_tokens": 90, "stopping_criteria": stopping_criteria, "do_sample": True, "use_cache": True,
                       "min_length": 1, "top_p": 0.01, "temperature": 1.0}           # evaluation_aqa_dataset.py:289-301

    model.eval()
    records, all_time, sampled = [], 0.0, 0
    for testid, data_sample in enumerate(loader):
        if testid < args.start:
            continue
        if args.limit and testid >= args.start + args.limit:
            break
        with torch.no_grad():
            t1 = time.time()
            outputs = model.generate(data_sample, **generate_kwargs)
            torch.cuda.synchronize()
            all_time += time.time() - t1
        sampled += model.last_generate_stats["sampled_rows"]
        texts = EP.postprocess_generation(outputs["token_ids"], model.llama_tokenizer)
        maps = outputs.get("ve_anomaly_maps")
        for ind, text in enumerate(texts):
            amax = None
            if maps is not None:                         # anomaly_map_handler (:93-104): uint8(map * 255) then max
                amax = float((maps[ind].detach().float().cpu() * 255.0).to(torch.uint8).max())
            records.append(EP.make_ad_record(int(data_sample["image_id"][ind]), data_sample["img_path"][ind],
                                             bool(data_sample["is_anomaly"][ind]), text, amax))
    EP.write_jsonl(save_path, records)
    n_batches = max(1, len(records) // max(1, args.bs))
    print("CUDA Memory:", torch.cuda.max_memory_allocated() / (1024 * 1024))
    print("Mean Time: ", all_time / n_batches)
    if sampled:
        print(f"note: {sampled} generated tokens had p_max < top_p and were drawn, not arg-maxed (see LlamaHIP.greedy_generate)")
    return save_path, records


if __name__ == "__main__":
    main()
