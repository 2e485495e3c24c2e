"""Builds, in a temporary directory, every file the reference's `Myriad.from_config` reads -- in the reference's own
formats -- from seeded weights (tests/golden_utils.py), plus YAML configs using the reference's keys:

  eva_vit_g.pth                      flat EVA ViT state dict (eva_vit.py:429-441); 2 blocks, the config pins vit_depth: 1
  blip2_qformer.pth                  {"model": {Qformer.*, query_tokens, ln_vision.*, + keys that must be ignored}}
  pretrained_minigpt4_7b.pth         {"model": {llama_proj.*}} (myriad.py:210-217)
  vicuna/                            HF save_pretrained layout: config.json, tokenizer.model (SentencePiece, trained here on a
                                     tiny corpus), model.safetensors.index.json + 2 shards
  checkpoint_0.pth                   a fine-tuned checkpoint of the trainables (runner_base.py:592-628)
  train.yaml / eval.yaml             model / datasets / run sections as the reference's shipped YAMLs
Full width (the VE adapters hard-code 1408 / 768 / 4096), reduced depth, small MLP: ~200 MB, a few seconds.
"""
import json
import os

import torch

from tests import golden_utils as gu

CORPUS = ("###Human: <Img><ImageHere></Img> This image may be simulated by photo editing. According to IAD expert opinions and "
          "corresponding visual descriptions, find out if there are defects in this image. ###Assistant: "
          "Yes, there exists anomalies in the image. No, there exists no anomalies in the image, at the upper left top right "
          "center bottom lower of the image.\n")


def build(root: str, seeds=(11, 12, 13, 14, 15), inter: int = 256, use_lora: bool = False) -> dict:
    import sentencepiece as spm
    from safetensors.torch import save_file
    os.makedirs(root, exist_ok=True)
    vic = os.path.join(root, "vicuna")
    os.makedirs(vic, exist_ok=True)
    corpus = os.path.join(root, "corpus.txt")
    open(corpus, "w").write(CORPUS * 20)
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(root, "tok"), vocab_size=120, model_type="bpe",
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, character_coverage=1.0, minloglevel=2)
    os.replace(os.path.join(root, "tok.model"), os.path.join(vic, "tokenizer.model"))
    V = 120
    sd = {}
    sd.update(gu.vit_weights(1408, 2, 16, int(1408 * 4.3637), 14, 257, seed=seeds[0]))
    sd.update(gu.qformer_weights(768, 2, 3072, 1408, seed=seeds[1]))
    sd.update(gu.llama_weights(4096, 1, inter, V, seed=seeds[2]))
    sd.update(gu.adapter_weights(seed=seeds[3]))
    sd.update(gu.glue_weights(seed=seeds[4]))
    # ---- the reference's files
    vit = {k[len("visual_encoder."):]: v.half() for k, v in sd.items() if k.startswith("visual_encoder.")}
    torch.save(vit, os.path.join(root, "eva_vit_g.pth"))
    qf = {k: v for k, v in sd.items() if k.startswith(("Qformer.", "ln_vision.")) or k == "query_tokens"}
    qf["opt_proj.weight"] = torch.zeros(4, 4)                         # present in the BLIP-2 file, ignored (strict=False)
    qf["visual_encoder.cls_token"] = torch.ones(1, 1, 1408)           # the BLIP-2 file carries a ViT copy: the ViT file wins
    torch.save({"model": qf}, os.path.join(root, "blip2_qformer.pth"))
    torch.save({"model": {k: v for k, v in sd.items() if k.startswith("llama_proj.")}}, os.path.join(root, "pretrained_minigpt4_7b.pth"))
    llama = {k[len("llama_model."):]: v.to(torch.bfloat16) for k, v in sd.items() if k.startswith("llama_model.")}
    names = sorted(llama)
    shards = {"model-00001-of-00002.safetensors": names[:len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for fn, ks in shards.items():
        save_file({k: llama[k].contiguous() for k in ks}, os.path.join(vic, fn))
    json.dump({"metadata": {}, "weight_map": {k: fn for fn, ks in shards.items() for k in ks}},
              open(os.path.join(vic, "model.safetensors.index.json"), "w"))
    json.dump({"architectures": ["LlamaForCausalLM"], "hidden_size": 4096, "intermediate_size": inter, "num_attention_heads": 32,
               "num_hidden_layers": 1, "rms_norm_eps": 1e-6, "vocab_size": V, "bos_token_id": 1, "eos_token_id": 2,
               "pad_token_id": 0, "model_type": "llama"}, open(os.path.join(vic, "config.json"), "w"))
    trainable = {k: v for k, v in sd.items() if k.startswith(("expert_adaptor.", "VETokenizer.", "VEInstructor."))}
    torch.save({"model": trainable, "optimizer": None, "config": {}, "scaler": None, "epoch": 0}, os.path.join(root, "checkpoint_0.pth"))
    prompts = os.path.join(root, "alignment.txt")
    open(prompts, "w").write("<Img><ImageHere></Img> Describe this image in detail.\nno image placeholder here\n")
    # ---- YAMLs with the reference's keys (train_configs/loraadapter_simple_myriad_finetune.yaml, eval_configs/myriad.yaml)
    model = {"arch": "myriad", "model_type": "pretrain_vicuna", "freeze_vit": True, "freeze_qformer": True, "max_txt_len": 160,
             "end_sym": "###", "prompt_path": prompts, "prompt_template": "###Human: {} ###Assistant: ",
             "ckpt": os.path.join(root, "checkpoint_0.pth"), "llama_model": vic,
             "q_former_model": os.path.join(root, "blip2_qformer.pth"), "vit_model": os.path.join(root, "eva_vit_g.pth"),
             # keys of this build: the reduced-depth fixture and the file myriad.py:210 hard-codes
             "vit_depth": 1, "minigpt4_ckpt": os.path.join(root, "pretrained_minigpt4_7b.pth")}
    if use_lora:
        model["use_lora"] = True
    run = {"task": "image_text_pretrain", "lr_sched": "linear_warmup_cosine_lr", "init_lr": "1e-4", "min_lr": 0, "warmup_lr": "1e-6",
           "weight_decay": 0.05, "max_epoch": 2, "iters_per_epoch": 2, "batch_size_train": 4, "batch_size_eval": 4, "num_workers": 0,
           "warmup_steps": 0, "seed": 42, "output_dir": os.path.join(root, "out"), "amp": True, "resume_ckpt_path": None,
           "evaluate": False, "train_splits": ["train"], "device": "cuda", "world_size": 1, "dist_url": "env://", "distributed": True,
           "max_checkpoints": 20}
    import yaml
    yaml.safe_dump({"model": model, "datasets": {"synthetic": {"num_samples": 16, "seed": 3}}, "run": run},
                   open(os.path.join(root, "train.yaml"), "w"), sort_keys=False)
    ev = dict(model, use_ve=True, noise_level=0.15, round_index=14, k_shot=0)
    yaml.safe_dump({"model": ev, "datasets": {"synthetic": {"num_samples": 4, "seed": 5}}, "run": {"task": "image_text_pretrain"}},
                   open(os.path.join(root, "eval.yaml"), "w"), sort_keys=False)
    sd_model = {k: v for k, v in sd.items() if not (k.startswith("visual_encoder.blocks.") and int(k.split(".")[2]) >= 1)}
    # what the model computes with: the ViT file is fp16, the LLaMA shards bf16 (both exactly representable round trips)
    for k in list(sd_model):
        if k.startswith("visual_encoder."):
            sd_model[k] = sd_model[k].half().float()
        elif k.startswith("llama_model."):
            sd_model[k] = sd_model[k].to(torch.bfloat16).float()
    return dict(root=root, train_yaml=os.path.join(root, "train.yaml"), eval_yaml=os.path.join(root, "eval.yaml"), sd=sd_model,
                vicuna=vic, vocab=V)
