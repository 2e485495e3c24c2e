"""Checkpoint I/O at the reference's file formats.  Host-side only: nothing here touches the GPU.

What the reference loads, and from where (all `torch.load(..., map_location="cpu")`):
  * `eva_vit_g.pth`               flat ViT state dict, position embedding interpolated to the run's grid
                                  (eva_vit.py:373-394, 429-441)  -> keys `visual_encoder.*`
  * BLIP-2 Q-Former checkpoint    {"model": {...}} loaded with strict=False (blip2.py:91-110)
                                  -> keys `Qformer.*`, `query_tokens`, `ln_vision.*`
  * `pretrained_minigpt4_7b.pth`  {"model": {"llama_proj.weight", "llama_proj.bias", ...}} (myriad.py:210-217)
  * Vicuna / LLaMA HF checkpoint  sharded `pytorch_model-0000x-of-0000y.bin` or `*.safetensors` with an index json
                                  (`LlamaForCausalLM.from_pretrained`, myriad.py:186-196) -> keys `llama_model.*`
  * a fine-tuned `checkpoint_N.pth` {"model": trainable-only state dict, "optimizer", "config", "scaler", "epoch"}
                                  (runner_base.py:592-628, 649-672; myriad.py:511-515 `cfg.ckpt`)
and what it writes: the same `checkpoint_N.pth`, keeping at most `max_checkpoints` files (runner_base.py:618-626).

`assemble_reference_weights` merges the first four into ONE mapping keyed by the reference model's own state_dict
names -- exactly the `cfg["weights"]` that `MyriadHIP.from_config` consumes.  `save_checkpoint` / `load_checkpoint`
round-trip the fifth, with the optimizer state expressed in torch.optim.AdamW's state_dict layout (parameter order =
`RunnerBase.optimizer`'s two groups: weight-decay parameters first, then the rest, runner_base.py:104-139) so a
checkpoint written here resumes in the reference's runner and vice versa.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, List, Mapping, Optional

import torch

from .networks import from_reference_layout, to_reference_layout


# ------------------------------------------------------------------------------------------------ ViT position grid
def interpolate_pos_embed(pos_embed: torch.Tensor, num_patches: int, num_extra_tokens: int = 1) -> torch.Tensor:
    """eva_vit.py:373-394: bicubic (align_corners=False) resize of the patch-position grid of a [1, extra + g*g, D]
    table to `num_patches` = G*G positions; the class token rows are kept.  Returns fp32 like the reference."""
    pe = pos_embed.float()
    D = pe.shape[-1]
    orig = int((pe.shape[-2] - num_extra_tokens) ** 0.5)
    new = int(num_patches ** 0.5)
    if orig == new:
        return pe
    extra = pe[:, :num_extra_tokens]
    grid = pe[:, num_extra_tokens:].reshape(-1, orig, orig, D).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(new, new), mode="bicubic", align_corners=False)
    return torch.cat((extra, grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)


# ------------------------------------------------------------------------------------------------ HF shards
def _load_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu")


def load_hf_shards(path: str) -> Dict[str, torch.Tensor]:
    """A HF `save_pretrained` directory (index json + shards, or one `pytorch_model.bin` / `model.safetensors`) or a
    single weight file -> flat {name: tensor}.  Every tensor named by the index must be present exactly once."""
    if os.path.isfile(path):
        return dict(_load_file(path))
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ip = os.path.join(path, index)
        if os.path.exists(ip):
            weight_map = json.load(open(ip))["weight_map"]
            out: Dict[str, torch.Tensor] = {}
            for shard in sorted(set(weight_map.values())):
                part = _load_file(os.path.join(path, shard))
                for k, v in part.items():
                    if weight_map.get(k) == shard:
                        out[k] = v
            missing = [k for k in weight_map if k not in out]
            if missing:
                raise KeyError(f"{index} names tensors no shard holds: {missing[:4]}{'...' if len(missing) > 4 else ''}")
            return out
    for single in ("model.safetensors", "pytorch_model.bin"):
        sp = os.path.join(path, single)
        if os.path.exists(sp):
            return dict(_load_file(sp))
    raise FileNotFoundError(f"no HF weight files under {path}")


# ------------------------------------------------------------------------------------------------ frozen weights
PEFT_INFIX = "base_model.model."


def assemble_reference_weights(eva_vit: Optional[Mapping] = None, qformer: Optional[Mapping] = None,
                               minigpt4: Optional[Mapping] = None, llama: Optional[Mapping] = None,
                               num_patches: int = 256) -> "OrderedDict[str, torch.Tensor]":
    """Merge the reference's four weight sources into one mapping keyed by `Myriad.state_dict()` names.
    Arguments are the loaded objects (flat dicts, or {"model": dict} wrappers as the files store them)."""
    def unwrap(o):
        return o["model"] if isinstance(o, Mapping) and "model" in o and isinstance(o["model"], Mapping) else o

    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    if eva_vit is not None:
        for k, v in unwrap(eva_vit).items():
            if k == "pos_embed":
                v = interpolate_pos_embed(v, num_patches, 1)      # one class token (eva_vit.py:377-378)
            out["visual_encoder." + k] = v
    if qformer is not None:
        for k, v in unwrap(qformer).items():           # strict=False: everything else in the BLIP-2 file is ignored
            if k.startswith(("Qformer.", "ln_vision.")) or k == "query_tokens":
                out[k] = v
    if minigpt4 is not None:
        for k, v in unwrap(minigpt4).items():
            if k.startswith("llama_proj."):
                out[k] = v
    if llama is not None:
        for k, v in unwrap(llama).items():
            k = k.replace(PEFT_INFIX, "")              # a merged/peft-saved tree carries the wrapper's infix
            if "lora_" in k or k.endswith("rotary_emb.inv_freq"):
                continue                               # adapters travel in checkpoint_N.pth; inv_freq is recomputed
            out[k if k.startswith("llama_model.") else "llama_model." + k] = v
    return out


# ------------------------------------------------------------------------------------------------ fine-tune checkpoints
def optimizer_state_dict(store, lr: float, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8) -> dict:
    """The flat AdamW state as torch.optim.AdamW.state_dict() would hold it for `RunnerBase.optimizer`'s two groups."""
    state, idx, groups = {}, 0, [[], []]
    n_wd_params = sum(1 for name, _, rshape in store.specs if store.offsets[name][0] < store.n_wd)
    for name, ishape, rshape in store.specs:
        o, n = store.offsets[name]
        state[idx] = {"step": torch.tensor(float(store.step)),
                      "exp_avg": to_reference_layout(store.flat_m[o:o + n].view(ishape), rshape).cpu().clone(),
                      "exp_avg_sq": to_reference_layout(store.flat_v[o:o + n].view(ishape), rshape).cpu().clone()}
        groups[0 if idx < n_wd_params else 1].append(idx)
        idx += 1
    common = dict(lr=lr, betas=tuple(betas), eps=eps, amsgrad=False, maximize=False, foreach=None, capturable=False,
                  differentiable=False, fused=None)
    return {"state": state,
            "param_groups": [dict(common, weight_decay=weight_decay, params=groups[0]),
                             dict(common, weight_decay=0.0, params=groups[1])]}


def load_optimizer_state_dict(store, sd: dict) -> None:
    names = [name for name, _, _ in store.specs]
    order: List[int] = [i for g in sd["param_groups"] for i in g["params"]]
    if len(order) != len(names):
        raise ValueError(f"optimizer state has {len(order)} parameters, the model {len(names)}")
    step = 0
    for pos, idx in enumerate(order):
        name, ishape, _ = store.specs[pos]
        st = sd["state"].get(idx, sd["state"].get(str(idx)))
        if st is None:
            continue                                   # parameter never stepped (torch creates state lazily)
        o, n = store.offsets[name]
        store.flat_m[o:o + n].view(ishape).copy_(from_reference_layout(st["exp_avg"].to(store.flat_m.device, torch.float32), ishape))
        store.flat_v[o:o + n].view(ishape).copy_(from_reference_layout(st["exp_avg_sq"].to(store.flat_v.device, torch.float32), ishape))
        step = max(step, int(float(st["step"])))
    store.step = step


class CheckpointManager:
    """`RunnerBase._save_checkpoint` / `_load_checkpoint` (runner_base.py:592-628, 649-672) for the HIP model."""

    def __init__(self, output_dir: str, max_checkpoints: int = 1):
        self.output_dir = output_dir
        self.max_checkpoints = max_checkpoints
        self.saved_history: List[str] = []
        os.makedirs(output_dir, exist_ok=True)

    def save(self, model, cur_epoch, lr: float, weight_decay: float = 0.05, config: Optional[dict] = None,
             is_best: bool = False) -> str:
        if hasattr(model, "finish_update"):
            model.finish_update()                      # an overlapped optimiser step must land before parameters are read
        save_obj = {"model": model.state_dict(),       # trainable parameters only, reference key names and layouts
                    "optimizer": optimizer_state_dict(model.store, lr, weight_decay),
                    "config": config or {}, "scaler": None, "epoch": cur_epoch}
        save_to = os.path.join(self.output_dir, "checkpoint_{}.pth".format("best" if is_best else cur_epoch))
        if len(self.saved_history) >= self.max_checkpoints:
            old = self.saved_history.pop(0)
            if old != save_to and os.path.exists(old):
                os.remove(old)
        self.saved_history.append(save_to)
        torch.save(save_obj, save_to)
        return save_to

    @staticmethod
    def load(model, path: str, with_optimizer: bool = True) -> int:
        """Returns the epoch to resume at (checkpoint epoch + 1, runner_base.py:671)."""
        if not os.path.isfile(path):
            raise RuntimeError("checkpoint url or path is invalid")
        ck = torch.load(path, map_location="cpu")
        model.load_state_dict(ck["model"], strict=False)
        if with_optimizer and ck.get("optimizer") is not None:
            load_optimizer_state_dict(model.store, ck["optimizer"])
        return int(ck["epoch"]) + 1 if isinstance(ck.get("epoch"), int) else 0
