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


# ------------------------------------------------------------------------------------------------ the reference's weight sources
EVA_VIT_URL_BASENAME = "eva_vit_g.pth"          # eva_vit.py:430 downloads .../BLIP2/eva_vit_g.pth into the hub cache
MINIGPT4_PROJ_PATH = "./pretrained_models/pretrained_minigpt4_7b.pth"      # hard-coded at myriad.py:210


def _hub_dirs() -> List[str]:
    dirs = []
    if os.environ.get("TORCH_HOME"):
        dirs.append(os.path.join(os.environ["TORCH_HOME"], "hub", "checkpoints"))
    dirs.append(os.path.join(os.path.expanduser("~"), ".cache", "torch", "hub", "checkpoints"))
    dirs.append("./pretrained_models")
    return dirs


def resolve_checkpoint_file(url_or_filename: str) -> str:
    """`load_from_pretrained` (blip2.py:91-110) / `download_cached_file` (dist_utils.py:93-137) without the network:
    a file path is used as is; a URL maps to its basename in the torch hub cache (where the reference's download
    would have put it) or ./pretrained_models."""
    if os.path.isfile(url_or_filename):
        return url_or_filename
    if "://" in url_or_filename:
        base = os.path.basename(url_or_filename.split("?")[0])
        for d in _hub_dirs():
            cand = os.path.join(d, base)
            if os.path.isfile(cand):
                return cand
        raise RuntimeError(f"checkpoint url or path is invalid: {url_or_filename} is a URL, this build does not download; "
                           f"put {base} into one of {_hub_dirs()}")
    raise RuntimeError(f"checkpoint url or path is invalid: {url_or_filename}")


def load_llama_tokenizer(llama_model: str):
    """myriad.py:182-183: `LlamaTokenizer.from_pretrained(llama_model, use_fast=False)`, pad_token = eos_token."""
    from transformers import LlamaTokenizer
    tok = LlamaTokenizer.from_pretrained(llama_model, use_fast=False)
    tok.pad_token = tok.eos_token
    return tok


def load_reference_weights(cfg, arch: str = "myriad"):
    """Everything `Myriad.__init__` / `MiniGPT4.__init__` load from disk (myriad.py:104-217, mini_gpt4.py:43-121), from the
    reference's own config keys: `vit_model` ("eva_clip_g" or a file), `q_former_model` (URL or file), `llama_model` (HF
    directory), the MiniGPT-4 projection file.  Returns (weights keyed by the reference's state_dict names, meta)."""
    get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    llama_dir = get("llama_model")
    if not llama_dir or not os.path.isdir(llama_dir):
        raise FileNotFoundError(f"model.llama_model must be a HuggingFace checkpoint directory, got {llama_dir!r}")
    hf_cfg = json.load(open(os.path.join(llama_dir, "config.json")))
    vit_model = get("vit_model", "eva_clip_g")
    if vit_model == "eva_clip_g":
        vit_path = get("vit_ckpt") or resolve_checkpoint_file("https://hub/" + EVA_VIT_URL_BASENAME)
    elif os.path.isfile(str(vit_model)):
        vit_path = vit_model
    else:
        raise AssertionError("vit model must be eva_clip_g (or a path to its state dict)")      # blip2.py:68-70
    qf_path = resolve_checkpoint_file(get("q_former_model", "https://storage.googleapis.com/sfr-vision-language-research/"
                                                            "LAVIS/models/BLIP2/blip2_pretrained_flant5xxl.pth"))
    image_size = int(get("image_size", 224))
    vit_sd = torch.load(vit_path, map_location="cpu")
    depth = int(get("vit_depth", 39))             # create_eva_vit_g builds depth=39 and loads strict=False (eva_vit.py:421,436)
    vit_sd = {k: v for k, v in (vit_sd["model"] if "model" in vit_sd and isinstance(vit_sd["model"], Mapping) else vit_sd).items()
              if not (k.startswith("blocks.") and int(k.split(".")[1]) >= depth)}
    n_blocks = len({k.split(".")[1] for k in vit_sd if k.startswith("blocks.")})
    if n_blocks != depth:
        raise ValueError(f"EVA ViT checkpoint has {n_blocks} of the {depth} blocks the reference builds")
    proj = None
    proj_path = get("minigpt4_ckpt", MINIGPT4_PROJ_PATH)
    if arch == "myriad" or os.path.isfile(proj_path):
        if not os.path.isfile(proj_path):
            raise FileNotFoundError(f"{proj_path}: Myriad loads llama_proj from it (myriad.py:210-217)")
        proj = torch.load(proj_path, map_location="cpu")
    weights = assemble_reference_weights(eva_vit=vit_sd, qformer=torch.load(qf_path, map_location="cpu"), minigpt4=proj,
                                         llama=load_hf_shards(llama_dir), num_patches=(image_size // 14) ** 2)
    meta = dict(llm_heads=int(hf_cfg.get("num_attention_heads", 32)), llm_eps=float(hf_cfg.get("rms_norm_eps", 1e-6)),
                bos_token_id=int(hf_cfg.get("bos_token_id", 1)), pad_token_id=int(hf_cfg.get("eos_token_id", 2)),
                vit_heads=16, qf_heads=12)
    return weights, meta


# ------------------------------------------------------------------------------------------------ fine-tune checkpoints
def reference_param_order(names) -> List[str]:
    """The trainable parameters in the order `model.named_parameters()` yields them in the REFERENCE -- the order
    `RunnerBase.optimizer` builds its two param groups in (runner_base.py:110-119) and therefore the indices of
    torch.optim.AdamW's state_dict.  nn.Module yields a module's own parameters before its children's, children in
    registration order: Myriad.__init__ registers expert_adaptor, VETokenizer (direct parameter `base_prompts` before
    `meta_net.*`), VEInstructor, Qformer, llama_model (peft: per layer q_proj.lora_A, q_proj.lora_B, v_proj.lora_A,
    v_proj.lora_B), llama_proj (myriad.py:117-125, 148, 186-207)."""
    import re

    def key(n: str):
        if n.startswith("expert_adaptor."):
            return (0, 0 if ".conv1." in n else 1, 0, 0)
        for rank, pre in ((1, "VETokenizer."), (2, "VEInstructor.")):
            if n.startswith(pre):
                if n.endswith("base_prompts"):
                    return (rank, -1, 0, 0)
                m = re.search(r"meta_net\.(\d+)\.(weight|bias)$", n)
                return (rank, int(m.group(1)), 0 if m.group(2) == "weight" else 1, 0)
        if "lora_" in n:
            m = re.search(r"layers\.(\d+)\.self_attn\.([qv])_proj\.lora_([AB])\.", n)
            return (3, int(m.group(1)), 0 if m.group(2) == "q" else 1, 0 if m.group(3) == "A" else 1)
        if n.startswith("llama_proj."):
            return (4, 0 if n.endswith("weight") else 1, 0, 0)
        raise KeyError(n)

    return sorted(names, key=key)


def _optimizer_index(store) -> List[str]:
    """Parameter name of each torch optimizer index: weight-decay group first, then the rest, each in reference order."""
    from .myriad import uses_weight_decay
    names = reference_param_order([n for n, _, _ in store.specs])
    shapes = {n: r for n, _, r in store.specs}
    wd = [n for n in names if uses_weight_decay(n, len(shapes[n]))]
    return wd + [n for n in names if n not in set(wd)], len(wd)


def optimizer_state_dict(store, lr: float, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8) -> dict:
    """The flat AdamW state as torch.optim.AdamW.state_dict() holds it for `RunnerBase.optimizer`'s two groups
    (runner_base.py:104-139): indices follow the reference's named_parameters() order, NOT the flat buffer's; a module
    that was never stepped has no state entry (torch creates state lazily), each entry carries its module's own step."""
    from .myriad import module_of
    if not getattr(store, "moments_complete", True):
        raise RuntimeError("optimizer_state_dict: this rank holds Adam moments for its own shard only (DataParallel mode "
                           "'rs_ag'); call dp.gather_state(store) on every rank first")
    order, n_wd = _optimizer_index(store)
    ishape = {n: i for n, i, _ in store.specs}
    rshape = {n: r for n, _, r in store.specs}
    steps = store.module_steps()
    state, groups = {}, [[], []]
    for idx, name in enumerate(order):
        o, n = store.offsets[name]
        st = steps.get(module_of(name), 0)
        if st > 0:
            state[idx] = {"step": torch.tensor(float(st)),
                          "exp_avg": to_reference_layout(store.flat_m[o:o + n].view(ishape[name]), rshape[name]).cpu().clone(),
                          "exp_avg_sq": to_reference_layout(store.flat_v[o:o + n].view(ishape[name]), rshape[name]).cpu().clone()}
        groups[0 if idx < n_wd else 1].append(idx)
    common = dict(lr=lr, betas=tuple(betas), eps=eps, amsgrad=False, maximize=False, foreach=None, capturable=False,
                  differentiable=False, fused=None)
    pg = []
    if groups[0]:
        pg.append(dict(common, weight_decay=weight_decay, params=groups[0]))
    if groups[1]:
        pg.append(dict(common, weight_decay=0.0, params=groups[1]))
    return {"state": state, "param_groups": pg}


def load_optimizer_state_dict(store, sd: dict) -> None:
    """Inverse of optimizer_state_dict: accepts what the reference's runner saved (`self.optimizer.state_dict()`,
    runner_base.py:606-612)."""
    from .myriad import module_of
    order, _ = _optimizer_index(store)
    flat: List[int] = [i for g in sd["param_groups"] for i in g["params"]]
    if len(flat) != len(order):
        raise ValueError(f"optimizer state has {len(flat)} parameters, the model {len(order)}")
    ishape = {n: i for n, i, _ in store.specs}
    steps: Dict[str, int] = {}
    for pos, idx in enumerate(flat):
        name = order[pos]
        st = sd["state"].get(idx, sd["state"].get(str(idx)))
        if st is None:
            continue                                   # parameter never stepped (torch creates state lazily)
        o, n = store.offsets[name]
        if tuple(st["exp_avg"].shape) != tuple({n_: r for n_, _, r in store.specs}[name]):
            raise ValueError(f"optimizer state {idx} has shape {tuple(st['exp_avg'].shape)}, parameter {name} "
                             f"{tuple({n_: r for n_, _, r in store.specs}[name])}: parameter order mismatch")
        store.flat_m[o:o + n].view(ishape[name]).copy_(from_reference_layout(st["exp_avg"].to(store.flat_m.device, torch.float32), ishape[name]))
        store.flat_v[o:o + n].view(ishape[name]).copy_(from_reference_layout(st["exp_avg_sq"].to(store.flat_v.device, torch.float32), ishape[name]))
        m = module_of(name)
        steps[m] = max(steps.get(m, 0), int(float(st["step"])))
    store.set_module_steps(steps)
    store.step = max(steps.values()) if steps else 0


class CheckpointManager:
    """`RunnerBase._save_checkpoint` / `_load_checkpoint` (runner_base.py:592-628, 649-672) for the HIP model."""

    def __init__(self, output_dir: str, max_checkpoints: int = 1):
        self.output_dir = output_dir
        self.max_checkpoints = max_checkpoints
        self.saved_history: List[str] = []
        os.makedirs(output_dir, exist_ok=True)

    def save(self, model, cur_epoch, lr: float, weight_decay: float = 0.05, config: Optional[dict] = None,
             is_best: bool = False, dp=None) -> str:
        """`dp`: the DataParallel in use.  In mode 'rs_ag' the Adam moments are sharded; with `dp` the gather (a collective:
        every rank must then call save) happens here, without it a save of incomplete moments raises instead of writing them."""
        if hasattr(model, "finish_update"):
            model.finish_update()                      # an overlapped optimiser step must land before parameters are read
        if dp is not None and not getattr(model.store, "moments_complete", True):
            dp.gather_state(model.store)
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
