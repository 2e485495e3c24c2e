"""`Myriad` / `MiniGPT4` model classes on the MI355X HIP path -- the drop-in boundary.

Mirror of the reference's registered model classes (minigpt4/models/myriad.py:62-517, mini_gpt4.py:14-307):
`from_config(cfg)`, `forward(samples) -> {"loss"}`, `generate(samples, **kw) -> {"token_ids", "ve_anomaly_maps"}`,
`named_parameters()` / `requires_grad`, `state_dict()` / `load_state_dict(strict=False)` with the reference's key
names, `.device`, `.train()/.eval()`.  The forward follows `Myriad.forward` (myriad.py:377-431) ->
`encode_img` (:241-272) -> `prompt_wrap` (:354-375) -> label/mask assembly (:395-421) -> LLaMA loss; every tensor
op runs in libmyriad_hip.so.  The backward is explicit (frozen ViT/Q-Former/LLaMA => dgrad only) and
is also reachable through `loss.backward()` via a thin autograd bridge so the reference's step loop
(tasks/base_task.py:236-271) works unchanged.

Out of scope here: the ImageBind vision expert.  Its outputs are inputs to this model:
`samples["anomaly_maps"]` / `samples["oneshot_anomaly_maps"]` [B,1,224,224] in [0,1].
"""
from __future__ import annotations

import os
import random
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from .eva_vit import EvaViTHIP
from .llama import LlamaHIP
from .networks import LoraAdaptor, VENet, from_reference_layout, to_reference_layout, ve_param_specs
from .qformer import QFormerHIP
from .registry import registry

BF16, F32 = torch.bfloat16, torch.float32


def uses_weight_decay(name: str, ndim: int) -> bool:
    """Parameter grouping of `RunnerBase.optimizer` (reference runners/runner_base.py:115-118)."""
    return not (ndim < 2 or "bias" in name or "ln" in name or "bn" in name)


MODULES = ("expert_adaptor", "VETokenizer", "VEInstructor", "llama_proj", "lora")


def module_of(name: str) -> str:
    """The top-level reference module a trainable parameter belongs to (the unit torch's AdamW skips when unused)."""
    for m in MODULES[:4]:
        if name.startswith(m + "."):
            return m
    if "lora_" in name:
        return "lora"
    raise KeyError(name)



def accum_update_due(calls_since_update: int, accum_grad_iters: int, accum_index=None) -> bool:
    """Whether the optimiser steps after this backward (base_task.py:262-271).  With the iteration index of the epoch the
    reference's rule `(i + 1) % accum_grad_iters == 0`; without it, every accum_grad_iters-th call."""
    n = max(int(accum_grad_iters), 1)
    if accum_index is not None:
        return (int(accum_index) + 1) % n == 0
    return calls_since_update >= n


class ParamStore:
    """All trainable parameters in ONE flat fp32 buffer (+ grad, Adam m, v): a single RCCL all-reduce and a handful of
    fused AdamW launches per step.  Layout: [weight-decay group | no-decay group | per-module use flags]; inside each
    group the parameters of one reference module are contiguous, segments 16-byte aligned.

    torch.optim.AdamW (runner_base.py:132-137) skips a parameter whose .grad is None: a module that no rank used in a
    step (VEInstructor at prompt stage 0, VETokenizer at stage 2; myriad.py:378,252,265) keeps its weights, moments and
    step count.  Here every module has a device-resident use flag at the tail of the gradient buffer -- so the data-
    parallel all-reduce sums it over ranks for free -- and a device-resident step counter; the gated AdamW kernel
    (mh_adamw_gated) reads both, the host never synchronises."""

    def __init__(self, specs: List[Tuple[str, Tuple[int, ...], Tuple[int, ...]]], device):
        self.dev = torch.device(device)
        order = {m: i for i, m in enumerate(MODULES)}
        self.modules = [m for m in MODULES if any(module_of(s[0]) == m for s in specs)]
        midx = {m: i for i, m in enumerate(self.modules)}
        key = lambda s: order[module_of(s[0])]                    # stable: keeps each module's own order
        wd = sorted([s for s in specs if uses_weight_decay(s[0], len(s[2]))], key=key)
        nwd = sorted([s for s in specs if not uses_weight_decay(s[0], len(s[2]))], key=key)
        self.specs = wd + nwd
        self.offsets: Dict[str, Tuple[int, int]] = {}
        self.ranges: List[Tuple[int, int, int, bool]] = []        # (module index, start, end, weight decay?)
        off = 0
        for group, decays in ((wd, True), (nwd, False)):
            if not decays:
                self.n_wd = off
            for name, ishape, _ in group:
                n = 1
                for sdim in ishape:
                    n *= sdim
                self.offsets[name] = (off, n)
                mi = midx[module_of(name)]
                if self.ranges and self.ranges[-1][0] == mi and self.ranges[-1][3] == decays:
                    self.ranges[-1] = (mi, self.ranges[-1][1], off + ops.round_up(n, 4), decays)
                else:
                    self.ranges.append((mi, off, off + ops.round_up(n, 4), decays))
                off += ops.round_up(n, 4)
        # padded to a multiple of 32 elements: at 2 / 4 / 8 ranks the reduce-scatter / all-gather shards (multiples of 4
        # elements) tile the buffer exactly, so DataParallel mode 'rs_ag' never pads or copies it (the tail belongs to no
        # module range: zero gradient, never updated)
        self.n_used = off                                         # the parameters proper end here
        off = ops.round_up(off, 32)
        self.total = off
        self.moments_complete = True                              # False while Adam moments exist for this rank's shard only
        nm = len(self.modules)
        self.flat_p = torch.zeros(off, dtype=F32, device=self.dev)
        self.flat_g_comm = torch.zeros(off + ops.round_up(max(nm, 1), 4), dtype=F32, device=self.dev)   # what DP exchanges
        self.flat_g = self.flat_g_comm[:off]
        self.used = self.flat_g_comm[off:off + nm]                # per-module use count (summed over ranks by the all-reduce)
        self.steps_dev = torch.zeros(max(nm, 1), dtype=torch.int32, device=self.dev)
        self.flat_m = torch.zeros(off, dtype=F32, device=self.dev)
        self.flat_v = torch.zeros(off, dtype=F32, device=self.dev)
        self.p, self.g = {}, {}
        self.ref_shape = {}
        for name, ishape, rshape in self.specs:
            o, n = self.offsets[name]
            self.p[name] = self.flat_p[o:o + n].view(ishape)
            self.g[name] = self.flat_g[o:o + n].view(ishape)
            self.ref_shape[name] = rshape
        self.step = 0                                             # optimiser calls so far (host)

    def n_params(self) -> int:
        return sum(n for _, n in self.offsets.values())

    def mark_used(self, used_modules) -> None:
        """Write this rank's use flags (1.0 per used module) into the tail of the gradient buffer."""
        if not self.modules:
            return
        flags = torch.tensor([1.0 if m in used_modules else 0.0 for m in self.modules], dtype=F32)
        self.used.copy_(ops.h2d(flags, self.dev))

    def module_steps(self) -> Dict[str, int]:
        """Updates applied per module so far (device -> host: checkpointing only)."""
        st = self.steps_dev.cpu().tolist()
        return {m: int(st[i]) for i, m in enumerate(self.modules)}

    def set_module_steps(self, steps: Dict[str, int]) -> None:
        host = torch.tensor([int(steps.get(m, 0)) for m in self.modules] or [0], dtype=torch.int32)
        self.steps_dev.copy_(host.to(self.dev))

    def module_range(self, module: str, decays: bool = True):
        """(start, end) of one module's contiguous run inside a weight-decay group of the flat buffers, or None."""
        if module not in self.modules:
            return None
        mi = self.modules.index(module)
        for m2, a, b, d in self.ranges:
            if m2 == mi and d == decays:
                return a, b
        return None

    def adamw_module(self, module: str, lr: float, weight_decay: float = 0.05, beta2: float = 0.999) -> None:
        """The gated AdamW of ONE module's ranges, on the current stream (single-process training: a module whose gradient is
        complete may be updated while the rest of the backward still runs; adamw_step(..., skip={module}) then leaves it out
        and bumps every module's step counter as usual)."""
        mi = self.modules.index(module)
        for m2, a, b, decays in self.ranges:
            if m2 == mi:
                ops.adamw_gated(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], lr,
                                weight_decay if decays else 0.0, self.used[mi:mi + 1], self.steps_dev[mi:mi + 1], beta2=beta2)

    def adamw_pieces(self, pieces, lr: float, weight_decay: float = 0.05, beta2: float = 0.999, grad_scale: float = 1.0) -> None:
        """The gated AdamW on the slices `pieces` [(lo, hi)] of the flat buffers only -- no step-counter bump: a part of the step's
        update issued early (a data-parallel segment whose exchange is complete); adamw_step(..., exclude=pieces' span) does the
        rest and bumps the counters.  Same kernel, same per-module flags and step counts as the one-shot update."""
        for mi, a0, b0, decays in self.ranges:
            for lo, hi in pieces:
                a, b = max(a0, lo), min(b0, hi)
                if a < b:
                    ops.adamw_gated(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], lr,
                                    weight_decay if decays else 0.0, self.used[mi:mi + 1], self.steps_dev[mi:mi + 1], beta2=beta2,
                                    grad_scale=grad_scale)

    def adamw_step(self, lr: float, weight_decay: float = 0.05, beta2: float = 0.999, grad_scale: float = 1.0, shard=None,
                   skip=(), exclude=()):
        """torch.optim.AdamW semantics (runner_base.py:132-137), fused, on the flat buffers; modules unused on every rank
        this step are left alone (see class docstring).  `shard` = (lo, hi) or a list of such: update only those slices of the
        flat buffer (DataParallel mode 'rs_ag': each rank owns 1/world of every exchange segment's optimiser state).  `exclude`:
        spans adamw_pieces() already updated this step.  `skip`: modules adamw_module() already
        updated this step."""
        self.step += 1
        pieces = None
        if shard is not None:
            pieces = [tuple(shard)] if isinstance(shard[0], int) else [tuple(p) for p in shard]
            if sum(hi - lo for lo, hi in pieces) < self.total:
                self.moments_complete = False                     # DataParallel.gather_state() restores it
        skip_idx = {self.modules.index(m) for m in skip if m in self.modules}
        for mi, a0, b0, decays in self.ranges:
            if mi in skip_idx:
                continue
            for lo, hi in (pieces if pieces is not None else [(a0, b0)]):
                a, b = max(a0, lo), min(b0, hi)
                if a >= b:
                    continue
                todo = [(a, b)]
                for ex_lo, ex_hi in exclude:                      # spans adamw_pieces() already updated this step
                    todo = [q for (x, y) in todo for q in ((x, min(y, ex_lo)), (max(x, ex_hi), y)) if q[0] < q[1]]
                for a, b in todo:
                    ops.adamw_gated(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], lr,
                                    weight_decay if decays else 0.0, self.used[mi:mi + 1], self.steps_dev[mi:mi + 1], beta2=beta2,
                                    grad_scale=grad_scale)
        if self.modules:
            ops.adamw_bump(self.used, self.steps_dev)


def init_trainable(name: str, rshape) -> torch.Tensor:
    """Fresh trainable parameter, drawn from torch's global CPU RNG with the distribution the reference's constructor uses:
    LoraAdaptorV2 N(0, 0.02) (networks.py:78-79), nn.Conv2d default (kaiming-uniform, a = sqrt(5): U(+-1/sqrt(fan_in)) for
    weight and bias, networks.py:98-127,159-189), base_prompts N(0, 1) (:189).  llama_proj (MiniGPT-4 stage 2) is an
    nn.Linear whose reset_parameters the reference disables (common/utils.py:41-47): it must come from a checkpoint."""
    import math
    if name.startswith("expert_adaptor."):
        return torch.randn(rshape) * 0.02
    if name.endswith("base_prompts"):
        return torch.randn(rshape)
    if ".meta_net." in name:
        if name.endswith(".weight"):
            fan_in = rshape[1] * rshape[2] * rshape[3]
        else:                                   # bias bound uses the layer's fan-in: recover it from the stack's geometry
            idx = int(name.split(".meta_net.")[1].split(".")[0])
            cin = {0: 1, 3: 4, 6: 16, 9: 64, 12: 256, 15: 1024}[idx]
            k = 3 if idx < 15 else (5 if name.startswith("VETokenizer.") else 1)
            fan_in = cin * k * k
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(rshape) * 2 - 1) * bound
    raise KeyError(f"{name}: not in the weight files and no constructor init exists (load it with `ckpt`)")


class _LossBridge(torch.autograd.Function):
    """Lets `loss.backward()` (reference base_task.py:256-259) drive the explicit HIP backward: gradients are
    written straight into the parameters' .grad views of the flat buffer (no extra copy)."""

    @staticmethod
    def forward(ctx, anchor, model, loss):
        ctx.model = model
        return loss.detach().clone()

    @staticmethod
    def backward(ctx, gout):
        # gout is the scalar the caller back-propagates (1.0, or the GradScaler's loss scale under amp=True,
        # base_task.py:256-259): the gradients are multiplied by it so that scaler.unscale_() / step() see what they expect.
        # The read-back is a host sync per step, like the reference's own loss.item() (base_task.py:276).
        ctx.model.backward(float(gout), accumulate=ctx.model._grads_live())
        return None, None, None


class MyriadHIP(nn.Module):
    arch = "myriad"

    def __init__(self, weights, cfg: Optional[dict] = None, device="cuda:0"):
        super().__init__()
        cfg = dict(cfg or {})
        self.cfg = cfg
        self._dev = torch.device(device)
        self.max_txt_len = cfg.get("max_txt_len", 160)
        self.end_sym = cfg.get("end_sym", "###")
        self.k_shot = cfg.get("k_shot", 0)
        self.fixed_stage = cfg.get("fixed_stage", None)        # None => random.choice like the reference
        self.fixed_taskstage = cfg.get("fixed_taskstage", None)
        self.bos_id, self.pad_id = cfg.get("bos_token_id", 1), cfg.get("pad_token_id", 2)
        self.llama_tokenizer = cfg.get("tokenizer", None)
        self.prompt_list = cfg.get("prompt_list", [])
        need_bwd = cfg.get("need_backward", True)
        ops.ensure_workspace(self._dev)                        # scratch for the automatic split-K GEMM path
        self.visual_encoder = EvaViTHIP(weights, cfg.get("vit_heads", 16), self._dev)
        self.qformer = QFormerHIP(weights, cfg.get("qf_heads", 12), self._dev,
                                  need_backward=need_bwd and self.arch == "myriad")
        self.llama = LlamaHIP(weights, cfg.get("llm_heads", 32), self._dev, eps=float(cfg.get("llm_eps", 1e-6)),
                              need_backward=need_bwd)
        self.ln_w = weights["ln_vision.weight"].to(self._dev, F32).contiguous()
        self.ln_b = weights["ln_vision.bias"].to(self._dev, F32).contiguous()
        self.query_tokens_f32 = weights["query_tokens"].to(self._dev, F32).contiguous()      # frozen (myriad.py:165)
        self.proj_w = weights["llama_proj.weight"].to(self._dev, BF16).contiguous()
        self.proj_b = weights["llama_proj.bias"].to(self._dev, F32).contiguous()
        self.proj_wT = self.proj_w.t().contiguous()
        self.Dv, self.Dq, self.Dl = self.visual_encoder.D, self.qformer.D, self.llama.D
        self.nq0 = self.query_tokens_f32.shape[1]
        # ---- trainables
        specs = []
        if self.arch == "myriad":
            specs.append(("expert_adaptor.conv1.weight", (4, self.Dv), (4, self.Dv)))
            specs.append(("expert_adaptor.conv2.weight", (self.Dv, 4), (self.Dv, 4)))
            specs += ve_param_specs("VETokenizer.", self.Dl, 5)
            specs.append(("VETokenizer.base_prompts", (9, self.Dl), (9, self.Dl)))
            specs += ve_param_specs("VEInstructor.", self.Dq, 1)
        else:
            specs.append(("llama_proj.weight", (self.Dl, self.Dq), (self.Dl, self.Dq)))
            specs.append(("llama_proj.bias", (self.Dl,), (self.Dl,)))
        # PEFT LoRA on q_proj/v_proj (myriad.py:170-180): off in the shipped recipe, on with cfg use_lora
        self.use_lora = bool(cfg.get("use_lora", False))
        lora_init = {}
        if self.use_lora:
            from .lora import LoraQV, init_lora_weights, lora_param_specs
            r = int(cfg.get("lora_r", 8))
            n_l = len(self.llama.layers)
            specs = lora_param_specs(n_l, self.Dl, r) + specs      # LoRA tensors first: keeps A_q|A_v adjacent
            lora_init = init_lora_weights(n_l, self.Dl, r, seed=int(cfg.get("lora_seed", 1234)), device=self._dev)
        self.store = ParamStore(specs, self._dev)
        self._params = OrderedDict()
        for name, ishape, rshape in self.store.specs:
            if name in lora_init and name not in weights:
                src = lora_init[name]
            elif name in weights:
                src = weights[name]
            else:
                src = init_trainable(name, rshape)      # built from the frozen files only: the reference's constructors' init
            self.store.p[name].copy_(from_reference_layout(src.to(self._dev, F32), ishape))
            prm = nn.Parameter(self.store.p[name], requires_grad=True)
            prm.grad = self.store.g[name]
            self._params[name] = prm
            self._register_dotted(name, prm)
        if self.arch == "myriad":
            self.adaptor = LoraAdaptor(self.store.p, self.store.g)
            self.ve_tok = VENet("VETokenizer.", 5, self.Dl, self.store.p, self.store.g, self._dev)
            self.ve_ins = VENet("VEInstructor.", 1, self.Dq, self.store.p, self.store.g, self._dev)
        if self.use_lora:
            self.lora = LoraQV(len(self.llama.layers), self.Dl, int(cfg.get("lora_r", 8)),
                               float(cfg.get("lora_alpha", 16)), float(cfg.get("lora_dropout", 0.05)),
                               self.store.p, self.store.g, self._dev)
            # peft's nn.Dropout draws from torch's RNG, which train.py:63-72 seeds with seed + rank: the counter-based masks
            # take that seed, so ranks (and runs with another seed) drop different elements (cfg lora_dropout_seed overrides)
            self.lora.base_seed = int(cfg.get("lora_dropout_seed", torch.initial_seed())) & ((1 << 31) - 1)
            self.llama.attach_lora(self.lora)
        self._pending_update = None
        self._vit_stream, self._vit_prefetched = None, None
        self._vit_graphs, self._vit_seen, self._vit_rest = {}, {}, None
        # blocks in the look-ahead's first piece (of 39): measured 39 -> 50.9, 30 -> 49.6, 24 -> 49.1, 20 -> 49.1, 12 -> 50.8, 0 -> 52.3 ms
        self._vit_split = int(os.environ.get("MYRIAD_VIT_SPLIT", "22"))
        self._vit_graph_on = os.environ.get("MYRIAD_VIT_GRAPH", "1") != "0"
        self._leaf_aside = os.environ.get("MYRIAD_LEAF_STREAM", "1") != "0"
        self._anchor = torch.zeros((), device=self._dev, requires_grad=True)
        self._ctx = None
        self._has_grads, self._bwd_gscale, self._bwd_prev = False, 1.0, None
        self._accum_count = 0                                  # train_step calls since the last optimiser update (accum_grad_iters)
        self._bridge_used = set()

    # ------------------------------------------------------------------ nn.Module plumbing
    def _register_dotted(self, name: str, prm: nn.Parameter):
        mod = self
        parts = name.split(".")
        for part in parts[:-1]:
            if not hasattr(mod, part) or not isinstance(getattr(mod, part), nn.Module):
                setattr(mod, part, nn.Module())
            mod = getattr(mod, part)
        mod.register_parameter(parts[-1], prm)

    @property
    def device(self):
        return self._dev

    def to(self, *args, **kwargs):   # the model is built in place on its device
        dev = args[0] if args else kwargs.get("device", self._dev)
        if isinstance(dev, (str, torch.device)):
            d = torch.device(dev)
            if d.type != self._dev.type or (d.index is not None and d.index != self._dev.index):
                raise RuntimeError(f"{type(self).__name__} lives on {self._dev}; build it there (cfg.device; no CPU path exists)")
        return self

    def before_evaluation(self, **kwargs):   # base_model.py:102-103
        pass

    def state_dict(self, *a, **k):
        """Trainable parameters under the reference's key names and tensor layouts (runner_base.py:598-605 keeps
        exactly the requires_grad parameters)."""
        self.finish_update()
        out = OrderedDict()
        for name, _, rshape in self.store.specs:
            out[name] = to_reference_layout(self.store.p[name].detach(), rshape).cpu()
        return out

    def load_state_dict(self, sd, strict: bool = False):
        self.finish_update()                              # a delayed update must not land on top of the loaded values
        missing = []
        for name, ishape, _ in self.store.specs:
            if name in sd:
                self.store.p[name].copy_(from_reference_layout(sd[name].to(self._dev, F32), ishape))
            else:
                missing.append(name)
        if strict and missing:
            raise KeyError(missing)
        return missing

    @classmethod
    def from_config(cls, cfg):
        """`Myriad.from_config` / `MiniGPT4.from_config` (myriad.py:456-517, mini_gpt4.py:259-307) with the reference's own
        keys: vit_model, q_former_model, image_size, num_query_token, llama_model, drop_path_rate, use_grad_checkpoint,
        vit_precision, freeze_vit, freeze_qformer, freeze_llama, use_lora, k_shot, round_index, prompt_path,
        prompt_template, max_txt_len, end_sym, low_resource, device_8bit, ckpt.  The frozen weights and the tokenizer are
        loaded from the files those keys name (checkpoint.load_reference_weights); `cfg["weights"]` / `cfg["tokenizer"]`
        (an in-memory mapping keyed by the reference's state_dict names, e.g. synthetic.SyntheticWeights) replace the
        files in tests and in bench.py."""
        get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        if get("low_resource", False):
            raise NotImplementedError("low_resource (8-bit LLaMA + ViT on the CPU, myriad.py:186-192) is not part of the MI355X path")
        for k in ("freeze_vit", "freeze_qformer", "freeze_llama"):
            if get(k, True) is False:
                raise NotImplementedError(f"{k}: False -- the HIP path keeps ViT / Q-Former / LLaMA frozen (dgrad only), as every "
                                          "shipped recipe does (train_configs/*.yaml)")
        if get("drop_path_rate", 0) or get("use_grad_checkpoint", False):
            raise NotImplementedError("drop_path_rate / use_grad_checkpoint only matter for an unfrozen ViT")
        keys = ("max_txt_len", "end_sym", "k_shot", "round_index", "fixed_stage", "fixed_taskstage", "tokenizer",
                "vit_heads", "qf_heads", "llm_heads", "need_backward", "bos_token_id", "pad_token_id", "use_lora",
                "lora_r", "lora_alpha", "lora_dropout", "lora_seed", "num_query_token", "llm_eps")
        sub = {k: get(k) for k in keys if get(k) is not None}
        if get("max_txt_len") is None:
            sub["max_txt_len"] = 32                       # from_config defaults (myriad.py:483-484)
        if get("end_sym") is None:
            sub["end_sym"] = "\n"
        weights = get("weights")
        if weights is None:
            from .checkpoint import load_llama_tokenizer, load_reference_weights
            weights, meta = load_reference_weights(cfg, arch=cls.arch)
            for k, v in meta.items():
                sub.setdefault(k, v)
            if "tokenizer" not in sub:
                sub["tokenizer"] = load_llama_tokenizer(get("llama_model"))
        prompt_path, template = get("prompt_path", ""), get("prompt_template", "")
        if prompt_path:                                   # myriad.py:221-231
            with open(prompt_path, "r") as f:
                raw = f.read().splitlines()
            sub["prompt_list"] = [template.format(p_) for p_ in raw if "<ImageHere>" in p_]
        dev = get("device")
        if dev is None or dev == "cuda":
            dev = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0"
        model = cls(weights, sub, device=dev)
        ckpt = get("ckpt", "")
        if ckpt:
            print("Load BLIP2-LLM Checkpoint: {}".format(ckpt))
            model.load_state_dict(torch.load(ckpt, map_location="cpu")["model"], strict=False)
        return model

    # ------------------------------------------------------------------ hot path
    def _tokenize(self, samples, B, stage, training):
        """prompt_wrap tokenisation (myriad.py:358-366) + targets (:395-405); integer ids may be supplied directly."""
        if "before_ids" in samples:
            before, after = samples["before_ids"], samples["after_ids"]
            tgt, tmask = samples.get("target_ids"), samples.get("target_mask")
            return before, after, tgt, tmask
        tok = self.llama_tokenizer
        if tok is None:
            raise RuntimeError("no tokenizer: pass integer ids (before_ids/after_ids/target_ids/target_mask) in samples")
        key = {0: "question", 1: "question2", 2: "question3"}[stage] if self.arch == "myriad" else "question"
        qs = samples[key]
        if training and "aug_image" in samples:
            qs = list(qs) + list(qs)
        prompts = ["###Human: " + q + " ###Assistant: " for q in qs]
        bs, as_ = [], []
        for p in prompts:
            pb, pa = p.split("<ImageHere>")
            bs.append(tok(pb, return_tensors="pt", add_special_tokens=False).input_ids[0])
            as_.append(tok(pa, return_tensors="pt", add_special_tokens=False).input_ids[0])
        before, after = torch.stack(bs), torch.stack(as_)     # torch.stack => equal lengths, like myriad.py:371
        tgt = tmask = None
        if training:
            texts = list(samples["text_input"]) + list(samples.get("aug_text_input", []))
            tok.padding_side = "right"
            enc = tok([t + self.end_sym for t in texts], return_tensors="pt", padding="longest", truncation=True,
                      max_length=self.max_txt_len, add_special_tokens=False)
            tgt, tmask = enc.input_ids, enc.attention_mask
        return before, after, tgt, tmask

    def encode_img(self, image, maps, stage, save=True, vit_out=None):
        """`Myriad.encode_img` (myriad.py:241-272).  Returns img tokens [B, n_img, Dl] f32."""
        B = image.shape[0]
        x = vit_out if vit_out is not None else self.visual_encoder.forward(image)   # [B,257,Dv] f32, frozen
        N = x.shape[1]
        x2 = x.view(B * N, self.Dv)
        y = self.adaptor.forward(x2, save) if self.arch == "myriad" else x2
        enc_b, _ = ops.layernorm_fwd(y, self.ln_w, self.ln_b, 1e-5)   # ln_vision (blip2.py:119-125)
        use_ins = self.arch == "myriad" and stage in (1, 2)
        use_tok = self.arch == "myriad" and stage in (0, 1)
        nq = self.nq0 + (49 if use_ins else 0)
        # The map tokenizer's conv stack (networks.py:159-197) reads only the anomaly map and its tokens are first needed by the
        # prompt assembly, behind the whole Q-Former forward: it runs on the leaf side stream (its own split-K scratch; the stream
        # its backward runs on) beside that chain instead of ~0.3 ms behind it.  Joined before the tokens are used.
        tok_out, tok_ev = None, None
        if use_tok and self._leaf_aside and self._dev.type == "cuda":
            aux, main = self._side_stream("leaf"), torch.cuda.current_stream()
            aux.wait_stream(main)                                     # parameters (AdamW) and the map are ordered on the main stream
            maps.record_stream(aux)
            with torch.cuda.stream(aux):
                tok_out = self.ve_tok.forward(maps, save)
                tok_ev = torch.cuda.Event()
                tok_ev.record()
        q = torch.empty((B, nq, self.Dq), dtype=F32, device=self._dev)
        ops.copy3d(self.query_tokens_f32.expand(B, -1, -1), q[:, :self.nq0])
        if use_ins:
            ops.copy3d(self.ve_ins.forward(maps, save), q[:, self.nq0:])
        qo = self.qformer.forward(q, enc_b.view(B, N, self.Dv), save and self.arch == "myriad")
        qo_b = ops.to_bf16(qo.view(B * nq, self.Dq))
        if self.arch == "myriad":
            pw, pb = self.proj_w, self.proj_b                         # frozen (myriad.py:218-219)
        else:
            pw, pb = ops.to_bf16(self.store.p["llama_proj.weight"]), self.store.p["llama_proj.bias"]
        img = ops.gemm(qo_b, pw, bias=pb, out_dtype=F32)
        parts = [img.view(B, nq, self.Dl)]
        if use_tok:
            parts.append(self.store.p["VETokenizer.base_prompts"].view(1, 9, self.Dl).expand(B, -1, -1))
            if tok_ev is not None:
                torch.cuda.current_stream().wait_event(tok_ev)
                tok_out.record_stream(torch.cuda.current_stream())    # allocated on the side stream, read here
                parts.append(tok_out)
            else:
                parts.append(self.ve_tok.forward(maps, save))
        if save:
            self._ctx = dict(B=B, N=N, y=y, nq=nq, use_ins=use_ins, use_tok=use_tok, qo_b=qo_b)
        return parts

    def _assemble(self, parts, before, after, tgt, tmask):
        """inputs_embeds = cat(bos, before, img, after, text) (myriad.py:370-371,413-419) + mask/labels (:395-421)."""
        B = parts[0].shape[0]
        nb, na = before.shape[1], after.shape[1]
        n_img = sum(p.shape[1] for p in parts)
        T = 0 if tgt is None else tgt.shape[1]
        S = 1 + nb + n_img + na + T
        emb = torch.empty((B, S, self.Dl), dtype=F32, device=self._dev)
        col = 1 + nb
        img_slices = []
        for p in parts:
            ops.copy3d(p, emb[:, col:col + p.shape[1]])
            img_slices.append((col, p.shape[1]))
            col += p.shape[1]
        ids, rows = [], []
        for b in range(B):
            base = b * S
            ids.append(torch.tensor([self.bos_id]))
            rows.append(torch.tensor([base]))
            ids.append(before[b]); rows.append(base + 1 + torch.arange(nb))
            ids.append(after[b]); rows.append(base + 1 + nb + n_img + torch.arange(na))
            if T:
                ids.append(tgt[b]); rows.append(base + 1 + nb + n_img + na + torch.arange(T))
        ids = ops.h2d(torch.cat(ids).long(), self._dev)
        rows = ops.h2d(torch.cat(rows).int(), self._dev)
        self.llama.embed_tokens_into(ids, emb.view(B * S, self.Dl), rows)
        attn = labels = None
        if T:
            attn = torch.cat([torch.ones(B, 1 + nb + n_img + na, dtype=torch.long), tmask.long().cpu()], 1)
            targets = tgt.cpu().masked_fill(tgt.cpu() == self.pad_id, -100)
            labels = torch.cat([torch.full((B, 1 + nb + n_img + na), -100, dtype=torch.long), targets], 1)
        return emb, attn, labels, img_slices

    def _image_of(self, samples):
        """[B(+B aug), 3, H, W] f32 on the device.  The uploaded tensor is remembered in the batch dict (keyed by the identity
        of its host tensors): a look-ahead batch is touched twice -- its ViT forward one step early, the rest of its step
        later -- and must cross PCIe once."""
        src, aug = samples["image"], (samples.get("aug_image") if self.training else None)
        cached = samples.get("_image_dev") if isinstance(samples, dict) else None
        if cached is not None and cached[0] is src and cached[1] is aug:
            return cached[2]
        image = src if aug is None else torch.cat([src, aug])         # myriad.py:315-316
        image = image.to(self._dev, F32, non_blocking=True)
        if isinstance(samples, dict) and not src.is_cuda:
            samples["_image_dev"] = (src, aug, image)
        return image

    def attach_vision_expert(self, expert) -> None:
        """Optional in-model producer of the anomaly maps (`self.vision_expert` of the reference, myriad.py:83-90):
        a `myriad_amd.vision_expert.VisionExpertHIP`.  With it attached, samples may carry `expert_text_feats`
        [B,2,C] (the cached per-class [normal, abnormal] text embeddings) and `ref_images` [B*k,3,224,224] instead of
        ready-made maps; both map pairs then come from one trunk pass (myriad.py:331-345)."""
        self.vision_expert = expert

    def _maps_for(self, samples, key: str, image: torch.Tensor) -> torch.Tensor:
        if key in samples:
            return samples[key].to(self._dev, F32)
        expert = getattr(self, "vision_expert", None)
        if expert is None or "expert_text_feats" not in samples or "ref_images" not in samples:
            raise KeyError(f"samples['{key}'] is required (or attach_vision_expert() + samples['expert_text_feats'] and "
                           "samples['ref_images']): the vision expert is an upstream producer (SURVEY 2.1 row 10)")
        (zs, _), (os_, _) = expert.forward(image, samples["expert_text_feats"], samples["ref_images"])
        samples["anomaly_maps"], samples["oneshot_anomaly_maps"] = zs, os_      # both are computed once per batch
        return samples[key]

    def _forward_impl(self, samples, need_grad: bool, vit_out=None):
        if self.use_lora:
            self.lora.step_seed = (self.lora.step_seed + 1) if need_grad else self.lora.step_seed
        stage = self.fixed_stage if self.fixed_stage is not None else random.choice([0, 1, 2])   # myriad.py:378
        if self.arch != "myriad":
            stage = 0
        image = self._image_of(samples)
        maps = None
        if self.arch == "myriad":
            task = self.fixed_taskstage if self.fixed_taskstage is not None else random.choice([0, 1])  # :381
            key = "anomaly_maps" if task == 0 else "oneshot_anomaly_maps"
            maps = self._maps_for(samples, key, image)
        before, after, tgt, tmask = self._tokenize(samples, image.shape[0], stage, True)
        parts = self.encode_img(image, maps, stage, need_grad, vit_out=vit_out)
        emb, attn, labels, img_slices = self._assemble(parts, before, after, tgt, tmask)
        loss = self.llama.forward_loss(emb, attn, labels, save_for_backward=need_grad,
                                       lora_training=need_grad and self.training)   # peft's Dropout is off in eval()
        if need_grad:
            self._ctx["img_slices"] = img_slices
        return loss

    def forward(self, samples):
        """`Myriad.forward` (myriad.py:377-431).  Returns {"loss": 0-d tensor}; `loss.backward()` works."""
        need_grad = torch.is_grad_enabled() and self.training and self.store.total > 0
        self.finish_update()                              # a delayed update of train_step() lands before parameters are read
        with torch.no_grad():
            loss = self._forward_impl(samples, need_grad)
        if need_grad:
            return {"loss": _LossBridge.apply(self._anchor, self, loss)}
        return {"loss": loss}

    def _grads_live(self) -> bool:
        """True when gradients of an earlier backward are still attached (no optimizer.zero_grad() since): the reference
        loop accumulates over `accum_grad_iters` backward calls before it steps (base_task.py:256-271)."""
        return self._has_grads and any(prm.grad is not None for prm in self._params.values())

    def backward(self, gscale: float = 1.0, accumulate: bool = False, early_adamw=None, early_exchange=None):
        """Explicit backward of the last forward: fills the flat gradient buffer, multiplied by `gscale` (a GradScaler's
        loss scale; 1.0 otherwise).  Modules this step did not use keep zero gradients AND a zero use flag, so the gated
        AdamW leaves them untouched unless another rank used them (ParamStore).  With `accumulate` the result is added to
        the gradients already in the buffer.
        early_exchange = (DataParallel, segment index): the data-parallel exchange of that segment of the gradient buffer -- the
        map tokenizer's conv head, 91 % of the bytes -- is started as soon as the tokenizer's backward is queued, on the stream
        that ran it, so the collective runs under the Q-Former / adaptor backward (runner_base.py:94-98: DDP's bucketed overlap;
        here two buckets, cut where the backward's own order puts the bytes).  Every rank issues it at the same point, also a rank
        whose prompt stage skipped the tokenizer (its zeros are its contribution)."""
        c = self._ctx
        if c is None:
            raise RuntimeError("backward() without a training forward")
        # Accumulation (base_task.py:262-271): this call's gradients are added to the buffer's.  The map tokenizer's conv-head
        # weight -- 420 of the buffer's 460 MB -- accumulates IN PLACE in its weight-gradient GEMM (VENet.backward: the residual
        # operand), so only the other ~10 M gradients are set aside, zero-filled and added back (the former form
        # cloned, zero-filled and re-added the whole buffer).  A loss scale other than 1 keeps the whole-buffer form (it multiplies
        # this call's gradients only).
        head = None
        if accumulate and float(gscale) == 1.0 and self.arch == "myriad" and self.ve_tok.head_accumulates_in_place:
            o_n = self.store.offsets.get("VETokenizer.meta_net.15.weight")
            if o_n is not None and o_n[1] >= (1 << 20):
                head = (o_n[0], o_n[0] + o_n[1])
        self._acc_head = head
        fg = self.store.flat_g_comm
        if not accumulate:
            prev = None
        elif head is None:
            prev = fg.clone()
        else:
            prev = (fg[:head[0]].clone(), fg[head[1]:].clone())
        zeroed, self._g_zeroed = getattr(self, "_g_zeroed", None), None
        if zeroed is not None and not accumulate:
            torch.cuda.current_stream().wait_event(zeroed)        # train_step filled it on the leaf stream during the forward
        elif head is None:
            fg.zero_()
        else:
            fg[:head[0]].zero_()
            fg[head[1]:].zero_()
        used = {"lora"} if self.use_lora else set()
        if self.arch == "myriad":
            used.add("expert_adaptor")
            if c["use_tok"]:
                used.add("VETokenizer")
            if c["use_ins"]:
                used.add("VEInstructor")
        else:
            used.add("llama_proj")
        self.store.mark_used(used)
        # modules whose parameters receive a gradient in this accumulation window (the bridge path hands torch's optimiser
        # `.grad is None` for the others, exactly what autograd leaves for a module the forward never touched)
        self._bridge_used = (self._bridge_used | used) if accumulate else set(used)
        self._bwd_gscale, self._bwd_prev = float(gscale), prev
        self._early_done = set()
        demb = self.llama.backward(defer_lora_join=True)              # [B,S,Dl] f32; LoRA wgrads run on a side stream
        self._prefetch_vit_rest()                                     # the look-ahead's second piece: beside the light tail of the step
        B, nq = c["B"], c["nq"]
        (c0, n0) = c["img_slices"][0]
        dimg = torch.empty((B, nq, self.Dl), dtype=F32, device=self._dev)
        ops.copy3d(demb[:, c0:c0 + n0], dimg)
        dimg_b = ops.to_bf16(dimg.view(B * nq, self.Dl))
        if self.arch != "myriad":
            # MiniGPT-4 stage-2: llama_proj is the trainable piece (mini_gpt4.py:98-100): wgrad only
            ops.gemm_auto_f32(ops.transpose_to_bf16(dimg_b, 64), ops.transpose_to_bf16(c["qo_b"], 64),
                              self.store.g["llama_proj.weight"])
            self.store.g["llama_proj.bias"].copy_(ops.colsum(dimg.view(B * nq, self.Dl)))
            if self.llama.lora is not None:
                self.llama.lora.join_wgrads()
            self._finish_backward()
            return
        dqo = ops.gemm(dimg_b, self.proj_wT, out_dtype=F32).view(B, nq, self.Dq)
        if c["use_tok"]:
            (cb, _), (ct, _) = c["img_slices"][1], c["img_slices"][2]
            dbase = torch.empty((B, 9, self.Dl), dtype=F32, device=self._dev)
            ops.copy3d(demb[:, cb:cb + 9], dbase)
            self.store.g["VETokenizer.base_prompts"].view(-1).copy_(ops.colsum(dbase.view(B, 9 * self.Dl)))
            dtok = torch.empty((B, 9, self.Dl), dtype=F32, device=self._dev)
            ops.copy3d(demb[:, ct:ct + 9], dtok)
            # the tokenizer's conv stack is a leaf (its input is the anomaly map): its backward feeds only the optimiser and
            # runs on a side stream beside the Q-Former backward
            aux, main = (self._side_stream("leaf") if self._leaf_aside else torch.cuda.current_stream()), torch.cuda.current_stream()
            aux.wait_stream(main)
            leaf_keep = (dtok, self.ve_tok._saved)       # main-stream allocations the side stream reads: alive until the join
            with torch.cuda.stream(aux):
                self.ve_tok.backward(dtok, accumulate_head=self._acc_head is not None)
                if early_adamw is not None and self._leaf_aside:
                    # single-process step: the map tokenizer holds 91 % of the trainable parameters (its 105 M-weight head) and its
                    # gradient is complete here, right behind the LLaMA backward -- its AdamW (0.55 ms of HBM traffic) runs on the
                    # leaf stream beside the Q-Former backward instead of at the tail of the step.  Same kernel, same inputs.
                    self.store.adamw_module("VETokenizer", early_adamw[0], early_adamw[1])
                    self._early_done = {"VETokenizer"}
                if early_exchange is not None:
                    self._start_early_exchange(early_exchange)
                leaf_ev = torch.cuda.Event()
                leaf_ev.record()
        else:
            leaf_ev, leaf_keep = None, None
            if early_exchange is not None:                # unused at this rank's prompt stage: the segment's zeros travel now
                self._start_early_exchange(early_exchange)
        dq, denc = self.qformer.backward(dqo)
        ins_ev, ins_keep = None, None
        if c["use_ins"]:
            dins = torch.empty((B, 49, self.Dq), dtype=F32, device=self._dev)
            ops.copy3d(dq[:, self.nq0:], dins)
            if self._leaf_aside and self._dev.type == "cuda" and os.environ.get("MYRIAD_INS_ASIDE", "1") != "0":
                # the instructor's conv stack is a leaf too (its input is the anomaly map): its backward runs on the leaf stream
                # beside ln_vision's and the adaptor's backward
                aux, main = self._side_stream("leaf"), torch.cuda.current_stream()
                aux.wait_stream(main)
                ins_keep = (dins, self.ve_ins._saved)     # main-stream allocations the side stream reads: alive until the join
                with torch.cuda.stream(aux):
                    self.ve_ins.backward(dins)
                    ins_ev = torch.cuda.Event()
                    ins_ev.record()
            else:
                self.ve_ins.backward(dins)
        dy, _ = ops.layernorm_bwd(denc.view(B * c["N"], self.Dv), c["y"], self.ln_w, 1e-5)
        self.adaptor.backward(dy)
        if self.llama.lora is not None:
            self.llama.lora.join_wgrads()                 # the side-stream LoRA weight gradients land before anyone reads flat_g
        if leaf_ev is not None:
            torch.cuda.current_stream().wait_event(leaf_ev)
        if ins_ev is not None:
            torch.cuda.current_stream().wait_event(ins_ev)
        del leaf_keep, ins_keep                           # freed only now: later main-stream work is ordered behind the events
        self._finish_backward()

    def _start_early_exchange(self, early_exchange) -> None:
        """early_exchange = (dp, segment index[, (lr, weight_decay)]): start the segment's collective behind the current stream;
        with the third item the segment's optimiser update follows its collective on the exchange's own stream order -- the gated
        AdamW on this rank's slices of the segment (its 1/world piece in 'rs_ag') with grad_scale 1/world, then in 'rs_ag' the
        all-gather of the updated parameters -- so both run under the Q-Former backward as well (finish_update() does the rest of
        the buffer and bumps the step counters)."""
        dp, k = early_exchange[0], early_exchange[1]
        st = self.store
        if len(early_exchange) > 2 and getattr(dp, "supports_early_update", False):
            lr, wd = early_exchange[2]

            def then():
                st.adamw_pieces(dp.part_shards(st.total, k), lr, wd, grad_scale=1.0 / dp.world)
                dp.gather_part(st.flat_p, k)
            dp.start_part(st.flat_g_comm, st.total, k, then=then)
        else:
            dp.start_part(st.flat_g_comm, st.total, k)

    def _finish_backward(self):
        if self._bwd_gscale != 1.0:
            ops.scale_(self.store.flat_g, self._bwd_gscale)
        if self._bwd_prev is not None:
            fg = self.store.flat_g_comm
            if isinstance(self._bwd_prev, tuple):                 # the head's segment accumulated in place (backward())
                a, b = self._acc_head
                for pv, dst in ((self._bwd_prev[0], fg[:a]), (self._bwd_prev[1], fg[b:])):
                    if pv.numel():
                        ops.copy2d(pv.view(1, -1), dst.view(1, -1), accumulate=True)
            else:
                n = fg.numel()
                ops.copy2d(self._bwd_prev.view(1, n), fg.view(1, n), accumulate=True)
            self._bwd_prev = None
        self._has_grads = True
        self._reattach_grads()
        self._ctx = None

    def _reattach_grads(self):
        """Autograd-bridge path (`model(samples)["loss"].backward()` + an external torch optimiser): `.grad` aliases the flat
        buffer for the parameters of modules this step (or accumulation window) used, and is None for the others -- a module
        unused at this prompt stage (VEInstructor at stage 0, VETokenizer at stage 2; myriad.py:378,252,265) then gets no
        weight decay, no moment decay and no step increment from torch.optim.AdamW, as in the reference."""
        used = getattr(self, "_bridge_used", None)
        for name, prm in self._params.items():
            if used is not None and module_of(name) not in used:
                prm.grad = None
            elif prm.grad is None or prm.grad.data_ptr() != self.store.g[name].data_ptr():
                prm.grad = self.store.g[name]

    def _side_stream(self, name: str):
        return ops.side_stream(self._dev, name)

    def prefetch_vit(self, samples) -> None:
        """Launch the frozen ViT forward of a LATER step on a side stream.  The result is picked up by the train_step() that
        receives the same `samples` object.  Its split-K GEMMs use their own scratch (mh_set_stream_workspace).

        The forward is issued in two pieces.  Its GEMMs are work the chip has to do -- run beside the LLaMA part they cost
        almost their own duration -- but the step has two phases in which the main stream leaves most CUs idle (chains of
        5-30 us launches): the adaptor / Q-Former forward at its start, and the Q-Former / adaptor backward + AdamW at its
        end.  The first `MYRIAD_VIT_SPLIT` blocks are launched here, at the start of the step; the rest by
        _prefetch_vit_rest(), which backward() calls once the LLaMA backward is enqueued.
        From the second batch of a given size on, both pieces are replayed from hipGraphs captured on that stream (fixed
        shapes, frozen weights, no host-side arguments): enqueueing ~290 launches costs the launch thread ~0.2 ms instead of
        ~2.8 ms during which the main stream had nothing to run (MYRIAD_VIT_GRAPH=0 disables)."""
        if self._vit_stream is None:
            self._vit_stream = self._side_stream("vit")
        self._prefetch_vit_rest()                        # a look-ahead that was never consumed / finished: finish it first
        main = torch.cuda.current_stream()
        image = self._image_of(samples)
        key = tuple(image.shape)
        graph = self._vit_graphs.get(key)
        if graph is None and self._vit_graph_on and self._vit_seen.get(key, 0) >= 1 and image.dtype == F32:
            graph = self._capture_vit(image)
        self._vit_seen[key] = self._vit_seen.get(key, 0) + 1
        nb = len(self.visual_encoder.blocks)
        split = min(max(self._vit_split, 0), nb)
        self._vit_stream.wait_stream(main)               # inputs uploaded / buffers freed on the main stream so far
        with torch.cuda.stream(self._vit_stream), torch.no_grad():
            if graph is not None:
                graph["in"].copy_(image)
                graph["a"].replay()
                state = None
            else:
                state = self.visual_encoder.run_blocks(self.visual_encoder.embed(image), 0, split)
        self._vit_rest = (samples, graph, state, split)

    def _prefetch_vit_rest(self) -> None:
        """Second piece of a look-ahead ViT forward (see prefetch_vit): ordered behind everything the main stream has queued."""
        if self._vit_rest is None:
            return
        samples, graph, state, split = self._vit_rest
        self._vit_rest = None
        self._vit_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._vit_stream), torch.no_grad():
            if graph is not None:
                if graph["b"] is not None:
                    graph["b"].replay()
                out = graph["out"].clone()               # the next replay overwrites it while this step's backward reads `out`
            else:
                ve = self.visual_encoder
                out = ve.finish(ve.run_blocks(state, split, len(ve.blocks)))
            ev = torch.cuda.Event()
            ev.record()
        self._vit_prefetched = (samples, out, ev)

    def prepare_vit_graph(self, samples) -> None:
        """Capture the look-ahead graphs for this batch's image shape now (one eager pass + the capture: ~0.1 s), so that no
        later train_step pays for it.  Without this call the capture happens at the second look-ahead of a shape."""
        image = self._image_of(samples)
        key = tuple(image.shape)
        if not self._vit_graph_on or key in self._vit_graphs or image.dtype != F32:
            return
        if self._vit_stream is None:
            self._vit_stream = self._side_stream("vit")
        self._vit_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._vit_stream), torch.no_grad():
            self.visual_encoder.forward(image)            # first-call set-up of every kernel happens outside the capture
        self._vit_seen[key] = max(1, self._vit_seen.get(key, 0))
        self._capture_vit(image)

    def _capture_vit(self, image):
        """Capture the two pieces of visual_encoder.forward at this input shape into hipGraphs on the ViT side stream (whose
        split-K scratch the captured launches keep using on replay).  Called after one eager pass at the shape, so no kernel
        does first-call set-up inside the capture.  The pieces share one memory pool: piece b reads piece a's state."""
        torch.cuda.synchronize()
        ve = self.visual_encoder
        split = min(max(self._vit_split, 0), len(ve.blocks))
        static_in = torch.empty_like(image).contiguous()
        static_in.copy_(image)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: a data-parallel run has RCCL's watchdog thread polling events while this thread captures
        with torch.no_grad(), torch.cuda.graph(ga, stream=self._vit_stream, capture_error_mode="thread_local"):
            state = ve.run_blocks(ve.embed(static_in), 0, split)
        if split < len(ve.blocks):
            with torch.no_grad(), torch.cuda.graph(gb, pool=ga.pool(), stream=self._vit_stream, capture_error_mode="thread_local"):
                static_out = ve.finish(ve.run_blocks(state, split, len(ve.blocks)))
        else:
            gb, static_out = None, ve.finish(state)          # a shallow encoder: everything is in the first piece
        g = dict(a=ga, b=gb, out=static_out, state=state)
        g["in"] = static_in
        self._vit_graphs[tuple(image.shape)] = g
        return g

    def _take_prefetched_vit(self, samples):
        if self._vit_rest is not None and self._vit_rest[0] is samples:
            self._prefetch_vit_rest()                    # its second piece was never issued (no backward in between)
        if self._vit_prefetched is None or self._vit_prefetched[0] is not samples:
            return None
        _, out, ev = self._vit_prefetched
        self._vit_prefetched = None
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        out.record_stream(main)                          # allocated on the side stream, consumed (and freed) on this one
        return out

    def train_step(self, samples, lr: float, weight_decay: float = 0.05, allreduce=None, world: int = 1, dp=None,
                   overlap: bool = True, next_samples=None, accum_grad_iters: int = 1, accum_index=None):
        """forward + backward (+ gradient all-reduce) + fused AdamW: one optimisation step of
        `BaseTask._train_inner_loop` (base_task.py:233-271) without the autograd bridge.
        With a `DataParallel` (`dp`) and overlap=True the all-reduce + AdamW of step t are hidden behind the frozen
        ViT forward of step t+1 (95 % of the gradient bytes only exist after the whole LLaMA backward, so
        overlapping with backward hides nothing); call `finish_update()` after the last step.

        accum_grad_iters > 1 (base_task.py:262-271, runner_base.py:311-312): the gradients -- and the per-module use flags -- of
        that many consecutive calls are summed in the flat buffer (no division, as the reference); the exchange and the AdamW
        run on the last call of a window, with that call's lr.  A module is skipped by the gated AdamW only if no call of the
        window (on any rank) used it, which is what `optimizer.zero_grad()` (set_to_none) + DDP leave torch's AdamW.  The
        reference all-reduces every backward; one exchange of the window's sum is the same gradient.
        `accum_index` = the iteration index inside the epoch (RunnerBase passes it): the update then lands where the reference
        puts it, `(i + 1) % accum_grad_iters == 0` with i restarting every epoch (base_task.py:265), so when iters_per_epoch is
        not a multiple of accum_grad_iters the left-over gradients of an epoch's tail are folded into the next epoch's first
        update, as the reference's un-zeroed .grad does.  Without it the window is counted in calls."""
        with torch.no_grad():
            vit_out = self._take_prefetched_vit(samples)
            if next_samples is not None:
                self.prefetch_vit(next_samples)         # runs beside everything below
            if self._pending_update is not None and vit_out is not None:
                self.finish_update()
            elif self._pending_update is not None:
                # The previous step's gradient all-reduce is still in flight on the side stream: run this step's
                # frozen ViT forward (independent of the update) under it, then apply the delayed AdamW.
                vit_out = self.visual_encoder.forward(self._image_of(samples))
                self.finish_update()
            if (self._accum_count == 0 and self._leaf_aside and self._dev.type == "cuda"
                    and os.environ.get("MYRIAD_EARLY_ZERO", "1") != "0"):
                # the gradient buffer's zero fill (460 MB, ~55 us) leaves the chain between forward and backward: nothing writes a
                # gradient before the backward, and the last update (main stream, and the leaf stream itself) has read them
                aux, main = self._side_stream("leaf"), torch.cuda.current_stream()
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    self.store.flat_g_comm.zero_()
                    self._g_zeroed = torch.cuda.Event()
                    self._g_zeroed.record(aux)
            try:
                loss = self._forward_impl(samples, True, vit_out=vit_out)
            except Exception:
                self._g_zeroed = None                 # a forward that raised ("no valid labels"): no stale event for a later backward()
                raise
            accumulate = self._accum_count > 0                   # > 0: the flat buffer holds gradients no update has consumed
            due = accum_update_due(self._accum_count + 1, accum_grad_iters, accum_index)
            # no gradient exchange, no accumulation window: modules whose gradient is complete early are updated early (backward())
            # (world == 1 as well: a caller-scaled update -- `world` > 1 without dp / allreduce -- must see ONE grad_scale)
            early = (lr, weight_decay) if (due and not accumulate and (dp is None or dp.world == 1) and allreduce is None
                                           and world == 1 and os.environ.get("MYRIAD_EARLY_ADAMW", "1") != "0") else None
            # data parallel with the exchange overlapped: the tokenizer's segment of the gradient buffer starts its collective from
            # inside the backward (backward(): early_exchange); accumulation windows exchange the window's sum at its end instead
            early_x = None
            if (dp is not None and dp.world > 1 and overlap and due and not accumulate and self.arch == "myriad"
                    and hasattr(dp, "start_part") and os.environ.get("MYRIAD_DP_EARLY", "1") != "0"):
                k = self._dp_segment(dp)
                if k is not None:
                    early_x = (dp, k, (lr, weight_decay)) if os.environ.get("MYRIAD_DP_EARLY_ADAMW", "1") != "0" else (dp, k)
            self.backward(accumulate=accumulate, early_adamw=early, early_exchange=early_x)
            self._accum_count += 1
            if not due:
                return loss                                   # inside an accumulation window: no exchange, no update
            self._accum_count = 0
            if dp is not None and dp.world > 1 and overlap:
                dp.start(self.store.flat_g_comm, self.store.total)   # RCCL exchange on the side HIP stream (grads + use flags)
                self._pending_update = (dp, lr, weight_decay)
            else:
                shard = None
                if dp is not None and dp.world > 1:
                    dp.allreduce(self.store.flat_g_comm, self.store.total)
                    shard = self._dp_shards(dp)
                elif allreduce is not None:
                    allreduce(self.store.flat_g_comm)
                    world = max(world, 1)
                self.store.adamw_step(lr, weight_decay, grad_scale=1.0 / (dp.world if dp is not None else world), shard=shard,
                                      skip=self._early_done)
                if shard is not None:
                    dp.gather_params(self.store.flat_p)
        return loss

    def _dp_shards(self, dp):
        """rs_ag: the slices of the flat buffers this rank's AdamW owns (one per exchange segment); None: the whole buffer."""
        if getattr(dp, "mode", "allreduce") != "rs_ag":
            return None
        return dp.shards(self.store.total) if hasattr(dp, "shards") else [dp.shard(self.store.total)[:2]]

    def _dp_segment(self, dp):
        """Cut the exchange at the map tokenizer's conv-head weights (its weight-decay run of the flat buffer: 105 M of the 115 M
        trainables) and return that segment's index, or None when the model has no such run worth an exchange of its own.  The
        cut points are rounded inwards to multiples of 32 elements, so that at 2 / 4 / 8 ranks every segment's reduce-scatter
        tiles it exactly (ParamStore pads the total the same way)."""
        if getattr(self, "_dp_seg_of", None) is not None and self._dp_seg_of[0] is dp:
            return self._dp_seg_of[1]
        k = None
        rng = self.store.module_range("VETokenizer", True)
        if rng is not None:
            a, b = ops.round_up(rng[0], 32), rng[1] // 32 * 32
            if b - a >= (1 << 20):
                dp.set_segments(self.store.total, [a, b])
                segs = dp.segments(self.store.total)
                k = segs.index((a, b)) if (a, b) in segs else None
        self._dp_seg_of = (dp, k)
        return k

    def finish_update(self):
        """Apply a delayed optimiser update (overlap mode): wait for the gradient exchange, if any, then fused AdamW."""
        if self._pending_update is not None:
            dp, lr, wd = self._pending_update
            self._pending_update = None
            if dp is None:
                self.store.adamw_step(lr, wd)
                return
            dp.wait()
            shard = self._dp_shards(dp)
            done = dp.take_early_done() if hasattr(dp, "take_early_done") else set()      # segments updated behind their collective
            segs = dp.segments(self.store.total) if done else []
            self.store.adamw_step(lr, wd, grad_scale=1.0 / dp.world, shard=shard, exclude=[segs[k] for k in done])
            if shard is not None:
                dp.gather_params(self.store.flat_p, skip=done) if done else dp.gather_params(self.store.flat_p)

    @torch.no_grad()
    def generate(self, samples, **generate_kwargs):
        """`Myriad.generate` (myriad.py:433-454): stage-1 prompt layout, then the HF `generate(inputs_embeds=..., **kw)`
        contract for the arguments the evaluation script passes (evaluation_aqa_dataset.py:289-301):
        max_new_tokens, stopping_criteria (a list whose items carry `.stops` = id tensors, conversation.py:96-107, applied
        to batch row 0), do_sample + top_p + temperature (see LlamaHIP.greedy_generate: arg-max whenever p_max >= top_p,
        a host-side draw otherwise), min_length, use_cache.  Anything that would change the decoding rule and is not
        implemented raises instead of being ignored."""
        self.finish_update()
        kw = dict(generate_kwargs)
        stops = kw.pop("stop_ids", None)
        crit = kw.pop("stopping_criteria", None)
        if crit is not None:
            stops = [tuple(int(t) for t in torch.as_tensor(st).reshape(-1).tolist()) for c in crit for st in getattr(c, "stops", [])]
        if stops is None:
            stops = ()                                     # HF without a criterion stops on EOS / max_new_tokens only
        max_new = kw.pop("max_new_tokens", None)
        if max_new is None:
            max_len = kw.pop("max_length", None)
            max_new = 20 if max_len is None else None
        do_sample = bool(kw.pop("do_sample", False))
        top_p = float(kw.pop("top_p", 1.0))
        temperature = float(kw.pop("temperature", 1.0))
        min_length = int(kw.pop("min_length", 0))
        eos_id = kw.pop("eos_token_id", self.pad_id)
        kw.pop("pad_token_id", None)
        if not kw.pop("use_cache", True):
            raise NotImplementedError("use_cache=False: decode here always keeps a KV cache (same tokens)")
        top_k = kw.pop("top_k", 50)                         # HF's generation default; only a sampled (host-drawn) row sees it
        top_k = 0 if top_k is None else int(top_k)
        for k, neutral in (("num_beams", 1), ("repetition_penalty", 1.0), ("length_penalty", 1), ("num_return_sequences", 1)):
            v = kw.pop(k, neutral)
            if v not in (neutral, None):
                raise NotImplementedError(f"generate({k}={v}) is not implemented on the HIP decode path")
        generator = kw.pop("generator", None)
        if kw:
            raise TypeError(f"generate() got unsupported arguments: {sorted(kw)}")
        stage = 1 if self.arch == "myriad" else 0
        image = samples["image"].to(self._dev, F32)
        maps = None
        if self.arch == "myriad":
            key = "oneshot_anomaly_maps" if self.k_shot > 0 else "anomaly_maps"
            maps = self._maps_for(samples, key, image)
        before, after, _, _ = self._tokenize(samples, image.shape[0], stage, False)
        parts = self.encode_img(image, maps, stage, False)
        emb, _, _, _ = self._assemble(parts, before, after, None, None)
        emb = emb[:, 1:].contiguous()         # generate() wraps without BOS (myriad.py:446-449)
        if max_new is None:
            max_new = max(1, max_len - emb.shape[1])
        ids = self.llama.greedy_generate(emb, max_new_tokens=max_new, stop_ids=stops, min_length=min_length, eos_id=eos_id,
                                         do_sample=do_sample, top_p=top_p, temperature=temperature, generator=generator,
                                         top_k=top_k)
        self.last_generate_stats = self.llama.last_generate_stats
        return {"token_ids": ids, "ve_anomaly_maps": maps}


class StoppingCriteriaSub:
    """`StoppingCriteriaSub` (conversation.py:96-107) for callers that build the criterion themselves: holds `stops`, and
    `__call__(input_ids, scores)` is the reference's row-0 suffix test.  `generate()` reads `.stops`."""

    def __init__(self, stops=(), encounters=1):
        self.stops = list(stops)

    def __call__(self, input_ids, scores=None):
        for stop in self.stops:
            stop = torch.as_tensor(stop).to(input_ids.device)
            if torch.all((stop == input_ids[0][-len(stop):])).item():
                return True
        return False


class MiniGPT4HIP(MyriadHIP):
    """`MiniGPT4` (reference mini_gpt4.py:14-307): ViT -> ln_vision -> Q-Former(32 queries) -> llama_proj -> LLaMA."""
    arch = "mini_gpt4"


registry.register_model("myriad")(MyriadHIP)
registry.register_model("mini_gpt4")(MiniGPT4HIP)
Myriad = MyriadHIP
MiniGPT4 = MiniGPT4HIP
