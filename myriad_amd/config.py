"""Configuration loading for the entry points: the reference's `Config` (minigpt4/common/config.py:16-176) restated
on PyYAML (OmegaConf is not a dependency here).

Same precedence as the reference: model default YAML (chosen by `model.arch` + `model.model_type`) < the user's
`--cfg-path` YAML (sections `model`, `datasets`, `run`) < `--options key=value ...` dot-list overrides
(config.py:23-51, 60-88, 110-176).  The reference's shipped files (train_configs/*.yaml, eval_configs/*.yaml) are read
unchanged.  Sections are `Node`s: mappings with attribute access and `.get`, like an OmegaConf DictConfig.
"""
from __future__ import annotations

import copy
import json
import os
import re
from typing import Any, Dict, Iterable, List, Optional

import yaml

_PKG = os.path.dirname(os.path.abspath(__file__))
# model default YAMLs (PRETRAINED_MODEL_CONFIG_DICT of the registered classes, myriad.py:68-70, mini_gpt4.py:19-21)
MODEL_DEFAULTS = {("myriad", "pretrain_vicuna"): "configs/models/minigpt4.yaml",
                  ("mini_gpt4", "pretrain_vicuna"): "configs/models/minigpt4.yaml"}


class Node(dict):
    """dict with attribute access (cfg.model_cfg.arch, cfg.run_cfg.get("amp", False))."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_node(o: Any) -> Any:
    if isinstance(o, dict):
        return Node({k: to_node(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_node(v) for v in o]
    return o


def merge(base: dict, over: Optional[dict]) -> dict:
    """Recursive merge, `over` wins (OmegaConf.merge semantics for mappings; lists are replaced)."""
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


_SCI = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _coerce(o: Any) -> Any:
    """PyYAML (YAML 1.1) reads `1e-4` as a string where OmegaConf (the reference) reads a float: normalise."""
    if isinstance(o, dict):
        return {k: _coerce(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_coerce(v) for v in o]
    if isinstance(o, str) and _SCI.match(o.strip()):
        return float(o)
    return o


def load_yaml(path: str) -> dict:
    with open(path, "r") as f:
        return _coerce(yaml.safe_load(f) or {})


def _parse_scalar(s: str) -> Any:
    try:
        return _coerce(yaml.safe_load(s))
    except yaml.YAMLError:
        return s


def dotlist_to_dict(opts: Optional[Iterable[str]]) -> dict:
    """`--options a.b=1 c=x` (config.py:122-145: also accepts `a.b 1` pairs) -> nested dict."""
    opts = list(opts or [])
    if opts and all("=" not in o for o in opts):
        opts = [f"{k}={v}" for k, v in zip(opts[0::2], opts[1::2])]
    out: dict = {}
    for o in opts:
        key, _, val = o.partition("=")
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_scalar(val)
    return out


class Config:
    """`Config(args)` with `args.cfg_path`, `args.options` (argparse namespace) or `Config(path)`."""

    def __init__(self, args):
        if isinstance(args, str):
            cfg_path, options = args, None
        else:
            cfg_path, options = args.cfg_path, getattr(args, "options", None)
        self.args = args
        user = load_yaml(cfg_path)
        over = dotlist_to_dict(options)
        if "model" not in user:
            raise AssertionError("Missing model configuration file.")
        arch = user["model"].get("arch")
        model_type = (over.get("model", {}) or {}).get("model_type") or user["model"].get("model_type")
        if model_type is None:
            raise AssertionError("Missing model_type.")
        default_rel = MODEL_DEFAULTS.get((arch, model_type))
        if default_rel is None:
            raise AssertionError(f"Model '{arch}' / model_type '{model_type}' has no default configuration.")
        cfg = merge(load_yaml(os.path.join(_PKG, default_rel)), {"model": user["model"]})
        cfg = merge(cfg, {"run": user.get("run", {}) or {}})
        if "datasets" not in user:
            raise KeyError("Expecting 'datasets' as the root key for dataset configuration.")
        cfg = merge(cfg, {"datasets": user["datasets"] or {}})
        cfg = merge(cfg, over)
        self.config = to_node(cfg)

    @property
    def run_cfg(self) -> Node:
        return self.config["run"]

    @property
    def datasets_cfg(self) -> Node:
        return self.config["datasets"]

    @property
    def model_cfg(self) -> Node:
        return self.config["model"]

    def get_config(self) -> Node:
        return self.config

    def to_dict(self) -> dict:
        return json.loads(json.dumps(self.config))

    def pretty_print(self, log=print) -> None:
        log("\n=====  Running Parameters    =====")
        log(json.dumps(self.run_cfg, indent=4, sort_keys=True))
        log("\n======  Dataset Attributes  ======")
        for name, d in self.datasets_cfg.items():
            log(f"\n======== {name} =======")
            log(json.dumps(d, indent=4, sort_keys=True))
        log("\n======  Model Attributes  ======")
        log(json.dumps(self.model_cfg, indent=4, sort_keys=True))
