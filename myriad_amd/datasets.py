"""Datasets for the entry points (train.py / eval_aqa.py): the sample-dict schema of the reference's
`AnomalyDetectionDataset.__getitem__` (minigpt4/datasets/datasets/anomaly_detection.py:231-362) that `Myriad.forward`
and `Myriad.generate` consume.

Two builders are registered under the YAML's `datasets:` keys:
  anomaly_detection   the reference's dataset: jsonl annotations (`img_path`, `caption`, `is_anomaly`) under `vis_root`,
                      Resize(224, BICUBIC) -> CenterCrop -> [NSA / CutPaste self-supervised anomaly for the training
                      split, myriad_amd.self_sup.patch_ex] -> ToTensor -> CLIP Normalize; the question / answer strings
                      of anomaly_detection.py:35-39,299-347.
  synthetic           the same schema from seeded random tensors (no files): what the tests and bench use.

The vision expert's maps are inputs of the model (samples['anomaly_maps'] / ['oneshot_anomaly_maps'], SURVEY 2.1 row
10); a dataset may provide them (`synthetic` does) or the model computes them with an attached VisionExpertHIP.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

QUESTION_PROMPTS = [   # anomaly_detection.py:35-39 (prompt strings are data of the task, like a tokenizer vocabulary)
    "This image may be simulated by photo editing. According on IAD expert opinions, find out if there are defects in this image.",
    "This image may be simulated by photo editing. According to IAD expert opinions and corresponding visual descriptions, find out if there are defects in this image.",
    "This image may be simulated by photo editing. According to IAD expert visual descriptions, find out if there are defects in this image.",
]
NORMAL_DESCRIBE = "No, there exists no anomalies in the image."
ABNORMAL_DESCRIBE = "Yes, there exists anomalies in the image."
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

_BUILDERS: Dict[str, Callable] = {}


def register_dataset(name: str):
    def wrap(fn):
        _BUILDERS[name] = fn
        return fn
    return wrap


def build_datasets(datasets_cfg, split: str = "train") -> Dict[str, Dataset]:
    """`BaseTask.build_datasets` (tasks/base_task.py:36-66): one dataset per key of the YAML's `datasets:` section."""
    out = {}
    for name, cfg in (datasets_cfg or {}).items():
        if name not in _BUILDERS:
            raise KeyError(f"dataset '{name}' is not registered (have: {sorted(_BUILDERS)})")
        out[name] = _BUILDERS[name](cfg or {}, split)
    if not out:
        raise KeyError("Expecting at least one entry under 'datasets'")
    return out


def position_words(centers, size: int = 224) -> List[str]:
    """anomaly_detection.py:269-292: coarse 3x3 position of each pasted patch (center = (dim1, dim2) in pixels)."""
    names = [["upper left", "top", "upper right"], ["left", "center", "right"], ["lower left", "bottom", "lower right"]]
    out = []
    for c in centers:
        cx, cy = c[0] / size, c[1] / size
        i = 0 if cx <= 1 / 3 else (1 if cx <= 2 / 3 else 2)
        j = 0 if cy <= 1 / 3 else (1 if cy <= 2 / 3 else 2)
        out.append(names[i][j])
    return out


def sample_strings(is_aug_anomalous: Optional[bool]) -> dict:
    """The string fields of one sample (anomaly_detection.py:333-360, version 0)."""
    q = "<Img><ImageHere></Img>" + QUESTION_PROMPTS[1]
    d = {"question": q, "question2": q, "question3": q, "text_input": NORMAL_DESCRIBE}
    if is_aug_anomalous is not None:
        d["aug_text_input"] = ABNORMAL_DESCRIBE if is_aug_anomalous else NORMAL_DESCRIBE
    return d


class SyntheticAnomalyDataset(Dataset):
    """Seeded random samples in the reference schema: image / aug_image N(0,1) [3,224,224] (CLIP-normalised images are
    about unit variance), maps U[0,1) [1,224,224]."""
    DatasetName = "AnomalyDetection"          # the loader halves the batch for this name (runner_base.py:546-549)

    def __init__(self, n: int = 64, seed: int = 0, train: bool = True, image_size: int = 224, scene: str = "bottle"):
        self.n, self.seed, self.train, self.size, self.scene = n, seed, train, image_size, scene

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        S = self.size
        ret = {"image": torch.randn(3, S, S, generator=g), "scene": self.scene, "image_id": index,
               "is_anomaly": bool(index % 2), "img_path": f"synthetic/{self.scene}/test/good/{index:04d}.png",
               "anomaly_maps": torch.rand(1, S, S, generator=g), "oneshot_anomaly_maps": torch.rand(1, S, S, generator=g)}
        if self.train:
            ret["aug_image"] = torch.randn(3, S, S, generator=g)
            ret["aug_anomaly_maps"] = torch.rand(1, S, S, generator=g)
            ret["aug_oneshot_anomaly_maps"] = torch.rand(1, S, S, generator=g)
        ret.update(sample_strings(True if self.train else None))
        return ret


@register_dataset("synthetic")
def _build_synthetic(cfg, split):
    return SyntheticAnomalyDataset(n=int(cfg.get("num_samples", 64)), seed=int(cfg.get("seed", 0)), train=(split == "train"),
                                   image_size=int(cfg.get("image_size", 224)))


class AnomalyDetectionDataset(Dataset):
    """`AnomalyDetectionDataset` (anomaly_detection.py:104-362), host side: PIL decode + Resize/CenterCrop, the
    NSA / CutPaste augmentation on the uint8 crop, ToTensor + CLIP Normalize.  (`image_frontend.ImageFrontEndHIP` is the
    on-GPU, Pillow-bit-exact form of the resize/normalise pair for loaders that hand over decoded uint8 batches.)"""
    DatasetName = "AnomalyDetection"

    def __init__(self, vis_root: str, ann_paths: List[str], img_size: int = 224, crop_size: int = 224, stage: str = "train",
                 self_sup_mode: Optional[str] = None, seed: Optional[int] = None):
        """self_sup_mode None = the reference's recipe (anomaly_detection.py:118-141,254-264): per-class arguments, the patch
        resampled (`resize=True`) and blended with cv2.NORMAL_CLONE.  'swap' / 'uniform' are explicit opt-outs that keep the
        same geometry arguments but paste arithmetically without resampling (bit-pinned to the reference, no OpenCV step)."""
        self.vis_root, self.ann_paths, self.stage = vis_root, list(ann_paths), stage
        self.img_size, self.crop_size = img_size, crop_size
        self.annotation = []
        for ap in self.ann_paths:
            with open(os.path.join(vis_root, ap)) as f:
                self.annotation.extend(json.loads(l) for l in f if l.strip())
        self.self_sup_mode = self_sup_mode
        self.rng = np.random.RandomState(seed) if seed is not None else np.random

    def __len__(self):
        return len(self.annotation)

    def get_class_name(self, index):
        """anomaly_detection.py:224-230: 'mvtec' when the first annotation file's name contains MVTEC, else 'visa'; the class
        is the second path component of the image."""
        ds = "mvtec" if "MVTEC" in self.ann_paths[0] else "visa"
        return ds, self.annotation[index]["img_path"].split("/")[1]

    def _crop(self, index) -> np.ndarray:
        from PIL import Image
        img = Image.open(os.path.join(self.vis_root, self.annotation[index]["img_path"])).convert("RGB")
        w, h = img.size
        s = self.img_size                       # torchvision Resize(int): shorter side -> s, BICUBIC
        nw, nh = (s, int(s * h / w)) if w <= h else (int(s * w / h), s)
        img = img.resize((nw, nh), Image.BICUBIC)
        c = self.crop_size
        left, top = int(round((nw - c) / 2.0)), int(round((nh - c) / 2.0))
        return np.asarray(img.crop((left, top, left + c, top + c)))

    @staticmethod
    def _to_tensor(u8: np.ndarray) -> torch.Tensor:
        x = torch.from_numpy(np.ascontiguousarray(u8)).permute(2, 0, 1).float() / 255.0
        return (x - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]

    def __getitem__(self, index):
        ann = self.annotation[index]
        image = self._crop(index)
        ret = {"image": self._to_tensor(image), "scene": ann["img_path"].split("/")[1], "image_id": index,
               "is_anomaly": ann["is_anomaly"] == "1", "img_path": os.path.join(self.vis_root, ann["img_path"])}
        aug_anom = None
        if self.stage == "train":
            from . import self_sup
            src_index = int(self.rng.randint(len(self)))
            while src_index == index and len(self) > 1:
                src_index = int(self.rng.randint(len(self)))
            src = self._crop(src_index)
            ds, class_name = self.get_class_name(index)
            args = self_sup.self_sup_args(ds, class_name, visa_base="VISA" in self.ann_paths[0])
            if ds == "mvtec" and (args.get("width_bounds_pct") is None or args.get("intensity_logistic_params") is None):
                raise KeyError(f"MVTec class {class_name!r} is in neither argument table (the reference fails on the None it looks up)")
            if self.self_sup_mode is not None:
                args.update(mode=self.self_sup_mode, resize=False)
            aug, mask, boxes = self_sup.patch_ex(image, src, rng=self.rng, **args)
            while mask.sum() == 0:                      # anomaly_detection.py:263-265
                aug, mask, boxes = self_sup.patch_ex(image, src, rng=self.rng, **args)
            ret["aug_image"] = self._to_tensor(aug)
            aug_anom = bool(mask.sum() != 0)
        ret.update(sample_strings(aug_anom))
        return ret


@register_dataset("anomaly_detection")
def _build_anomaly_detection(cfg, split):
    info = cfg.get("build_info", {}) or {}
    root = info.get("vis_root", info.get("storage", "./data"))
    return AnomalyDetectionDataset(root, info.get("ann_paths", []), stage="train" if split == "train" else "test",
                                   img_size=int(cfg.get("img_size", 224)), crop_size=int(cfg.get("crop_size", 224)),
                                   self_sup_mode=cfg.get("self_sup_mode", None))


def collate(batch: List[dict]) -> dict:
    """torch's default collate for this schema: tensors stacked, numbers -> tensors, strings -> lists."""
    out = {}
    for k in batch[0]:
        v = [b[k] for b in batch]
        if isinstance(v[0], torch.Tensor):
            out[k] = torch.stack(v)
        elif isinstance(v[0], (bool, int, float)):
            out[k] = torch.tensor(v)
        else:
            out[k] = v
    # the model doubles the batch with aug_image (myriad.py:315-316): the maps of the augmented half ride along
    for key in ("anomaly_maps", "oneshot_anomaly_maps"):
        if key in out and ("aug_" + key) in out:
            out[key] = torch.cat([out[key], out.pop("aug_" + key)])
    return out
