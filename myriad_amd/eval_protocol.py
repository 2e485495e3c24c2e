"""AQA evaluation protocol: generated text -> anomaly label, per-scene accuracy / AUROC, jsonl records,
rank-sharded evaluation.  Host-side numpy; mirrors

  * `get_model_answer` / `get_performance`, scripts/eval_protocol/summary_results.py:8-182
  * the per-sample record the evaluation driver writes, evaluation_aqa_dataset.py:339-387
  * the result-directory summary table, scripts/eval_protocol/summary_results.py:185-245

The keyword lists that define the protocol are data (`eval_rules.json`, extracted from the reference script by
tools/make_golden_host.py); parity is pinned by tests/golden/eval_protocol.json, produced by the reference's own
functions on seeded synthetic answers."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_RULES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_rules.json")


@dataclass(frozen=True)
class AnswerRules:
    abnormal_words: Tuple[str, ...]
    normal_words: Tuple[str, ...]

    @staticmethod
    def default() -> "AnswerRules":
        d = json.load(open(_RULES_PATH))
        return AnswerRules(tuple(d["abnormal_words"]), tuple(d["normal_words"]))


def classify_answer(text: str, mode: int = 0, rules: Optional[AnswerRules] = None) -> int:
    """1 = anomalous, 0 = normal, -1 = undecided.  mode 0: free text, the abnormal phrases are tried FIRST (so
    'has no defect but is damaged' is anomalous); modes 2 / 3: three- / four-way multiple choice where the last
    letter is the 'normal' option (summary_results.py:8-123)."""
    if mode == 0:
        rules = rules or AnswerRules.default()
        if any(w in text for w in rules.abnormal_words):
            return 1
        if any(w in text for w in rules.normal_words):
            return 0
        return -1
    if mode in (2, 3):
        normal_letter = "C" if mode == 2 else "D"
        if normal_letter in text:
            return 0
        letters = "AB" if mode == 2 else "ABC"
        return 1 if any(f"is {c}." in text for c in letters) else -1
    raise NotImplementedError(f"answer mode {mode}")


def auroc(gt: Sequence[int], score: Sequence[float]) -> float:
    """Area under the ROC curve = Mann-Whitney U / (n_pos * n_neg) with average ranks for ties (what
    sklearn.metrics.roc_auc_score computes for binary labels)."""
    gt = np.asarray(gt)
    score = np.asarray(score, dtype=np.float64)
    n_pos, n_neg = int((gt == 1).sum()), int((gt == 0).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(score, kind="mergesort")
    ranks = np.empty(len(score), dtype=np.float64)
    s = score[order]
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[gt == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def scene_performance(records: Iterable[dict], rules: Optional[AnswerRules] = None) -> Tuple[float, float, float]:
    """(mean accuracy, mean AUROC, mean thresholded accuracy) over scenes (summary_results.py:126-182).
    Undecided answers (-1) are dropped from all three; the threshold is the largest score of a normal sample."""
    rules = rules or AnswerRules.default()
    scenes: Dict[str, Dict[str, list]] = {}
    for r in records:
        pred = classify_answer(r["output"], 0, rules)
        scene = r["scene"] if "scene" in r else r["image_path"].split("/")[1]
        s = scenes.setdefault(scene, {"gt": [], "pred": [], "score": []})
        key = "anomaly_map_scores" if "anomaly_map_scores" in r else "anomaly_score"
        if pred != -1:
            s["gt"].append(1 if r["is_anomaly"] else 0)
            s["pred"].append(pred)
            s["score"].append(float(r[key]))
    accs, aucs, th_accs = [], [], []
    for s in scenes.values():
        gt, pred, score = np.array(s["gt"]), np.array(s["pred"]), np.array(s["score"])
        th = score[gt == 0].max()
        accs.append(float((gt == pred).mean()))
        aucs.append(auroc(gt, score))
        th_accs.append(float((gt == (score > th).astype(gt.dtype)).mean()))
    return float(np.mean(accs)), float(np.mean(aucs)), float(np.mean(th_accs))


# ------------------------------------------------------------------------------------------------ records
def postprocess_generation(token_ids, tokenizer) -> List[str]:
    """evaluation_aqa_dataset.py:339-341, 369: ids clamped to [1, 40000], decoded, cut at the first '###'."""
    import torch
    ids = torch.clamp(torch.as_tensor(token_ids), 1, 40000)
    return [t.split("###")[0] for t in tokenizer.batch_decode(ids, add_special_tokens=False)]


def make_ad_record(image_id: int, img_path: str, is_anomaly: bool, output_text: str,
                   anomaly_map_max: Optional[float] = None) -> dict:
    """One jsonl line for the 'ad' / '1cls' / 'shot' task types (evaluation_aqa_dataset.py:361-384).  `output_text`
    is already cut at '###'; `anomaly_map_max` is the maximum of the 0..255 expert map."""
    item = {"image_id": int(image_id), "image_path": "/".join(img_path.split("/")[-5:]), "is_anomaly": bool(is_anomaly)}
    if anomaly_map_max is not None:
        ok = ("Yes" in output_text and bool(is_anomaly)) or ("No" in output_text and not bool(is_anomaly))
        item["error"] = "0" if ok else "1"
        item["output"] = output_text
        item["anomaly_score"] = str(round(float(anomaly_map_max) / 255.0, 4))
    return item


def write_jsonl(path: str, records: Iterable[dict]) -> None:
    with open(path, "w") as f:
        for r in records:
            f.write(json.dumps(r) + "\n")


def read_jsonl(path: str) -> List[dict]:
    return [json.loads(l) for l in open(path) if l.strip()]


# ------------------------------------------------------------------------------------------------ multi-GPU eval
def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Replicas only: rank r evaluates samples r, r+world, ... ; no collective on the data path."""
    return list(range(rank, n, world))


def merge_shards(paths: Sequence[str], out_path: str) -> int:
    """Concatenate the per-rank jsonl files and restore dataset order (image_id)."""
    recs = [r for p in paths for r in read_jsonl(p)]
    recs.sort(key=lambda r: r["image_id"])
    write_jsonl(out_path, recs)
    return len(recs)


# ------------------------------------------------------------------------------------------------ summary table
def summarize_result_dir(result_path: str, rules: Optional[AnswerRules] = None) -> List[str]:
    """`summary_v2.txt` rows (summary_results.py:185-245): one row per checkpoint ordinal parsed from
    `<prefix>_ckpt<N>_...jsonl`, columns 1cls / 1shot / 2shot / 4shot accuracy then the same four AUROCs.
    With several files for one cell the reference scores the FIRST file each time (`f_names[0]` inside its loop);
    that is kept so the table is identical."""
    files: Dict[int, Dict[str, List[str]]] = {}
    for f in sorted(os.listdir(result_path)):
        if not f.endswith("jsonl"):
            continue
        k = int(f.split("_")[1].replace("ckpt", ""))
        cell = files.setdefault(k, {"1cls": [], "1shot": [], "2shot": [], "4shot": []})
        if "1cls" in f:
            cell["1cls"].append(f)
        else:
            few = int(f.split("kshot=")[-1].split("_")[0])
            cell[f"{few}shot"].append(f)
    order = ["1cls", "1shot", "2shot", "4shot"]
    rows = ["Head 1cls(Myriad) 1shot(Myriad) 2shot(Myriad) 4shot(Myriad)   1cls(Expert) 1shot(Expert) 2shot(Expert) 4shot(Expert)"]
    for k in sorted(files):
        rec = {}
        for proc, names in files[k].items():
            rec[proc] = ["-", "-"]
            if names:
                try:
                    acc, auc, _ = scene_performance(read_jsonl(os.path.join(result_path, names[0])), rules)
                    rec[proc] = [f"{acc:.4f}", f"{auc:.4f}"]
                except Exception:
                    pass
        rows.append(f"{k:03d} " + "{} {} {} {}   {} {} {} {}".format(*([rec[p][0] for p in order] + [rec[p][1] for p in order])))
    return rows
