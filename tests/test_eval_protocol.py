"""the AQA evaluation protocol against golden outputs of the reference's own functions
(tools/make_golden_host.py ran scripts/eval_protocol/summary_results.py in the build container)."""
import json
import os

import numpy as np
import pytest

from myriad_amd import eval_protocol as E

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eval_protocol.json")))


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_answer_labels_match_reference(mode):
    got = [E.classify_answer(t, mode) for t in G["texts"]]
    assert got == G["labels"][str(mode)]


def test_abnormal_phrases_win_over_normal_ones():
    assert E.classify_answer("the cable has no defect but is damaged") == 1       # order of the two scans matters
    assert E.classify_answer("I cannot tell.") == -1
    with pytest.raises(NotImplementedError):
        E.classify_answer("x", mode=1)


@pytest.mark.parametrize("case", range(3))
def test_scene_performance_matches_reference(case):
    c = G["performance_cases"][case]
    acc, auc, th = E.scene_performance(c["records"])
    assert abs(acc - c["acc"]) < 1e-12 and abs(auc - c["auroc"]) < 1e-12 and abs(th - c["th_acc"]) < 1e-12


def test_auroc_ties_and_degenerate_input():
    assert E.auroc([0, 0, 1, 1], [0.1, 0.4, 0.35, 0.8]) == pytest.approx(0.75)
    assert E.auroc([0, 1, 0, 1], [0.5, 0.5, 0.5, 0.5]) == pytest.approx(0.5)     # all tied: average ranks
    with pytest.raises(ValueError):
        E.auroc([1, 1], [0.2, 0.3])


def test_records_sharding_and_summary(tmp_path):
    class Tok:
        def batch_decode(self, ids, add_special_tokens=False):
            return ["Yes, broken ### Human: next" if int(r[0]) == 1 else "No, fine" for r in ids]

    import torch
    texts = E.postprocess_generation(torch.tensor([[0, 5, 70000], [9, 2, 3]]), Tok())      # 0 -> 1 by the clamp
    assert texts == ["Yes, broken ", "No, fine"]
    r = E.make_ad_record(7, "/a/b/data/mvtec/bottle/test/bad/000.png", True, texts[0], anomaly_map_max=204.0)
    assert r == {"image_id": 7, "image_path": "mvtec/bottle/test/bad/000.png", "is_anomaly": True, "error": "0",
                 "output": "Yes, broken ", "anomaly_score": "0.8"}
    assert E.make_ad_record(8, "x/y.png", False, "Yes", anomaly_map_max=10)["error"] == "1"
    # replicas-only multi-GPU eval: disjoint cover, merge restores order
    n, world = 23, 4
    shards = [E.shard_indices(n, r, world) for r in range(world)]
    assert sorted(i for s in shards for i in s) == list(range(n))
    recs = G["performance_cases"][2]["records"]
    paths = []
    (tmp_path / "parts").mkdir()
    for rk in range(world):
        p = tmp_path / "parts" / f"part{rk}.jsonl"
        E.write_jsonl(str(p), [recs[i] for i in E.shard_indices(len(recs), rk, world)])
        paths.append(str(p))
    out = tmp_path / "Myriad_ckpt3_1cls.jsonl"
    assert E.merge_shards(paths, str(out)) == len(recs)
    acc, auc, th = E.scene_performance(E.read_jsonl(str(out)))
    assert abs(acc - G["performance_cases"][2]["acc"]) < 1e-12 and abs(auc - G["performance_cases"][2]["auroc"]) < 1e-12
    E.write_jsonl(str(tmp_path / "Myriad_ckpt3_kshot=2_x.jsonl"), G["performance_cases"][0]["records"])
    rows = E.summarize_result_dir(str(tmp_path))
    c0, c2 = G["performance_cases"][0], G["performance_cases"][2]
    assert rows[1] == f"003 {c2['acc']:.4f} - {c0['acc']:.4f} -   {c2['auroc']:.4f} - {c0['auroc']:.4f} -"
