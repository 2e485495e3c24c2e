#!/usr/bin/env python3
"""Golden vectors for the host-side rows of SURVEY 8(f): eval protocol (f-3) and checkpoint I/O (f-4).

Runs the REFERENCE's own functions in this container (they cannot travel to the GPU box) on seeded synthetic inputs
and commits only inputs + expected outputs:
  tests/golden/eval_protocol.json   texts -> get_model_answer labels (modes 0/2/3); jsonl records -> get_performance
  tests/golden/ckpt_io.npz          pos_embed tables -> interpolate_pos_embed outputs
  myriad_amd/eval_rules.json        the protocol's keyword lists (data, extracted from the reference script's AST:
                                    scripts/eval_protocol/summary_results.py:8-95) -- the label vocabulary of the
                                    AQA protocol, shipped as configuration like a tokenizer vocabulary would be
python tools/make_golden_host.py [--ref /root/reference]
"""
import argparse
import ast
import importlib.util
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_summary_module(ref):
    # the reference script imports `jsonlines` (not installed here): a reader shim with the same open()/iterate surface
    jl = types.ModuleType("jsonlines")

    class _Reader:
        def __init__(self, path):
            self.rows = [json.loads(l) for l in open(path) if l.strip()]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __iter__(self):
            return iter(self.rows)

    jl.open = lambda path, mode="r": _Reader(path)
    sys.modules["jsonlines"] = jl
    spec = importlib.util.spec_from_file_location("ref_summary", os.path.join(ref, "scripts/eval_protocol/summary_results.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract_rules(ref):
    src = open(os.path.join(ref, "scripts/eval_protocol/summary_results.py")).read()
    rules = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in ("abnormal_words", "normal_words") and isinstance(node.value, ast.List):
            rules[node.targets[0].id] = [ast.literal_eval(e) for e in node.value.elts]
    assert set(rules) == {"abnormal_words", "normal_words"}
    return rules


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    S = load_summary_module(a.ref)
    rules = extract_rules(a.ref)
    json.dump({"source": "scripts/eval_protocol/summary_results.py:8-95 (get_model_answer, mode 0)",
               "abnormal_words": rules["abnormal_words"], "normal_words": rules["normal_words"]},
              open(os.path.join(ROOT, "myriad_amd", "eval_rules.json"), "w"), indent=1)

    rng = random.Random(7)
    fillers = ["The image shows a metal nut on a table.", "Looking at the texture,", "I think", "Overall,", "", "the object in the picture"]
    texts = []
    for w in rules["abnormal_words"] + rules["normal_words"]:
        texts.append(f"{rng.choice(fillers)} it {w} near the edge.")
    texts += ["Yes, there is an anomaly in the image.", "No, this hazelnut has no defect.", "The answer is A.", "The answer is C.",
              "It is B. because of the crack", "Option D", "I cannot tell.", "", "the cable has no defect but is damaged",
              "There are no visible issues", "yes", "no,", "is D.", "the correct choice is A. or C"]
    labels = {str(m): [S.get_model_answer(t, mode=m) for t in texts] for m in (0, 2, 3)}

    # records for get_performance: several scenes, unknown answers, both score keys
    cases = []
    for case_id, (n_scene, n_img, score_key, with_scene) in enumerate([(3, 40, "anomaly_score", False), (2, 25, "anomaly_map_scores", True),
                                                                       (4, 30, "anomaly_score", True)]):
        recs = []
        for s in range(n_scene):
            for i in range(n_img):
                anomaly = rng.random() < 0.5
                r = rng.random()
                if r < 0.7:
                    out = "Yes, there is an anomaly in the image." if (anomaly ^ (rng.random() < 0.2)) else "No, there is no defect."
                elif r < 0.9:
                    out = rng.choice(texts)
                else:
                    out = "I cannot tell."
                score = min(1.0, max(0.0, (0.65 if anomaly else 0.35) + rng.gauss(0, 0.2)))
                rec = {"image_id": s * 1000 + i, "image_path": f"data/scene{s}/test/bad/{i:03d}.png", "is_anomaly": anomaly,
                       "output": out, score_key: str(round(score, 4))}
                if with_scene:
                    rec["scene"] = f"scene{s}"
                recs.append(rec)
        with tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False) as f:
            for r in recs:
                f.write(json.dumps(r) + "\n")
            path = f.name
        acc, auroc, th_acc = S.get_performance(path)
        os.unlink(path)
        cases.append({"records": recs, "acc": float(acc), "auroc": float(auroc), "th_acc": float(th_acc)})
    json.dump({"texts": texts, "labels": labels, "performance_cases": cases},
              open(os.path.join(ROOT, "tests", "golden", "eval_protocol.json"), "w"))

    # ---- interpolate_pos_embed (eva_vit.py:373-394) through the reference function
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as MG
    M = MG.load_reference(a.ref)
    E = M["eva_vit"]
    out = {}
    g = torch.Generator().manual_seed(3)
    for tag, (orig, new, D) in {"up": (4, 7, 24), "down": (16, 8, 16), "same": (5, 5, 8), "real": (16, 16, 32)}.items():
        table = torch.randn(1, 1 + orig * orig, D, generator=g)
        fake = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=new * new), pos_embed=torch.zeros(1, 1 + new * new, D))
        ck = {"pos_embed": table.clone()}
        E.interpolate_pos_embed(fake, ck)
        out[f"{tag}_in"] = table.numpy()
        out[f"{tag}_out"] = ck["pos_embed"].float().numpy()
        out[f"{tag}_n"] = np.array(new * new)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ckpt_io.npz"), **out)
    print("wrote eval_rules.json, eval_protocol.json, ckpt_io.npz")


def param_order_golden(ref):
    """tests/golden/param_order.json: `named_parameters()` order, shapes and RunnerBase.optimizer's two-group split
    (runner_base.py:110-119) of Myriad's trainable modules, built from the REFERENCE's own networks.py classes registered
    in Myriad.__init__'s order (myriad.py:117-125).  The LoRA block follows peft's module tree (q_proj before v_proj,
    lora_A before lora_B; peft itself is not installed here, so that part is a restated order, marked as such)."""
    spec = importlib.util.spec_from_file_location("ref_networks", os.path.join(ref, "minigpt4/models/networks.py"))
    N = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(N)

    class Tree(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.expert_adaptor = N.LoraAdaptorV2(dims=1408, input_dim=4)
            self.VETokenizer = N.VETokenizer()
            self.VEInstructor = N.VEInstructorV2()

    t = Tree()
    names = [(n, list(p.shape)) for n, p in t.named_parameters() if p.requires_grad]
    wd, nwd = [], []
    for n, shp in names:
        (nwd if (len(shp) < 2 or "bias" in n or "ln" in n or "bn" in n) else wd).append(n)
    out = {"named_parameters": names, "weight_decay_group": wd, "no_decay_group": nwd,
           "peft_lora_order_restated": ["q_proj.lora_A", "q_proj.lora_B", "v_proj.lora_A", "v_proj.lora_B"]}
    path = os.path.join(ROOT, "tests", "golden", "param_order.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(names), "parameters")


if __name__ == "__main__":
    if "--param-order" in sys.argv:
        param_order_golden("/root/reference")
    else:
        main()
        param_order_golden("/root/reference")
