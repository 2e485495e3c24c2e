"""Drop-in boundary, end to end: the model built by `from_config` from a YAML with the REFERENCE's keys
and on-disk weight files in the reference's formats (tests/disk_fixture.py), driven through the reference's `samples`
schema with strings and a real LlamaTokenizer, the evaluation script's generate() arguments, the train / eval entry
points and a checkpoint round trip on the real model."""
import json
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from myriad_amd import checkpoint as C  # noqa: E402
from myriad_amd import datasets as D  # noqa: E402
from myriad_amd.config import Config  # noqa: E402
from myriad_amd.myriad import MyriadHIP, StoppingCriteriaSub  # noqa: E402
from myriad_amd.registry import registry  # noqa: E402
from oracle import myriad_ref as R  # noqa: E402
from tests import disk_fixture  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def fx(tmp_path_factory):
    return disk_fixture.build(str(tmp_path_factory.mktemp("ref_files")))


@pytest.fixture(scope="module")
def model(fx):
    cfg = Config(fx["train_yaml"])
    m = registry.get_model_class(cfg.model_cfg.arch).from_config(cfg.model_cfg)
    assert isinstance(m, MyriadHIP)
    return m


def _batch(n, train, seed=3):
    ds = D.SyntheticAnomalyDataset(n=n, seed=seed, train=train)
    return D.collate([ds[i] for i in range(n)])


def _ids(tok, samples, stage, training, end_sym="###"):
    """The reference's tokenisation (myriad.py:358-366, 395-405) done by hand, for the oracle's integer-id interface."""
    key = {0: "question", 1: "question2", 2: "question3"}[stage]
    qs = list(samples[key]) * (2 if training else 1)
    bs, as_ = [], []
    for q in qs:
        pb, pa = ("###Human: " + q + " ###Assistant: ").split("<ImageHere>")
        bs.append(tok(pb, return_tensors="pt", add_special_tokens=False).input_ids[0])
        as_.append(tok(pa, return_tensors="pt", add_special_tokens=False).input_ids[0])
    before, after = torch.stack(bs), torch.stack(as_)
    if not training:
        return before, after, None, None
    tok.padding_side = "right"
    enc = tok([t + end_sym for t in list(samples["text_input"]) + list(samples["aug_text_input"])], return_tensors="pt",
              padding="longest", truncation=True, max_length=160, add_special_tokens=False)
    return before, after, enc.input_ids, enc.attention_mask


def test_from_config_reads_the_reference_keys_and_files(model, fx):
    assert model.max_txt_len == 160 and model.end_sym == "###" and model.k_shot == 0
    assert model.prompt_list == ["###Human: <Img><ImageHere></Img> Describe this image in detail. ###Assistant: "]   # myriad.py:221-231
    assert len(model.visual_encoder.blocks) == 1                       # vit_depth pins the block count; the file has 2
    assert type(model.llama_tokenizer).__name__ == "LlamaTokenizer" and model.llama_tokenizer.pad_token_id == 2
    # the fine-tuned checkpoint named by `ckpt` was loaded over the trainables
    want = fx["sd"]["expert_adaptor.conv1.weight"]
    assert torch.equal(model.state_dict()["expert_adaptor.conv1.weight"], want)
    names = [n for n, _ in model.named_parameters()]
    assert "VETokenizer.base_prompts" in names and all(p.requires_grad for _, p in model.named_parameters())
    with pytest.raises(NotImplementedError):
        cfg = Config(fx["train_yaml"]).model_cfg
        cfg["low_resource"] = True
        MyriadHIP.from_config(cfg)


@pytest.mark.parametrize("seed", [0, 1, 5])
def test_string_samples_schema_forward_matches_the_oracle(model, fx, seed):
    """model(samples) with the dataset's dict (strings, aug_image doubling, question2 at stage 1, random stage draws) against
    the oracle fed with hand-tokenised ids."""
    samples = _batch(2, train=True)
    model.train()
    random.seed(seed)
    stage, task = random.choice([0, 1, 2]), random.choice([0, 1])       # the draws Myriad.forward makes (myriad.py:378,381)
    random.seed(seed)
    loss = model(samples)["loss"]
    image = torch.cat([samples["image"], samples["aug_image"]])
    maps = samples["anomaly_maps" if task == 0 else "oneshot_anomaly_maps"]
    before, after, tgt, tmask = _ids(model.llama_tokenizer, samples, stage, True)
    assert image.shape[0] == 4 and maps.shape[0] == 4 and before.shape[0] == 4 and tgt.shape[0] == 4
    with torch.no_grad():
        ref = R.model_forward(fx["sd"], image, maps, stage, before, after, tgt, tmask, arch="myriad")
    assert abs(float(loss.detach()) - float(ref)) < 5e-3 * abs(float(ref)), (stage, task, float(loss.detach()), float(ref))
    loss.backward()
    assert float(model.store.flat_g.abs().max()) > 0


def test_generate_takes_the_eval_scripts_arguments(model, fx):
    """evaluation_aqa_dataset.py:268-301: StoppingCriteriaList([StoppingCriteriaSub(stops=[...])]), do_sample=True,
    top_p=0.01, temperature=1.0, min_length=1, use_cache=True, max_new_tokens=90 -- passed through unchanged."""
    tok = model.llama_tokenizer
    samples = _batch(2, train=False, seed=5)
    model.eval()
    try:
        hashes = tok("###", add_special_tokens=False).input_ids
        stops = [torch.tensor(hashes).to(DEV), torch.tensor([7, 9]).to(DEV)]
        kw = {"max_new_tokens": 12, "stopping_criteria": [StoppingCriteriaSub(stops=stops)], "do_sample": True, "use_cache": True,
              "min_length": 1, "top_p": 0.01, "temperature": 1.0}
        out = model.generate(samples, **kw)
        ids = out["token_ids"]
        assert out["ve_anomaly_maps"].shape == samples["anomaly_maps"].shape and ids.shape[0] == 2
        st = model.last_generate_stats
        assert st["steps"] == ids.shape[1]
        before, after, _, _ = _ids(tok, samples, 1, False)
        with torch.no_grad():
            img = R.encode_img(fx["sd"], samples["image"], samples["anomaly_maps"], 1, "myriad")
            ew = fx["sd"]["llama_model.model.embed_tokens.weight"]
            wrapped = torch.cat([ew[before], img, ew[after]], 1)
            ids_ref, margins = R.greedy_generate(fx["sd"], wrapped, 32, max_new_tokens=12, stop_ids=[tuple(hashes), (7, 9)],
                                                 min_length=1, return_margins=True)
        if st["sampled_rows"] == 0:                       # every step had p_max >= top_p: the call WAS greedy decoding
            for t in range(min(ids.shape[1], ids_ref.shape[1])):
                if float(margins[:, t].min()) < 0.1:      # random weights: flat logits, compare while the oracle is decisive
                    break
                assert torch.equal(ids[:, t], ids_ref[:, t]), (t, ids, ids_ref)
        texts = tok.batch_decode(torch.clamp(ids, 1, 40000), add_special_tokens=False)   # evaluation_aqa_dataset.py:339-340
        assert len(texts) == 2
        for bad in ({"num_beams": 4}, {"repetition_penalty": 1.3}, {"bogus_flag": 1}):
            with pytest.raises((NotImplementedError, TypeError)):
                model.generate(samples, **dict(kw, **bad))
    finally:
        model.train()


def test_unused_modules_are_skipped_like_torch_adamw(model):
    """A module that took no part in the step has grad None in the reference: torch.optim.AdamW leaves its weights, moments
    and step count alone (no weight decay either).  Prompt stage 0 does not use VEInstructor, stage 2 not VETokenizer."""
    samples = _batch(2, train=True)
    model.train()
    saved = (model.fixed_stage, model.fixed_taskstage)
    try:
        model.fixed_taskstage = 0
        before = {n: model.store.p[n].clone() for n, _, _ in model.store.specs}
        steps0 = model.store.module_steps()
        model.fixed_stage = 0
        model.train_step(samples, lr=1e-3)
        steps1 = model.store.module_steps()
        assert steps1["VEInstructor"] == steps0["VEInstructor"] and steps1["VETokenizer"] == steps0["VETokenizer"] + 1
        for n, _, _ in model.store.specs:
            moved = not torch.equal(before[n], model.store.p[n])
            assert moved == (not n.startswith("VEInstructor.")), n
        o, cnt = model.store.offsets["VEInstructor.meta_net.15.weight"]
        assert float(model.store.flat_m[o:o + cnt].abs().max()) == 0.0
        model.fixed_stage = 2
        mid = {n: model.store.p[n].clone() for n, _, _ in model.store.specs}
        model.train_step(samples, lr=1e-3)
        steps2 = model.store.module_steps()
        assert steps2["VETokenizer"] == steps1["VETokenizer"] and steps2["VEInstructor"] == steps1["VEInstructor"] + 1
        assert all(torch.equal(mid[n], model.store.p[n]) == n.startswith("VETokenizer.") for n, _, _ in model.store.specs)
    finally:
        model.fixed_stage, model.fixed_taskstage = saved


def test_bridge_path_leaves_unused_modules_without_a_gradient(model):
    """The autograd-bridge path -- `model(samples)["loss"].backward()` with an external torch.optim.AdamW, what RunnerBase uses
    for accum_grad_iters > 1 -- hands torch `.grad is None` for a module the step did not use, so AdamW applies neither weight
    decay nor a step to it (the reference: autograd never touches VEInstructor at prompt stage 0)."""
    samples = _batch(2, train=True, seed=12)
    model.train()
    saved = (model.fixed_stage, model.fixed_taskstage)
    try:
        model.fixed_stage, model.fixed_taskstage = 0, 0
        params = dict(model.named_parameters())
        opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3, weight_decay=0.05)
        before = {n: p.detach().clone() for n, p in params.items()}
        opt.zero_grad(set_to_none=True)
        model(samples)["loss"].backward()
        none = {n for n, p in params.items() if p.requires_grad and p.grad is None}
        assert none and all(n.startswith("VEInstructor.") for n in none)
        assert all(params[n].grad is not None for n in params if n.startswith(("VETokenizer.", "expert_adaptor.")))
        opt.step()
        torch.cuda.synchronize()
        for n, p in params.items():
            if p.requires_grad:
                assert torch.equal(p.detach(), before[n]) == n.startswith("VEInstructor."), n
    finally:
        model.fixed_stage, model.fixed_taskstage = saved


def test_checkpoint_round_trip_on_the_real_model(model, fx, tmp_path):
    """runner_base.py:592-672 on the HIP model: two training steps, CheckpointManager.save, a FRESH model built from the same
    files + that checkpoint, identical loss and identical next step (parameters, moments, per-module step counts)."""
    samples = _batch(2, train=True, seed=8)
    model.train()
    saved = (model.fixed_stage, model.fixed_taskstage)
    model.fixed_stage, model.fixed_taskstage = 1, 0
    try:
        for _ in range(2):
            model.train_step(samples, lr=5e-4)
        mgr = C.CheckpointManager(str(tmp_path), max_checkpoints=1)
        path = mgr.save(model, 0, lr=5e-4, config={"run": {"seed": 42}})
        ck = torch.load(path, map_location="cpu")
        assert set(ck["model"]) == {n for n, _ in model.named_parameters()}            # trainable parameters only
        cfg = Config(fx["train_yaml"]).model_cfg
        cfg["ckpt"] = ""
        fresh = MyriadHIP.from_config(cfg)
        fresh.fixed_stage, fresh.fixed_taskstage = 1, 0
        assert C.CheckpointManager.load(fresh, path) == 1
        assert torch.equal(fresh.store.flat_p, model.store.flat_p) and torch.equal(fresh.store.flat_m, model.store.flat_m)
        assert torch.equal(fresh.store.flat_v, model.store.flat_v) and fresh.store.module_steps() == model.store.module_steps()
        with torch.no_grad():
            la, lb = float(model._forward_impl(samples, False)), float(fresh._forward_impl(samples, False))
        assert la == lb
        a, b = float(model.train_step(samples, lr=5e-4)), float(fresh.train_step(samples, lr=5e-4))
        assert a == b and torch.equal(fresh.store.flat_p, model.store.flat_p)
    finally:
        model.fixed_stage, model.fixed_taskstage = saved


def test_train_and_eval_entry_points(fx, tmp_path):
    """train.py / eval_aqa.py counterparts read the YAMLs, build the model through the registry, run the epoch loop with
    checkpoints + log.txt, then evaluate the saved checkpoint into the reference's jsonl record schema."""
    import eval_aqa
    import train
    runner = train.main(["--cfg-path", fx["train_yaml"], "--options", "run.max_epoch=2", "run.iters_per_epoch=2"])
    out = runner.output_dir
    assert sorted(f for f in os.listdir(out) if f.endswith(".pth")) == ["checkpoint_0.pth", "checkpoint_1.pth"]
    lines = open(os.path.join(out, "log.txt")).read().strip().split("\n")
    stats = [json.loads(l) for l in lines if l.startswith("{\"train_")]
    assert len(stats) == 2 and set(stats[0]) == {"train_lr", "train_loss"}
    ck = torch.load(os.path.join(out, "checkpoint_1.pth"), map_location="cpu")
    assert ck["epoch"] == 1 and ck["config"]["run"]["iters_per_epoch"] == 2
    res = str(tmp_path / "res.jsonl")
    path, records = eval_aqa.main(["--cfg-path", fx["eval_yaml"], "--dataset", "synthetic", "--bs", "2", "--out", res, "--options",
                                   "model.ckpt=" + os.path.join(out, "checkpoint_0.pth"), "--ckpt", "1"])
    rows = [json.loads(l) for l in open(path)]
    assert len(rows) == 4 and set(rows[0]) == {"image_id", "image_path", "is_anomaly", "error", "output", "anomaly_score"}
    assert all(r["error"] in ("0", "1") for r in rows)
