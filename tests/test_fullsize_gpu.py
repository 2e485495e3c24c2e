"""Size-independent properties at the FULL benchmark configuration (BASELINE.json config 3 per-GPU shard: EVA-ViT-g 39L +
Q-Former 12L + Vicuna-7B 32L, Myriad stage 1, S = 148), where the fp32 oracle is too slow to be the checker:
  * determinism: two runs of the same step give bit-identical loss and gradients (fixed-order split-K, no atomics);
  * batch independence: a sample's loss does not depend on what else is in the batch (different M picks different
    GEMM kernels / K splits, so this also cross-checks the three GEMM kernels against each other at full size);
  * decode consistency: the hipGraph-replayed KV-cache decode emits exactly the ids of the eager token loop, and its first
    token is the arg-max of the prefill logits."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from myriad_amd.myriad import MyriadHIP  # noqa: E402
from myriad_amd.synthetic import SyntheticWeights, full_config  # noqa: E402

DEV = "cuda:0"


def samples(B, seed, vocab=32000):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    maps = torch.rand(B, 1, 224, 224, generator=g)
    before = torch.randint(3, vocab, (1, 4), generator=g).expand(B, -1).contiguous()
    after = torch.randint(3, vocab, (1, 28), generator=g).expand(B, -1).contiguous()
    tgt = torch.randint(3, vocab, (B, 16), generator=g)
    return dict(image=image.to(DEV), anomaly_maps=maps.to(DEV), oneshot_anomaly_maps=maps.to(DEV), before_ids=before,
                after_ids=after, target_ids=tgt, target_mask=torch.ones(B, 16, dtype=torch.long))


def pick(s, idx):
    return {k: (v[idx] if v.shape[0] > 1 else v) for k, v in s.items()}


@pytest.fixture(scope="module")
def model():
    cfg = full_config()
    m = MyriadHIP(SyntheticWeights(cfg, DEV, seed=0), dict(fixed_stage=1, fixed_taskstage=0, use_lora=True, lora_dropout=0.0),
                  device=DEV)
    m.train()
    return m


def loss_and_grad(model, s):
    model.store.flat_g.zero_()
    with torch.no_grad():
        loss = model._forward_impl(s, True)
    model.backward()
    torch.cuda.synchronize()
    return float(loss), model.store.flat_g.clone()


def test_step_is_deterministic(model):
    s = samples(4, seed=11)
    l1, g1 = loss_and_grad(model, s)
    l2, g2 = loss_and_grad(model, s)
    assert l1 == l2 and torch.equal(g1, g2)                      # bit-identical run to run
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    # a different batch must not reproduce it (the comparison above is not vacuous)
    l3, g3 = loss_and_grad(model, samples(4, seed=99))
    assert l3 != l1 and not torch.equal(g3, g1)


def test_loss_is_independent_of_batch_composition(model):
    s = samples(4, seed=12)
    with torch.no_grad():
        full = float(model._forward_impl(s, False))
        singles = [float(model._forward_impl(pick(s, slice(i, i + 1)), False)) for i in range(4)]
        pair = float(model._forward_impl(pick(s, slice(1, 3)), False))
    # every sample has 16 label tokens, so the batch loss is the mean of the per-sample losses
    assert abs(full - sum(singles) / 4) < 2e-3 * abs(full), (full, singles)
    assert abs(pair - (singles[1] + singles[2]) / 2) < 2e-3 * abs(pair)


def test_benchmark_shape_b8_is_deterministic_and_batch_independent(model):
    """The bench.py shape itself: 8 images per GPU, B*S = 1184 rows (the 256x256 kernel's 5 clamped row tiles, its K splits
    and slab-summing consumers, the 430-tile gate|up launch).  Bit-identical run to run; the batch-8 loss is the mean of
    the eight batch-1 losses and of the two batch-4 halves (M = 148 and 592 pick other kernels and split counts)."""
    s = samples(8, seed=21)
    l1, g1 = loss_and_grad(model, s)
    l2, g2 = loss_and_grad(model, s)
    assert l1 == l2 and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    with torch.no_grad():
        full = float(model._forward_impl(s, False))
        singles = [float(model._forward_impl(pick(s, slice(i, i + 1)), False)) for i in range(8)]
        halves = [float(model._forward_impl(pick(s, slice(i, i + 4)), False)) for i in (0, 4)]
    assert abs(full - sum(singles) / 8) < 2e-3 * abs(full), (full, singles)
    assert abs(full - sum(halves) / 2) < 2e-3 * abs(full), (full, halves)
    # the gradient of the batch-8 mean loss is the mean of the two half-batch gradients (linearity of the backward)
    _, ga = loss_and_grad(model, pick(s, slice(0, 4)))
    _, gb = loss_and_grad(model, pick(s, slice(4, 8)))
    gm = (ga + gb) / 2
    # M = 592 and M = 1184 round differently (other K splits, bf16 slabs): stated tolerance 1e-1 of max-abs on the worst
    # element (measured 5e-2: conv-stem sums over 1e5 positions) and direction cosine >= 0.995 over all 115 M gradients
    # (measured 0.9986: two bf16 backward passes through 32 layers, the same size as the 5e-2 gradient tolerance vs fp32)
    assert float((g1 - gm).abs().max()) < 1e-1 * float(g1.abs().max())
    cos = float((g1.double() @ gm.double()) / (g1.double().norm() * gm.double().norm()))
    assert cos >= 0.995, cos


def test_b1_full_size_backward_is_deterministic_and_linear(model):
    """BASELINE configs[1]: the same fine-tune step at batch 1 (M = 148 LLaMA rows, 257 ViT rows) takes other kernels than the batch-8
    step in BOTH directions -- the 160-row weight-streaming tiles with up to 16 K splits, their bf16 slabs, the SiLU-gate backward
    and the norm backwards summing many slabs, the two-workgroups-per-head attention backward -- and VERDICT r5 found no full-size
    assertion on its backward.  Here: loss and all 115 M gradients bit-identical run to run; the mean of four batch-1 gradients
    against the batch-4 gradient of the same samples (linearity of the backward across kernel families; the bounds of the
    batch-8-against-its-halves test: 1e-1 of max-abs on the worst element, cosine >= 0.995), and the batch-4 loss is the mean of the
    four batch-1 losses."""
    s = samples(4, seed=31)
    singles = []
    for i in range(4):
        si = pick(s, slice(i, i + 1))
        l1, g1 = loss_and_grad(model, si)
        l2, g2 = loss_and_grad(model, si)
        assert l1 == l2 and torch.equal(g1, g2), i                  # fixed-order slab sums at every split count: same bits
        assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
        singles.append((l1, g1))
    assert not torch.equal(singles[0][1], singles[1][1])            # not vacuous: other samples, other gradients
    l4, g4 = loss_and_grad(model, s)
    assert abs(l4 - sum(l for l, _ in singles) / 4) < 2e-3 * abs(l4), (l4, [l for l, _ in singles])
    gm = sum(g for _, g in singles) / 4
    worst = float((g4 - gm).abs().max()) / float(g4.abs().max())
    cos = float((g4.double() @ gm.double()) / (g4.double().norm() * gm.double().norm()))
    print(f"batch-1 mean vs batch-4 gradient: worst element {worst:.3e} of max-abs, cosine {cos:.5f}")
    assert worst < 1e-1 and cos >= 0.995, (worst, cos)
    # per module: the LoRA pairs and the adaptor are small sums that a loose global cosine could hide
    st = model.store
    # (peft's B = 0 at init: the A gradients are exactly zero, the B gradients are not)
    for name in (model.lora.names(0)[2], model.lora.names(31)[3], "expert_adaptor.conv1.weight", "VETokenizer.base_prompts"):
        o, n = st.offsets[name]
        a, b = g4[o:o + n].double(), gm[o:o + n].double()
        c = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        assert c >= 0.99, (name, c)


def test_bf16_split_k_slabs_stay_inside_the_stated_tolerances(model):
    """The 256x256 kernel hands its split-K partial sums over as bf16 slabs (each partial rounded once, 2^-9) -- also where
    the consumer writes the fp32 residual stream or an fp32 gradient (gemm.hip run_splitk; MYRIAD_SLAB_BF16=0 keeps fp32).
    Full-size model at the bench shape, same batch, both slab types: the loss moves by < 1e-3 relative and the 115 M
    gradients stay within the tolerance the gradients carry against the fp32 oracle anyway (5e-2 of max-abs, cosine >= 0.999)."""
    from myriad_amd import _lib
    lib = _lib.load()
    s = samples(8, seed=23)
    try:
        lib.mh_set_option(b"slab_bf16", 0)
        l32, g32 = loss_and_grad(model, s)
        lib.mh_set_option(b"slab_bf16", 1)
        l16, g16 = loss_and_grad(model, s)
    finally:
        lib.mh_set_option(b"slab_bf16", 1)
    assert l32 != l16                                             # the switch does reach the kernels
    assert abs(l16 - l32) < 1e-3 * abs(l32), (l16, l32)
    assert float((g16 - g32).abs().max()) < 5e-2 * float(g32.abs().max())
    cos = float((g16.double() @ g32.double()) / (g16.double().norm() * g32.double().norm()))
    assert cos >= 0.999, cos


def test_benchmark_shape_b8_train_step_with_lora_dropout(model):
    """One full optimisation step at the bench shape with `use_lora` and peft's dropout (p = 0.05) ON: finite, repeatable
    from the same state and step seed, the dropout changes the loss, and the LoRA / adapter parameters move."""
    st, lora = model.store, model.lora
    keep = (st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone(), st.step, st.steps_dev.clone(), lora.p, lora.step_seed)
    s = samples(8, seed=22)

    def restore():
        st.flat_p.copy_(keep[0]); st.flat_m.copy_(keep[1]); st.flat_v.copy_(keep[2]); st.step = keep[3]; st.steps_dev.copy_(keep[4])
        lora.step_seed = keep[6]

    try:
        # peft's B = 0 at init would make dropout invisible in the first forward: give B a value first
        for i in range(len(model.llama.layers)):
            for n in lora.names(i)[2:]:
                st.p[n].normal_(0.0, 0.02, generator=torch.Generator(device=DEV).manual_seed(100 + i))
        keep = (st.flat_p.clone(),) + keep[1:]
        lora.p = 0.0
        l_plain = float(model.train_step(s, 1e-3, 0.05))
        restore()
        lora.p = 0.05
        l_a = float(model.train_step(s, 1e-3, 0.05))
        torch.cuda.synchronize()
        p_a = st.flat_p.clone()
        restore()
        l_b = float(model.train_step(s, 1e-3, 0.05))
        torch.cuda.synchronize()
        assert l_a == l_b and torch.equal(p_a, st.flat_p)        # same state + same step seed -> same masks -> same bits
        assert l_a != l_plain and abs(l_a - l_plain) < 0.05 * abs(l_plain)
        assert torch.isfinite(st.flat_p).all()
        moved = (p_a - keep[0]).abs()
        for n in (lora.names(0)[0], lora.names(31)[3], "expert_adaptor.conv1.weight"):
            o, cnt = st.offsets[n]
            assert float(moved[o:o + cnt].max()) > 0, n
    finally:
        lora.p = keep[5]
        st.flat_p.zero_()
        model_init = keep[0]
        st.flat_p.copy_(model_init); st.flat_m.copy_(keep[1]); st.flat_v.copy_(keep[2]); st.step = keep[3]; st.steps_dev.copy_(keep[4])
        lora.step_seed = keep[6]
        for i in range(len(model.llama.layers)):                  # back to peft's B = 0
            for n in lora.names(i)[2:]:
                st.p[n].zero_()


def test_graph_replayed_decode_equals_eager_decode(model):
    model.eval()
    try:
        s = samples(2, seed=13)
        s = {k: v for k, v in s.items() if k not in ("target_ids", "target_mask")}
        kw = dict(max_new_tokens=12, stop_ids=((-1,),), min_length=0, eos_token_id=-5)
        a = model.generate(s, **kw)["token_ids"]
        import myriad_amd.llama as L
        orig = L.LlamaHIP.greedy_generate

        def eager(self, *args, **kwargs):
            kwargs["use_graph"] = False
            return orig(self, *args, **kwargs)

        L.LlamaHIP.greedy_generate = eager
        try:
            b = model.generate(s, **kw)["token_ids"]
        finally:
            L.LlamaHIP.greedy_generate = orig
        assert a.shape == (2, 12) and torch.equal(a, b)
        # the single-token step streams packed weight copies by default; the row-major path must give the same tokens
        llm = model.llama
        assert llm._packed is not None and llm.pack_decode
        llm.pack_decode = False
        try:
            d = model.generate(s, **kw)["token_ids"]
            assert llm._packed is None
        finally:
            llm.pack_decode = True
        assert torch.equal(a, d)
        c = model.generate(s, **dict(kw, max_new_tokens=1))["token_ids"]
        assert torch.equal(a[:, :1], c)                              # first token = arg-max of the prefill logits
    finally:
        model.train()


def test_decode_with_lora_attached_fused_step_equals_the_separate_launches(model):
    """The model that generates is the fine-tuned one: LoRA on q_proj / v_proj stays attached (unmerged, as peft leaves it).  Its
    single-token step runs the fused launches too -- norm + LoRA down projection as one launch in front of the bordered packed qkv
    weight, rotary + append inside the attention, norm / SiLU gate inside the MLP products -- and must give exactly the tokens of
    the launch-per-op path (MYRIAD_DECODE_FUSED=0), at batch 1 (every fusion active) and batch 3 (the <= 2-row ones fall back)."""
    model.eval()
    llm = model.llama
    assert llm.lora is not None
    try:
        kw = dict(max_new_tokens=10, stop_ids=((-1,),), min_length=0, eos_token_id=-5)
        for B in (1, 3):
            s = {k: v for k, v in samples(B, seed=31 + B).items() if k not in ("target_ids", "target_mask")}
            outs = []
            for fused in (True, False):
                llm.decode_fused = fused
                llm._decode_ws.clear()
                outs.append(model.generate(s, **kw)["token_ids"])
            assert outs[0].shape == (B, 10) and torch.equal(outs[0], outs[1])
    finally:
        llm.decode_fused = True
        llm._decode_ws.clear()
        model.train()


def test_decode_graph_and_buffers_survive_across_generate_calls(model):
    """The token-step graph, KV caches and device counters are kept per batch size and re-used by later generate() calls:
    interleaved calls with different prompts, lengths and batch sizes must give exactly the tokens a fresh state gives."""
    model.eval()
    try:
        kw = dict(stop_ids=((-1,),), min_length=0, eos_token_id=-5)
        strip = lambda d: {k: v for k, v in d.items() if k not in ("target_ids", "target_mask")}
        a_in, b_in, c_in = strip(samples(2, seed=21)), strip(samples(2, seed=22)), strip(samples(1, seed=23))
        llm = model.llama

        def fresh(inp, n):
            llm._decode_ws.clear()
            return model.generate(inp, max_new_tokens=n, **kw)["token_ids"]

        ref_a, ref_b, ref_c, ref_a_long = fresh(a_in, 10), fresh(b_in, 10), fresh(c_in, 10), fresh(a_in, 40)
        llm._decode_ws.clear()
        seq = [(a_in, 10, ref_a), (b_in, 10, ref_b), (c_in, 10, ref_c), (a_in, 10, ref_a), (a_in, 40, ref_a_long), (b_in, 10, ref_b)]
        for inp, n, want in seq:
            got = model.generate(inp, max_new_tokens=n, **kw)["token_ids"]
            assert torch.equal(got, want)
        assert any(ws["graph"] is not None for ws in llm._decode_ws.values())      # and the replayed graph was what ran
        assert not torch.equal(ref_a, ref_b)
    finally:
        model.train()


def test_vit_prefetch_on_a_side_stream_changes_nothing(model):
    """train_step(next_samples=...) issues the NEXT batch's frozen ViT forward on a side stream (its own split-K scratch)
    while this step runs, replayed from a hipGraph from the second batch of a shape on: four optimisation steps over four
    different batches must leave bit-identical losses, parameters and AdamW moments compared with the inline order."""
    st = model.store
    keep = (st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone(), st.step, st.steps_dev.clone())
    batches = [samples(2, seed=31 + i) for i in range(4)]

    def run(lookahead):
        st.flat_p.copy_(keep[0]); st.flat_m.copy_(keep[1]); st.flat_v.copy_(keep[2]); st.step = keep[3]; st.steps_dev.copy_(keep[4])
        losses = []
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if (lookahead and i + 1 < len(batches)) else None
            losses.append(float(model.train_step(b, 1e-3, 0.05, next_samples=nxt)))
        torch.cuda.synchronize()
        return losses, st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone()

    try:
        a = run(False)
        b = run(True)
        assert model._vit_stream is not None                     # the side stream was really used
        assert model._vit_graphs                                 # ... and the later prefetches were graph replays
        assert a[0] == b[0]
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert not torch.equal(a[1], keep[0])                    # and the steps did move the parameters
    finally:
        st.flat_p.copy_(keep[0]); st.flat_m.copy_(keep[1]); st.flat_v.copy_(keep[2]); st.step = keep[3]; st.steps_dev.copy_(keep[4])
        model._vit_prefetched, model._vit_rest = None, None


def test_minigpt4_arch_b8_is_deterministic_batch_independent_and_trains():
    """The MiniGPT-4 baseline arch (SURVEY 8d's second reported workload; mini_gpt4.py:153-257, minigpt4_stage2_finetune.yaml:
    32 queries, no expert tokens, only llama_proj trainable) at full size and the bench batch: B * S = 8 * 81 = 648 rows --
    2.53 row tiles of 256, other tile counts and K splits than the Myriad stage-1 shape.  Bit-identical run to run; the batch-8
    loss is the mean of the eight batch-1 losses and of the two batch-4 halves; the batch gradient is the mean of the halves'
    gradients; one optimisation step moves llama_proj and lowers the loss on the same batch."""
    from myriad_amd.myriad import MiniGPT4HIP
    cfg = full_config()
    m = MiniGPT4HIP(SyntheticWeights(cfg, DEV, seed=0, arch="mini_gpt4"), dict(fixed_stage=0, fixed_taskstage=0, use_lora=False),
                    device=DEV)
    m.train()
    assert m.store.n_params() == 768 * 4096 + 4096                # llama_proj.weight + bias and nothing else
    s = samples(8, seed=41)
    l1, g1 = loss_and_grad(m, s)
    l2, g2 = loss_and_grad(m, s)
    assert l1 == l2 and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    with torch.no_grad():
        full = float(m._forward_impl(s, False))
        singles = [float(m._forward_impl(pick(s, slice(i, i + 1)), False)) for i in range(8)]
        halves = [float(m._forward_impl(pick(s, slice(i, i + 4)), False)) for i in (0, 4)]
    assert abs(full - l1) < 1e-6 * abs(full)
    assert abs(full - sum(singles) / 8) < 2e-3 * abs(full), (full, singles)
    assert abs(full - sum(halves) / 2) < 2e-3 * abs(full), (full, halves)
    _, ga = loss_and_grad(m, pick(s, slice(0, 4)))
    _, gb = loss_and_grad(m, pick(s, slice(4, 8)))
    gm = (ga + gb) / 2
    assert float((g1 - gm).abs().max()) < 1e-1 * float(g1.abs().max())
    cos = float((g1.double() @ gm.double()) / (g1.double().norm() * gm.double().norm()))
    assert cos >= 0.995, cos
    p0 = m.store.flat_p.clone()
    la = float(m.train_step(s, 1e-3, 0.05))
    lb = float(m.train_step(s, 1e-3, 0.05))
    m.finish_update()
    assert la == l1 and lb < la and not torch.equal(m.store.flat_p, p0)


def test_last_layer_on_label_rows_only_changes_nothing_but_rounding(model):
    """forward_loss runs the LAST decoder layer's o_proj / post-attention norm / MLP on the label-bearing rows only (the other rows
    of its output feed nothing; in the backward their gradient is exactly zero down to that layer's attention).  Against the same
    model with every row kept (llama.last_layer_rows = False; other row counts pick other GEMM kernels and K splits, so the bits
    differ as between any two batch compositions): loss within 1e-3 relative, the 115 M gradients within 5e-2 of max-abs and
    cosine >= 0.999 -- and the d(inputs_embeds) the adapters receive likewise."""
    s = samples(8, seed=24)
    llm = model.llama
    try:
        llm.last_layer_rows = True
        la, ga = loss_and_grad(model, s)
        llm.last_layer_rows = False
        lb, gb = loss_and_grad(model, s)
    finally:
        llm.last_layer_rows = True
    assert la != lb or not torch.equal(ga, gb)                   # the switch reaches the kernels
    assert abs(la - lb) < 1e-3 * abs(lb), (la, lb)
    assert float((ga - gb).abs().max()) < 5e-2 * float(gb.abs().max())
    cos = float((ga.double() @ gb.double()) / (ga.double().norm() * gb.double().norm()))
    assert cos >= 0.999, cos
