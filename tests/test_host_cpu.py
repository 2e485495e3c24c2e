"""CPU tests of the host logic: LR schedule, parameter grouping, flat parameter store layout, checkpoint-layout
conversion, registry, synthetic-weight tables, and the N>1 data-parallel gradient exchange over gloo (world 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from myriad_amd.myriad import ParamStore, uses_weight_decay
from myriad_amd.networks import from_reference_layout, to_reference_layout, ve_param_specs
from myriad_amd.registry import registry
from myriad_amd.runner import DataParallel, LinearWarmupCosineLRScheduler
from myriad_amd.synthetic import full_config, shape_table
from oracle import myriad_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")


def test_lr_scheduler_matches_reference_golden():
    g = np.load(os.path.join(G, "optim.npz"))
    s = LinearWarmupCosineLRScheduler(None, max_epoch=10, iters_per_epoch=1600, min_lr=0.0, init_lr=1e-4,
                                      warmup_steps=0, warmup_start_lr=1e-6)
    for (e, i), lr in zip(g["pts"].tolist(), g["lrs"].tolist()):
        assert abs(s.step(e, i) - lr) <= 1e-12
    s2 = LinearWarmupCosineLRScheduler(None, max_epoch=2, iters_per_epoch=100, min_lr=1e-5, init_lr=1e-3,
                                       warmup_steps=20, warmup_start_lr=1e-6)
    for (e, i), lr in zip(g["pts2"].tolist(), g["lrs2"].tolist()):
        assert abs(s2.step(e, i) - lr) <= 1e-12
    assert registry.get_lr_scheduler_class("linear_warmup_cosine_lr") is LinearWarmupCosineLRScheduler


def test_registry_has_reference_model_names():
    assert {"myriad", "mini_gpt4"} <= set(registry.list_models())


def test_weight_decay_grouping_and_flat_store_layout():
    specs = [("expert_adaptor.conv1.weight", (4, 1408), (4, 1408)), ("expert_adaptor.conv2.weight", (1408, 4), (1408, 4))]
    specs += ve_param_specs("VETokenizer.", 4096, 5) + [("VETokenizer.base_prompts", (9, 4096), (9, 4096))]
    specs += ve_param_specs("VEInstructor.", 768, 1)
    for name, _, rshape in specs:
        assert uses_weight_decay(name, len(rshape)) == R.uses_weight_decay(name, len(rshape))
    st = ParamStore(specs, "cpu")
    assert st.n_params() == 110_732_912            # trainable parameters of the shipped Myriad recipe
    names = [s[0] for s in st.specs]
    n_wd_names = [n for n in names if uses_weight_decay(n, len(st.ref_shape[n]))]
    assert names[:len(n_wd_names)] == n_wd_names    # decay group first
    off_first_nowd = st.offsets[names[len(n_wd_names)]][0]
    assert off_first_nowd == st.n_wd and st.n_wd % 4 == 0 and st.total % 4 == 0
    # views alias the flat buffers
    st.p["VETokenizer.base_prompts"].fill_(3.0)
    o, n = st.offsets["VETokenizer.base_prompts"]
    assert float(st.flat_p[o:o + n].min()) == 3.0


def test_conv_layout_round_trip():
    w = torch.randn(16, 4, 3, 3)
    m = from_reference_layout(w, (16, 36))
    assert torch.equal(to_reference_layout(m, (16, 4, 3, 3)), w)
    # GEMM order is (ky, kx, ci)
    assert m[5, (1 * 3 + 2) * 4 + 3] == w[5, 3, 1, 2]


def test_synthetic_table_matches_parameter_count():
    t = shape_table(full_config(), "myriad")
    n = 0
    for k, (shape, _) in t.items():
        if k.startswith("llama_model.model.layers.") or k in ("llama_model.lm_head.weight",
                                                               "llama_model.model.embed_tokens.weight",
                                                               "llama_model.model.norm.weight"):
            c = 1
            for s in shape:
                c *= s
            n += c
    assert n == 6_738_415_616   # LLaMA-7B


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    flat = torch.randn(1000)
    # rank-dependent "unused module": rank 0 skips segment [200,500), rank 1 skips [600,700)  => zeros (DDP
    # find_unused_parameters semantics, reference runner_base.py:96-98 / myriad.py:378)
    if rank == 0:
        flat[200:500] = 0
    else:
        flat[600:700] = 0
    mine = flat.clone()
    dp = DataParallel(device=None)
    assert dp.world == world and dp.rank == rank
    dp.allreduce(flat)
    q.put((rank, mine.numpy().copy(), flat.numpy().copy()))          # by value: a tensor travels as an fd the parent must fetch
                                                                      # while this process is still alive
    dp.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, mine, red = q.get(timeout=120)
        got[r] = (torch.from_numpy(mine), torch.from_numpy(red))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = R.allreduce_mean_reference([{"g": got[0][0]}, {"g": got[1][0]}])["g"] * 2   # sum; AdamW applies 1/world
    for r in range(2):
        assert torch.allclose(got[r][1], want, atol=1e-6)
    assert torch.equal(got[0][1], got[1][1])


def _dp_flags_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [("expert_adaptor.conv1.weight", (4, 8), (4, 8)), ("VETokenizer.base_prompts", (9, 16), (9, 16)),
             ("VEInstructor.meta_net.0.bias", (4,), (4,)), ("llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight", (8, 16), (8, 16))]
    st = ParamStore(specs, "cpu")
    torch.manual_seed(7 + rank)
    st.flat_g.copy_(torch.randn(st.total))
    # per-rank random prompt stage (myriad.py:378, seeds differ per rank): rank 0 uses the tokenizer, rank 1 the instructor;
    # nobody uses LoRA this step
    used = {"expert_adaptor", "VETokenizer"} if rank == 0 else {"expert_adaptor", "VEInstructor"}
    st.used.copy_(torch.tensor([1.0 if m in used else 0.0 for m in st.modules]))
    dp = DataParallel(device=None)
    dp.allreduce(st.flat_g_comm)                       # ONE exchange carries gradients and use flags
    q.put((rank, list(st.modules), st.used.numpy().copy(), st.flat_g.numpy().copy()))     # by value (see _dp_worker)
    dp.barrier()
    dist.destroy_process_group()


def test_module_use_flags_ride_the_gradient_allreduce_world2_gloo():
    """The gated AdamW (mh_adamw_gated) must skip a module only when NO rank used it (DDP leaves the grad of a globally unused
    parameter None and torch's AdamW skips it; a module used on one rank gets that rank's gradient / world).  The per-module
    use flags sit at the tail of the flat gradient buffer, so the data-parallel sum delivers the global use count."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_flags_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, modules, used, grad = q.get(timeout=120)
        got[r] = (modules, torch.from_numpy(used), torch.from_numpy(grad))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    modules = got[0][0]
    assert modules == ["expert_adaptor", "VETokenizer", "VEInstructor", "lora"]
    for r in range(2):
        assert got[r][1].tolist() == [2.0, 1.0, 1.0, 0.0]      # used on both / one / one / no rank -> only LoRA is skipped
    assert torch.equal(got[0][2], got[1][2])


def _dp_modes_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_grad, n_flags = 1003 * 4, 4                        # not a multiple of world * 4 per shard: exercises the padding
    out = {}
    for mode, gd in (("allreduce", "f32"), ("allreduce", "bf16"), ("rs_ag", "f32"), ("rs_ag", "bf16")):
        torch.manual_seed(100 + rank)
        comm = torch.cat([torch.randn(n_grad), torch.tensor([1.0, float(rank), 0.0, 1.0 - rank])])
        mine = comm.clone()
        dp = DataParallel(device=None, mode=mode, grad_dtype=gd)
        dp.allreduce(comm, n_grad)
        lo, hi, per = dp.shard(n_grad)
        # sharded optimiser stand-in: p[shard] = -grad_sum[shard]; then every rank gathers every shard
        p = torch.zeros(n_grad)
        if mode == "rs_ag":
            p[lo:hi] = -comm[lo:hi]
            dp.gather_params(p)
        out[(mode, gd)] = (mine.numpy().copy(), comm.numpy().copy(), (lo, hi, per), p.numpy().copy())   # by value: no shm handles
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_modes_world2_gloo():
    """DataParallel modes behind the same train_step: fp32 / bf16 all-reduce and reduce-scatter + sharded update + all-gather
    (the two-phase exchange sized for xGMI's point-to-point links), world 2 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_modes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_grad = 1003 * 4
    for key in got[0]:
        mode, gd = key
        for r in range(2):
            m_, c_, sh_, p_ = got[r][key]
            got[r][key] = (torch.from_numpy(m_), torch.from_numpy(c_), sh_, torch.from_numpy(p_))
        total = got[0][key][0] + got[1][key][0]                   # what the sum must be (fp32)
        tol = 1e-6 if gd == "f32" else 2e-2
        for r in range(2):
            mine, red, (lo, hi, per), p = got[r][key]
            assert torch.allclose(red[n_grad:], total[n_grad:], atol=tol)           # flags: summed on every rank, every mode
            if mode == "allreduce":
                assert torch.allclose(red[:n_grad], total[:n_grad], atol=tol * (1 + total[:n_grad].abs().max()))
            else:
                assert per % 4 == 0 and lo == r * per and hi == min(lo + per, n_grad)
                assert torch.allclose(red[lo:hi], total[lo:hi], atol=tol * (1 + total.abs().max()))   # own shard reduced
                assert torch.allclose(p, -total[:n_grad], atol=tol * (1 + total.abs().max()))          # all shards gathered
        if mode == "rs_ag":
            assert torch.equal(got[0][key][3], got[1][key][3])        # every rank ends with the same parameters


def _dp_segments_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_grad, n_flags = 4096 + 32 * 37, 4                  # total and both cut points multiples of 32, as ParamStore / _dp_segment make them
    cuts = [256, 256 + 32 * 100]                         # [small head | the early segment | tail]
    out = {}
    for mode, gd in (("allreduce", "f32"), ("allreduce", "bf16"), ("rs_ag", "f32"), ("rs_ag", "bf16")):
        res = []
        for segmented in (False, True):
            torch.manual_seed(300 + rank)
            comm = torch.cat([torch.randn(n_grad), torch.tensor([1.0, float(rank), 0.0, 1.0])])
            dp = DataParallel(device=None, mode=mode, grad_dtype=gd)
            if segmented:
                dp.set_segments(n_grad, cuts)
                assert dp.segments(n_grad) == [(0, 256), (256, 3456), (3456, n_grad)]
                p_early = torch.zeros(n_grad)

                def then():                              # the segment's update behind its collective: own slices, then its all-gather
                    for lo, hi in dp.part_shards(n_grad, 1):
                        p_early[lo:hi] = -comm[lo:hi]
                    dp.gather_part(p_early, 1)
                dp.start_part(comm, n_grad, 1, then=then)   # the early segment first, as MyriadHIP.backward issues it (+ the flags) ...
                flags_after_part = comm[n_grad:].clone()
                dp.start(comm, n_grad)                   # ... then everything else (the flags are NOT summed a second time)
                dp.wait()
                assert torch.equal(flags_after_part, comm[n_grad:])
                assert dp.take_early_done() == {1} and dp.take_early_done() == set()
                try:
                    dp.set_segments(n_grad + 32, cuts)   # the segmentation is fixed once set
                    raise AssertionError("set_segments accepted a second geometry")
                except ValueError:
                    pass
            else:
                dp.allreduce(comm, n_grad)
            pieces = dp.shards(n_grad)
            assert len(pieces) == (3 if segmented else 1)
            p = torch.zeros(n_grad)
            if mode == "rs_ag":
                for lo, hi in pieces:
                    p[lo:hi] = -comm[lo:hi]
                if segmented:                            # segment 1 was updated and gathered behind its collective
                    p[256:3456] = p_early[256:3456]
                    dp.gather_params(p, skip={1})
                else:
                    dp.gather_params(p)
            else:
                p = -comm[:n_grad].clone()
                if segmented:
                    assert torch.equal(p_early[256:3456], p[256:3456])
            res.append((comm[n_grad:].numpy().copy(), p.numpy().copy(), pieces))
        out[(mode, gd)] = res
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_segmented_exchange_equals_the_single_exchange_gloo(world):
    """the gradient buffer exchanged in segments -- the map tokenizer's segment started early (start_part),
    the rest + the use flags at the end of the backward (start) -- hands every rank the same sums as ONE exchange of the whole
    buffer, in both modes and both wire types (element-wise sums: bit-equal also on the bf16 wire), and in rs_ag every rank owns
    one piece of EVERY segment (its AdamW shards) and gathers all of them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_segments_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for key in got[0]:
        for r in range(world):
            (f0, p0, pc0), (f1, p1, pc1) = got[r][key]
            assert (f0 == f1).all() and f0.tolist() == [float(world), sum(range(world)), 0.0, float(world)]
            if world == 2:
                assert (p0 == p1).all(), (key, r)                  # segmented == single, bit for bit (a two-term sum has one order)
            else:                                                  # gloo's ring adds four terms in an order that depends on the chunking
                tol = 1e-5 if key[1] == "f32" else 3e-2
                assert np.allclose(p0, p1, rtol=0, atol=tol * (1 + np.abs(p0).max())), (key, r)
            assert (p1 == got[0][key][1][1]).all()                 # and every rank holds the same result
            if key[0] == "rs_ag":
                assert pc1 == [(lo + r * ((hi - lo) // world), lo + (r + 1) * ((hi - lo) // world)) for lo, hi in
                               ((0, 256), (256, 3456), (3456, 4096 + 32 * 37))]


def _dp_world4_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [("expert_adaptor.conv1.weight", (4, 1408), (4, 1408)), ("VETokenizer.base_prompts", (9, 4096), (9, 4096)),
             ("VEInstructor.meta_net.0.bias", (4,), (4,)), ("VEInstructor.meta_net.15.weight", (768, 1024, 1, 1), (768, 1024, 1, 1)),
             ("llama_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight", (8, 4096), (8, 4096))]
    out = {}
    # ragged use: the prompt stage is drawn per rank (myriad.py:378): tokenizer on ranks 0 and 3, instructor on rank 1 only,
    # adaptor everywhere, LoRA nowhere
    used_by_rank = [{"expert_adaptor", "VETokenizer"}, {"expert_adaptor", "VEInstructor"}, {"expert_adaptor"},
                    {"expert_adaptor", "VETokenizer"}] + [{"expert_adaptor"}] * (world - 4)
    for mode in ("allreduce", "rs_ag"):
        st = ParamStore(specs, "cpu")
        assert st.total % 32 == 0 and st.total >= st.n_used       # padded once: every shard of 2 / 4 / 8 ranks tiles it exactly
        torch.manual_seed(50 + rank)
        st.flat_g[:st.n_used].copy_(torch.randn(st.n_used))
        for name, _, _ in st.specs:                               # an unused module contributes an exact zero gradient
            from myriad_amd.myriad import module_of
            if module_of(name) not in used_by_rank[rank]:
                st.g[name].zero_()
        flags = torch.tensor([1.0 if m in used_by_rank[rank] else 0.0 for m in st.modules])
        st.used.copy_(flags)
        mine = st.flat_g.clone()
        dp = DataParallel(device=None, mode=mode)
        n_alloc0 = len(dp._bufs)
        for _ in range(2):                                        # two exchanges: the staging buffers are allocated once
            st.flat_g.copy_(mine)
            st.used.copy_(flags)
            dp.allreduce(st.flat_g_comm, st.total)
        lo, hi, per = dp.shard(st.total)
        assert per * world == st.total
        p = torch.zeros(st.total)
        if mode == "rs_ag":
            p[lo:hi] = -st.flat_g[lo:hi]
            dp.gather_params(p)
            st.moments_complete = False
            dp.gather_state(st)
            assert st.moments_complete
        out[mode] = (mine.numpy().copy(), st.flat_g.numpy().copy(), st.used.numpy().copy(), p.numpy().copy(), (lo, hi),
                     len(dp._bufs) - n_alloc0, list(st.modules))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_exchange_world4_gloo_with_ragged_module_use(world):
    """Four ranks -- and eight, the rank count of BASELINE configs[2] --, each with its own prompt stage: the per-module use
    flags are summed by the same exchange that sums the gradients (all-reduce) or ride beside it (reduce-scatter + all-gather);
    the padded flat buffer needs no per-step concatenation, and the staging buffers of `rs_ag` are allocated once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_world4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for mode in ("allreduce", "rs_ag"):
        mine = [torch.from_numpy(got[r][mode][0]) for r in range(world)]
        total = sum(mine)
        modules = got[0][mode][6]
        want_flags = [float(world) if m == "expert_adaptor" else 2.0 if m == "VETokenizer" else 1.0 if m == "VEInstructor" else 0.0
                      for m in modules]
        for r in range(world):
            _, red, used, p, (lo, hi), n_new, _ = got[r][mode]
            red, p = torch.from_numpy(red), torch.from_numpy(p)
            assert used.tolist() == want_flags, (mode, r, used)
            if mode == "allreduce":
                assert torch.allclose(red, total, atol=1e-5) and n_new == 0
            else:
                assert torch.allclose(red[lo:hi], total[lo:hi], atol=1e-5)
                assert torch.allclose(p, -total, atol=1e-5)
                assert n_new <= 1                                     # gloo path: one persistent staging buffer for the all-gather
        if mode == "rs_ag":
            assert all(np.array_equal(got[0][mode][3], got[r][mode][3]) for r in range(1, world))


def test_gemm_policy_for_the_steps_shapes():
    """mh_gemm_plan is host-only logic (no launch): which kernel / how many K splits the library picks.  Pins the
    policy for the shapes of the fine-tune step (B*S = 1184 LLaMA rows, 2056 ViT rows) and the decode token."""
    import ctypes

    from myriad_amd import _lib, ops
    lib = _lib.load()
    try:
        # the plan only looks at the workspace SIZE; the pointer is never dereferenced without a launch
        lib.mh_set_workspace(ctypes.c_void_p(0x1000), 256 << 20)
        plan = ops.gemm_plan
        assert plan(1, 12288, 4096) == (0, 1) and plan(16, 32000, 4096) == (0, 1)        # decode rows: weight streaming
        assert plan(1184, 12288, 4096) == (2, 1)                                         # qkv: 5 x 48 = 240 tiles of 256^2
        assert plan(1184, 22016, 4096) == (2, 1)                                         # gate|up: 430 tiles, no split
        assert plan(1184, 4096, 22016) == (2, 3)                                         # dgrad: 80 tiles -> 240 workgroups
        assert plan(1184, 4096, 4096) == (2, 3)
        assert plan(2056, 6144, 1408) == (2, 1)                                          # ViT fc1
        assert plan(2056, 1408, 1408) == (3, 1)                                          # ViT proj: 187 tiles of 128^2 < 256 CUs -> 128x64 tiles
        assert plan(4096, 25600, 128)[0] == 1                                            # K = 128 (VETokenizer head wgrad): 128x128 kernel
        assert plan(648, 2304, 768)[0] == 3 and plan(648, 3072, 768)[0] == 3             # Q-Former sizes: 128x64 tiles fill more CUs
        assert plan(256, 768, 768) == (6, 1) and plan(648, 768, 2304) == (6, 1)          # one round of 64x64 tiles, short K: deep ring, never split
        assert plan(81, 768, 768) == (6, 1) and plan(81, 768, 4096)[0] != 6              # ... only up to K = 3072
        assert plan(72, 4096, 25664)[0] == 1                                             # conv-stem head: 16 K splits fill the chip
        # batch-1 step: 148 LLaMA rows / 257 ViT rows as one / two 160-row tiles, the weight streamed once, one round of workgroups
        assert plan(148, 22016, 4096) == (5, 1) and plan(148, 12288, 4160) == (5, 2) and plan(148, 4096, 22016) == (4, 8)
        assert plan(257, 6144, 1408) == (5, 2) and plan(129, 4096, 4096)[0] == 4 and plan(321, 4096, 4096)[0] in (1, 2)
        lib.mh_set_workspace(None, 0)
        assert plan(1184, 4096, 22016) == (1, 1)                                         # no workspace: nothing may split
        kernel, splits = ctypes.c_int(), ctypes.c_int()
        assert lib.mh_gemm_plan(0, 8, 64, 0, ctypes.addressof(kernel), ctypes.addressof(splits)) != 0   # bad dims rejected
    finally:
        lib.mh_set_workspace(None, 0)


def test_train_loop_looks_one_batch_ahead():
    """runner.train_loop hands train_step the CURRENT batch and, as `next_samples`, the batch of the following step (whose
    frozen ViT forward the model issues on a side stream); the iterator is advanced exactly once per step, in order, the
    lr is stepped before each step (base_task.py:229) and the delayed update is flushed at the end."""
    from myriad_amd.runner import LinearWarmupCosineLRScheduler, train_loop

    class Recorder:
        def __init__(self):
            self.calls, self.flushed = [], 0

        def train_step(self, samples, lr, weight_decay, dp=None, world=1, next_samples=None):
            self.calls.append((samples["id"], None if next_samples is None else next_samples["id"], lr, weight_decay))
            return torch.tensor(float(samples["id"]))

        def finish_update(self):
            self.flushed += 1

    drawn = []

    def data_iter():
        drawn.append(len(drawn))
        return {"id": drawn[-1]}

    sched = LinearWarmupCosineLRScheduler(None, max_epoch=10, iters_per_epoch=1600, min_lr=0.0, init_lr=1e-4, warmup_steps=0,
                                          warmup_start_lr=1e-6)
    m = Recorder()
    logged = []
    losses = train_loop(m, data_iter, 4, sched, weight_decay=0.05, log=lambda i, l, lr: logged.append((i, l)))
    assert drawn == [0, 1, 2, 3]                                          # one draw per step, no extra batch consumed
    assert [(c[0], c[1]) for c in m.calls] == [(0, 1), (1, 2), (2, 3), (3, None)]
    assert [float(x) for x in losses] == [0.0, 1.0, 2.0, 3.0] and logged == [(0, 0.0), (1, 1.0), (2, 2.0), (3, 3.0)]
    assert all(c[3] == 0.05 for c in m.calls) and m.calls[0][2] == pytest.approx(1e-4)
    assert m.flushed == 1
    # without lookahead every step gets next_samples=None and the same batches
    drawn.clear()
    m2 = Recorder()
    train_loop(m2, data_iter, 3, sched, lookahead=False)
    assert drawn == [0, 1, 2] and [(c[0], c[1]) for c in m2.calls] == [(0, None), (1, None), (2, None)]
    assert train_loop(Recorder(), data_iter, 0, sched) == []


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")
@pytest.mark.parametrize("rel", ["train_configs/loraadapter_simple_myriad_finetune.yaml", "train_configs/minigpt4_stage2_finetune.yaml",
                                 "eval_configs/myriad.yaml"])
def test_config_reads_the_reference_yaml_files_unchanged(rel):
    """`Config` (the OmegaConf-free counterpart of minigpt4/common/config.py:16-51) on the reference's own YAML files: model
    defaults <- file <- --options, scientific-notation floats as floats, every section reachable by attribute and key."""
    import argparse
    from myriad_amd.config import Config
    path = os.path.join(REF, rel)
    cfg = Config(argparse.Namespace(cfg_path=path, options=["run.max_epoch=3", "model.max_txt_len=77"]))
    m, r = cfg.model_cfg, cfg.run_cfg
    assert m.arch in ("myriad", "mini_gpt4") and m["arch"] == m.arch
    assert m.max_txt_len == 77                                       # --options wins over the file
    assert m.get("llama_model") and m.get("image_size") == 224       # keys that only the model's default YAML carries
    assert m.get("freeze_vit", True) is True
    if "train_configs" in rel:
        assert r.max_epoch == 3 and isinstance(r.init_lr, float) and 0 < r.init_lr < 1e-3      # written as 1e-4 / 3e-5 in the files
        assert isinstance(r.warmup_lr, float) and r.weight_decay == 0.05 and r.lr_sched == "linear_warmup_cosine_lr"
        assert r.seed == 42 and r.distributed is True and r.resume_ckpt_path is None
    assert len(cfg.datasets_cfg) >= 1
    d = cfg.to_dict()
    assert set(d) >= {"model", "run", "datasets"}


def test_accum_grad_iters_follows_the_reference_schedule_across_epochs():
    """base_task.py:262-271: the optimiser steps when (i + 1) % accum_grad_iters == 0 with i the index INSIDE the epoch, and
    .grad is only zeroed by a step -- so with iters_per_epoch = 5, accum = 2 the fifth batch of an epoch is folded into the
    next epoch's first update (three micro-batches).  The same walk with train_step's rule gives the same windows."""
    from myriad_amd.myriad import accum_update_due
    iters, accum, epochs = 5, 2, 3
    ref_windows, cur = [], []
    for ep in range(epochs):                      # the reference's loop, literally
        for i in range(iters):
            cur.append((ep, i))                   # loss.backward(): accumulates into .grad
            if (i + 1) % accum == 0:
                ref_windows.append(cur)           # optimizer.step(); optimizer.zero_grad()
                cur = []
    got, cur, count = [], [], 0
    for ep in range(epochs):
        for i in range(iters):
            cur.append((ep, i))
            count += 1
            if accum_update_due(count, accum, accum_index=i):
                got.append(cur)
                cur, count = [], 0
    assert got == ref_windows and [len(w) for w in got] == [2, 2, 3, 2, 3, 2]
    # without the epoch index: every accum-th call
    assert [accum_update_due(c, 2) for c in (1, 2, 3)] == [False, True, True]
