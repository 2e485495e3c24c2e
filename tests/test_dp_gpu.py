"""Two data-parallel ranks of the HIP model on ONE GPU (the only N > 1 run a one-GPU box can make): SURVEY 8(e), reference
runners/runner_base.py:94-98 (DDP, find_unused_parameters=True) + train.py:63-72 (seed + rank).  Two processes on cuda:0
exchange the flat gradient buffer over gloo through runner.DataParallel with the overlap on (exchange on the side stream,
AdamW delayed under the next step's look-ahead ViT); the result must be what ONE process gets from summing the two ranks'
gradients and applying AdamW with 1/world -- bit for bit -- in both exchange modes."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wire_sum(g0, g1, wire, n_grad=None):
    """What the exchange hands back for two ranks' buffers: the plain fp32 sum, or with a bf16 wire each rank's buffer rounded
    to bf16, added in fp32 and rounded to bf16 once more (what RCCL's bf16 sum and the stand-in do); with n_grad only the
    first n_grad elements travel in bf16 (rs_ag: the use flags behind them are all-reduced in fp32)."""
    if wire != "bf16":
        return g0 + g1
    def rt(t):
        return t.to(torch.bfloat16).float()
    out = g0 + g1
    n = out.numel() if n_grad is None else n_grad
    out[:n] = rt(rt(g0[:n]) + rt(g1[:n]))
    return out


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    """The stand-in RCCL of tests/fake_rccl, built once per session with hipcc (host code only)."""
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-shared", "-fPIC", "-o", out, src])
    return out


def _reference(dev, wire="f32", mode="allreduce"):
    """One process: per step, run both ranks' forward + backward (their batch, their prompt stage, their dropout seed), add the
    two [gradients | use flags] buffers (as the wire would), one gated AdamW with grad_scale = 1/2."""
    from tests import dp_common as C
    model, cfg = C.build_model(dev)
    st = model.store
    losses = {0: [], 1: []}
    for i in range(C.N_STEPS):
        gs = []
        for r in (0, 1):
            model.lora.base_seed = C.BASE_SEED + r
            model.lora.step_seed = i                              # the training forward increments it: step i draws seed i + 1
            model.fixed_stage = C.STAGES[r][i]
            with torch.no_grad():
                losses[r].append(float(model._forward_impl(C.batch(r, i, cfg["vocab"], dev), True)))
                model.backward()
                if model.use_lora:
                    model.lora.join_wgrads()
            torch.cuda.synchronize()
            gs.append(st.flat_g_comm.clone())
        st.flat_g_comm.copy_(_wire_sum(gs[0], gs[1], wire, st.total if mode == "rs_ag" else None))
        st.adamw_step(C.LRS[i], 0.05, grad_scale=0.5)
    return C.snapshot(model), losses


CASES = [("allreduce", "torch", "f32"), ("rs_ag", "torch", "f32"),
         # the path an 8-GPU node runs by default: the mh_ctx verbs, both modes, both wire types
         ("allreduce", "ctx", "f32"), ("rs_ag", "ctx", "f32"), ("allreduce", "ctx", "bf16"), ("rs_ag", "ctx", "bf16"),
         # rank 0 cannot bring its context up: no rank may enter the communicator rendezvous, all fall back to torch.distributed
         ("allreduce", "ctx_bad0", "f32")]


@pytest.mark.parametrize("mode,collective,wire", CASES)
def test_two_ranks_on_one_gpu_equal_one_process_summing_their_gradients(mode, collective, wire, tmp_path, fake_rccl):
    from tests import dp_common as C
    port = str(29600 + (os.getpid() % 200) + 7 * CASES.index((mode, collective, wire)))
    outs = [str(tmp_path / f"rank{r}.pt") for r in (0, 1)]
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), "2", port, mode, outs[r],
                               collective, wire, fake_rccl],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    s0, s1 = (torch.load(o) for o in outs)
    ref, ref_losses = _reference(torch.device("cuda:0"), wire, mode)
    # the two ranks end with the same parameters and optimiser state
    for k in ("p", "m", "v"):
        assert torch.equal(s0[k], s1[k]), k
    assert s0["steps"] == s1["steps"]
    # ... which are the single-process result of averaging the two gradients
    assert s0["losses"] == ref_losses[0] and s1["losses"] == ref_losses[1]
    for k in ("p", "m", "v"):
        assert torch.equal(s0[k], ref[k]), (k, (s0[k] - ref[k]).abs().max().item())
    assert s0["steps"] == ref["steps"]
    # a module no rank used in a step (VEInstructor at step 1) kept its step count: one update fewer than the others
    assert s0["steps"]["VEInstructor"] == C.N_STEPS - 1 and s0["steps"]["VETokenizer"] == C.N_STEPS
    assert s0["steps"]["expert_adaptor"] == C.N_STEPS and s0["steps"]["lora"] == C.N_STEPS
    # and the parameters did move
    model0, _ = C.build_model(torch.device("cuda:0"))
    assert (model0.store.flat_p.cpu() - s0["p"]).abs().max().item() > 1e-5


def test_accum_grad_iters_sums_the_window_and_steps_once():
    """base_task.py:262-271: with accum_grad_iters = 2 the optimiser steps on every second iteration with the SUM of the two
    backward passes (no division) and that iteration's lr; a module used by only one call of the window is updated, one used
    by none keeps its weights, moments and step count.  train_step(..., accum_grad_iters=2) twice == one AdamW on g1 + g2."""
    from tests import dp_common as C
    dev = torch.device("cuda:0")
    stages = [2, 1, 0, 0]                                     # window 1: VETokenizer used by the 2nd call only; window 2: VEInstructor by none
    got, cfg = C.build_model(dev)
    got.lora.base_seed = 7
    for i in range(4):
        got.fixed_stage = stages[i]
        got.train_step(C.batch(0, i, cfg["vocab"], dev), C.LRS[i % 3], 0.05, accum_grad_iters=2)
    got.finish_update()
    ref, _ = C.build_model(dev)
    ref.lora.base_seed = 7
    st = ref.store
    for w in range(2):
        gsum = None
        for j in range(2):
            i = 2 * w + j
            ref.fixed_stage = stages[i]
            with torch.no_grad():
                ref._forward_impl(C.batch(0, i, cfg["vocab"], dev), True)
                ref.backward()
            torch.cuda.synchronize()
            g = st.flat_g_comm.clone()
            gsum = g if gsum is None else g + gsum
        st.flat_g_comm.copy_(gsum)
        st.adamw_step(C.LRS[(2 * w + 1) % 3], 0.05)
    a, b = C.snapshot(got), C.snapshot(ref)
    for k in ("p", "m", "v"):
        assert torch.equal(a[k], b[k]), k
    assert a["steps"] == b["steps"] and a["steps"]["VETokenizer"] == 2 and a["steps"]["VEInstructor"] == 1


def test_early_adamw_of_the_tokenizer_on_the_leaf_stream_is_the_same_update(monkeypatch):
    """Single-process train_step updates the map tokenizer (91 % of the trainable parameters) on the leaf side stream as soon as
    its backward is done, beside the Q-Former backward, instead of at the tail of the step (MYRIAD_EARLY_ADAMW=0: all modules at
    the tail).  Same kernel, same inputs: three steps with ragged prompt stages end with bit-identical parameters, moments and
    per-module step counts, and a step whose stage skips the tokenizer leaves it untouched either way."""
    from tests import dp_common as C
    dev = torch.device("cuda:0")
    stages = [1, 2, 0]                                        # step 1 (stage 2): the tokenizer is unused
    snaps = []
    for early in ("1", "0"):
        monkeypatch.setenv("MYRIAD_EARLY_ADAMW", early)
        m, cfg = C.build_model(dev)
        m.lora.base_seed = 11
        for i in range(3):
            m.fixed_stage = stages[i]
            nxt = C.batch(0, i + 1, cfg["vocab"], dev) if i < 2 else None
            m.train_step(C.batch(0, i, cfg["vocab"], dev), C.LRS[i], 0.05, next_samples=nxt)
        m.finish_update()
        snaps.append(C.snapshot(m))
    a, b = snaps
    for k in ("p", "m", "v"):
        assert torch.equal(a[k], b[k]), k
    assert a["steps"] == b["steps"] and a["steps"]["VETokenizer"] == 2 and a["steps"]["lora"] == 3
