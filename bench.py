#!/usr/bin/env python3
"""Headline benchmark: images/s of one Myriad fine-tune step (EVA-ViT-g -> Q-Former + vision-expert adapters ->
Vicuna-7B, fwd + dgrad/wgrad bwd + gradient all-reduce + AdamW) on N MI355X GPUs, data parallel (weak scaling,
fixed per-GPU batch).  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Synthetic data and seeded random weights of the real architecture: no dataset / checkpoint exists in
the environment.  Nothing here reads /root/reference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory: HIP's default places the kernarg segment in host memory, and every kernel begins by fetching
# it over the host link -- 0.9 ms of a 42 ms step, 1.5 ms of the 20 ms batch-1 step (profiles/r04_gemm_x4.md).  Must be set before
# the HIP runtime initialises (first device call); a value the user set is respected.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)


def flops_per_sample(arch: str, stage: int, cfg: dict, n_before=4, n_after=28, n_tgt=16) -> dict:
    """Algorithmic FLOPs per sample per training step, SURVEY 8(d) formulas (1 MAC = 2 FLOP)."""
    Dv, Hv, dv = cfg["vit_dim"], cfg["vit_hidden"], cfg["vit_depth"]
    N = (cfg["image_size"] // cfg["patch"]) ** 2 + 1
    f_vit = dv * (2 * N * (4 * Dv * Dv + 2 * Dv * Hv) + 4 * N * N * Dv) + 2 * (N - 1) * 3 * cfg["patch"] ** 2 * Dv
    nq = cfg["num_query_token"] + (49 if (arch == "myriad" and stage in (1, 2)) else 0)
    Q, QI, ql = cfg["qf_dim"], cfg["qf_inter"], cfg["qf_layers"]
    f_qf = ql * (2 * nq * 4 * Q * Q + 4 * nq * nq * Q + 4 * nq * Q * QI) + (ql // 2) * (
        4 * nq * Q * Q + 4 * N * Dv * Q + 4 * nq * N * Q)
    L, LI, ll, V = cfg["llm_dim"], cfg["llm_inter"], cfg["llm_layers"], cfg["vocab"]
    n_img = nq + (18 if (arch == "myriad" and stage in (0, 1)) else 0)
    S = 1 + n_before + n_img + n_after + n_tgt
    f_llm = 2 * S * ll * (4 * L * L + 3 * L * LI) + 4 * S * S * L * ll + 2 * S * L * V
    f_proj = 2 * nq * Q * L
    f_ve = 0.0
    if arch == "myriad":
        stem = sum(2 * 9 * ci * co * hw * hw for (ci, co), hw in zip([(1, 4), (4, 16), (16, 64), (64, 256), (256, 1024)],
                                                                       [224, 112, 56, 28, 14]))
        if stage in (1, 2):
            f_ve += stem + 2 * 49 * 1024 * Q
        if stage in (0, 1):
            f_ve += stem + 2 * 9 * 25 * 1024 * L
    total = f_vit + 2 * (f_qf + f_proj + f_llm) + 3 * f_ve
    return dict(total=total, S=S, n_img=n_img, vit=f_vit, qformer=f_qf, llm=f_llm, ve=f_ve)


def make_samples(B: int, vocab: int, seed: int, device):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    maps = torch.rand(B, 1, 224, 224, generator=g)
    before = torch.randint(3, vocab, (1, 4), generator=g).expand(B, -1).contiguous()
    after = torch.randint(3, vocab, (1, 28), generator=g).expand(B, -1).contiguous()
    tgt = torch.randint(3, vocab, (B, 16), generator=g)
    return dict(image=image.to(device), anomaly_maps=maps.to(device), oneshot_anomaly_maps=maps.to(device),
                before_ids=before, after_ids=after, target_ids=tgt, target_mask=torch.ones(B, 16, dtype=torch.long))


KERNEL_NAMES = {1: "gemm_nt_kernel<128x128>", 2: "gemm_x8_kernel", 3: "gemm_nt_kernel<128x64>", 4: "gemm_nt_kernel<160x128>",
                5: "gemm_nt_kernel<160x96>", 6: "gemm_nt_kernel<64x64>"}
GEMM_OUT_F32 = 1


class LaunchProfile:
    """The library's launch profiler (include/myriad_hip.h: mh_prof_start / mh_prof_stop): two HIP events on the launch
    stream around EVERY launch of the GEMM kernels -- the kernel alone, also when it is the partial-product launch of a
    split-K op (the launch that sums the slabs is outside the pair).  This is what `roofline.achieved` is computed from and
    what the rocprofv3 kernel trace under profiles/ must agree with (tools/roofline_from_profiles.py)."""

    def __init__(self, capacity: int = 8192):
        from myriad_amd import _lib
        self.lib, self.check, self.cap = _lib.load(), _lib.check, capacity
        self.records = []

    def __enter__(self):
        self.check(self.lib.mh_prof_start(self.cap, torch.cuda.current_stream().cuda_stream), "mh_prof_start")
        return self

    def __exit__(self, *exc):
        import ctypes
        meta = (ctypes.c_int * (6 * self.cap))()
        ms = (ctypes.c_float * self.cap)()
        n = self.lib.mh_prof_stop(ctypes.cast(meta, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p), self.cap,
                                  torch.cuda.current_stream().cuda_stream)
        if n < 0:
            raise RuntimeError(f"mh_prof_stop failed: {n}")
        # an event pair around NOTHING still reads the queue's marker-to-marker time (calibrated by mh_prof_start: median of 33
        # empty pairs); every bracketed launch carries it on top of the kernel's own duration, so it is taken off each record
        self.overhead_ms = float(self.lib.mh_prof_overhead_ms())
        self.records = [(tuple(meta[6 * i:6 * i + 6]), max(float(ms[i]) - self.overhead_ms, 1e-4)) for i in range(n)]

    def summary(self):
        """per kernel and per (kernel, M, N, K, splits): launches, total ms, TFLOP/s, algorithmic bytes per launch
        (both bf16 operands once + the output once; a split launch writes `splits` slabs)."""
        per, shapes = {}, {}
        for (kid, M, N, K, sp, flags), ms in self.records:
            fl = 2.0 * M * N * K
            by = 2.0 * (M + N) * K + M * N * (4 if flags & GEMM_OUT_F32 else 2) * sp
            for d, key in ((per, KERNEL_NAMES.get(kid, str(kid))), (shapes, (kid, M, N, K, sp))):
                e = d.setdefault(key, [0, 0.0, 0.0, 0.0])
                e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
        self.per_kernel = {k: dict(launches=v[0], total_ms=round(v[1], 3), avg_us=round(1e3 * v[1] / v[0], 2),
                                   tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                                   algorithmic_mb_per_launch=round(v[3] / v[0] / 1e6, 1)) for k, v in per.items()}
        self.shapes = sorted(((k, v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12, v[3] / v[0]) for k, v in shapes.items()),
                             key=lambda r: -r[2])
        return self.per_kernel

    def subset(self, kid, pred):
        v = [0, 0.0, 0.0]
        for (k, M, N, K, sp, flags), ms in self.records:
            if k == kid and pred(sp, flags):
                v[0] += 1; v[1] += ms; v[2] += 2.0 * M * N * K
        if not v[0]:
            return None
        tf = v[2] / (v[1] * 1e-3) / 1e12
        return dict(launches=v[0], total_ms=round(v[1], 3), avg_us=round(1e3 * v[1] / v[0], 2), tflops=round(tf, 1),
                    frac=round(tf / PEAK_BF16_TFLOPS, 4))


def cpu_baseline(arch: str, stage: int, cfg: dict, budget_note: str = "") -> dict:
    """Reference-equivalent CPU path (the oracle, pinned to the reference's modules by tests/golden) timed on the
    host cores on a bounded sample: full width, reduced depth, B=1; per-component times are scaled linearly to the
    full depth (ViT 39 blocks fwd; Q-Former 12 layers, LLaMA 32 layers, VE nets fwd+bwd).
    Protocol: the thread count is chosen by a short scan on a shallower sample (at S = 148 rows the host BLAS
    is slower on 128 threads than on 16); at that count the sample runs 1 warm-up + 3 timed times and the MEDIAN is
    reported, with the three values beside it."""
    from oracle import myriad_ref as R
    from tests import golden_utils as gu
    torch.manual_seed(0)
    nth0 = torch.get_num_threads()
    kv, kq, kl = 8, 6, 8          # sampled depths; deeper samples only move the extrapolation by percents
    V = 2048
    sd = {}
    sd.update(gu.vit_weights(cfg["vit_dim"], kv, cfg["vit_heads"], cfg["vit_hidden"], cfg["patch"], 257, seed=1))
    sd.update(gu.qformer_weights(cfg["qf_dim"], kq, cfg["qf_inter"], cfg["vit_dim"], seed=2))
    sd.update(gu.llama_weights(cfg["llm_dim"], kl, cfg["llm_inter"], V, seed=3))
    sd.update(gu.adapter_weights(seed=4))
    sd.update(gu.glue_weights(seed=5))
    train = [k for k in sd if k.startswith(("expert_adaptor.", "VEInstructor.", "VETokenizer."))]
    for k in train:
        sd[k] = sd[k].clone().requires_grad_(True)
    batch = gu.synthetic_batch(1, V, seed=6)

    def run(depths):
        for k in train:
            sd[k].grad = None
        return _cpu_baseline_once(R, sd, train, batch, arch, stage, cfg, depths)

    scan = {}
    for nth in sorted({t for t in (8, 16, 32, 64, nth0) if t <= max(nth0, 8)}):
        torch.set_num_threads(nth)
        scan[nth] = run((2, 2, 2))[0]                     # shallow: ranks the thread counts, is not the reported number
    nth = min(scan, key=scan.get)
    torch.set_num_threads(nth)
    run((kv, kq, kl))                                     # warm-up
    timed = [run((kv, kq, kl)) for _ in range(3)]
    torch.set_num_threads(nth0)
    totals = sorted(t[0] for t in timed)
    total, measured = totals[1], sum(t[1] for t in timed) / 3
    return dict(value=1.0 / total, unit="images/s", cores=nth, kind="port",
                samples_images_per_s=[round(1.0 / t[0], 4) for t in timed], warmup=1, timed=3,
                thread_scan_s_per_step_shallow={str(k): round(v, 2) for k, v in scan.items()},
                sample=f"oracle (CPU restatement pinned to the reference modules) fp32 B=1 {arch} stage {stage}: "
                       f"ViT {kv}/{cfg['vit_depth']} blocks, Q-Former {kq}/{cfg['qf_layers']}, LLaMA {kl}/"
                       f"{cfg['llm_layers']} layers timed fwd+bwd and scaled linearly to full depth, AdamW on a 16 M-element slice "
                       f"scaled to the trainable count; 1 warm-up + 3 timed runs on {nth} threads (chosen by a shallow scan over "
                       f"{sorted(scan)}), median {total:.1f} s/step extrapolated from {measured:.1f} s measured per run; host has "
                       f"{os.cpu_count()} logical cores")


def _truncate_depth(sd, kv, kq, kl):
    """A view of the weight dict with only the first kv / kq / kl blocks (the oracle walks the layers it finds)."""
    import re
    lim = (("visual_encoder.blocks.", kv), ("Qformer.bert.encoder.layer.", kq), ("llama_model.model.layers.", kl))
    out = {}
    for k, v in sd.items():
        keep = True
        for pre, n in lim:
            if k.startswith(pre):
                keep = int(re.match(r"(\d+)", k[len(pre):]).group(1)) < n
        if keep:
            out[k] = v
    return out


def _cpu_baseline_once(R, sd, train, batch, arch, stage, cfg, depths):
    kv, kq, kl = depths
    image, maps, before, after, tgt, tmask = batch
    sd = _truncate_depth(sd, kv, kq, kl)
    t = {}

    def clock(name, fn):
        t0 = time.perf_counter()
        out = fn()
        t[name] = time.perf_counter() - t0
        return out

    with torch.no_grad():
        x = clock("vit", lambda: R.vit_forward(sd, image, cfg["vit_heads"]))
    x = R.ln_vision(sd, R.lora_adaptor(sd, x) if arch == "myriad" else x)
    q = sd["query_tokens"].expand(1, -1, -1)
    if arch == "myriad" and stage in (1, 2):
        q = torch.cat([q, clock("ve_ins_f", lambda: R.ve_instructor(sd, maps))], 1)
    qo = clock("qf_f", lambda: R.qformer_forward(sd, q, x, cfg["qf_heads"]))
    img = torch.nn.functional.linear(qo, sd["llama_proj.weight"], sd["llama_proj.bias"])
    if arch == "myriad" and stage in (0, 1):
        img = torch.cat([img, clock("ve_tok_f", lambda: R.ve_tokenizer(sd, maps))], 1)
    ew = sd["llama_model.model.embed_tokens.weight"]
    emb, attn, labels = R.assemble_inputs(ew, img, before, after, tgt, tmask, 1, 2)
    loss = clock("llm_f", lambda: R.llama_causal_lm(sd, emb, attn, labels, cfg["llm_heads"])[0])
    clock("bwd_all", lambda: loss.backward())
    # backward time is attributed proportionally to forward time of the differentiable parts
    f_parts = {k: v for k, v in t.items() if k not in ("vit", "bwd_all", "adamw_scaled")}
    fsum = sum(f_parts.values())
    scale = dict(vit=cfg["vit_depth"] / kv, qf_f=cfg["qf_layers"] / kq, llm_f=cfg["llm_layers"] / kl, ve_ins_f=1.0,
                 ve_tok_f=1.0)
    total = t["vit"] * scale["vit"]
    for k, v in f_parts.items():
        total += (v + t["bwd_all"] * v / fsum) * scale[k]
    n_train = sum(sd[k].numel() for k in train)
    # AdamW: the oracle's own update timed on a 16 M-element slice of the host buffers, scaled to the trainable count
    ns = 1 << 24
    pp, gg, mm, vv = (torch.randn(ns) for _ in range(4))
    vv.abs_()
    R.adamw_step(pp, gg, mm, vv, 1, 1e-4, 0.05)
    t0 = time.perf_counter()
    R.adamw_step(pp, gg, mm, vv, 2, 1e-4, 0.05)
    t_adam = (time.perf_counter() - t0) * n_train / ns
    t["adamw_scaled"] = t_adam
    total += t_adam
    return total, sum(t.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU images per step (reference: 4 + 4 augmented)")
    ap.add_argument("--arch", default="myriad", choices=["myriad", "mini_gpt4"])
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--lora", type=int, default=1, help="PEFT LoRA r=8 on q_proj/v_proj (BASELINE metric: Vicuna-7B+LoRA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--vit-depth", type=int, default=39)
    ap.add_argument("--qf-layers", type=int, default=12)
    ap.add_argument("--no-prefetch-vit", action="store_true",
                    help="run the frozen ViT forward inline at the start of its own step instead of one step ahead on a side stream")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 line (BASELINE configs[1]) reported as config1_b1")
    ap.add_argument("--no-minigpt4", action="store_true",
                    help="skip the MiniGPT-4 arch line (SURVEY 8d's second reported workload, S = 81) reported as config_minigpt4_b8")
    ap.add_argument("--side-steps", type=int, default=150,
                    help="timed steps of the two side workloads (config1_b1, config_minigpt4_b8): ~3 + 4.5 s of GPU work")
    ap.add_argument("--host-inputs", action="store_true",
                    help="inputs start in (pinned) host memory and every step uploads a fresh batch: the PCIe-inclusive rate "
                         "(DESIGN.md; never the headline `value`, whose inputs are resident in HBM)")
    a = ap.parse_args()

    from myriad_amd import _lib
    from myriad_amd.myriad import MiniGPT4HIP, MyriadHIP
    from myriad_amd.runner import DataParallel, LinearWarmupCosineLRScheduler, init_distributed, setup_seeds
    from myriad_amd.synthetic import SyntheticWeights, full_config

    _lib.load()   # fail loudly if the HIP library is missing
    rank, world, local = init_distributed()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists for the product)")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if os.environ.get("MYRIAD_MAIN_PRIO") == "1":
        # experiment switch: the step's main chain on a high-priority HIP stream (the look-ahead ViT / leaf / LoRA side streams
        # keep the default priority), so that the dispatcher prefers the chain's workgroups whenever both have some pending
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    setup_seeds(42, rank)                                  # reference train.py:63-72
    cfg = full_config(llm_layers=a.llm_layers, vit_depth=a.vit_depth, qf_layers=a.qf_layers)
    weights = SyntheticWeights(cfg, dev, seed=0, arch=a.arch)   # identical frozen weights on every rank
    cls = MyriadHIP if a.arch == "myriad" else MiniGPT4HIP
    t0 = time.time()
    model = cls(weights, dict(fixed_stage=a.stage, fixed_taskstage=0, vit_heads=cfg["vit_heads"], qf_heads=cfg["qf_heads"],
                              llm_heads=cfg["llm_heads"], use_lora=bool(a.lora) and a.arch == "myriad"), device=dev)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    dp = DataParallel(dev)
    # N > 1 is an RCCL run by contract: observe it instead of asserting it.  Every rank contributes a one; the sum is the
    # number of ranks the exchange really spans, and the backend must be nccl (= RCCL on ROCm) unless the single-device test
    # harness (MYRIAD_SINGLE_DEVICE=1, all ranks on one GPU over gloo) is selected explicitly.
    rccl_ranks = 0
    if world > 1:
        backend = torch.distributed.get_backend()
        if backend != "nccl" and os.environ.get("MYRIAD_SINGLE_DEVICE") != "1":
            raise SystemExit(f"bench.py --gpus {a.gpus}: process group backend is {backend!r}, not nccl (RCCL); "
                             "set MYRIAD_SINGLE_DEVICE=1 only for the one-GPU control-flow harness")
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        rccl_ranks = int(ones.item()) if backend == "nccl" else 0
    n_trainable = model.store.n_params()
    sched = LinearWarmupCosineLRScheduler(None, max_epoch=10, iters_per_epoch=1600, min_lr=0.0, init_lr=1e-4,
                                          warmup_steps=0, warmup_start_lr=1e-6)   # shipped recipe
    samples = make_samples(a.batch, cfg["vocab"], 42 + rank, dev)
    overlap = os.environ.get("MYRIAD_NO_OVERLAP") != "1"

    prefetch = not a.no_prefetch_vit

    host_batches = None
    if a.host_inputs:
        def host_batch(seed):
            d = make_samples(a.batch, cfg["vocab"], seed, "cpu")
            return {k: (v.pin_memory() if torch.is_tensor(v) and v.dtype.is_floating_point else v) for k, v in d.items()}
        host_batches = [host_batch(1000 + rank * 100 + j) for j in range(4)]

    def step(i, smp=samples):
        if host_batches is not None and smp is samples:
            cur, nxt = dict(host_batches[i % 4]), dict(host_batches[(i + 1) % 4])     # fresh dicts: nothing cached across steps
            if step.nxt is not None:
                cur = step.nxt                                        # the batch whose ViT forward was issued one step ago
            step.nxt = nxt
            return model.train_step(cur, sched.step(0, i), 0.05, dp=dp if world > 1 else None, world=world, overlap=overlap,
                                    next_samples=nxt if prefetch else None)
        # Input-pipeline lookahead of one batch (what a DataLoader with prefetch gives): every step issues the frozen ViT
        # forward of the NEXT step's batch on a side stream, where it fills the CUs this step leaves idle; one ViT forward
        # per step, as before.  N>1: the gradient all-reduce (RCCL, side stream) + AdamW of step i run while step i+1 waits
        # for them with that side-stream ViT forward in flight.
        return model.train_step(smp, sched.step(0, i), 0.05, dp=dp if world > 1 else None, world=world, overlap=overlap,
                                next_samples=smp if prefetch else None)

    step.nxt = None
    if prefetch:
        model.prepare_vit_graph(samples)      # the look-ahead's hipGraph is captured here, not inside a timed step (any --warmup)
    for i in range(a.warmup):
        step(i)
    model.finish_update()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    model.finish_update()                 # the last step's optimiser update is inside the timed region
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt.item())
    ms_per_step = 1e3 * dt / a.steps
    value = a.batch * world * a.steps / dt

    fl = flops_per_sample(a.arch, a.stage, cfg)
    step_tflops = a.batch * fl["total"] / (ms_per_step * 1e-3) / 1e12          # per GPU: every rank runs the same shard
    roof = None
    if not a.no_probe:
        # Two more steps on every rank (the gradient exchange is collective), outside the timed region, with the frozen ViT
        # forward INLINE and launched kernel by kernel: the look-ahead's launches are replayed from a hipGraph, where an event
        # pair cannot sit.  The first consumes the look-ahead the last timed step issued, the second is the profiled one:
        # one whole step's launches, ViT included -- the population of one step window of the rocprofv3 trace.
        def inline_step(i):
            out = model.train_step(samples, sched.step(0, i), 0.05, dp=dp if world > 1 else None, world=world, overlap=overlap,
                                   next_samples=None)
            model.finish_update()
            return out
        try:
            inline_step(a.warmup + a.steps)
            if rank == 0:
                with LaunchProfile() as lp:
                    inline_step(a.warmup + a.steps + 1)
                per = lp.summary()
                if os.environ.get("BENCH_SHAPES"):
                    with open(os.environ["BENCH_SHAPES"], "w") as f:
                        f.write("kernel,M,N,K,splits,launches,total_ms,TFLOPs\n")
                        for (kid, m, n, k, sp), cnt, ms, tf, _ in lp.shapes:
                            f.write(f"{KERNEL_NAMES.get(kid, kid)},{m},{n},{k},{sp},{cnt},{ms:.3f},{tf:.1f}\n")
                dname = max(per, key=lambda k: per[k]["total_ms"])          # dominant kernel = most time over ALL its launches
                dk = per[dname]
                did = [k for k, v in KERNEL_NAMES.items() if v == dname][0]
                # HBM-side bytes per launch: separate rocprofv3 --pmc passes (tools/pmc_traffic.sh), averaged over exactly the
                # launch grids profiled here (tiles x K splits)
                traffic = None
                pdir = os.path.join(ROOT, "profiles")
                # the NEWEST committed PMC summary (rNN[x]_gemm256_traffic.json); a lookup, not a measurement of this run:
                # the line says so (`from_committed_profile`)
                cands = sorted(f for f in (os.listdir(pdir) if os.path.isdir(pdir) else []) if f.endswith("_gemm256_traffic.json"))
                tpath = os.path.join(pdir, cands[-1]) if cands else None
                if tpath and did == 2:
                    tj = json.load(open(tpath))
                    rd = wr_ = 0.0
                    src_n = 0
                    for (kid, m_, n_, k_, sp), cnt, _, _, _ in lp.shapes:
                        tiles = ((m_ + 255) // 256) * ((n_ + 255) // 256)
                        ent = tj.get("by_grid", {}).get(f"{tiles}x{sp}") or (tj.get("by_workgroups", {}).get(str(tiles)) if sp == 1 else None)
                        if kid == 2 and ent:
                            rd += cnt * ent["read_bytes_per_launch"]
                            wr_ += cnt * ent["write_bytes_per_launch"]
                            src_n += cnt
                    if src_n:
                        traffic = dict(bytes_per_launch=round((rd + wr_) / src_n), read=round(rd / src_n), write=round(wr_ / src_n),
                                       algorithmic_bytes_per_launch=round(dk["algorithmic_mb_per_launch"] * 1e6),
                                       launches_matched=src_n, of_launches=dk["launches"], from_committed_profile=True,
                                       source=f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                              f"passes; read = 2 x FETCH_SIZE on gfx950)")
                # the same figure from the NEWEST committed rocprofv3 profile (profiles/rNN_step_breakdown.md total of the kernel in
                # the profiled step + rNN_step_gemm_shapes.csv FLOPs; tools/roofline_from_profiles.py): another box, another day --
                # beside the live number so the two are compared in the line itself
                committed = None
                try:
                    import csv as _csv, re as _re
                    rounds = sorted({f.split("_")[0] for f in os.listdir(pdir) if f.endswith("_step_gemm_shapes.csv")
                                     and os.path.exists(os.path.join(pdir, f.split("_")[0] + "_step_breakdown.md"))})
                    if rounds:
                        rnd = rounds[-1]
                        fl_c = sum(int(r_["launches"]) * 2.0 * int(r_["M"]) * int(r_["N"]) * int(r_["K"])
                                   for r_ in _csv.DictReader(open(os.path.join(pdir, f"{rnd}_step_gemm_shapes.csv")))
                                   if r_["kernel"] == dname)
                        for ln in open(os.path.join(pdir, f"{rnd}_step_breakdown.md")):
                            m_ = _re.match(r"\| `(?:void )?" + dname + r"[^|]*\| (\d+) \| ([\d.]+) \|", ln)
                            if m_:
                                tf_c = fl_c / (float(m_.group(2)) * 1e-3) / 1e12
                                committed = dict(round=rnd, launches=int(m_.group(1)), kernel_ms=float(m_.group(2)),
                                                 tflops=round(tf_c, 1), frac=round(tf_c / PEAK_BF16_TFLOPS, 4))
                                break
                except Exception:                                  # noqa: BLE001 -- evidence lookup only
                    committed = None
                all_ms = sum(v["total_ms"] for v in per.values())
                all_fl = sum(v["tflops"] * v["total_ms"] for v in per.values())
                roof = dict(bound="mfma", kernel=dname, achieved=dk["tflops"], peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                            frac=round(dk["tflops"] / PEAK_BF16_TFLOPS, 4), traffic=traffic, rocprof_frac_committed=committed,
                            population="every launch of the kernel in one step (LLaMA fwd + dgrad, ViT fwd; split-K launches "
                                       "included, each timed alone by a HIP-event pair on its stream, minus the calibrated "
                                       "duration of an empty pair)",
                            launches_per_step=dk["launches"], avg_launch_us=dk["avg_us"], kernel_ms_per_step=dk["total_ms"],
                            event_pair_overhead_us=round(1e3 * lp.overhead_ms, 2),
                            unsplit_launches=lp.subset(did, lambda sp, fl: sp == 1), split_launches=lp.subset(did, lambda sp, fl: sp > 1),
                            # launches whose read-out also does the SiLU gate of the MLP (forward: gate|up product, backward: the
                            # down projection's dgrad; csrc/gemm.hip mh_gemm_swiglu_*) carry elementwise work the FLOP count
                            # above does not credit: shown apart, counted in `achieved` / `frac` like every other launch
                            launches_with_fused_gate=lp.subset(did, lambda sp, fl: bool(fl & 48)),
                            launches_without_fused_gate=lp.subset(did, lambda sp, fl: not (fl & 48)),
                            per_kernel=per,
                            by_shape=[dict(kernel=KERNEL_NAMES.get(kid, kid), M=m, N=n, K=k, splits=sp, launches=cnt,
                                           total_ms=round(ms, 3), tflops=round(tf, 1))
                                      for (kid, m, n, k, sp), cnt, ms, tf, _ in lp.shapes[:10]],
                            all_gemm=dict(launches=sum(v["launches"] for v in per.values()), total_ms=round(all_ms, 3),
                                          tflops=round(all_fl / all_ms, 1), frac=round(all_fl / all_ms / PEAK_BF16_TFLOPS, 4)),
                            step_algorithmic_tflops=round(step_tflops, 1))
            else:
                inline_step(a.warmup + a.steps + 1)
        except Exception as e:                                    # noqa: BLE001
            roof = dict(bound="mfma", achieved=None, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=None, traffic=None,
                        error=repr(e))
    # The side measurements below must never cost the headline line: a failure is reported in place of the number.
    extra = {}
    if not a.no_b1 and rank == 0 and world == 1:
        try:
            s1 = make_samples(1, cfg["vocab"], 42, dev)
            if prefetch:
                model.prepare_vit_graph(s1)
            # >= 150 timed steps (~3 s): long enough for a 5-s utilisation sampler beside the run to see the leg
            n1 = a.side_steps
            for i in range(3):
                step(i, s1)
            model.finish_update()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n1):
                step(i, s1)
            model.finish_update()
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t0) / n1
            extra["config1_b1"] = dict(value=round(1.0 / d1, 2), unit="images/s", ms_per_step=round(1e3 * d1, 2), steps=n1,
                                       warmup=3, workload="BASELINE configs[1]: the same fine-tune step at batch 1 "
                                                          "(weight-streaming regime)")
        except Exception as e:                                    # noqa: BLE001
            extra["config1_b1"] = dict(value=None, error=repr(e))
    if not a.no_minigpt4 and rank == 0 and world == 1 and a.arch == "myriad":
        # SURVEY 8(d) / BASELINE.md section 2 name two reported workloads: Myriad stage 1 (the headline above, the worst case)
        # and the MiniGPT-4 baseline arch (mini_gpt4.py:153-257, minigpt4_stage2_finetune.yaml: no expert tokens, S = 81, only
        # llama_proj trainable) at the same per-GPU batch -- same protocol as the headline (look-ahead ViT, AdamW inside), 3 + 5 steps.
        try:
            model = None                                          # release the headline model (the closures above see None)
            torch.cuda.empty_cache()
            w2 = SyntheticWeights(cfg, dev, seed=0, arch="mini_gpt4")
            m2 = MiniGPT4HIP(w2, dict(fixed_stage=0, fixed_taskstage=0, vit_heads=cfg["vit_heads"], qf_heads=cfg["qf_heads"],
                                      llm_heads=cfg["llm_heads"], use_lora=False), device=dev)
            s2 = make_samples(a.batch, cfg["vocab"], 43, dev)
            if prefetch:
                m2.prepare_vit_graph(s2)

            def step2(i):
                return m2.train_step(s2, sched.step(0, i), 0.05, next_samples=s2 if prefetch else None)
            for i in range(3):
                step2(i)
            m2.finish_update()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n2 = a.side_steps
            for i in range(n2):
                l2 = step2(3 + i)
            m2.finish_update()
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t0) / n2
            fl2 = flops_per_sample("mini_gpt4", 0, cfg)
            extra["config_minigpt4_b8"] = dict(
                value=round(a.batch / d2, 2), unit="images/s", ms_per_step=round(1e3 * d2, 2), steps=n2, warmup=3,
                per_gpu_batch=a.batch, seq_len=fl2["S"], algorithmic_tflop_per_sample=round(fl2["total"] / 1e12, 3),
                step_frac_of_peak=round(a.batch * fl2["total"] / d2 / 1e12 / PEAK_BF16_TFLOPS, 4), loss=round(float(l2), 4),
                trainable_params=m2.store.n_params(),
                workload="MiniGPT-4 arch (SURVEY 8d: the config-2 baseline arch): EVA-ViT-g + Q-Former (32 queries) + llama_proj "
                         "(trainable) + Vicuna-7B, no expert tokens, no LoRA; fwd+bwd+AdamW")
            del m2, w2
        except Exception as e:                                    # noqa: BLE001
            extra["config_minigpt4_b8"] = dict(value=None, error=repr(e))
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(a.arch, a.stage, cfg)
            cpu["value"] = round(cpu["value"], 4)
        except Exception as e:                                    # noqa: BLE001
            cpu = dict(value=None, unit="images/s", cores=torch.get_num_threads(), kind="port", sample="failed: " + repr(e))
    if rank == 0:
        out = {
            "metric": "images/sec fine-tune step (Vicuna-7B + LoRA-style adapters, 224px)", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Myriad fine-tune step ({a.arch}, prompt stage {a.stage}): EVA-ViT-g/14 {cfg['vit_depth']}L "
                                   f"+ Q-Former {cfg['qf_layers']}L + adapters + Vicuna-7B {cfg['llm_layers']}L, fwd+bwd+AdamW; "
                                   f"224x224 image, 32-token prompt, 16-token target, S={fl['S']}",
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "seq_len": fl["S"],
                       "parallelism": f"dp{world}", "rccl_ranks": rccl_ranks,
                       "dp_exchange": (dp.mode + ("+bf16" if dp.grad_dtype == torch.bfloat16 else "")) if world > 1 else None,
                       "trainable_params": n_trainable,
                       # peft's LoraLayer formula on q_proj / v_proj, r = 8 (myriad.py:170-180).  peft itself is absent from the image
                       # and from the reference tree: the formula is pinned to hand-computed known-answer vectors, not to a peft run
                       "lora_qv_r8": bool(a.lora) and a.arch == "myriad",
                       "lora_parity": "peft formula pinned to hand-computed KATs (peft not installable: reference-unpinnable)",
                       "algorithmic_tflop_per_sample": round(fl["total"] / 1e12, 3)},
            "loss": round(float(loss), 4), "model_build_s": round(build_s, 1),
            # whole-step algorithmic FLOPs (SURVEY 8d F_step x per-GPU batch) / measured step time / dense bf16 peak, per GPU
            "step_frac_of_peak": round(step_tflops / PEAK_BF16_TFLOPS, 4),
            # F_step is the REFERENCE's arithmetic (full S x S attention, lm_head and the last layer over all S
            # rows); the HIP path runs causal attention, the lm_head on the 16 label rows per sample and -- round 5 -- the LAST
            # decoder layer's o_proj / MLP on those rows only (identical results: the other rows feed nothing and carry exactly-zero
            # gradients), ~4 % fewer executed FLOPs
            "flop_accounting": "reference-algorithmic (SURVEY 8d); executed FLOPs are ~4 % lower (causal attention; lm_head and the last layer's o_proj / MLP on label rows only)",
            "dp_collective": ("mh_ctx" if getattr(dp, "ctx", None) is not None else "torch.distributed") if world > 1 else None,
        }
        if roof is not None:
            out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out.update(extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
