"""Training-loop counterpart of the reference's runner/task pair for the hot path:
`LinearWarmupCosineLRScheduler` (reference common/optims.py:57-125), the step loop of
`BaseTask._train_inner_loop` (tasks/base_task.py:202-294: lr step -> forward -> backward -> optimizer step),
seed rule `seed + rank` (train.py:63-72) and data-parallel gradient averaging replacing
`DDP(find_unused_parameters=True)` (runners/runner_base.py:94-98).

MI355X data parallelism: one process per GPU; the trainable gradients live in ONE flat fp32 buffer, so the
exchange is a single RCCL all-reduce(sum) over xGMI (or a reduce-scatter / sharded-AdamW / all-gather pair, DataParallel
mode 'rs_ag') issued on a side HIP stream as soon as the backward has produced the buffer; the fused AdamW consumes
sum/world.  A module unused on this rank contributes zeros and a zero use flag; the flags ride the same exchange, and a
module no rank used is skipped by the gated AdamW exactly as torch's AdamW skips a parameter whose grad is None.
"""
from __future__ import annotations

import json
import math
import os
import random
from typing import Callable, Optional

import torch

from .registry import registry


@registry.register_lr_scheduler("linear_warmup_cosine_lr")
class LinearWarmupCosineLRScheduler:
    """Same arithmetic as reference optims.py:79-125; `step()` returns the lr (and sets it on an optional
    torch optimizer's param groups, honouring a per-group 'init_lr' like the reference)."""

    def __init__(self, optimizer=None, max_epoch=10, iters_per_epoch=1600, min_lr=0.0, init_lr=1e-4, warmup_steps=0,
                 warmup_start_lr=-1, **kwargs):
        self.optimizer = optimizer
        self.max_epoch, self.iters_per_epoch = max_epoch, iters_per_epoch
        self.min_lr, self.init_lr = min_lr, init_lr
        self.warmup_steps = warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr

    def lr_at(self, cur_epoch: int, cur_step: int) -> float:
        total = cur_epoch * self.iters_per_epoch + cur_step
        if total < self.warmup_steps:
            return min(self.init_lr, self.warmup_start_lr + (self.init_lr - self.warmup_start_lr) * cur_step /
                       max(self.warmup_steps, 1))
        max_step = self.max_epoch * self.iters_per_epoch
        return (self.init_lr - self.min_lr) * 0.5 * (1.0 + math.cos(math.pi * total / max_step)) + self.min_lr

    def step(self, cur_epoch: int, cur_step: int) -> float:
        lr = self.lr_at(cur_epoch, cur_step)
        if self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        return lr


def setup_seeds(seed: int, rank: int = 0) -> None:
    """reference train.py:63-72: every RNG seeded with seed + rank."""
    s = seed + rank
    random.seed(s)
    try:
        import numpy as np
        np.random.seed(s)
    except Exception:
        pass
    torch.manual_seed(s)


class CtxCollective:
    """The gradient exchange through the C ABI's context object (include/myriad_hip.h: mh_ctx_*, mh_allreduce_start_dt,
    mh_reduce_scatter_start, mh_allgather_start, mh_allreduce_wait): RCCL communicator, side stream and events live in the
    library; a `*_start` orders the verb behind the current stream and returns, `wait` makes the current stream depend on
    everything started.  `dist` (any initialised process group) only carries the 128-byte communicator id."""

    DT = {torch.float32: 0, torch.bfloat16: 1}

    def __init__(self, device, rank: int, world: int, dist=None):
        """Collective: every rank of `dist` must call this together.  Raises on EVERY rank if any rank could not create its
        context or rank 0 could not produce a communicator id -- the ranks agree on that through the process group BEFORE anyone
        enters mh_ctx_comm_init (ncclCommInitRank is a rendezvous: a rank that skipped it would leave the others inside it),
        and every rank runs the same sequence of process-group collectives on every path."""
        import ctypes
        from . import _lib
        self.lib, self.check = _lib.load(), _lib.check
        self.dev = torch.device(device)
        self.world = world
        self.h = None
        err = None
        h = ctypes.c_void_p()
        try:
            with torch.cuda.device(self.dev):             # the context's side stream and events belong to THIS device
                self.check(self.lib.mh_ctx_create(ctypes.addressof(h)), "mh_ctx_create")
            self.h = h
        except Exception as e:                            # noqa: BLE001
            err = e
        msg = torch.zeros(129, dtype=torch.uint8)         # [communicator id (128) | rank 0's "id is valid" flag]
        if rank == 0 and err is None:
            raw = (ctypes.c_ubyte * 128)()
            try:
                self.check(self.lib.mh_ctx_comm_id(ctypes.addressof(raw)), "mh_ctx_comm_id")
                msg[:128] = torch.tensor(list(raw), dtype=torch.uint8)
                msg[128] = 1
            except Exception as e:                        # noqa: BLE001
                err = e
        if world > 1:
            on_dev = dist.get_backend() == "nccl"
            t = msg.to(self.dev) if on_dev else msg
            dist.broadcast(t, src=0)
            msg = t.cpu()
            ok = torch.tensor([1.0 if (err is None and int(msg[128]) == 1) else 0.0])
            ok = ok.to(self.dev) if on_dev else ok
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            all_ok = float(ok.item()) == 1.0
        else:
            all_ok = err is None and int(msg[128]) == 1
        if not all_ok:
            self.close()
            raise RuntimeError(f"mh_ctx bring-up failed on at least one rank (this rank: {err if err is not None else 'ok'})")
        raw = (ctypes.c_ubyte * 128)(*msg[:128].tolist())
        try:
            with torch.cuda.device(self.dev):
                self.check(self.lib.mh_ctx_comm_init(self.h, ctypes.addressof(raw), rank, world), "mh_ctx_comm_init")
        except Exception:
            self.close()
            raise

    @staticmethod
    def _s():
        return torch.cuda.current_stream().cuda_stream

    def start(self, flat: torch.Tensor) -> None:
        """all-reduce(sum) in place, wire type = the tensor's dtype (f32 or bf16)."""
        self.check(self.lib.mh_allreduce_start_dt(self.h, flat.data_ptr(), flat.numel(), self.DT[flat.dtype], self._s()),
                   "mh_allreduce_start_dt")

    def reduce_scatter(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        assert send.dtype == recv.dtype and send.numel() == recv.numel() * self.world
        self.check(self.lib.mh_reduce_scatter_start(self.h, send.data_ptr(), recv.data_ptr(), recv.numel(), self.DT[send.dtype],
                                                    self._s()), "mh_reduce_scatter_start")

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        assert send.dtype == recv.dtype and recv.numel() == send.numel() * self.world
        self.check(self.lib.mh_allgather_start(self.h, send.data_ptr(), recv.data_ptr(), send.numel(), self.DT[send.dtype],
                                               self._s()), "mh_allgather_start")

    def wait(self) -> None:
        self.check(self.lib.mh_allreduce_wait(self.h, self._s()), "mh_allreduce_wait")

    def close(self) -> None:
        if getattr(self, "h", None) is not None:
            self.lib.mh_ctx_destroy(self.h)
            self.h = None


class DataParallel:
    """Gradient exchange for one-process-per-GPU data parallelism over torch.distributed (backend 'nccl' is RCCL on ROCm;
    'gloo' in the CPU tests).  The unit of exchange is the model's flat buffer `store.flat_g_comm` = [gradients | per-module
    use flags]; everything is issued on a side HIP stream (it overlaps whatever the main stream does next, e.g. the next
    step's frozen ViT forward) and `wait()` makes the main stream depend on it.

    mode (MYRIAD_DP_MODE):
      'allreduce'  one all-reduce(sum) of the whole buffer; every rank then runs the full AdamW.
      'rs_ag'      two phases sized for xGMI's point-to-point links (a ring all-reduce of 443 MB is per-link bound,
                   ~5 ms; reduce-scatter + all-gather drive all 7 links at once): reduce-scatter the gradients, run AdamW on
                   this rank's 1/world shard only (1/world of the optimiser's HBM traffic), all-gather the updated parameters.
                   The tiny flag tail is all-reduced.  Adam moments exist only for the own shard; `gather_state` collects
                   them for a checkpoint.
    grad_dtype (MYRIAD_DP_GRAD_DTYPE=bf16): exchange the gradients in bf16 (half the bytes on the wire; the sum is rounded to
      bf16 once per hop -- a numerics change the reference does not make, off by default)."""

    def __init__(self, device=None, use_side_stream: bool = True, mode: Optional[str] = None, grad_dtype: Optional[str] = None):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device
        # default: one all-reduce at 2 ranks; reduce-scatter + sharded AdamW + all-gather from 4 ranks on (every rank then runs
        # 1/world of the optimiser's HBM traffic and the second half of the ring moves parameters instead of gradients)
        self.mode = mode or os.environ.get("MYRIAD_DP_MODE") or ("rs_ag" if self.world >= 4 else "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"MYRIAD_DP_MODE={self.mode}: expected allreduce or rs_ag")
        gd = grad_dtype or os.environ.get("MYRIAD_DP_GRAD_DTYPE", "f32")
        self.grad_dtype = torch.bfloat16 if gd in ("bf16", "bfloat16") else torch.float32
        self.side = None
        if use_side_stream and device is not None and torch.device(device).type == "cuda" and self.world > 1:
            self.side = torch.cuda.Stream(device=device)
        self._pending = None
        self._bufs = {}
        # The gradient buffer may be exchanged in SEGMENTS (set_segments): a segment whose gradient is complete early -- the map
        # tokenizer's 105 M-weight head, 91 % of the bytes, done right behind the LLaMA backward -- starts its collective while the
        # rest of the backward still runs (start_part); start() then exchanges what is left.  The segmentation is fixed for the life
        # of the object: in 'rs_ag' mode it also defines which slices of parameters and Adam moments this rank owns (shards()).
        self._segs = None                 # [(lo, hi)] tiling [0, total)
        self._seg_total = None
        self._started = set()             # segment indices already started in the exchange under way
        self._flags_started = False       # the use flags travelled with an early part (start() then leaves them alone)
        self._early_done = set()          # segments whose optimiser update (and all-gather) already followed their collective
        self._upd = None                  # ctx path: the stream an early update runs on (it waits for the context on the device)
        # on RCCL (backend nccl) the exchange goes through the library's own verbs -- mh_allreduce_start_dt /
        # mh_reduce_scatter_start / mh_allgather_start / mh_allreduce_wait on an mh_ctx that owns the communicator and the side
        # stream (include/myriad_hip.h) -- both modes, both wire types; the process group is then only the channel that hands
        # rank 0's communicator id to the other ranks.  torch.distributed carries the data on gloo (CPU tests, the one-GPU
        # harness) or with MYRIAD_DP_COLLECTIVE=torch.  The context is checked once against the process group (a sum of ones):
        # a build whose RCCL cannot be loaded or disagrees falls back to torch.distributed with a warning instead of failing the run.
        self.ctx = None
        self._gloo = dist.is_initialized() and dist.get_backend() == "gloo"
        want = os.environ.get("MYRIAD_DP_COLLECTIVE", "ctx" if (dist.is_initialized() and dist.get_backend() == "nccl") else "torch")
        explicit = os.environ.get("MYRIAD_DP_COLLECTIVE") == "ctx"      # asked for by name: also beside a gloo process group (tests)
        if want == "ctx" and self.world > 1 and device is not None and torch.device(device).type == "cuda" and (explicit or not self._gloo):
            try:
                ctx = CtxCollective(device, self.rank, self.world, dist)
                probe = torch.ones(8, dtype=torch.float32, device=device)
                ctx.start(probe)
                ctx.wait()
                torch.cuda.current_stream().synchronize()
                if not bool((probe == float(self.world)).all()):
                    raise RuntimeError(f"context all-reduce of ones gave {probe[0].item()} on {self.world} ranks")
                self.ctx = ctx
            except Exception as e:                          # noqa: BLE001 -- any failure here must not take the job down
                import warnings
                warnings.warn(f"mh_ctx gradient exchange unavailable ({e}); using torch.distributed")
                self.ctx = None
            # the choice has to be the same on every rank (a rank on torch.distributed and a rank on the context would wait for
            # each other forever): agree through the process group, fall back everywhere if anyone fell back
            ok = torch.tensor([1.0 if self.ctx is not None else 0.0], device=None if self._gloo else device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and self.ctx is not None:
                self.ctx.close()
                self.ctx = None

    def _persistent(self, key: str, n: int, dtype, device) -> torch.Tensor:
        """Exchange staging buffers are allocated once per (purpose, size, dtype) and reused every step."""
        k = (key, n, dtype, str(device))
        buf = self._bufs.get(k)
        if buf is None:
            buf = self._bufs[k] = torch.zeros(n, dtype=dtype, device=device)
        return buf

    # ---- segments and shard geometry
    def set_segments(self, total: int, cuts) -> None:
        """Cut the gradient buffer [0, total) at the interior points `cuts` (each a multiple of 4; multiples of 4 * world keep
        the reduce-scatter in place).  Call once, before the first exchange, identically on every rank."""
        if self._segs is not None:
            if self._seg_total != total:
                raise ValueError("DataParallel.set_segments: the segmentation is fixed once set (Adam moments are sharded by it)")
            return
        pts = [0] + sorted({int(c) for c in cuts if 0 < int(c) < total}) + [total]
        if any(c % 4 for c in pts):
            raise ValueError("DataParallel.set_segments: cut points must be multiples of 4 elements")
        self._segs = [(a, b) for a, b in zip(pts[:-1], pts[1:]) if b > a]
        self._seg_total = total

    def segments(self, total: int):
        return self._segs if (self._segs is not None and self._seg_total == total) else [(0, total)]

    def _piece(self, lo: int, hi: int):
        """This rank's piece of segment [lo, hi): `world` equal pieces of a multiple of 4 elements.  -> (lo_r, hi_r, per)"""
        n = hi - lo
        per = (n + self.world - 1) // self.world
        per = (per + 3) // 4 * 4
        a = min(lo + self.rank * per, hi)
        return a, min(a + per, hi), per

    def shards(self, total: int):
        """rs_ag: the slices of the flat buffers this rank owns (one per segment): its AdamW runs on exactly these."""
        return [self._piece(lo, hi)[:2] for lo, hi in self.segments(total)]

    def shard(self, total: int):
        """The single-segment form of shards(): (lo, hi, per) of this rank's piece of [0, total)."""
        return self._piece(0, total)

    def _run(self, fn):
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                fn()
        else:
            fn()

    def _cast(self, t: torch.Tensor, dtype):
        if t.dtype == dtype:
            return t
        if t.is_cuda:                                     # the library's cast kernels (no torch arithmetic on the data path)
            from . import ops
            return ops.to_bf16(t) if dtype == torch.bfloat16 else ops.to_f32(t)
        return t.to(dtype)

    def _all_reduce(self, t: torch.Tensor) -> None:
        if self.grad_dtype != t.dtype:
            w = self._cast(t, self.grad_dtype)
            self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM)
            t.copy_(self._cast(w, t.dtype))
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def _reduce_scatter(self, body: torch.Tensor, k: int = 0) -> None:
        """Sum over ranks of one segment; rank r ends up with the sum in its own piece of `body` (the rest of `body` is left
        unspecified)."""
        n = body.numel()
        lo, hi, per = self._piece(0, n)
        if n == per * self.world:                         # the store pads its buffers so that this holds at 2 / 4 / 8 ranks
            padded = body
        else:                                             # other rank counts: one persistent padded copy, no per-step allocation
            padded = self._persistent(f"rs_pad{k}", per * self.world, body.dtype, body.device)
            padded[:n].copy_(body)
        w = self._cast(padded, self.grad_dtype)
        if self._gloo:                                    # gloo has no reduce_scatter: all-reduce and keep the own piece
            self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM)
            mine = w[self.rank * per:(self.rank + 1) * per]
        else:
            mine = self._persistent(f"rs_mine{k}", per, w.dtype, w.device)
            self.dist.reduce_scatter_tensor(mine, w, op=self.dist.ReduceOp.SUM)
        body[lo:hi].copy_(self._cast(mine, body.dtype)[:hi - lo])

    def _exchange(self, flat_g_comm: torch.Tensor, n_grad: int, ks, with_flags: bool) -> None:
        """Queue the collectives of the segments `ks` (indices into segments(n_grad)) and, with_flags, of the use-flag tail
        behind the current stream."""
        segs = self.segments(n_grad)
        n_all = flat_g_comm.numel()
        if self.ctx is not None:
            self._ctx_exchange(flat_g_comm, n_grad, ks, with_flags)
            return

        def go():
            for k in ks:
                lo, hi = segs[k]
                if self.mode == "allreduce":
                    # the flag tail rides the all-reduce of the last segment (one call for [.. gradients | flags])
                    ext = n_all if (with_flags and hi == n_grad) else hi
                    self._all_reduce(flat_g_comm[lo:ext])
                else:
                    self._reduce_scatter(flat_g_comm[lo:hi], k)
            if with_flags and n_grad < n_all and (self.mode != "allreduce" or (len(segs) - 1) not in ks):
                self.dist.all_reduce(flat_g_comm[n_grad:], op=self.dist.ReduceOp.SUM)
            self._pending = True
        self._run(go)

    supports_early_update = True

    def start_part(self, flat_g_comm: torch.Tensor, n_grad: int, k: int, then=None) -> None:
        """Start the exchange of segment k (see set_segments) NOW, ordered behind the current stream: the caller guarantees that
        this segment's gradient is complete there.  Every rank calls it at the same point of its step, also a rank whose step
        did not touch the segment's module (its zeros are its contribution).  start() later exchanges the other segments.
        `then` (a callable): run right behind this segment's collective, on the exchange's own stream order -- the segment's
        optimiser update (sharded in 'rs_ag') and, in 'rs_ag', the all-gather of its updated parameters (gather_part), so that
        they too run under the rest of the backward.  The use flags then travel with this part (a gated update reads them)."""
        if self.world == 1:
            return
        if k in self._started:
            raise RuntimeError(f"DataParallel.start_part: segment {k} was already started in this exchange")
        self._started.add(k)
        self._exchange(flat_g_comm, n_grad, [k], with_flags=then is not None)
        if then is None:
            return
        self._flags_started = True
        self._early_done.add(k)
        if self.ctx is not None:
            # the context's side stream belongs to the library: the update runs on a torch stream that waits for it ON THE DEVICE
            if self._upd is None:
                self._upd = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._upd):
                self.ctx.wait()                               # _upd waits for the collective(s) started so far
                for fn in (self._pending or []):              # bf16 wire: cast back; rs_ag: the own piece into the buffer
                    fn()
                self._pending = []
                then()                                        # AdamW (+ the all-gather, queued through the context behind it)
        else:
            self._run(then)                                   # in order behind the collective on the exchange stream

    def part_shards(self, total: int, k: int):
        """The slices of segment k this rank's optimiser owns: its piece in 'rs_ag', the whole segment otherwise."""
        lo, hi = self.segments(total)[k]
        return [self._piece(lo, hi)[:2]] if self.mode == "rs_ag" else [(lo, hi)]

    def gather_part(self, flat_p: torch.Tensor, k: int) -> None:
        """rs_ag: all-gather the updated parameters of segment k, ordered behind the current stream, without waiting (wait()
        covers it).  No-op in 'allreduce' mode."""
        if self.world == 1 or self.mode != "rs_ag":
            return
        self._gather_segment(flat_p, k, wait=False)

    def start(self, flat_g_comm: torch.Tensor, n_grad: Optional[int] = None) -> None:
        """Launch the exchange of [gradients (n_grad elements) | flags] -- every segment start_part() has not started yet, and the
        flags.  n_grad=None: the whole buffer is gradients."""
        if self.world == 1:
            return
        n_grad = flat_g_comm.numel() if n_grad is None else n_grad
        ks = [k for k in range(len(self.segments(n_grad))) if k not in self._started]
        flags = not self._flags_started
        self._started, self._flags_started = set(), False
        self._exchange(flat_g_comm, n_grad, ks, with_flags=flags)

    def _ctx_exchange(self, flat_g_comm: torch.Tensor, n_grad: int, ks, with_flags: bool) -> None:
        """The exchange through the library's verbs.  Casts (bf16 wire) run on the current stream in front of / behind the
        verbs, which queue on the context's side stream; what has to happen after the wait is kept in self._pending."""
        ctx, bf = self.ctx, self.grad_dtype == torch.bfloat16
        segs = self.segments(n_grad)
        n_all = flat_g_comm.numel()
        after = self._pending if isinstance(self._pending, list) else []
        from . import ops
        # the wire copy lives in a persistent buffer: the verb reads it on the context's side stream, which torch's caching
        # allocator knows nothing about -- a per-step tensor could be handed out again while RCCL still reads it
        wire = self._persistent("wire", n_all, torch.bfloat16, flat_g_comm.device) if bf else None
        for k in ks:
            lo, hi = segs[k]
            if self.mode == "allreduce":
                ext = n_all if (with_flags and hi == n_grad) else hi
                t = flat_g_comm[lo:ext]
                if bf:
                    w = ops.to_bf16(t, out=wire[lo:ext])
                    ctx.start(w)
                    after.append(lambda w=w, t=t: ops.to_f32(w, out=t))
                else:
                    ctx.start(t)
            else:
                body = flat_g_comm[lo:hi]
                n = hi - lo
                a, b, per = self._piece(0, n)
                if n == per * self.world:                         # the store pads its buffers so that this holds at 2 / 4 / 8 ranks
                    padded = body
                else:
                    padded = self._persistent(f"rs_pad{k}", per * self.world, body.dtype, body.device)
                    padded[:n].copy_(body)
                if bf:
                    w = wire[lo:hi] if padded is body else self._persistent(f"rs_wire{k}", padded.numel(), torch.bfloat16, padded.device)
                    ops.to_bf16(padded, out=w)
                else:
                    w = padded
                mine = self._persistent(f"rs_mine{k}", per, w.dtype, w.device)
                ctx.reduce_scatter(w, mine)
                after.append(lambda body=body, mine=mine, a=a, b=b: body[a:b].copy_(self._cast(mine, body.dtype)[:b - a]))
        if with_flags and n_grad < n_all and (self.mode != "allreduce" or (len(segs) - 1) not in ks):
            ctx.start(flat_g_comm[n_grad:])                       # the use flags: a few floats, always fp32
        self._pending = after

    def wait(self) -> None:
        if self._upd is not None:
            torch.cuda.current_stream().wait_stream(self._upd)      # an early update (ctx path) ran there
        if self._pending is not None and self.ctx is not None:
            self.ctx.wait()
            for fn in self._pending:
                fn()
            self._pending = None
            return
        if self._pending is not None:
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
            self._pending = None

    def take_early_done(self):
        """Segments whose update already followed their collective in the exchange just waited for (cleared by the call)."""
        done, self._early_done = self._early_done, set()
        return done

    def allreduce(self, flat_grad: torch.Tensor, n_grad: Optional[int] = None) -> None:
        self.start(flat_grad, n_grad)
        self.wait()

    def _gather_segment(self, flat_p: torch.Tensor, k: int, wait: bool = True) -> bool:
        """All-gather of one segment's parameters in place.  -> True when a context verb was queued and nobody waited for it."""
        s_lo, s_hi = self.segments(flat_p.numel())[k]
        seg = flat_p[s_lo:s_hi]
        n = s_hi - s_lo
        lo, hi, per = self._piece(0, n)
        if self.ctx is not None and n == per * self.world:
            mine = self._persistent(f"ag_mine{k}", per, flat_p.dtype, flat_p.device)
            mine.copy_(seg[lo:hi])
            self.ctx.all_gather(mine, seg)
            if wait:
                self.ctx.wait()
            else:
                self._pending = self._pending if isinstance(self._pending, list) else []
            return not wait
        if self._gloo or n != per * self.world:
            mine = torch.zeros(per, dtype=flat_p.dtype, device=flat_p.device)
            mine[:hi - lo].copy_(seg[lo:hi])
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine)
            seg.copy_(torch.cat(parts)[:n])
        else:
            mine = self._persistent(f"ag_mine{k}", per, flat_p.dtype, flat_p.device)
            mine.copy_(seg[lo:hi])                    # a copy: input and output may not alias
            self.dist.all_gather_into_tensor(seg, mine)
        return False

    def gather_params(self, flat_p: torch.Tensor, skip=()) -> None:
        """rs_ag: after the sharded AdamW every rank holds fresh parameters for its pieces only; all-gather them in place,
        segment by segment (`skip`: segments gather_part() already gathered)."""
        if self.world == 1 or self.mode != "rs_ag":
            return
        queued = False
        for k in range(len(self.segments(flat_p.numel()))):
            if k not in skip:
                queued |= self._gather_segment(flat_p, k, wait=False)
        if queued:
            self.ctx.wait()

    def gather_state(self, store) -> None:
        """rs_ag: collect the Adam moments of every shard (before a checkpoint is written).  A collective: every rank calls it."""
        self.gather_params(store.flat_m)
        self.gather_params(store.flat_v)
        store.moments_complete = True

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def close(self) -> None:
        """Release the context's communicator, stream and events (they leaked)."""
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa: BLE001 -- interpreter shutdown
            pass


def init_distributed(backend: Optional[str] = None):
    """env:// rendezvous from torchrun (reference common/dist_utils.py:57-90)."""
    import torch.distributed as dist
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 1, 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        # MYRIAD_DIST_BACKEND=gloo + MYRIAD_SINGLE_DEVICE=1: exercise the N>1 control flow with every rank on cuda:0
        # (one-GPU test boxes; NCCL/RCCL refuses two ranks per device).  Production: nccl (= RCCL on ROCm).
        backend = os.environ.get("MYRIAD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("MYRIAD_SINGLE_DEVICE") == "1":
        local = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    dist.barrier()
    return rank, world, local


def train_loop(model, data_iter: Callable[[], dict], n_steps: int, scheduler: LinearWarmupCosineLRScheduler,
               weight_decay: float = 0.05, dp: Optional[DataParallel] = None, epoch: int = 0,
               log: Optional[Callable[[int, float, float], None]] = None, lookahead: bool = True):
    """`BaseTask._train_inner_loop` for the HIP model: lr is stepped BEFORE the forward with (epoch, i)
    (base_task.py:229), then forward/backward/all-reduce/AdamW."""
    world = dp.world if dp is not None else 1
    losses = []
    samples = data_iter() if n_steps > 0 else None
    for i in range(n_steps):
        lr = scheduler.step(cur_epoch=epoch, cur_step=i)
        nxt = data_iter() if (i + 1 < n_steps and lookahead) else None     # one batch ahead: its frozen ViT forward is
        loss = model.train_step(samples, lr, weight_decay, dp=(dp if dp and world > 1 else None), world=world,
                                next_samples=nxt)                           # issued beside this step (side stream)
        losses.append(loss)
        if log is not None:
            log(i, float(loss), lr)      # device->host sync per step like metric_logger.update(loss.item())
        samples = nxt if nxt is not None else (data_iter() if i + 1 < n_steps else None)
    model.finish_update()                # flush the delayed (overlapped) optimiser update of the last step
    return losses


# ------------------------------------------------------------------------------------------------ the runner
class IterLoader:
    """`IterLoader` (datasets/datasets/dataloader_utils.py:145-181): endless iterator over a DataLoader that bumps the
    DistributedSampler's epoch at every wrap-around."""

    def __init__(self, loader, use_distributed: bool = False):
        self.loader, self.use_distributed = loader, use_distributed
        self._it, self.epoch = None, 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.loader)

    def __next__(self):
        if self._it is None:
            self._it = iter(self.loader)
        try:
            return next(self._it)
        except StopIteration:
            self.epoch += 1
            if hasattr(self.loader, "sampler") and hasattr(self.loader.sampler, "set_epoch") and self.use_distributed:
                self.loader.sampler.set_epoch(self.epoch)
            self._it = iter(self.loader)
            return next(self._it)


class SmoothedValue:
    """`SmoothedValue` (common/logger.py:19-74) reduced to what the train log prints: last value and global average."""

    def __init__(self):
        self.total, self.count, self.value = 0.0, 0, 0.0

    def update(self, v: float):
        self.value = float(v)
        self.total += float(v)
        self.count += 1

    @property
    def global_avg(self) -> float:
        return self.total / max(self.count, 1)


class RunnerBase:
    """`RunnerBase` (runners/runner_base.py:42-686) for the HIP model: device / distributed setup, loaders
    (DistributedSampler; the AnomalyDetection loader yields batch_size_train // 2 samples which the model doubles with
    their augmented copies, :546-549), the epoch loop with resume (:374-432), the step loop of
    `BaseTask._train_inner_loop` (base_task.py:202-294), `checkpoint_N.pth` after every epoch (:592-628) and the JSON-lines
    `log.txt` (:674-686).  The optimiser is the model's fused AdamW on the flat buffer; data parallelism is
    `DataParallel` (one RCCL all-reduce per step on a side stream) instead of a DDP wrapper."""

    def __init__(self, cfg, job_id: str, model, datasets: dict, rank: int = 0, world: int = 1, device=None):
        self.config, self.job_id, self._model, self.datasets = cfg, job_id, model, datasets
        self.rank, self.world = rank, world
        self.device = torch.device(device if device is not None else getattr(model, "device", "cuda"))
        run = cfg.run_cfg
        self.max_epoch = int(run.get("max_epoch", 1))
        self.start_epoch = 0
        self.log_freq = int(run.get("log_freq", 50))
        self.resume_ckpt_path = run.get("resume_ckpt_path", None)
        self.evaluate_only = bool(run.get("evaluate", False))
        self.accum_grad_iters = max(int(run.get("accum_grad_iters", 1)), 1)      # base_task.py:262-271: step every n-th iteration
        out_root = run.get("output_dir", "output")
        self.output_dir = os.path.join(out_root, job_id)
        if self.rank == 0:
            os.makedirs(self.output_dir, exist_ok=True)
        from .checkpoint import CheckpointManager
        self.ckpt = CheckpointManager(self.output_dir, max_checkpoints=int(run.get("max_checkpoints", 1)))
        self.dp = DataParallel(device=self.device) if world > 1 else None
        self._sched, self._loader = None, None

    # ---- properties mirroring the reference's lazily built members
    @property
    def model(self):
        return self._model

    @property
    def lr_scheduler(self):
        if self._sched is None:
            run = self.config.run_cfg
            cls = registry.get_lr_scheduler_class(run.get("lr_sched", "linear_warmup_cosine_lr"))
            if cls is None:
                raise KeyError(f"lr_sched '{run.get('lr_sched')}' is not registered")
            ipe = run.get("iters_per_epoch", None)
            if ipe is None:
                ipe = len(self.train_loader.loader) if hasattr(self.train_loader, "loader") else 10000
            self._sched = cls(optimizer=None, max_epoch=self.max_epoch, iters_per_epoch=int(ipe), min_lr=float(run.get("min_lr", 0)),
                              init_lr=float(run.get("init_lr", 1e-4)), warmup_start_lr=float(run.get("warmup_lr", -1)),
                              warmup_steps=int(run.get("warmup_steps", 0)))
        return self._sched

    @property
    def train_loader(self):
        if self._loader is None:
            from torch.utils.data import DataLoader
            from torch.utils.data.distributed import DistributedSampler
            from .datasets import collate
            run = self.config.run_cfg
            names = list(self.datasets)
            if len(names) != 1:
                raise NotImplementedError("one training dataset per run (the shipped recipes use one)")
            ds = self.datasets[names[0]]
            bsz = int(run.get("batch_size_train", 4))
            if getattr(ds, "DatasetName", "") == "AnomalyDetection":
                bsz = max(1, bsz // 2)                       # runner_base.py:546-549
            sampler = DistributedSampler(ds, shuffle=True, num_replicas=self.world, rank=self.rank) if self.world > 1 else None
            loader = DataLoader(ds, batch_size=bsz, num_workers=int(run.get("num_workers", 0)), pin_memory=False, sampler=sampler,
                                shuffle=sampler is None, collate_fn=collate, drop_last=True)
            self._loader = IterLoader(loader, use_distributed=self.world > 1)
        return self._loader

    # ---- logging
    def log_stats(self, stats: dict, split_name: str):
        if self.rank != 0:
            return
        with open(os.path.join(self.output_dir, "log.txt"), "a") as f:
            f.write(json.dumps({f"{split_name}_{k}": v for k, v in stats.items()}) + "\n")

    def log_config(self):
        if self.rank != 0:
            return
        with open(os.path.join(self.output_dir, "log.txt"), "a") as f:
            f.write(json.dumps(self.config.to_dict(), indent=4) + "\n")

    # ---- training
    def train_epoch(self, epoch: int) -> dict:
        run = self.config.run_cfg
        model, sched, loader = self.model, self.lr_scheduler, self.train_loader
        model.train()
        iters = int(run.get("iters_per_epoch", len(loader)))
        wd = float(run.get("weight_decay", 0.05))
        meters = {"lr": SmoothedValue(), "loss": SmoothedValue()}
        pending = []
        samples = next(loader) if iters > 0 else None
        for i in range(iters):
            samples.update({"epoch": epoch, "num_iters_per_epoch": iters, "iters": i})      # base_task.py:217-223
            lr = sched.step(cur_epoch=epoch, cur_step=i)                                    # stepped BEFORE the forward (:229)
            nxt = next(loader) if i + 1 < iters else None                                   # one batch of lookahead: its frozen
            loss = model.train_step(samples, lr, wd, dp=self.dp, next_samples=nxt,          # ViT forward runs on a side stream
                                    accum_grad_iters=self.accum_grad_iters, accum_index=i)
            pending.append(loss)                          # the reference reads loss.item() every step (:276), which stalls the
            meters["lr"].update(lr)                       # launch thread behind the GPU; here the values are fetched at log points
            if i % self.log_freq == 0 or i == iters - 1:
                for t in pending:
                    meters["loss"].update(float(t))
                pending.clear()
                if self.rank == 0:
                    print(f"Train: data epoch: [{epoch}]  [{i}/{iters}]  lr: {lr:.6f}  loss: {meters['loss'].value:.4f}", flush=True)
            samples = nxt
        model.finish_update()
        if self.dp is not None:                                                             # logger.py:43-48 cross-rank average
            t = torch.tensor([meters["loss"].count, meters["loss"].total], dtype=torch.float64, device=self.device)
            self.dp.dist.all_reduce(t)
            meters["loss"].count, meters["loss"].total = int(t[0].item()), float(t[1].item())
        return {k: "{:.3f}".format(m.global_avg) for k, m in meters.items()}

    def train(self):
        self.log_config()
        if not self.evaluate_only and self.resume_ckpt_path:
            self.start_epoch = self.ckpt.load(self.model, self.resume_ckpt_path)
        cur_epoch = self.start_epoch
        for cur_epoch in range(self.start_epoch, self.max_epoch):
            if not self.evaluate_only:
                stats = self.train_epoch(cur_epoch)
                self.log_stats(stats, "train")
                if self.dp is not None and getattr(self.dp, "mode", "") == "rs_ag":
                    self.dp.gather_state(self.model.store)     # sharded AdamW: every rank owns 1/world of the moments (collective)
                if self.rank == 0:                             # no validation split in the shipped recipes: save every epoch
                    lr = self.lr_scheduler.lr_at(cur_epoch, int(self.config.run_cfg.get("iters_per_epoch", 1)) - 1)
                    self.ckpt.save(self.model, cur_epoch, lr, float(self.config.run_cfg.get("weight_decay", 0.05)),
                                   config=self.config.to_dict())
            if self.evaluate_only:
                break
            if self.dp is not None:
                self.dp.barrier()
        return cur_epoch
