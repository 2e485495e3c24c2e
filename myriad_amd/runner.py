"""Training-loop counterpart of the reference's runner/task pair for the hot path:
`LinearWarmupCosineLRScheduler` (reference common/optims.py:57-125), the step loop of
`BaseTask._train_inner_loop` (tasks/base_task.py:202-294: lr step -> forward -> backward -> optimizer step),
seed rule `seed + rank` (train.py:63-72) and data-parallel gradient averaging replacing
`DDP(find_unused_parameters=True)` (runners/runner_base.py:94-98).

MI355X data parallelism: one process per GPU; the trainable gradients live in ONE flat fp32 buffer, so the
exchange is a single RCCL all-reduce(sum) over xGMI issued on a side HIP stream as soon as the backward has
produced the buffer; the fused AdamW consumes sum/world.  Unused-this-step modules contribute zeros (the buffer
is zero-filled each backward), which is exactly DDP's find_unused_parameters semantics.
"""
from __future__ import annotations

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


class DataParallel:
    """Gradient averaging for one-process-per-GPU data parallelism over torch.distributed (backend 'nccl' is
    RCCL on ROCm; 'gloo' in the CPU tests).  `start(flat_grad)` launches the all-reduce on a side stream
    (overlapping whatever the main stream does next, e.g. the next step's frozen ViT forward); `wait()` makes
    the main stream depend on it."""

    def __init__(self, device=None, use_side_stream: bool = True):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = device
        self.side = None
        if use_side_stream and device is not None and torch.device(device).type == "cuda" and self.world > 1:
            self.side = torch.cuda.Stream(device=device)
        self._pending = None

    def start(self, flat_grad: torch.Tensor) -> None:
        if self.world == 1:
            return
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self._pending = self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, async_op=True)
        else:
            self._pending = self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, async_op=True)

    def wait(self) -> None:
        if self._pending is not None:
            self._pending.wait()
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
            self._pending = None

    def allreduce(self, flat_grad: torch.Tensor) -> None:
        self.start(flat_grad)
        self.wait()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


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
