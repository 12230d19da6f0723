"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm)
over xGMI.  The hot path shards over the batch axis (InstanceNorm, RoIAlign and every loss are
per-sample: SURVEY.md 8(e)); the only exchange is the gradient of each optimizer's arena,
which is ONE flat fp32 buffer per network (G 550 MB, D 11 MB) -> one all-reduce each, no
per-tensor launches.  The reference has no multi-GPU code at all; this is new design."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # SWAPNET_DIST_BACKEND=gloo lets the N>1 code path be smoke-tested on a single-GPU box
        backend = backend or os.environ.get("SWAPNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class GradExchange:
    """Averages a flat gradient arena across ranks.  `begin()` launches the collective on
    RCCL's own stream (it depends on the kernels already enqueued on the compute stream);
    `finish()` makes the compute stream wait for it -- work enqueued between the two calls
    (the other network's forward/backward) overlaps with the transfer."""

    def __init__(self, world=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self._pending = []

    def begin(self, flat):
        if self.world <= 1:
            return
        self._pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat))

    def finish(self):
        # the mean needs no extra pass: every loss gradient is pre-scaled by 1/world
        # (swn_hyper.grad_scale), so SUM over ranks is already the average
        for work, _ in self._pending:
            work.wait()
        self._pending = []

    def allreduce_mean(self, flat):
        self.begin(flat)
        self.finish()


def broadcast_arena(flat, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
