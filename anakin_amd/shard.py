"""Batch sharding across the GPUs of one node.

Every op on the path is per-image (BatchNorm is folded), so a batch splits into contiguous, independent
per-rank sub-batches with replicated weights and NO data-path collective; the only exchange is the
gather of the per-rank logits (RCCL all-gather over xGMI on GPUs, gloo on CPU in the tests)."""
import torch
import torch.distributed as dist


def shard_range(global_batch, world, rank):
    """Contiguous [start, start+count) of the images rank `rank` owns; remainders go to the first ranks."""
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_logits(local, world, out=None):
    """All-gather equal-sized per-rank logits [b, classes] -> [world*b, classes] (rank order)."""
    if world == 1:
        return local
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo":
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, local.contiguous())
    else:
        dist.all_gather_into_tensor(out, local.contiguous())
    return out


class AsyncLogitGather:
    """Per-step all-gather of the ranks' logits that overlaps with the next step: step i's collective is launched
    asynchronously from a private copy of the logits (double-buffered) and waited for one step later, so the step
    loop never stalls on the exchange; `flush()` waits for the last one. `latest()` is the newest complete result."""

    def __init__(self, like, world):
        self.world = world
        shape = tuple(like.shape)
        self.local = [torch.empty(shape, dtype=like.dtype, device=like.device) for _ in range(2)]
        self.full = [torch.empty((world * shape[0],) + shape[1:], dtype=like.dtype, device=like.device)
                     for _ in range(2)]
        self.i = 0
        self.pending = None
        self.done = None

    def step(self, logits):
        b = self.i & 1
        self.local[b].copy_(logits)
        if dist.get_backend() == "gloo":
            h = dist.all_gather(list(self.full[b].chunk(self.world, 0)), self.local[b], async_op=True)
        else:
            h = dist.all_gather_into_tensor(self.full[b], self.local[b], async_op=True)
        self.flush()
        self.pending = (h, b)
        self.i += 1

    def flush(self):
        if self.pending is not None:
            h, b = self.pending
            h.wait()
            self.done = b
            self.pending = None

    def latest(self):
        return None if self.done is None else self.full[self.done]


def max_over_ranks(seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
