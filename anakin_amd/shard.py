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


class BatchedLogitGather:
    """The same exchange with the collective amortised over `every` steps: each step's logits are copied (asynchronously, on
    the compute stream) into a slot of a device ring; once per `every` steps the whole ring is all-gathered in ONE
    asynchronous collective that overlaps the following steps (double-buffered rings). Every step's logits still reach every
    rank — `every` steps late at most — and the launching thread pays one small copy per step instead of one collective:
    a per-step Python-side collective costs more host time than a 0.24 ms forward pass leaves. With the gloo backend (CPU
    tests, the shared-GPU dry run) the ring is staged through pinned host memory, because gloo on device tensors
    synchronises the stream at every call."""

    def __init__(self, like, world, every=16):
        self.world, self.every = world, int(every)
        shape = (self.every,) + tuple(like.shape)
        self.gloo = dist.get_backend() == "gloo"
        self.ring = [torch.empty(shape, dtype=like.dtype, device=like.device) for _ in range(2)]
        full = (world * self.every,) + tuple(like.shape)
        if self.gloo:
            pin = like.device.type == "cuda"
            self.host = [torch.empty(shape, dtype=like.dtype, pin_memory=pin) for _ in range(2)]
            self.full = [torch.empty(full, dtype=like.dtype) for _ in range(2)]
            self.ev = [torch.cuda.Event() if pin else None for _ in range(2)]
        else:
            self.full = [torch.empty(full, dtype=like.dtype, device=like.device) for _ in range(2)]
        self.i = 0
        self.pending = None
        self.done = None
        self.gathers = 0

    def _launch(self, b, count):
        if self.gloo:
            if self.ev[b] is not None:
                self.host[b].copy_(self.ring[b], non_blocking=True)
                self.ev[b].record()
                self.ev[b].synchronize()      # once per `every` steps: the ring (not the pipeline) has to be on the host
            else:
                self.host[b].copy_(self.ring[b])
            h = dist.all_gather(list(self.full[b].chunk(self.world, 0)), self.host[b], async_op=True)
        else:
            h = dist.all_gather_into_tensor(self.full[b], self.ring[b], async_op=True)
        self.flush()
        self.pending = (h, b, count)
        self.gathers += 1

    def step(self, logits):
        b = (self.i // self.every) & 1
        self.ring[b][self.i % self.every].copy_(logits, non_blocking=True)
        self.i += 1
        if self.i % self.every == 0:
            self._launch(b, self.every)

    def flush(self):
        if self.pending is not None:
            h, b, count = self.pending
            h.wait()
            self.done = (b, count)
            self.pending = None

    def finish(self):
        """Gathers what is left in a partly filled ring and waits for everything (end of a timed region)."""
        rem = self.i % self.every
        if rem:
            self._launch((self.i // self.every) & 1, rem)
            self.i += self.every - rem
        self.flush()

    def latest(self):
        """[world, count, batch, classes] view of the newest complete gather (None before the first)."""
        if self.done is None:
            return None
        b, count = self.done
        return self.full[b].view((self.world, self.every) + tuple(self.full[b].shape[1:]))[:, :count]


def max_over_ranks(seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
