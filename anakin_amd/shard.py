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


def max_over_ranks(seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
