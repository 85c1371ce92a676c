"""Data-parallel gradient averaging: one process per GPU, each rank runs forward/backward on its shard of the batch,
then ONE sum-all-reduce over the flat gradient buffer (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests),
divided by the world size -- what fleet.distributed_model does for the reference (ppvector/trainer.py:105-107,318-320).
BatchNorm statistics stay rank-local, as in the reference (plain BatchNorm, no SyncBN)."""
import torch
import torch.distributed as dist


def shard_batch(n_items, rank, world):
    """Contiguous split of a global batch, like paddle.io.DistributedBatchSampler (trainer.py:105-107)."""
    per = (n_items + world - 1) // world
    return range(min(rank * per, n_items), min((rank + 1) * per, n_items))


def allreduce_mean_(flat_grad, bucket_bytes=256 << 20):
    """In-place average of a flat gradient buffer over all ranks, in buckets (one collective per bucket_bytes)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat_grad
    world = dist.get_world_size()
    n = flat_grad.numel()
    step = max(1, bucket_bytes // flat_grad.element_size())
    works = [dist.all_reduce(flat_grad[i:min(i + step, n)], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, n, step)]
    for w in works:
        w.wait()
    flat_grad.div_(world)
    return flat_grad
