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


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _Done:
    """Handle of a collective that already completed (the staged gloo path)."""

    def wait(self):
        return True


def all_reduce_sum_(t, async_op=False):
    """In-place sum over the ranks.  "nccl" (= RCCL) reduces device tensors directly; "gloo" (the CPU tests, and the 2-rank tests
    that share one GPU) gets a device tensor staged through host memory.  Returns a handle with .wait() when async_op."""
    if world_size() == 1:
        return _Done() if async_op else None
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def all_reduce_max_(t, async_op=False):
    """In-place maximum over the ranks (the grid-barrier bail-out flag that travels with the gradients: train/step.py)."""
    if world_size() == 1:
        return _Done() if async_op else None
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=async_op)


def allreduce_mean_(flat_grad, bucket_bytes=256 << 20):
    """In-place average of a flat gradient buffer over all ranks, in buckets (one collective per bucket_bytes)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat_grad
    world = dist.get_world_size()
    n = flat_grad.numel()
    step = max(1, bucket_bytes // flat_grad.element_size())
    works = [all_reduce_sum_(flat_grad[i:min(i + step, n)], async_op=True) for i in range(0, n, step)]
    for w in works:
        w.wait()
    flat_grad.div_(world)
    return flat_grad


class OverlappedReducer:
    """Gradient all-reduce overlapped with backward (BASELINE config 5: "grad all-reduce overlapped with backward").

    The optimiser's flat gradient buffer is cut into contiguous buckets of ~bucket_bytes, built from the LAST parameter
    backwards: backward produces gradients in reverse parameter order, so the bucket holding the head and the last layers
    completes first and its all-reduce runs while autograd is still computing the early layers.  A post-accumulate hook on
    every parameter counts the bucket's pending gradients; a complete bucket is all-reduced asynchronously.  `finish()` waits
    for the handles; the flat buffer then holds the SUM over the ranks -- the caller passes grad_scale = 1 / world to
    optimizer.step, which applies it inside the update kernel (no division pass over the buffer).  Default 16 MB: ECAPA's 26.9 MB of gradients become two buckets, CAM++'s
    33 MB three (a single 64 MB bucket would start its one collective only after backward ended = no overlap), while each
    message is still large enough for the xGMI ring (per-link bound: 8 ranks x 7 links x ~153 GB/s)."""

    def __init__(self, optimizer, bucket_bytes=16 << 20):
        self.opt = optimizer
        self.flat = optimizer.grad
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.buckets, self.bucket_of, self.members = [], {}, []
        ends = []
        off = 0
        for p in optimizer.params:
            off += p.numel()
            ends.append(off)
        end, size, members = off, 0, []
        for p, e in zip(reversed(optimizer.params), reversed(ends)):
            members.append(p)
            size += p.numel()
            if size * 4 >= bucket_bytes:
                self.buckets.append([e - p.numel(), end, len(members), 0, None])      # [begin, end, n_params, n_ready, handle]
                self.members.append(list(members))
                for q in members:
                    self.bucket_of[q] = len(self.buckets) - 1
                end, size, members = e - p.numel(), 0, []
        if members:
            self.buckets.append([0, end, len(members), 0, None])
            self.members.append(list(members))
            for q in members:
                self.bucket_of[q] = len(self.buckets) - 1
        self.hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in optimizer.params] if self.world > 1 else []

    def _ready(self, p):
        bi = self.bucket_of[p]
        b = self.buckets[bi]
        b[3] += 1
        if b[3] == b[2]:
            self._pack(bi)
            b[4] = all_reduce_sum_(self.flat[b[0]:b[1]], async_op=True)

    def finish(self):
        """Call after backward, before optimizer.step(grad_scale=1 / world): waits for every bucket's sum."""
        if self.world == 1:
            return
        for bi, b in enumerate(self.buckets):
            if b[4] is None:                      # a bucket with a parameter that received no gradient this step
                self._pack(bi)
                b[4] = all_reduce_sum_(self.flat[b[0]:b[1]], async_op=True)
        for b in self.buckets:
            b[4].wait()
            b[3], b[4] = 0, None
        if hasattr(self.opt, '_packed'):
            self.opt._packed = True               # every bucket was gathered before its all-reduce: step() must not re-pack

    def _pack(self, bi):
        """The bucket's gradients into the flat buffer (the optimiser gathers them by kernel: optimizer/adam.py) just before its
        all-reduce; an optimiser without pack_range keeps .grad views of the flat buffer and needs nothing."""
        pack = getattr(self.opt, 'pack_range', None)
        if pack is not None:
            pack(self.members[bi])

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []
