"""world_size-2 gloo test (CPU) of the N>1 bench path: the forward path shards by utterance with no
data-path collective, so what N>1 adds is the barrier-bracketed timing and the max-over-ranks
reduction -- and of the training path's data-parallel pieces: contiguous batch sharding and the bucketed all-reduce
average over a flat gradient buffer (ppvector/train/ddp.py; RCCL on the GPUs, gloo here)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    calls = []

    def step():
        time.sleep(0.01 * (rank + 1))          # rank 1 is the slow one
        calls.append(1)
        return len(calls)

    dt, out = bench.run_timed(step, steps=5, warmup=2, dist=dist, device=torch.device('cpu'))
    q.put((rank, dt, out, len(calls), bench.shard_seed(1000, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_is_max_over_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, dt0, out0, n0, s0), (r1, dt1, out1, n1, s1) = res
    assert n0 == n1 == 7 and out0 == out1 == 7            # W + K steps each, exactly
    assert abs(dt0 - dt1) < 1e-9                           # both report the reduced (max) time
    assert dt0 >= 5 * 0.02 * 0.9                           # ... which is the slow rank's
    assert s0 != s1                                        # distinct data shards


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppvector.train.ddp import allreduce_mean_, shard_batch
    idx = list(shard_batch(11, rank, world))
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r + 1) * [0 .. 999]
    allreduce_mean_(g, bucket_bytes=1024)                             # 4 buckets of 256 floats
    q.put((rank, idx, g[:3].tolist(), float(g[999])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, i0, h0, t0), (_, i1, h1, t1) = res
    assert i0 == [0, 1, 2, 3, 4, 5] and i1 == [6, 7, 8, 9, 10]        # contiguous split, last shard short
    assert h0 == h1 == [0.0, 1.5, 3.0] and t0 == t1 == 999 * 1.5     # mean of 1x and 2x


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppvector.optimizer.adam import Adam
    from ppvector.train.ddp import OverlappedReducer, shard_batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.Tanh(), torch.nn.Linear(32, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    x, y = torch.randn(10, 12), torch.randn(10, 4)
    full = torch.nn.Sequential(*[type(l)(l.in_features, l.out_features) if isinstance(l, torch.nn.Linear) else torch.nn.Tanh() for l in net])
    full.load_state_dict(net.state_dict())
    ((full(x) - y) ** 2).sum().div(10).backward()                      # single-process reference: mean over the global batch
    opt = Adam(net.parameters(), learning_rate=1e-3)                    # only the flat buffers are used here (no .step() on CPU)
    red = OverlappedReducer(opt, bucket_bytes=1024)                     # several buckets
    idx = list(shard_batch(10, rank, world))
    ((net(x[idx]) - y[idx]) ** 2).sum().div(len(idx)).backward()       # rank-local mean; equal shards -> mean of means = global mean
    red.finish()                                                       # the flat buffer now holds the SUM over the ranks; the 1 / world
    err = max((p.grad / world - r.grad).abs().max().item() for p, r in zip(net.parameters(), full.parameters()))   # rides on optimizer.step(grad_scale=)
    # second step as the trainer runs it: clear_grad() drops the .grad tensors, autograd hands over fresh ones, every bucket is
    # PACKED into the flat buffer just before its all-reduce, and the averaged gradients sit in the flat buffer only
    opt.clear_grad()
    assert all(p.grad is None for p in net.parameters())
    ((net(x[idx]) - y[idx]) ** 2).sum().div(len(idx)).backward()
    assert all(p.grad.data_ptr() != opt.grad.data_ptr() + 4 * opt._offset(p) for p in net.parameters())
    red.finish()
    assert opt._packed
    for p, r in zip(net.parameters(), full.parameters()):
        o = opt._offset(p)
        err = max(err, (opt.grad[o:o + p.numel()].view_as(r.grad) / world - r.grad).abs().max().item())
    q.put((rank, len(red.buckets), err))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_reducer_matches_full_batch_gradient():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, nb, err in res:
        assert nb >= 2 and err < 1e-6, (rank, nb, err)


def test_batch_loader_shards_like_distributed_batch_sampler():
    """The trainer's index batching (ppvector/trainer.py mirror of paddle.io.DistributedBatchSampler, trainer.py:105-107):
    every rank sees the same shuffled order, takes a contiguous 1/world shard (padded by wrapping), equal batch counts."""
    from ppvector.trainer import _BatchLoader

    class DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return i

    n, world, bs = 103, 4, 8
    loaders = [_BatchLoader(DS(n), batch_size=bs, shuffle=True, drop_last=True, num_workers=2, rank=r, world=world) for r in range(world)]
    assert len({len(l) for l in loaders}) == 1 and len(loaders[0]) == ((n + world - 1) // world) // bs
    for epoch in range(2):
        seen = []
        for l in loaders:
            batches = list(l)
            assert len(batches) == len(l) and all(len(b) == bs for b in batches)
            seen.append([i for b in batches for i in b])
        flat = [i for s in seen for i in s]
        assert len(set(flat)) >= len(flat) - (world * ((n + world - 1) // world) - n)      # disjoint except the wrap padding
        if epoch == 0:
            first = seen
    assert first != seen                                                                   # reshuffled per epoch
    again = [_BatchLoader(DS(n), batch_size=bs, shuffle=True, drop_last=True, rank=r, world=world) for r in range(world)]
    assert [[i for b in l for i in b] for l in again] == first                             # deterministic per (seed, epoch)
    tail = _BatchLoader(DS(10), batch_size=4, shuffle=False, drop_last=False)
    assert [b for b in tail] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and len(tail) == 3


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself, run a real all-reduce between them and
    print ONE JSON line with n_gpus = 2 (the driver invokes it exactly like this).  --dry-run swaps the HIP step for a CPU one
    and RCCL for gloo; launcher, rendezvous, timing, gradient all-reduce and reporting are the code the GPU run uses."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['steps'] == 3 and out['value'] > 0


def test_reducer_buckets_follow_backward_order():
    """Buckets are cut from the LAST parameter backwards (backward fills them in that order) and cover the flat buffer exactly."""
    sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
    from ppvector.train.ddp import OverlappedReducer

    class Opt:
        pass
    o = Opt()
    o.params = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 3000, 500, 4000, 200)]
    o.grad = torch.zeros(sum(p.numel() for p in o.params))
    r = OverlappedReducer(o, bucket_bytes=4 * 4000)
    spans = sorted((b[0], b[1]) for b in r.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == o.grad.numel()
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))              # contiguous, no overlap
    assert r.buckets[0][1] == o.grad.numel()                                # bucket 0 = the tail of the buffer
    assert r.bucket_of[o.params[-1]] == 0 and r.bucket_of[o.params[0]] == len(r.buckets) - 1
    assert len(r.buckets) >= 2


def test_cut_points_split_backward_into_a_chain_of_stages():
    """ppvector/train/segments.py on plain torch (CPU): a network with ECAPA's skip structure (every block output feeds the next
    block AND a final concatenation) differentiated stage by stage -- last stage first, `between` called after each -- gives
    exactly the gradients of one backward(); outside a Recorder `cut` is the identity."""
    sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
    from ppvector.train.segments import Recorder, cut
    torch.manual_seed(3)
    blocks = [torch.nn.Linear(8, 8) for _ in range(4)]
    tail = torch.nn.Linear(24, 5)

    def forward(x):
        x = torch.tanh(blocks[0](x))
        outs = []
        for b in blocks[1:]:
            x = torch.tanh(b(x)) + x
            outs.append(x)
            outs = list(cut(*outs))
            x = outs[-1]
        return tail(torch.cat(outs, dim=1)).square().mean()

    params = [p for m in blocks + [tail] for p in m.parameters()]
    x = torch.randn(6, 8)
    forward(x).backward()
    ref = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    rec = Recorder()
    order = []
    with rec:
        loss = forward(x)

        def between(i):
            order.append((i, sorted(k for k, p in enumerate(params) if p.grad is not None)))

        rec.backward(loss, between)
    assert rec.n_stages == 4 and [i for i, _ in order] == [0, 1, 2, 3]
    # stage 0 = the tail only; each later stage adds exactly one earlier block (the first stage of the forward holds two)
    assert order[0][1] == [8, 9] and order[1][1] == [6, 7, 8, 9] and order[2][1] == [4, 5, 6, 7, 8, 9] and order[3][1] == list(range(10))
    for p, r in zip(params, ref):
        assert torch.allclose(p.grad, r, rtol=1e-6, atol=1e-7)
    a, b = torch.randn(2, requires_grad=True), torch.randn(2)
    assert cut(a, b) == (a, b) or all(u is v for u, v in zip(cut(a, b), (a, b)))


def test_cut_carries_the_producers_bf16_twin_to_the_next_stage():
    """A tensor's `_vp_bf16` attribute (the bf16 GEMM operand its producer wrote, train/functions.py) must survive a stage cut: the
    leaf that replaces the tensor in the next stage is what the next block's tdnn1 / the MFA layer receive."""
    import torch
    from ppvector.train.segments import Recorder, cut
    a = torch.randn(6, 4, requires_grad=True)
    y = a * 2
    twin = y.detach().to(torch.bfloat16)
    y._vp_bf16 = twin
    z = torch.randn(6, 4)                                  # no grad: passes through unchanged
    with Recorder() as rec:
        y2, z2 = cut(y, z)
        assert y2 is not y and z2 is z
        assert getattr(y2, '_vp_bf16', None) is twin
        loss = (y2 * y2).sum()
        rec.backward(loss)
    assert torch.allclose(a.grad, 8 * a.detach())
    y3, = cut(y)                                           # outside a recorder: identity
    assert y3 is y


def test_bf16_only_activation_travels_as_a_placeholder_with_an_f32_gradient():
    """Round 4: an activation that exists as bf16 only is put on the autograd tape as an f32 tensor that owns one element (a zero expanded
    to the shape: ppvector/train/functions.py `_placeholder`); its values travel as the `_vp_bf16` twin with `_vp_bf16_only` set.  What
    this relies on, checked on plain torch (CPU): (a) a custom Function may return such a view and receive an f32 gradient of the FULL
    shape for it; (b) had the output been a bf16 tensor, autograd would hand its gradient over as bf16 -- the reason for the placeholder;
    (c) a stage cut carries both attributes; (d) a consumer without a bf16 path gets the values (`_f32c`)."""
    import torch
    from ppvector.train.segments import Recorder, cut
    seen = {}

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, as_bf16):
            ctx.save_for_backward(x)
            if as_bf16:
                return (x * 3).to(torch.bfloat16)
            return torch.zeros(1).expand(x.shape)            # placeholder; the caller hangs the twin on it

        @staticmethod
        def backward(ctx, g):
            seen['producer_grad'] = (g.dtype, tuple(g.shape), g.clone())
            return g.float() * 3, None

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            twin = y._vp_bf16 if getattr(y, '_vp_bf16_only', False) else y
            ctx.save_for_backward(twin)
            return twin.float().pow(2).sum()

        @staticmethod
        def backward(ctx, g):
            twin, = ctx.saved_tensors
            return g * 2 * twin.float()                        # f32 gradient of the full shape, whatever the input's storage is

    a = torch.randn(5, 7, requires_grad=True)
    y = Producer.apply(a, False)
    assert y.dtype == torch.float32 and tuple(y.shape) == (5, 7) and y.untyped_storage().nbytes() == 4
    y._vp_bf16, y._vp_bf16_only = (a.detach() * 3).to(torch.bfloat16), True
    with Recorder() as rec:
        y2, = cut(y)
        assert getattr(y2, '_vp_bf16_only', False) and y2._vp_bf16 is y._vp_bf16
        rec.backward(Consumer.apply(y2))
    dt, shape, g = seen['producer_grad']
    assert dt == torch.float32 and shape == (5, 7)
    ref = 2 * (a.detach() * 3).to(torch.bfloat16).float()
    assert torch.equal(g, ref) and torch.equal(a.grad, ref * 3)
    # (b) the same graph with a bf16 tensor on the tape: the consumer's f32 gradient arrives rounded to bf16
    a2 = a.detach().clone().requires_grad_()
    Consumer.apply(Producer.apply(a2, True)).backward()
    assert seen['producer_grad'][0] == torch.bfloat16
    # (d) _f32c on a placeholder returns the twin's values (GPU-only helper: checked for its branch logic on a stand-in)
    from ppvector.train import functions as F
    ph = torch.zeros(1).expand(5, 7)
    ph._vp_bf16, ph._vp_bf16_only = y._vp_bf16, True
    assert F._only16(ph) is y._vp_bf16 and F._only16(torch.zeros(2)) is None


def test_pass_through_gradients_are_handed_on_without_a_copy():
    """Recorder.backward: a leaf that only travels through a cut (consumed further down) receives the later leaf's gradient as the SAME
    tensor; what its own stage adds is accumulated by autograd.  Gradients equal the single-tape result."""
    import torch
    from ppvector.train.segments import Recorder, cut
    w1, w2, w3 = (torch.randn(4, 4, requires_grad=True) for _ in range(3))
    x = torch.randn(3, 4)

    def net(record):
        o1 = torch.tanh(x @ w1)
        o1c, = cut(o1) if record else (o1,)
        o2 = torch.tanh(o1c @ w2)
        o1d, o2d = cut(o1c, o2) if record else (o1c, o2)       # o1 travels through this cut to reach its second consumer
        return (torch.cat([o1d, o2d], dim=1) @ torch.cat([w3, w3], dim=0)).pow(2).sum()

    net(False).backward()
    want = [w.grad.clone() for w in (w1, w2, w3)]
    for w in (w1, w2, w3):
        w.grad = None
    with Recorder() as rec:
        loss = net(True)
        assert rec.n_stages == 3
        rec.backward(loss)
    for w, g in zip((w1, w2, w3), want):
        assert torch.allclose(w.grad, g, rtol=1e-5, atol=1e-6)


def test_collective_schedule_of_the_named_configs():
    """The fixed chunk list of the data-parallel step (train/step.py: reduce_chunks; the reference's fleet reducer, trainer.py:316-320) at the
    gradient sizes of the BASELINE configs: ECAPA + 2 796 classes (6.73 M parameters, 26.9 MB): seven chunks; CAM++ + 7 205 classes
    (8.2 M): eight; ERes2Net-large + the 200 000-class head (93.6 M + 38.4 M parameters, 528 MB of f32 gradients; 374 MB backbone alone):
    fifteen / sixteen chunks and the chunk that has to wait for the LAST backward stage -- the only one no later stage can hide -- is at most
    10 % of the buffer.  Chunks tile the buffer exactly, last parameters first, whatever the size."""
    from ppvector.train.step import MAX_CHUNKS, MIN_CHUNK, reduce_chunks
    for n, want in ((6_730_000, 7), (8_200_000, 8), (93_600_000, 15), (93_600_000 + 38_400_000, 16), (1, 1), (MIN_CHUNK, 1), (MIN_CHUNK + 1, 2)):
        ch = reduce_chunks(n)
        assert len(ch) == want <= MAX_CHUNKS, (n, len(ch))
        assert ch[0][1] == n and ch[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(ch, ch[1:]))
        assert all(hi > lo for lo, hi in ch)
    for n in (93_600_000, 93_600_000 + 38_400_000):
        lo, hi = reduce_chunks(n)[-1]                       # parameters [0, ...): complete only after the first layers' backward
        assert (hi - lo) / n <= 0.10, (n, (hi - lo) / n)
