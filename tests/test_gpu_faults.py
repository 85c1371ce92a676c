"""GPU tests of the grid-barrier bail-out machinery of the fused Res2Net training kernels (csrc/res2_train.hip, train/step.py) and of
their co-residency guard (ADVICE r04 mediums):

  * an injected bail-out drops the step on the DEVICE -- weights, Adam moments and EVERY BatchNorm running statistic unchanged -- in the
    replayed step and in the eager step; the host notices through the asynchronous per-step poll, switches to the per-chunk kernels,
    re-arms the words and training continues;
  * under two ranks the flag travels with the gradients: both ranks drop the same steps and end bit-identical;
  * the fused kernels beside a kernel of another queue that holds CUs (the stand-in for a collective): no fault, same results;
  * vp_set_grid_reserve_cus routes launches that would not fit beside the reserve to the per-chunk path.
The reference has no counterpart (paddle's kernels need no grid barrier, trainer.py:202-274); what these tests protect is OUR fused path."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import models as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from ppvector import _native as N
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: these tests must run on an MI355X (no CPU fallback exists)')
    N.ctx(0)
    return N


@pytest.fixture
def amp():
    import ppvector
    ppvector.set_train_amp(True)
    ppvector.set_fused_grid_kernels(True)
    yield
    ppvector.set_train_amp(False)
    ppvector.set_fused_grid_kernels(True)


def _make(n_cls=9, seed=21):
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
    m.load_state_dict(om.ecapa_params(80, seed=seed))
    head = SpeakerIdentification(192, n_cls)
    head.load_state_dict({'weight': om.head_params(192, n_cls, seed=seed + 1)})
    return torch.nn.Sequential(m, head).cuda()


def _state(model, opt):
    s = {k: v.detach().clone() for k, v in model.state_dict().items()}
    s['@m'], s['@v'], s['@flat'] = opt.m.clone(), opt.v.clone(), opt.flat.clone()
    return s


def _same(a, b):
    return [k for k in a if not torch.equal(a[k], b[k])]


@pytest.mark.parametrize('graphed', [True, False])
def test_injected_bail_out_drops_the_step_on_the_device_and_the_host_recovers(N, amp, graphed):
    import warnings
    import ppvector
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.optimizer.adam import Adam
    from ppvector.train.step import GraphedTrainStep, TrainStep
    g = torch.Generator().manual_seed(5)
    B, T = 6, 140
    xs = [(torch.randn(B, T, 80, generator=g) * 2).cuda() for _ in range(9)]
    ys = [torch.randint(0, 9, (B,), generator=g).cuda() for _ in range(9)]
    model = _make()
    opt = Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6)
    step = (GraphedTrainStep if graphed else TrainStep)(model, AAMLoss(margin=0.2, scale=32), opt, **({} if graphed else {'overlap_allreduce': False}))
    words = N.grid_words(0)
    assert words is not None and words.shape == (N.GRID_WORDS,) and int(words[N.FAULT_WORD]) == 0
    for i in range(5):                                    # three eager sightings, the capture, one replay
        step(xs[i], ys[i])
    torch.cuda.synchronize()
    if graphed:
        assert step.capture_error is None and len(step._plans) == 1
    assert N.lib().vp_grid_barrier_status(N.ctx(0)) == 0 and step.faults == 0
    before = _state(model, opt)
    assert any(k.endswith('_mean') for k in before)
    # --- the bail-out word as a grid barrier that gave up would leave it
    words[N.FAULT_WORD] = 0xdead
    step(xs[5], ys[5])
    torch.cuda.synchronize()
    changed = _same(before, _state(model, opt))
    assert not changed, f'a dropped step reached persistent state: {changed[:6]}'
    # --- the poll issued by that step is read by the next one (no host stall in between): per-chunk kernels from here on
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        step(xs[6], ys[6])                               # (still dropped: the word was set while it ran)
        torch.cuda.synchronize()
        step(xs[7], ys[7])
        torch.cuda.synchronize()
    assert step.faults == 1, step.faults
    assert any('grid barrier' in str(w.message) for w in rec)
    assert not ppvector.get_fused_grid_kernels()
    assert int(words[N.FAULT_WORD]) == 0 and N.lib().vp_grid_barrier_status(N.ctx(0)) == 0
    if graphed:
        assert len(step._plans) == 0                      # the captured graphs replayed the fused kernels: dropped
    after = _state(model, opt)
    moved = _same(before, after)
    assert '@flat' in moved and '@m' in moved and any(k.endswith('_mean') for k in moved), moved[:5]
    assert all(bool(torch.isfinite(v).all()) for v in after.values() if v.is_floating_point())
    loss, _ = step(xs[8], ys[8])
    assert np.isfinite(float(loss))
    # the synchronous form used before checkpoints finds nothing now
    assert step.check_faults() is False


def _two_rank_fault_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hashlib
    import warnings
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                                       # both ranks share the one GPU of the test box; gloo carries the sums
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ppvector
    from ppvector import _native as N
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.optimizer.adam import Adam
    from ppvector.train.ddp import shard_batch
    from ppvector.train.step import GraphedTrainStep
    ppvector.set_train_amp(True)
    g = torch.Generator().manual_seed(31)
    xs = [(torch.randn(8, 120, 80, generator=g) * 2) for _ in range(9)]
    ys = [torch.randint(0, 9, (8,), generator=g) for _ in range(9)]
    model = _make()
    opt = Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6)
    step = GraphedTrainStep(model, AAMLoss(margin=0.2, scale=32), opt)
    idx = list(shard_batch(8, rank, world))
    snaps = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for i, (x, y) in enumerate(zip(xs, ys)):
            if i == 5 and rank == 1:
                N.grid_words(0)[N.FAULT_WORD] = 0xdead              # rank 1's barrier "gave up" before step 5
            step(x[idx].cuda(), y[idx].cuda())
            torch.cuda.synchronize()
            snaps.append(hashlib.sha1(opt.flat.detach().cpu().numpy().tobytes()).hexdigest())   # (bit-exact comparisons, 40 bytes per step)
    q.put((rank, snaps, step.faults, bool(ppvector.get_fused_grid_kernels()), step.capture_error))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_drop_the_same_steps_when_one_ranks_barrier_gives_up(N):
    """the drop used to be rank-local -- the faulted rank skipped its update while its garbage gradient, already
    all-reduced, was applied by its peers.  The flag now travels with the gradients (MAX all-reduce of the bail-out word behind the
    last chunk): after rank 1's injected fault BOTH ranks leave step 5 out, both notice and move to the per-chunk kernels, and their
    parameters are bit-identical after every step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 39500 + os.getpid() % 2000
    procs = [ctx.Process(target=_two_rank_fault_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, s0, f0, fused0, e0), (_, s1, f1, fused1, e1) = res
    assert e0 is None and e1 is None, (e0, e1)
    for i, (a, b) in enumerate(zip(s0, s1)):
        assert a == b, f'ranks diverged at step {i}'
    assert s0[4] == s0[5], 'step 5 (the faulted one) reached the weights'
    assert s0[3] != s0[4] and s0[-2] != s0[-1], 'training did not continue'
    assert f0 >= 1 and f1 >= 1 and not fused0 and not fused1, (f0, f1, fused0, fused1)
    dropped = sum(int(a == b) for a, b in zip(s0, s0[1:]))
    print(f'[2-rank bail-out] steps dropped on both ranks: {dropped} of {len(s0)} (fault injected before step 5 on rank 1), '
          f'faults noticed {f0} / {f1}')
    assert dropped <= 3


def _run_steps(step, xs, ys, hog=None):
    out = []
    for x, y in zip(xs, ys):
        if hog is not None:
            hog()
        loss, _ = step(x, y)
        out.append(loss)
    torch.cuda.synchronize()
    return [float(v) for v in out]


@pytest.mark.parametrize('B,hog_wgs', [(32, 64), (224, 64)])
def test_fused_grid_kernels_beside_a_kernel_that_holds_cus(N, amp, B, hog_wgs):
    """the data-parallel step co-schedules a collective (a persistent kernel of another queue) with the 1-workgroup-
    per-CU grid-barrier kernels on purpose.  Stand-in: vp_occupy_cus -- `hog_wgs` workgroups that each pin a CU's whole LDS (nothing of
    ours can share the CU) for 10 ms, relaunched on a side stream before every step, so that training stages keep meeting it.  B = 32
    (the 8-GPU share of 256): the fused kernels fit beside it.  B = 224: they do NOT fit while the hog runs (224 + 64 > 256 CUs) -- the
    late workgroups wait for it to end, the barrier must not give up.  Results: no fault, no dropped step, and the same parameters as
    the quiet run (1e-6; printed: whether to the last bit -- the kernels' reductions are order-fixed)."""
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.optimizer.adam import Adam
    from ppvector.train.step import GraphedTrainStep
    T, steps = 150, 7
    g = torch.Generator().manual_seed(77)
    xs = [(torch.randn(B, T, 80, generator=g) * 2).cuda() for _ in range(steps)]
    ys = [torch.randint(0, 9, (B,), generator=g).cuda() for _ in range(steps)]
    side = torch.cuda.Stream()
    ctx = N.ctx(0)

    def hog():
        N.check(N.lib().vp_occupy_cus(ctx, hog_wgs, 160 * 1024, 10000, side.cuda_stream), ctx)

    finals, losses = [], []
    for with_hog in (False, True):
        model = _make()
        opt = Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6)
        step = GraphedTrainStep(model, AAMLoss(margin=0.2, scale=32), opt)
        losses.append(_run_steps(step, xs, ys, hog if with_hog else None))
        side.synchronize()
        assert step.capture_error is None and len(step._plans) == 1
        assert N.lib().vp_grid_barrier_status(ctx) == 0 and step.faults == 0 and step.check_faults() is False
        finals.append(opt.flat.detach().clone())
    print(f'[grid kernels beside a CU hog] B = {B}, hog = {hog_wgs} workgroups x 160 KB LDS x 10 ms per step: losses quiet '
          f'{[f"{v:.5f}" for v in losses[0]]} / beside the hog {[f"{v:.5f}" for v in losses[1]]}')
    dp = ((finals[0] - finals[1]).abs().max() / finals[0].abs().max()).item()
    print(f'    parameters after {steps} steps: max difference {dp:.2e} of the largest weight (bit-identical: {torch.equal(finals[0], finals[1])})')
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(losses[0], losses[1])), (losses[0], losses[1])
    assert dp <= 1e-6


def test_grid_reserve_sends_launches_that_do_not_fit_to_the_per_chunk_path(N, amp):
    """vp_set_grid_reserve_cus(n): a fused launch of more than #CUs - n workgroups returns VP_EUNSUP (train/functions.py then runs the
    per-chunk kernels); inside the limit it launches.  The data-parallel step sets 64 when it has peers (train/step.py: _reserve_cus)."""
    from ppvector.train.functions import Res2Fn
    lib, ctx = N.lib(), N.ctx(0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    S, T = 8, 64
    g = torch.Generator().manual_seed(3)
    params = []
    for _ in range(S - 1):
        params += [(torch.randn(64, 64, 3, generator=g) * 0.05).cuda(), torch.zeros(64).cuda(), torch.ones(64).cuda(), torch.zeros(64).cuda(),
                   torch.zeros(64).cuda(), torch.ones(64).cuda()]

    def launch(B):
        x = torch.randn(B * T, 64 * S, generator=g).cuda().to(torch.bfloat16)
        out16 = torch.empty_like(x)
        z = torch.empty((S - 1, B * T, 64), dtype=torch.float32, device='cuda')
        inb = torch.empty((S - 1, B * T, 64), dtype=torch.bfloat16, device='cuda')
        stats = torch.empty((S - 1, 2, 64), dtype=torch.float32, device='cuda')
        d = Res2Fn._fused_desc(x, None, dict(B=B, T=T, dilation=2, momentum=0.9, eps=1e-5), params, S)
        d.z, d.inb, d.stats, d.out_bf16 = z.data_ptr(), inb.data_ptr(), stats.data_ptr(), out16.data_ptr()
        ws = torch.empty(lib.vp_res2_train_workspace_bytes(B, S), dtype=torch.uint8, device='cuda')
        rc = lib.vp_res2_train_fwd(ctx, C.byref(d), ws.data_ptr(), ws.numel(), N.stream_ptr())
        torch.cuda.synchronize()
        return rc

    try:
        assert launch(cus) == 0 and launch(cus + 1) == N.VP_EUNSUP
        N.check(lib.vp_set_grid_reserve_cus(ctx, 64), ctx)
        assert launch(cus - 64) == 0 and launch(cus - 63) == N.VP_EUNSUP
    finally:
        N.check(lib.vp_set_grid_reserve_cus(ctx, 0), ctx)
    assert lib.vp_grid_barrier_status(ctx) == 0
