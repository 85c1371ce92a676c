"""BASELINE configs[4] as a TRAINING step at its named shape: ERes2Net-large (55.2 M parameters: m_channels 64, mul_channel 2, expansion 4,
base_width 24, scale 3) + 200 000-class cosine head + AAMLoss, 1024 utterances over 8 GPUs = 128 per GPU, 3 s utterances (T = 298, F = 80).
One GPU's share (the box has one MI355X): step time (enable_amp and f32), peak memory, the captured backward stages with their replay times,
and the gradient all-reduce's fixed chunk schedule against them -- which chunk becomes ready after which stage, how much backward is left
behind it, and how much of the 374 MB could travel under the remaining stages at the node's ring rate (7 xGMI links x ~153 GB/s = 1.07 TB/s
egress per GPU; a ring all-reduce moves 2 (N - 1) / N x bytes per rank = 1.75 x at N = 8).
    python tools/config5_train_probe.py [batch] [steps]          (VP_C5_PROFILE=1: a few eager steps only, for rocprofv3 --kernel-trace --stats)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
import ppvector  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.eres2net import ERes2Net  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.optimizer.adam import Adam  # noqa: E402
from ppvector.train.step import GraphedTrainStep, TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
NCLS, T, F = 200000, 298, 80
EGRESS_GBPS, RING = 1071.0, 1.75
GF_FWD = 93.8                        # forward GFLOP per 3 s utterance; a training step = 3 x (fwd + dgrad + wgrad)

for amp in ((True,) if os.environ.get('VP_C5_PROFILE') else (True, False)):
    ppvector.set_train_amp(amp)
    torch.manual_seed(0)
    m = ERes2Net(F, embd_dim=192, m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3)
    model = torch.nn.Sequential(m, SpeakerIdentification(192, NCLS)).cuda()
    nparam = sum(q.numel() for q in model.parameters())
    x = torch.randn(B, T, F, device='cuda') * 3
    y = torch.randint(0, NCLS, (B,), device='cuda')
    opt = Adam(model.parameters(), learning_rate=1e-5, weight_decay=1e-6)
    if os.environ.get('VP_C5_PROFILE'):
        step = TrainStep(model, AAMLoss(), opt)
        for _ in range(4):
            step(x, y)
        torch.cuda.synchronize()
        break
    step = GraphedTrainStep(model, AAMLoss(), opt)
    for _ in range(5):
        loss, acc = step(x, y)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(STEPS):
        loss, acc = step(x, y)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / STEPS * 1e3
    tag = 'enable_amp' if amp else 'f32'
    tf = 3 * GF_FWD * B / ms
    peak = 2500.0 if amp else 157.3
    print(f'# config 5, one GPU\'s share: ERes2Net-large ({(nparam - 192 * NCLS) / 1e6:.1f} M) + {NCLS}-class head ({192 * NCLS / 1e6:.1f} M), B = {B}, {tag}: '
          f'{ms:.1f} ms / step = {B / ms * 1e3:.0f} utt/s = {tf:.0f} TFLOP/s (3 x {GF_FWD} GFLOP / utt) = {tf / peak:.3f} of the {"bf16" if amp else "f32"} MFMA peak; '
          f'loss {float(loss):.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; capture error: {step.capture_error}', flush=True)
    plan = next(iter(step._plans.values()), None)
    if plan is not None:
        graphs = plan['graphs']
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(graphs) + 1)]
        for rep in range(2):
            evs[0].record()
            for k, g in enumerate(graphs):
                g.replay()
                evs[k + 1].record()
            torch.cuda.synchronize()
        st = [evs[k].elapsed_time(evs[k + 1]) for k in range(len(graphs))]
        total = sum(st)
        print(f'  {len(graphs)} captured stages (stage 0 = forward + loss + the first backward stretch), replay times ms: ' + ' '.join(f'{v:.1f}' for v in st) + f'  (sum {total:.1f})')
        nbytes = opt.grad.numel() * 4
        print(f'  flat gradient buffer {nbytes / 1e6:.1f} MB f32 in {len(plan["chunks"])} all-reduce chunks; ring all-reduce at N = 8 moves {RING} x = {nbytes * RING / 1e6:.0f} MB per rank '
              f'= {nbytes * RING / EGRESS_GBPS / 1e6:.2f} ms at {EGRESS_GBPS:.0f} GB/s egress')
        done_at = [sum(st[:k + 1]) for k in range(len(st))]
        hidden = 0.0
        wire = 0.0                                            # when the wire becomes free
        for c, ((lo, hi), r) in enumerate(zip(plan['chunks'], plan['ready'])):
            cb = (hi - lo) * 4
            t_ready = done_at[r]
            start = max(t_ready, wire)
            dur = cb * RING / EGRESS_GBPS / 1e6
            wire = start + dur
            left = total - t_ready
            hidden += max(0.0, min(wire, total) - min(start, total))
            print(f'    chunk {c:2d}: {cb / 1e6:6.1f} MB ready after stage {r} at {t_ready:6.1f} ms ({left:6.1f} ms of backward left); on the wire {start:6.1f} .. {wire:6.1f} ms')
        tot_wire = nbytes * RING / EGRESS_GBPS / 1e6
        print(f'  => collective {tot_wire:.2f} ms in all, {hidden:.2f} ms of it under the remaining backward stages ({hidden / tot_wire:.0%}); exposed tail {max(0.0, wire - total):.2f} ms '
              f'on a {total:.1f} ms step ({max(0.0, wire - total) / total:.1%}) -- a projection from one GPU\'s stage times and the link rate, NOT a measurement at N = 8', flush=True)
    del step, opt, model, m
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
ppvector.set_train_amp(False)
