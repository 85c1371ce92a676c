"""PCIe-inclusive rate of the bench workload: the same step as bench.py, but every batch of waveforms
starts in pinned host memory -- (a) copied on the compute stream in front of the step, (b) double-buffered on a copy stream
so batch i+1 uploads while batch i computes.  Not the headline `value` (bench.py times HBM-resident inputs)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from oracle import fbank as ofb  # noqa: E402
from oracle import models as om  # noqa: E402
from ppvector.data_utils.featurizer import AudioFeaturizer  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402

B, L, STEPS = 256, 48000, 30
dev = torch.device('cuda')
host = [torch.from_numpy(ofb.synth_waves(B, L, seed=s)).pin_memory() for s in (1, 2)]
labels = (torch.arange(B, device=dev) * 7) % 2796
fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
model = EcapaTdnn(80)
model.load_state_dict(om.ecapa_params(80, seed=1000))
eng = model.to(dev).eval().engine('bfloat16')
head = SpeakerIdentification(192, 2796).to(dev).eval()
crit = AAMLoss()


def step(w):
    return crit(head(eng.forward(fz(w, want_bf16=True))), labels)


def timed(fn):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS


resident = host[0].to(dev)
t_res = timed(lambda i: step(resident))
t_ser = timed(lambda i: step(host[i & 1].to(dev, non_blocking=True)))
copy_stream = torch.cuda.Stream()
bufs = [torch.empty((B, L), device=dev), torch.empty((B, L), device=dev)]
ready = [torch.cuda.Event(), torch.cuda.Event()]
done = [torch.cuda.Event(), torch.cuda.Event()]


def upload(i):
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(done[i & 1])                    # the step that last read this buffer has finished
        bufs[i & 1].copy_(host[i & 1], non_blocking=True)
        ready[i & 1].record(copy_stream)


for e in done:
    e.record()
upload(0)


def overlapped(i):
    upload(i + 1)
    torch.cuda.current_stream().wait_event(ready[i & 1])
    step(bufs[i & 1])
    done[i & 1].record()


t_ovl = timed(overlapped)
mb = B * L * 4 / 1e6
print(f'batch {B} x 3 s = {mb:.1f} MB of f32 samples per step')
print(f'inputs resident in HBM        : {t_res * 1e3:7.3f} ms/step  {B / t_res:10.0f} utt/s')
print(f'H2D on the compute stream      : {t_ser * 1e3:7.3f} ms/step  {B / t_ser:10.0f} utt/s   (copy alone ~{(t_ser - t_res) * 1e3:.3f} ms = {mb / 1e3 / max(t_ser - t_res, 1e-9):.1f} GB/s)')
print(f'H2D double-buffered, own stream: {t_ovl * 1e3:7.3f} ms/step  {B / t_ovl:10.0f} utt/s')

# 16-bit PCM as the decoder delivers it: half the bytes over PCIe, widened inside the Fbank frame kernel (vp_fbank_cmn_pcm16)
host16 = [torch.clamp((h / h.abs().max() * 20000.0).round(), -32768, 32767).to(torch.int16).pin_memory() for h in host]
bufs16 = [torch.empty((B, L), device=dev, dtype=torch.int16), torch.empty((B, L), device=dev, dtype=torch.int16)]


def upload16(i):
    with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(done[i & 1])
        bufs16[i & 1].copy_(host16[i & 1], non_blocking=True)
        ready[i & 1].record(copy_stream)


torch.cuda.synchronize()
for e in done:
    e.record()
upload16(0)


def overlapped16(i):
    upload16(i + 1)
    torch.cuda.current_stream().wait_event(ready[i & 1])
    step(bufs16[i & 1])
    done[i & 1].record()


t_ser16 = timed(lambda i: step(host16[i & 1].to(dev, non_blocking=True)))
t_ovl16 = timed(overlapped16)
print(f'int16 PCM ({mb / 2:.1f} MB per step), H2D on the compute stream : {t_ser16 * 1e3:7.3f} ms/step  {B / t_ser16:10.0f} utt/s')
print(f'int16 PCM, H2D double-buffered on its own stream      : {t_ovl16 * 1e3:7.3f} ms/step  {B / t_ovl16:10.0f} utt/s')
