"""Head + loss timing at the named class counts: class-tiled (csrc/head_tiled.hip) against the logits-tensor kernels (csrc/head.hip).
forward = evaluation of the objective (SpeakerIdentification.eval() -> AAMLoss); train = HeadLoss forward (value + both gradients).
Usage: python tools/head_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.train.functions import HeadLoss  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for Cn, B in ((2796, 256), (7205, 64), (200000, 128)):
    D = 192
    g = torch.Generator().manual_seed(1)
    emb = (torch.randn(B, D, generator=g) * 2).cuda()
    labels = torch.randint(0, Cn, (B,), generator=g).cuda()
    head = SpeakerIdentification(D, Cn).cuda()
    crit = AAMLoss(margin=0.2, scale=32)
    head.eval()
    f_t = timed(lambda: crit(head(emb), labels))

    def unfused():
        o = head(emb)
        o['logits']
        return crit(o, labels)
    f_u = timed(unfused)
    W = head.weight.detach()
    out = {}
    for mode in ('tiled', 'untiled'):
        if mode == 'untiled':
            os.environ['VPMI_HEAD_UNTILED'] = '1'
        else:
            os.environ.pop('VPMI_HEAD_UNTILED', None)
        out[mode] = timed(lambda: HeadLoss.apply(emb, W, labels, 0.2, 32.0, 0.0, False))
    os.environ.pop('VPMI_HEAD_UNTILED', None)
    print(f'C = {Cn:6d}, B = {B:3d}: forward value  tiled {f_t:6.0f} us | logits tensor {f_u:6.0f} us;   value + d emb + d W  tiled {out["tiled"]:6.0f} us | '
          f'logits tensor {out["untiled"]:6.0f} us   (host overhead of ~30 us per call included)', flush=True)
