"""Race screen: the same ECAPA forward many times, single stream and as concurrent launch sequences, every embedding compared
bit for bit with the first.  Usage: python tools/stress_identity.py [iterations] [schedule]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from ppvector import _native as N  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2:
    N.lib().vp_conv256_select(int(sys.argv[2]))
dev = torch.device('cuda', 0)
fz, model, head, _, _ = bench.build_ecapa(dev, 'bfloat16')
model.eval()
wav = torch.from_numpy(bench.synth_waves(256, 48000, seed=5)).to(dev)
for dt in ('bfloat16', 'float32x3', 'float32'):
    eng = model.engine(dt)
    w16 = dt == 'bfloat16'
    ref = eng.forward(fz(wav, want_bf16=w16)).clone()
    bad = {'single': 0, 's2': 0, 's4': 0}
    worst = 0.0
    for i in range(it):
        e = eng.forward(fz(wav, want_bf16=w16))
        if not torch.equal(e, ref):
            bad['single'] += 1; worst = max(worst, (e - ref).abs().max().item())
        for S, k in ((2, 's2'), (4, 's4')):
            e = eng.forward_streams(wav, S, producer=lambda w: fz(w, want_bf16=w16))
            torch.cuda.synchronize()
            if not torch.equal(e, ref):
                bad[k] += 1; worst = max(worst, (e - ref).abs().max().item())
    print(f'{dt}: mismatches over {it} iterations {bad}  worst |diff| {worst:.3e}', flush=True)
