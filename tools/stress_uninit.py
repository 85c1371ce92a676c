"""Does the bf16 ECAPA forward read scratch it never wrote?  Same input, workspace pre-filled with different garbage each time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from ppvector import _native as N  # noqa: E402

if len(sys.argv) > 1:
    N.lib().vp_conv256_select(int(sys.argv[1]))
dev = torch.device('cuda', 0)
fz, model, head, _, _ = bench.build_ecapa(dev, 'bfloat16')
model.eval()
for nb in (256, 128, 64):
    wav = torch.from_numpy(bench.synth_waves(nb, 48000, seed=5)).to(dev)
    for dt in ('bfloat16', 'float32'):
        eng = model.engine(dt)
        feats = fz(wav, want_bf16=(dt == 'bfloat16'))
        ref = eng.forward(feats).clone()
        torch.cuda.synchronize()
        res = []
        for fill in (0, 0x7f, 0xff, 0x3c):
            for w in eng.ws.bufs.values():
                w.fill_(fill)
            e = eng.forward(feats)
            torch.cuda.synchronize()
            res.append((fill, bool(torch.equal(e, ref)), float((e - ref).abs().max()), bool(torch.isnan(e).any())))
        print(f'B={nb} {dt}: (fill byte, identical, max diff, nan):', res, flush=True)
