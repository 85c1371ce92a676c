"""asp_utt (one kernel per utterance) against the two-launch path (VPMI_ASP_SPLIT=1, a second process): ECAPA bf16 embeddings for
several utterance lengths.  Usage: python tools/asp_probe.py [dump|check]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch, bench
dev = torch.device('cuda', 0)
fz, model, head, _, _ = bench.build_ecapa(dev, 'bfloat16')
model.eval()
eng = model.engine('bfloat16')
g = torch.Generator().manual_seed(7)
out = {}
for (B, T) in [tuple(int(v) for v in os.environ['SHAPES'].split('x'))] if os.environ.get('SHAPES') else ((2, 64), (3, 100), (2, 160), (5, 240), (4, 298), (2, 304), (64, 298)):
    feats = torch.randn((B, T, 80), generator=g).cuda()
    out[(B, T)] = eng.forward(feats).cpu()
path = '/tmp/asp_probe.pt'
if sys.argv[1] == 'dump':
    torch.save(out, path)
else:
    ref = torch.load(path)
    for k, e in out.items():
        r = ref[k]
        print(k, 'nan', int(torch.isnan(e).sum()), 'ref nan', int(torch.isnan(r).sum()), 'max |diff|', float((e - r).abs().max()), 'rel', float((e - r).norm() / r.norm()))
