"""Is the Fbank kernel a victim of the co-run hazard?  Stream A featurises a shard of waveforms while stream B runs the ECAPA
backbone (schedule from argv: 0 = 128-wide conv GEMMs) on other utterances; features compared bit for bit with a quiet run.
Usage: python tools/stress_fbank.py [schedule] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch, bench
from ppvector import _native as N
dev = torch.device('cuda', 0)
fz, model, head, _, _ = bench.build_ecapa(dev, 'bfloat16')
model.eval()
wav = torch.from_numpy(bench.synth_waves(256, 48000, seed=5)).to(dev)
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N.lib().vp_conv256_select(sched)
eng = model.engine('bfloat16')
f_all = fz(wav, want_bf16=True)
feats16 = f_all._vp_bf16
ref = fz(wav[:128], want_bf16=True)
ref32, ref16 = ref.clone(), ref._vp_bf16.clone()
emb_ref = eng.forward(feats16[128:].contiguous()).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad32 = bad16 = bade = 0
first = None
for it in range(iters):
    with torch.cuda.stream(sb):
        e = eng.forward(feats16[128:].contiguous())
    with torch.cuda.stream(sa):
        for rep in range(3):
            f = fz(wav[:128], want_bf16=True)
            if not torch.equal(f, ref32):
                bad32 += 1
                if first is None:
                    d = torch.nonzero((f != ref32).flatten()).flatten()
                    idx = [int(x) for x in d[:8]]
                    first = f'{d.numel()} values differ; (utt, frame, mel) of the first: {[(i // (298 * 80), (i // 80) % 298, i % 80) for i in idx]}  got {f.flatten()[idx[0]].item()} want {ref32.flatten()[idx[0]].item()}'
            if not torch.equal(f._vp_bf16, ref16):
                bad16 += 1
    torch.cuda.synchronize()
    bade += int(not torch.equal(e, emb_ref))
print(f'sched {sched}: fbank f32 differs {bad32}/{3 * iters}, bf16 twin differs {bad16}/{3 * iters}, backbone (fed quiet features) differs {bade}/{iters}')
if first:
    print('   ', first)
