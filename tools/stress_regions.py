"""Where does a concurrent ECAPA forward first differ from a quiet one?  Shard i of forward_streams(S = 2) runs its 128 utterances
with its own workspace; the same shard run alone (quiet) leaves the same workspace layout (csrc/ecapa.hip: plan_ecapa).  Every
region that survives to the end of the forward is compared byte for byte.  cat0 = block0; cat = the three block outputs; t1 / r2 / t2
= the LAST block's tdnn1 / Res2 / tdnn2; se_s = its gate; mfa, h (ASP tdnn), stats / rowbias (ASP context), pooled.
Usage: python tools/stress_regions.py [schedule] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch, bench
from ppvector import _native as N
dev = torch.device('cuda', 0)
fz, model, head, _, _ = bench.build_ecapa(dev, 'bfloat16')
model.eval()
wav = torch.from_numpy(bench.synth_waves(256, 48000, seed=5)).to(dev)
sched = int(sys.argv[1]) if len(sys.argv) > 1 else -1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N.lib().vp_conv256_select(sched)
eng = model.engine('bfloat16')
feats = fz(wav, want_bf16=True)._vp_bf16
B, T = 128, 298
M = B * T
C, Cm, nb, width, att, se_ch = 512, 1536, 3, 64, 128, 128
tiles, nseg = N.lib().vp_conv1d_tiles_m(B, T), N.lib().vp_conv1d_nseg(T)
al = lambda n: (max(n, 1) + 255) // 256 * 256
regions, off = [], 0
for name, nbytes, row in (('cat0', M * C * 2, C * 2), ('cat', M * nb * C * 2, nb * C * 2), ('t1', M * C * 2, C * 2), ('r2', M * C * 2, C * 2),
                          ('t2', M * C * 2, C * 2), ('tmpA', M * width * 2, width * 2), ('tmpB', M * width * 2, width * 2),
                          ('mfa', M * Cm * 2, Cm * 2), ('h', M * att * 2, att * 2), ('e', M * Cm * 4, Cm * 4),
                          ('psum', tiles * nseg * Cm * 4, Cm * 4), ('psumsq', tiles * nseg * Cm * 4, Cm * 4), ('stats', B * 2 * Cm * 4, 2 * Cm * 4),
                          ('se_h', B * se_ch * 4, se_ch * 4), ('se_s', B * C * 4, C * 4), ('rowbias', B * att * 4, att * 4),
                          ('pooled', B * 2 * Cm * 4, 2 * Cm * 4)):
    regions.append((name, off, nbytes, row)); off += al(nbytes)
quiet = []
for i in range(2):
    sh = feats[128 * i:128 * (i + 1)].contiguous()
    eng.forward(sh)
    torch.cuda.synchronize()
    quiet.append(next(iter(eng.ws.bufs.values()))[:off].clone())
first = {}
for it in range(iters):
    eng.forward_streams(feats, 2)
    torch.cuda.synchronize()
    for i in range(2):
        ws = next(iter(eng._slots[i + 1].bufs.values()))[:off]
        if torch.equal(ws, quiet[i]):
            continue
        names = []
        for name, o, n, row in regions:
            d = torch.nonzero(ws[o:o + n] != quiet[i][o:o + n]).flatten()
            if d.numel():
                names.append(f'{name}({d.numel()} B, first row {int(d[0]) // row} col-byte {int(d[0]) % row})')
        key = names[0].split('(')[0] if names else '?'
        first[key] = first.get(key, 0) + 1
        if sum(first.values()) <= 6:
            print(f'iter {it} shard {i}: ' + ' '.join(names), flush=True)
print(f'sched {sched}: earliest differing region over {iters} x 2 shard runs: {first}', flush=True)
