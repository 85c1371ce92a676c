"""vp_asp_utt_fwd on fixed random operands, many launches: every pooled vector must be bit-identical to the first; where they differ,
which (utterance, channel, mean / std) and by how much.  Usage: python tools/asp_determinism.py [launches] [B] [T]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch
from ppvector import _native as N
lib, ctx = N.lib(), N.ctx(0)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 298
Cc, att = 1536, 128
g = torch.Generator(device='cuda').manual_seed(3)
dev = torch.device('cuda')
x = (torch.randn((B * T, Cc), device=dev, generator=g) * 2 + 0.5).to(torch.bfloat16)
wt = (torch.randn((att, Cc), device=dev, generator=g) / Cc ** 0.5).to(torch.bfloat16)
wc = (torch.randn((Cc, att), device=dev, generator=g) * (3.0 / att ** 0.5)).to(torch.bfloat16)
bias = torch.randn((att,), device=dev, generator=g)
sc = torch.rand((att,), device=dev, generator=g) + 0.5
sh = torch.randn((att,), device=dev, generator=g) * 0.1
rb = torch.randn((B, att), device=dev, generator=g) * 0.1
cb = torch.zeros((Cc,), device=dev)
L = N.TdnnLayer()
L.w, L.bias, L.bn_scale, L.bn_shift, L.cin, L.cout, L.kw, L.dil = wt.data_ptr(), bias.data_ptr(), sc.data_ptr(), sh.data_ptr(), Cc, att, 1, 1


def run():
    out = torch.full((B, 2 * Cc), float('nan'), device=dev)
    N.check(lib.vp_asp_utt_fwd(ctx, x.data_ptr(), Cc, C.byref(L), rb.data_ptr(), wc.data_ptr(), cb.data_ptr(), B, T, Cc, att, 1e-12,
                               out.data_ptr(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    return out


ref = run()
assert not torch.isnan(ref).any()
bad = 0
for i in range(it):
    o = run()
    if not torch.equal(o, ref):
        bad += 1
        d = (o - ref).abs()
        idx = torch.nonzero(d > 0)
        utts = sorted(set(idx[:, 0].tolist()))
        cols = sorted(set(idx[:, 1].tolist()))
        if bad <= 6:
            print(f'launch {i}: {idx.shape[0]} values differ; utterances {utts[:8]}; columns {cols[:12]}{"..." if len(cols) > 12 else ""} '
                  f'(mean part < {Cc}); max |diff| {d.max().item():.3e}', flush=True)
print(f'B={B} T={T}: {bad} of {it} launches differ from the first', flush=True)
