"""Does splitting the batch over concurrent HIP streams pack the chip better?  ECAPA forward (bf16 engine) on B = 256
as one launch sequence vs S independent sequences of B / S utterances on S streams (each with its own engine workspace).
The 512 -> 512 GEMMs leave 596 tiles on 256 CUs (2.33 rounds) and the HBM-bound tail kernels idle the matrix cores; with
S > 1 the second round's idle CUs can take the other stream's work.  Usage: python tools/stream_probe.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.engine import EcapaEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda', 0)
model = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
model.load_state_dict(bench.random_state(model, seed=1000))
model = model.to(dev).eval()
feats = torch.randn((256, 298, 80), device=dev).to(torch.bfloat16)
ref = None
for S in (1, 2, 4):
    with torch.no_grad():
        engs = [EcapaEngine(model, 'bfloat16') for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = list(feats.chunk(S))

    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for e, st, p in zip(engs, streams, parts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(e.forward(p))
        for st in streams:
            cur.wait_stream(st)
        return outs

    for _ in range(5):
        outs = step()
    torch.cuda.synchronize()
    emb = torch.cat(outs)
    if ref is None:
        ref = emb
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    print(f'streams {S}: {ms:.3f} ms per 256 utterances ({256 / ms:.1f} k utt/s)   max |emb - emb(S=1)| {(emb - ref).abs().max().item():.2e}', flush=True)
