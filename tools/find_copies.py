"""Where do the memcpy / fill / aten kernels of the bench's infer step come from?  torch.profiler with Python stacks over a few
eager single-stream steps; prints every non-libvpmi device activity with the innermost repo frame.
Usage: python tools/find_copies.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
wav = torch.from_numpy(bench.synth_waves(bench.BATCH, bench.N_SAMPLES, seed=1000)).to(dev)
labels = (torch.arange(bench.BATCH, device=dev) * 7) % bench.N_CLASSES
run, info = bench.make_infer_step(dev, 'bfloat16', 1, wav, labels, graph=False)
for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        run()
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    name = ev.name
    if not (name.startswith('aten::') or 'Memcpy' in name or 'Memset' in name or 'memcpy' in name.lower()):
        continue
    if name in ('aten::empty', 'aten::empty_strided', 'aten::view', 'aten::slice', 'aten::as_strided', 'aten::select', 'aten::detach', 'aten::alias',
                'aten::reshape', 'aten::_unsafe_view', 'aten::empty_like', 'aten::lift_fresh', 'aten::unsqueeze', 'aten::resolve_conj', 'aten::resolve_neg'):
        continue
    frames = [f for f in (ev.stack or []) if 'repo' in f or 'bench.py' in f]
    key = (name, frames[0] if frames else (ev.stack[0] if ev.stack else '?'))
    seen[key] = seen.get(key, 0) + 1
for (name, frame), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:40s} {frame}')
