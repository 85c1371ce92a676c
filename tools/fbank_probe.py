"""Micro-benchmark of the Fbank+CMN kernels at the BASELINE shape (B=256, 3 s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import fbank as ofb  # noqa: E402
from ppvector.data_utils.featurizer import AudioFeaturizer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w = ofb.synth_waves(256, 48000, seed=1)
wav = torch.from_numpy(w).cuda()
fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
out = fz(wav, want_bf16=True)
torch.cuda.synchronize()
ref = ofb.featurize(w[:2], method_args=dict(sr=16000, n_mels=80))
print('max-abs err vs oracle', float(np.max(np.abs(out[:2].cpu().numpy() - ref))))
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in evs:
    a.record(); fz(wav, want_bf16=True); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
byts = 256 * (48000 * 4 + 298 * 80 * 4)
print(f'fbank+cmn: {ms * 1e3:.1f} us  {byts / ms / 1e6:.1f} GB/s algorithmic ({byts / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)')
