import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/voiceprintrecognition-paddlepaddle_amd')
import numpy as np, torch
from oracle import fbank as ofb
from ppvector.data_utils.featurizer import AudioFeaturizer
np.set_printoptions(precision=3, suppress=True, linewidth=250)
w = ofb.synth_waves(1, 16000, seed=3)
ref = ofb.featurize(w, method_args=dict(sr=16000, n_mels=80))[0]
fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
got = fz(torch.from_numpy(w).cuda()).cpu().numpy()[0]
err = got - ref
print('T', ref.shape)
print('per-frame max err (t=0..47):'); print(np.abs(err).max(1)[:48])
print('per-mel max err:'); print(np.abs(err).max(0))
print('err[t=0..7, m=0..9]'); print(err[:8, :10])
print('err[t=0..7, m=70..79]'); print(err[:8, 70:])
