"""CPU self-consistency pins of the front-end restatements the GPU tests are judged against (oracle/augment.py, oracle/fbank.py):
third-party algorithms (yeaudio, paddle.audio) restated from their published behaviour -- unpinned against the binaries, so every
identity that does not need them is checked here."""
import math

import numpy as np

from oracle import augment as oa
from oracle import fbank as ofb


def test_wave_batch_normalises_whole_utterance_then_crops_and_pads():
    rng = np.random.RandomState(0)
    waves = [rng.standard_normal(n).astype(np.float32) * s for n, s in ((5000, 0.3), (1200, 0.01), (800, 2.0))]
    out, nv = oa.wave_batch(waves, L=2000, starts=[1000, 0, 100], normalize=True, target_db=-20.0)
    assert out.shape == (3, 2000) and list(nv) == [2000, 1200, 700]
    for b, (w, st) in enumerate(zip(waves, (1000, 0, 100))):
        gain = 10.0 ** ((-20.0 - 10.0 * np.log10(np.mean(w.astype(np.float64) ** 2))) / 20.0)
        assert np.allclose(out[b, :nv[b]], (w[st:st + nv[b]] * gain), rtol=1e-6, atol=1e-9) and not out[b, nv[b]:].any()
        full = w.astype(np.float64) * gain                          # the WHOLE utterance sits at the target level, not the crop
        assert abs(10 * np.log10(np.mean(full ** 2)) + 20.0) < 1e-9
    raw, _ = oa.wave_batch(waves, L=800, normalize=False, gains_db=[6.0, 0.0, -6.0])
    assert np.allclose(raw[0], waves[0][:800] * 10 ** 0.3, rtol=1e-6) and np.array_equal(raw[1], waves[1][:800])
    z, nz = oa.wave_batch([np.zeros(10, np.float32)], L=16)
    assert not z.any() and nz[0] == 10 and np.isfinite(z).all()


def test_change_speed_is_linear_interpolation_onto_the_new_length():
    x = np.asarray([0.0, 1.0, 4.0, 9.0, 16.0, 25.0, 36.0, 49.0, 64.0, 81.0], np.float32)
    assert np.array_equal(oa.change_speed(x, 1.0), x)
    for rate in (0.9, 1.1, 0.5, 2.0):
        y = oa.change_speed(x, rate)
        m = int(len(x) / rate)
        assert y.shape == (m,) and y[0] == x[0]
        for i in range(m):                                        # the same numbers from a scalar loop
            p = i * len(x) / (m - 1)
            j = int(p)
            want = x[-1] if j >= len(x) - 1 else x[j] + (x[j + 1] - x[j]) * (p - j)
            assert abs(float(y[i]) - float(want)) < 1e-5
    t = np.arange(16000) / 16000.0
    tone = np.sin(2 * math.pi * 440.0 * t).astype(np.float32)
    fast = oa.change_speed(tone, 1.1)                               # played at the same rate: pitch up by 10 %
    spec = np.abs(np.fft.rfft(fast * np.hanning(len(fast))))
    assert abs(np.argmax(spec) * 16000.0 / len(fast) - 440.0 * 1.1) < 2.0


def test_log_mel_and_mfcc_relations():
    w = ofb.synth_waves(1, 6000, seed=3)[0]
    args = dict(sr=16000, n_fft=512, hop_length=160, n_mels=40, f_min=20.0)
    mel = ofb.mel_spectrogram(w, dtype=np.float64, **args)
    lm = ofb.log_mel_spectrogram(w, dtype=np.float64, **args)
    assert mel.shape == lm.shape == (1 + 6000 // 160, 40)
    assert np.allclose(lm, 10.0 * np.log10(np.maximum(mel, 1e-10)), atol=1e-9)
    ref = ofb.log_mel_spectrogram(w, dtype=np.float64, ref_value=4.0, **args)
    assert np.allclose(lm - ref, 10.0 * np.log10(4.0), atol=1e-9)
    assert ofb.log_mel_spectrogram(np.zeros(2000, np.float32), **args).max() == -100.0            # the amin floor
    full = ofb.mfcc(w, dtype=np.float64, n_mfcc=40, **args)          # n_mfcc == n_mels: the DCT is orthonormal
    assert np.allclose(np.sum(full ** 2, axis=1), np.sum(lm ** 2, axis=1), rtol=1e-9)
    assert np.allclose(ofb.mfcc(w, dtype=np.float64, n_mfcc=13, **args), full[:, :13], atol=1e-9)
    assert np.allclose(full[:, 0], lm.sum(axis=1) / math.sqrt(40.0), atol=1e-9)                   # coefficient 0 = scaled sum
    # paddle's default hop is win_length // 4
    assert ofb.mel_spectrogram(w, sr=16000, n_fft=512, n_mels=40).shape[0] == 1 + 6000 // 128
    assert ofb.featurize_mel(w, method_args=dict(sr=16000), log='mfcc').shape == (1, 1 + 6000 // 128, 40)
