"""CPU tests: pin the Fbank oracle (oracle/fbank.py) against independent derivations.

paddleaudio's kaldi.fbank is not in /root/reference (un-vendored, requirements.txt:8), so the
oracle is pinned against (i) a float64 direct-DFT derivation written from the Kaldi definition,
(ii) the Kaldi-mel helpers in transformers.audio_utils, (iii) structural facts the reference
states (feature_dim defaults featurizer.py:69-78; 298 frames for 3 s).
"""
import math

import numpy as np
import pytest

from oracle import fbank as ofb


def direct_fbank_f64(wave, n_mels=80, sr=16000):
    """Independent float64 derivation: explicit loops / DFT matrix, no shared helpers."""
    win, shift, nfft = 400, 160, 512
    T = 1 + (len(wave) - win) // shift
    n = np.arange(win)
    window = (0.5 - 0.5 * np.cos(2 * math.pi * n / (win - 1))) ** 0.85
    k = np.arange(nfft // 2 + 1)[:, None]
    dft = np.exp(-2j * math.pi * k * np.arange(nfft)[None, :] / nfft)
    # triangular filters in mel domain
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(sr / 2)
    d = (hi - lo) / (n_mels + 1)
    banks = np.zeros((n_mels, nfft // 2 + 1))
    for m in range(n_mels):
        l, c, r = lo + m * d, lo + (m + 1) * d, lo + (m + 2) * d
        for b in range(nfft // 2):
            x = mel(b * sr / nfft)
            if l < x < r:
                banks[m, b] = (x - l) / (c - l) if x <= c else (r - x) / (r - c)
    out = np.zeros((T, n_mels))
    for t in range(T):
        fr = np.asarray(wave[t * shift:t * shift + win], dtype=np.float64).copy()
        fr -= fr.mean()
        fr = np.concatenate([[fr[0] - 0.97 * fr[0]], fr[1:] - 0.97 * fr[:-1]])
        fr = fr * window
        fr = np.concatenate([fr, np.zeros(nfft - win)])
        p = np.abs(dft @ fr) ** 2
        out[t] = np.log(np.maximum(banks @ p, 1e-7))
    return out


def test_frame_geometry_and_dims():
    assert ofb.frame_geometry(16000) == (400, 160, 512)
    assert ofb.num_frames(48000, 400, 160) == 298
    assert ofb.num_frames(399, 400, 160) == 0
    assert ofb.num_frames(400, 400, 160) == 1
    assert ofb.feature_dim('Fbank', {}) == 23
    assert ofb.feature_dim('Fbank', {'n_mels': 80}) == 80
    assert ofb.feature_dim('MelSpectrogram', {}) == 64
    assert ofb.feature_dim('Spectrogram', {}) == 257
    with pytest.raises(Exception):
        ofb.feature_dim('Nope', {})


def test_oracle_matches_direct_f64():
    w = ofb.synth_waves(1, 4000, seed=3)[0]
    ref = direct_fbank_f64(w)
    o64 = ofb.kaldi_fbank(w, dtype=np.float64, sr=16000, n_mels=80)
    o32 = ofb.kaldi_fbank(w, dtype=np.float32, sr=16000, n_mels=80)
    assert o64.shape == ref.shape == (23, 80)
    assert np.max(np.abs(o64 - ref)) < 1e-9
    assert np.max(np.abs(o32 - ref)) < 2e-4           # float32 arithmetic vs float64


def test_oracle_matches_transformers_kaldi_mel():
    au = pytest.importorskip('transformers.audio_utils')
    w = ofb.synth_waves(1, 16000, seed=5, lowpass=0.9)[0]
    banks = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20.0,
                               max_frequency=8000.0, sampling_rate=16000, norm=None, mel_scale='kaldi',
                               triangularize_in_mel_space=True)                      # (257, 80)
    mine = ofb.mel_banks(80, 512, 16000).T
    assert np.max(np.abs(banks[:256] - mine[:256])) < 1e-6
    window = au.window_function(400, 'povey', periodic=False)
    assert np.max(np.abs(window - ofb.povey_window(400))) < 1e-7
    spec = au.spectrogram(w.astype(np.float64), window, frame_length=400, hop_length=160, fft_length=512,
                          power=2.0, center=False, preemphasis=0.97, mel_filters=banks, log_mel='log',
                          mel_floor=1e-7, remove_dc_offset=True, dtype=np.float64).T
    o64 = ofb.kaldi_fbank(w, dtype=np.float64, sr=16000, n_mels=80)
    assert spec.shape == o64.shape
    # the Nyquist bin carries weight in transformers' bank and none in Kaldi's: tolerance covers it
    assert np.max(np.abs(spec - o64)) < 5e-3
    assert np.mean(np.abs(spec - o64)) < 1e-4


def test_cmn_and_mask_semantics():
    w = ofb.synth_waves(3, 8000, seed=9)
    f = ofb.featurize(w, method_args=dict(sr=16000, n_mels=80))
    assert f.shape == (3, 48, 80)
    assert np.max(np.abs(f.mean(axis=1))) < 1e-4
    ratio = np.asarray([1.0, 0.5, 0.26], np.float32)
    fm = ofb.featurize(w, ratio, method_args=dict(sr=16000, n_mels=80))
    lens = (ratio * np.float32(48)).astype(np.int32)                # truncation: 48, 24, 12
    assert list(lens) == [48, 24, 12]
    for b in range(3):
        assert np.array_equal(fm[b, :lens[b]], f[b, :lens[b]])
        assert np.all(fm[b, lens[b]:] == 0)
    # 1-D input is unsqueezed (featurizer.py:43-44)
    assert ofb.featurize(w[0], method_args=dict(sr=16000, n_mels=80)).shape == (1, 48, 80)
    with pytest.raises(Exception):
        ofb.featurize(w, feature_method='Nope')


def test_silence_hits_log_floor():
    f = ofb.kaldi_fbank(np.zeros(1600, np.float32), sr=16000, n_mels=80)
    assert np.allclose(f, math.log(1e-7))


def test_melspectrogram_oracle_matches_transformers_librosa_semantics():
    au = pytest.importorskip('transformers.audio_utils')
    w = ofb.synth_waves(1, 16000, seed=21, lowpass=0.9)[0]
    args = dict(sr=16000, n_fft=1024, hop_length=320, win_length=1024, n_mels=64, f_min=50.0)
    mine = ofb.mel_spectrogram(w, dtype=np.float64, **args)
    assert mine.shape == (1 + 16000 // 320, 64)
    bank = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=64, min_frequency=50.0, max_frequency=8000.0,
                              sampling_rate=16000, norm='slaney', mel_scale='slaney')
    assert np.max(np.abs(bank.T - ofb.slaney_mel_bank(16000, 1024, 64, 50.0, None))) < 1e-9
    window = au.window_function(1024, 'hann', periodic=True)
    ref = au.spectrogram(w.astype(np.float64), window, frame_length=1024, hop_length=320, fft_length=1024, power=2.0,
                         center=True, pad_mode='reflect', mel_filters=bank, mel_floor=0.0, dtype=np.float64).T
    assert ref.shape == mine.shape
    assert np.max(np.abs(ref - mine)) < 1e-6 * max(1.0, np.max(np.abs(ref)))    # their STFT accumulates in complex64
    f = ofb.featurize_mel(w[None], method_args=args)
    assert f.shape == (1, 51, 64) and np.max(np.abs(f.mean(axis=1))) < 1e-5
    # win_length < n_fft: window is centred in the FFT frame
    m2 = ofb.mel_spectrogram(w, dtype=np.float64, sr=16000, n_fft=1024, hop_length=160, win_length=400, n_mels=80, f_min=20.0)
    assert m2.shape == (101, 80) and np.all(m2 >= 0)
