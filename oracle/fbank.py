"""Oracle (CPU, NumPy): Kaldi Fbank + CMN feature pipeline.  TEST INFRASTRUCTURE ONLY.

Follows
  * ppvector/data_utils/featurizer.py:83-101  (KaldiFbank: per-utterance loop calling
    ``paddleaudio.compliance.kaldi.fbank(waveform, **kwargs)``, transposed and stacked),
  * ppvector/data_utils/featurizer.py:33-60   (AudioFeaturizer.forward: transpose to (B,T,F),
    subtract the time mean over the padded length, optional ratio mask),
  * configs/ecapa_tdnn.yml:46-48              (method_args: sr 16000, n_mels 80).

``paddleaudio.compliance.kaldi.fbank`` (paddleaudio >= 1.0.1, requirements.txt:8) is NOT in
/root/reference; its published algorithm (a port of Kaldi ``compute-fbank-feats`` /
torchaudio.compliance.kaldi.fbank) is restated here with its defaults:
frame_length 25 ms, frame_shift 10 ms, snip_edges, dither 0, remove_dc_offset, preemphasis 0.97
(first sample replicated), povey window (symmetric hann ** 0.85), round_to_power_of_two (512),
power spectrum, low_freq 20, high_freq 0 (-> Nyquist), Kaldi mel scale 1127*ln(1+f/700),
triangular filters over the first n_fft/2 bins (Nyquist bin weight 0), log(max(e, eps)),
use_energy False, subtract_mean False.  eps: paddleaudio uses 1e-7 [3P-memory]
(torchaudio: float32 machine epsilon 1.19e-7); it only matters for (near-)silent frames.
Input is float in [-1, 1] (yeaudio samples), not int16-scaled.  Parity unpinned against the
paddleaudio binary; pinned against an independent float64 DFT derivation and the
``transformers.audio_utils`` Kaldi-mel helpers in tests/test_oracle_fbank.py.
"""
import math

import numpy as np

FBANK_DEFAULTS = dict(
    sr=16000, n_mels=23, frame_length=25.0, frame_shift=10.0, preemphasis_coefficient=0.97,
    remove_dc_offset=True, low_freq=20.0, high_freq=0.0, window_type='povey',
    round_to_power_of_two=True, snip_edges=True, use_power=True, use_log_fbank=True,
    dither=0.0, eps=1e-7)


def next_pow2(n):
    return 1 if n == 0 else 2 ** (n - 1).bit_length()


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def frame_geometry(sr=16000, frame_length=25.0, frame_shift=10.0, round_to_power_of_two=True):
    win = int(sr * frame_length * 0.001)
    shift = int(sr * frame_shift * 0.001)
    nfft = next_pow2(win) if round_to_power_of_two else win
    return win, shift, nfft


def num_frames(n_samples, win, shift):
    """snip_edges=True frame count (Kaldi NumFrames)."""
    if n_samples < win:
        return 0
    return 1 + (n_samples - win) // shift


def povey_window(win, dtype=np.float64):
    n = np.arange(win, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * math.pi * n / (win - 1))     # symmetric (periodic=False)
    return (hann ** 0.85).astype(dtype)


def mel_banks(n_mels, nfft, sr, low_freq=20.0, high_freq=0.0, dtype=np.float64):
    """(n_mels, nfft//2 + 1) triangular weights; the Nyquist column is zero (Kaldi/torchaudio
    build nfft//2 columns and right-pad one zero column)."""
    nbins = nfft // 2
    nyq = 0.5 * sr
    if high_freq <= 0.0:
        high_freq += nyq
    bin_w = sr / nfft
    mlo, mhi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mhi - mlo) / (n_mels + 1)
    b = np.arange(n_mels, dtype=np.float64)[:, None]
    left = mlo + b * delta
    center = left + delta
    right = left + 2.0 * delta
    mel = mel_scale(bin_w * np.arange(nbins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    w = np.concatenate([w, np.zeros((n_mels, 1))], axis=1)
    return w.astype(dtype)


def kaldi_fbank(wave, dtype=np.float32, **kwargs):
    """One utterance: wave (L,) float -> (T, n_mels) log-Mel filterbank energies.

    dtype float32 mirrors the reference's arithmetic type; float64 is the high-precision
    derivation used to bound the float32 error in tests.
    """
    o = dict(FBANK_DEFAULTS)
    o.update(kwargs)
    wave = np.asarray(wave, dtype=dtype).reshape(-1)
    win, shift, nfft = frame_geometry(o['sr'], o['frame_length'], o['frame_shift'],
                                      o['round_to_power_of_two'])
    T = num_frames(wave.shape[0], win, shift)
    if T == 0:
        return np.zeros((0, o['n_mels']), dtype=dtype)
    idx = np.arange(T)[:, None] * shift + np.arange(win)[None, :]
    fr = wave[idx].astype(dtype)                                  # (T, win) strided frames
    if o['remove_dc_offset']:
        fr = fr - fr.mean(axis=1, keepdims=True, dtype=dtype)
    pc = o['preemphasis_coefficient']
    if pc != 0.0:
        prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)    # replicate first sample
        fr = fr - dtype(pc) * prev
    fr = fr * povey_window(win, dtype)[None, :]
    if nfft != win:
        fr = np.concatenate([fr, np.zeros((T, nfft - win), dtype=dtype)], axis=1)
    spec = np.fft.rfft(fr.astype(np.float64) if dtype == np.float64 else fr, axis=1)
    if dtype == np.float32:
        spec = spec.astype(np.complex64)
    mag = np.abs(spec).astype(dtype)
    pw = mag * mag if o['use_power'] else mag
    banks = mel_banks(o['n_mels'], nfft, o['sr'], o['low_freq'], o['high_freq'], dtype)
    e = pw @ banks.T
    if o['use_log_fbank']:
        e = np.log(np.maximum(e, dtype(o['eps'])))
    return e.astype(dtype)


def featurize(waves, input_lens_ratio=None, feature_method='Fbank', method_args=None,
              dtype=np.float32):
    """AudioFeaturizer.forward (featurizer.py:33-60): (B, L) -> (B, T, F), CMN over the padded
    T, optional mask ``t < int32(ratio * T)`` (truncation, featurizer.py:51-59)."""
    if feature_method != 'Fbank':
        raise Exception(f'预处理方法 {feature_method} 不存在!')
    method_args = dict(method_args or {})
    waves = np.asarray(waves, dtype=dtype)
    if waves.ndim == 1:
        waves = waves[None, :]
    feats = np.stack([kaldi_fbank(w, dtype=dtype, **method_args) for w in waves])   # (B,T,F)
    feats = feats - feats.mean(axis=1, keepdims=True, dtype=dtype)
    if input_lens_ratio is not None:
        T = feats.shape[1]
        ratio = np.asarray(input_lens_ratio, dtype=np.float32)
        lens = (ratio * np.float32(T)).astype(np.int32)
        mask = np.arange(T)[None, :] < lens[:, None]
        feats = np.where(mask[:, :, None], feats, np.zeros_like(feats))
    return feats.astype(dtype)


def feature_dim(feature_method, method_args):
    """AudioFeaturizer.feature_dim (featurizer.py:63-80)."""
    defaults = {'LogMelSpectrogram': ('n_mels', 128), 'MelSpectrogram': ('n_mels', 64),
                'MFCC': ('n_mfcc', 40), 'Fbank': ('n_mels', 23)}
    if feature_method == 'Spectrogram':
        return method_args.get('n_fft', 512) // 2 + 1
    if feature_method not in defaults:
        raise Exception('没有{}预处理方法'.format(feature_method))
    k, d = defaults[feature_method]
    return method_args.get(k, d)


def synth_waves(batch, n_samples=48000, seed=1000, lowpass=0.0):
    """SURVEY.md section 8(d) synthetic input: Gaussian noise scaled per utterance to -20 dBFS
    RMS (sigma = 0.1), clipped to [-1, 1]; optional one-pole low-pass for a speech-like tilt."""
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((batch, n_samples)).astype(np.float32)
    if lowpass > 0.0:
        y = np.empty_like(x)
        acc = np.zeros(batch, dtype=np.float32)
        for i in range(n_samples):
            acc = lowpass * acc + (1.0 - lowpass) * x[:, i]
            y[:, i] = acc
        x = y
    rms = np.sqrt((x.astype(np.float64) ** 2).mean(axis=1, keepdims=True))
    x = (x * (0.1 / rms)).astype(np.float32)
    return np.clip(x, -1.0, 1.0)


# ----------------------------------------------------------------------------- MelSpectrogram
# paddle.audio.features.MelSpectrogram (call site ppvector/data_utils/featurizer.py:22-23) is inside the
# un-vendored paddle package; its published algorithm (librosa-compatible) is restated here [3P-memory]:
# STFT (hann window, periodic, win_length <= n_fft zero-padded symmetrically; center=True with reflect
# padding of n_fft//2; frames = 1 + L // hop) -> |X|**power -> Slaney-scale, Slaney-normalised mel bank
# (htk=False, norm='slaney') -> (n_mels, frames), linear power (no log).  Class defaults: sr 22050,
# n_fft 2048, hop_length 512, n_mels 64, f_min 50, f_max None (-> sr/2), power 2.0.
MEL_DEFAULTS = dict(sr=22050, n_fft=2048, hop_length=512, win_length=None, window='hann', power=2.0, center=True,
                    pad_mode='reflect', n_mels=64, f_min=50.0, f_max=None, htk=False, norm='slaney')


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_bank(sr, n_fft, n_mels, f_min, f_max, dtype=np.float64):
    """(n_mels, n_fft//2 + 1) weights, librosa.filters.mel(htk=False, norm='slaney')."""
    if f_max is None:
        f_max = sr / 2.0
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(f_min), hz_to_mel_slaney(f_max), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(dtype)


def mel_spectrogram(wave, dtype=np.float32, **kwargs):
    """One utterance: wave (L,) -> (frames, n_mels) linear mel power."""
    o = dict(MEL_DEFAULTS)
    o.update(kwargs)
    assert o['window'] == 'hann' and o['center'] and o['pad_mode'] == 'reflect' and not o['htk'] and o['norm'] == 'slaney'
    n_fft = int(o['n_fft'])
    win_length = int(o['win_length'] or n_fft)
    hop = int(o['hop_length']) if kwargs.get('hop_length') is not None else win_length // 4     # paddle's default rule
    x = np.asarray(wave, dtype=dtype).reshape(-1)
    xp = np.pad(x, (n_fft // 2, n_fft // 2), mode='reflect')
    T = 1 + x.shape[0] // hop
    n = np.arange(win_length, dtype=np.float64)
    win = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / win_length))          # periodic hann
    lpad = (n_fft - win_length) // 2
    win = np.pad(win, (lpad, n_fft - win_length - lpad)).astype(dtype)
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    fr = xp[idx] * win[None, :]
    spec = np.fft.rfft(fr.astype(np.float64) if dtype == np.float64 else fr, axis=1)
    mag = np.abs(spec).astype(dtype)
    pw = mag * mag if o['power'] == 2.0 else mag ** dtype(o['power'])
    bank = slaney_mel_bank(o['sr'], n_fft, o['n_mels'], o['f_min'], o['f_max'], dtype)
    return (pw @ bank.T).astype(dtype)


def log_mel_spectrogram(wave, dtype=np.float32, ref_value=1.0, amin=1e-10, top_db=None, **kwargs):
    """paddle.audio.features.LogMelSpectrogram (call site featurizer.py:20-21; third party, restated [3P-memory]): the mel
    spectrogram above with n_fft defaulting to 512, through power_to_db: 10 log10(max(x, amin)) - 10 log10(max(ref_value, amin))
    (top_db=None: no floor relative to the peak)."""
    assert top_db is None
    kwargs.setdefault('n_fft', 512)
    m = mel_spectrogram(wave, dtype=np.float64, **kwargs)
    return (10.0 * np.log10(np.maximum(m, amin)) - 10.0 * np.log10(max(ref_value, amin))).astype(dtype)


def mfcc(wave, dtype=np.float32, n_mfcc=40, **kwargs):
    """paddle.audio.features.MFCC (call site featurizer.py:26-27; third party, restated [3P-memory]): log-mel (above) times
    create_dct(n_mfcc, n_mels, norm='ortho') = cos(pi / n_mels (n + 0.5) k), row 0 scaled by 1/sqrt(2), all by sqrt(2 / n_mels)."""
    lm = log_mel_spectrogram(wave, dtype=np.float64, **kwargs)                       # (T, n_mels)
    n_mels = lm.shape[1]
    assert n_mfcc <= n_mels
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(math.pi / n_mels * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / n_mels)
    return (lm @ dct.T).astype(dtype)


def featurize_mel(waves, input_lens_ratio=None, method_args=None, dtype=np.float32, log=False):
    """AudioFeaturizer.forward with feature_method 'MelSpectrogram' / 'LogMelSpectrogram' (featurizer.py:20-23, :45-59)."""
    waves = np.asarray(waves, dtype=dtype)
    if waves.ndim == 1:
        waves = waves[None, :]
    fn = {False: mel_spectrogram, True: log_mel_spectrogram, 'mfcc': mfcc}[log]
    feats = np.stack([fn(w, dtype=dtype, **dict(method_args or {})) for w in waves])
    feats = feats - feats.mean(axis=1, keepdims=True, dtype=dtype)
    if input_lens_ratio is not None:
        T = feats.shape[1]
        lens = (np.asarray(input_lens_ratio, np.float32) * np.float32(T)).astype(np.int32)
        feats = np.where((np.arange(T)[None, :] < lens[:, None])[:, :, None], feats, np.zeros_like(feats))
    return feats.astype(dtype)
