"""Oracle (CPU, NumPy): SpecAugment and batch collation.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

SpecAugment -- PARITY UNPINNED.  The reference calls yeaudio.augmentation.SpecAugmentor(**aug_conf.spec_aug)
(ppvector/data_utils/reader.py:105-107,150-151; configs/augmentation.yml:36-48: prob 0.5, freq_mask_ratio 0.1,
n_freq_masks 1, time_mask_ratio 0.05, n_time_masks 1, max_time_warp 0).  yeaudio (requirements.txt:12, >= 0.0.6) is
not vendored and not installed, and the reference holds no test vectors for it, so this is a restatement of its
published algorithm [3P-memory]: with probability `prob`: (time warp when max_time_warp > 0 -- not built), then
n_freq_masks frequency masks, then n_time_masks time masks; a mask of width w = int(U(0, int(ratio * size))) starts
at int(U(0, size - w)) and is filled with the CURRENT mean of the (T, F) feature (or zero when replace_with_zero).
Draws come from Python's `random` in exactly this order, so a host that makes the same calls gets the same masks.

collate_fn -- ppvector/data_utils/collate_fn.py:5-23: zero-pad to the longest feature; labels / lengths as int64.
"""
import random

import numpy as np


def draw_masks(n_frames, n_bins, prob=0.5, freq_mask_ratio=0.1, n_freq_masks=1, time_mask_ratio=0.05, n_time_masks=1,
               rng=random):
    """-> (apply, [(f0, fw)] * n_freq_masks, [(t0, tw)] * n_time_masks) for ONE utterance."""
    if rng.random() > prob:
        return False, [(0, 0)] * n_freq_masks, [(0, 0)] * n_time_masks
    fm, tm = [], []
    fmax = int(freq_mask_ratio * n_bins)
    for _ in range(n_freq_masks):
        f = int(rng.uniform(0, fmax))
        f0 = int(rng.uniform(0, n_bins - f))
        fm.append((f0, f))
    tmax = int(time_mask_ratio * n_frames)
    for _ in range(n_time_masks):
        t = int(rng.uniform(0, tmax))
        t0 = int(rng.uniform(0, n_frames - t))
        tm.append((t0, t))
    return True, fm, tm


def apply_masks(x, fm, tm, replace_with_zero=False):
    """x (T, F) float32 -> masked copy; each mask takes the mean of the tensor as it is at that moment."""
    x = np.array(x, dtype=np.float32, copy=True)
    for f0, f in fm:
        if f > 0:
            x[:, f0:f0 + f] = 0.0 if replace_with_zero else x.mean(dtype=np.float64)
    for t0, t in tm:
        if t > 0:
            x[t0:t0 + t, :] = 0.0 if replace_with_zero else x.mean(dtype=np.float64)
    return x


def spec_augment(x, rng=random, **conf):
    conf = {k: v for k, v in conf.items() if k != 'max_time_warp'}
    rz = conf.pop('replace_with_zero', False)
    ok, fm, tm = draw_masks(x.shape[0], x.shape[1], rng=rng, **conf)
    return apply_masks(x, fm, tm, rz) if ok else np.array(x, dtype=np.float32, copy=True)


def collate(batch):
    """[(feature (T_i, F), label)] -> (features (B, Tmax, F) f32, labels int64, input_lens int64)."""
    tmax = max(f.shape[0] for f, _ in batch)
    out = np.zeros((len(batch), tmax, batch[0][0].shape[1]), np.float32)
    for i, (f, _) in enumerate(batch):
        out[i, :f.shape[0]] = f
    return out, np.asarray([int(l) for _, l in batch], np.int64), np.asarray([f.shape[0] for f, _ in batch], np.int64)


def wave_batch(waves, L=None, starts=None, normalize=True, target_db=-20.0, gains_db=None):
    """The waveform side of batch assembly, per utterance as the reference's workers do it (data_utils/reader.py:97-101):
    AudioSegment.normalize(target_db) over the whole utterance (yeaudio, third party: gain_dB = target_db - 10 log10(mean x^2),
    samples *= 10^(gain_dB / 20); restated from its published behaviour, silent input keeps gain 1), crop at starts[b], then
    predict_batch's zero padding (predict.py:246-254).  Returns (batch (B, L) f32, n_valid (B,) int32)."""
    L = L or max(len(w) for w in waves)
    out = np.zeros((len(waves), L), np.float32)
    nv = np.zeros(len(waves), np.int32)
    for b, w in enumerate(waves):
        w = np.asarray(w, np.float64)
        gain = 1.0
        if normalize:
            ms = float(np.mean(w ** 2)) if len(w) else 0.0
            if ms > 0:
                gain = 10.0 ** ((target_db - 10.0 * np.log10(ms)) / 20.0)
        elif gains_db is not None:
            gain = 10.0 ** (float(gains_db[b]) / 20.0)
        st = min(max(int(starts[b]) if starts is not None else 0, 0), len(w))
        n = min(L, len(w) - st)
        out[b, :n] = (w[st:st + n] * gain).astype(np.float32)
        nv[b] = n
    return out, nv


def change_speed(samples, speed_rate):
    """yeaudio AudioSegment.change_speed (third party, restated from its published behaviour; call site reader.py:155-156 via
    SpeedPerturbAugmentor): linear-interpolation resampling to int(len / rate) points spread over [0, len]."""
    if speed_rate == 1.0:
        return np.asarray(samples, np.float32)
    old_length = len(samples)
    new_length = int(old_length / speed_rate)
    new_indices = np.linspace(start=0, stop=old_length, num=new_length)
    return np.interp(new_indices, np.arange(old_length), samples).astype(np.float32)
