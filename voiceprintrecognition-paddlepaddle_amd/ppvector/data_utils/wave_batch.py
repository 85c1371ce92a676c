"""Waveform batch assembly on the GPU -- the work PPVectorDataset.__getitem__ does per utterance on CPU workers in front
of the featurizer (ppvector/data_utils/reader.py:97-101: decibel normalisation, crop to max_duration) plus the zero
padding of predict_batch (ppvector/predict.py:246-254), as one launch over the whole batch (csrc/augment.hip
wave_batch_kernel).  Decoding / resampling stay on the host; this takes the decoded utterances already on the device."""
import torch

from ppvector import _native as N


def assemble_waves(waves, max_len=None, starts=None, use_dB_normalization=True, target_dB=-20.0, gains_dB=None, with_valid=False):
    """waves: list of 1-D float GPU tensors (ragged).  Returns (batch (B, L) f32, input_lens_ratio (B,) f32) -- what
    AudioFeaturizer.forward(waveforms, input_lens_ratio) takes.  L = max_len or the longest utterance after its crop start.
    starts: per-utterance crop start in samples (the training-mode random crop; None = 0).  gains_dB: per-utterance gain when
    normalisation is off (the volume perturbation's draw).  with_valid adds the kept sample counts (B,) int32."""
    waves = [torch.as_tensor(w) for w in waves]
    if not waves or not all(w.is_cuda for w in waves):
        raise N.VpmiError('assemble_waves packs GPU waveforms: the engine has no CPU fallback')
    waves = [w.reshape(-1).contiguous().float() for w in waves]
    dev = waves[0].device
    B = len(waves)
    lens = [int(w.numel()) for w in waves]
    st = [0] * B if starts is None else [min(max(int(s), 0), n) for s, n in zip(starts, lens)]
    L = int(max_len) if max_len else max(n - s for n, s in zip(lens, st))
    if L <= 0:
        raise ValueError('assemble_waves: empty batch row length')
    out = torch.empty((B, L), dtype=torch.float32, device=dev)
    meta = torch.tensor([[w.data_ptr() for w in waves], lens, st], dtype=torch.int64)
    ptrs = meta[0].to(dev)
    lens_d, st_d = meta[1].to(torch.int32).to(dev), meta[2].to(torch.int32).to(dev)
    nv = torch.empty((B,), dtype=torch.int32, device=dev)
    g = None if gains_dB is None else torch.as_tensor(gains_dB, dtype=torch.float32).to(dev).contiguous()
    ctx = N.ctx(dev)
    N.check(N.lib().vp_wave_batch_f32(ctx, ptrs.data_ptr(), lens_d.data_ptr(), st_d.data_ptr(), B, L, int(bool(use_dB_normalization)),
                                      float(target_dB), None if g is None else g.data_ptr(), out.data_ptr(), nv.data_ptr(),
                                      N.stream_ptr()), ctx)
    ratio = nv.float() / float(L)
    return (out, ratio, nv) if with_valid else (out, ratio)


SPEEDS = (1.0, 0.9, 1.1)          # yeaudio SpeedPerturbAugmentor's rates; index = the class offset of speed_perturb_3_class


def speed_perturb(waves, rates):
    """Speed perturbation of a ragged batch on the GPU (SpeedPerturbAugmentor / AudioSegment.change_speed, reader.py:155-156):
    utterance b is resampled by linear interpolation to int(len / rates[b]) samples; rate 1.0 passes through untouched.
    waves: list of 1-D float GPU tensors.  Returns a list of the same length."""
    waves = [torch.as_tensor(w) for w in waves]
    if not waves or not all(w.is_cuda for w in waves):
        raise N.VpmiError('speed_perturb takes GPU waveforms: the engine has no CPU fallback')
    idx = [b for b, r in enumerate(rates) if float(r) != 1.0]
    out = list(waves)
    if not idx:
        return out
    src = [waves[b].reshape(-1).contiguous().float() for b in idx]
    lens = [int(w.numel()) for w in src]
    new_lens = [int(n / float(rates[b])) for n, b in zip(lens, idx)]
    if min(new_lens) <= 0:
        raise ValueError('speed_perturb: an utterance would become empty')
    dev = src[0].device
    dst = [torch.empty(m, dtype=torch.float32, device=dev) for m in new_lens]
    meta = torch.tensor([[w.data_ptr() for w in src], [w.data_ptr() for w in dst], lens, new_lens], dtype=torch.int64)
    sp, dp = meta[0].to(dev), meta[1].to(dev)
    ld, nd = meta[2].to(torch.int32).to(dev), meta[3].to(torch.int32).to(dev)
    ctx = N.ctx(dev)
    N.check(N.lib().vp_speed_perturb_f32(ctx, sp.data_ptr(), ld.data_ptr(), nd.data_ptr(), dp.data_ptr(), len(idx), max(new_lens),
                                         N.stream_ptr()), ctx)
    for b, w in zip(idx, dst):
        out[b] = w
    return out
