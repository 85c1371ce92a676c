"""SpecAugment on the MI355X engine.

Stands in for ``yeaudio.augmentation.SpecAugmentor`` as the reference constructs it
(ppvector/data_utils/reader.py:150-151: ``SpecAugmentor(**aug_conf.spec_aug)``; configs/augmentation.yml:36-48),
with two call forms:
  * ``aug(feature)`` on ONE (T, F) feature -- the reference's per-sample call (reader.py:105-107);
  * ``aug.batch(features)`` on a (B, T, F) GPU batch -- the same draws per utterance, one kernel for the batch.
Mask positions are drawn on the host with Python's ``random`` in the order documented in oracle/augment.py; the
masks are applied in place by csrc/augment.hip (vp_spec_augment).  Time warping (max_time_warp > 0) is not built.
"""
import random

import numpy as np
import torch

from ppvector import _native as N


class SpecAugmentor(object):
    def __init__(self, prob=0.5, freq_mask_ratio=0.15, n_freq_masks=2, time_mask_ratio=0.05, n_time_masks=2,
                 max_time_warp=0, replace_with_zero=False, inplace=True):
        if max_time_warp:
            raise NotImplementedError('time warping (max_time_warp > 0) is not built on the HIP engine; '
                                      'configs/augmentation.yml uses 0')
        self.prob = prob
        self.freq_mask_ratio, self.n_freq_masks = freq_mask_ratio, n_freq_masks
        self.time_mask_ratio, self.n_time_masks = time_mask_ratio, n_time_masks
        self.replace_with_zero = replace_with_zero
        self.inplace = inplace

    def _draw(self, n_frames, n_bins):
        fm = [(0, 0)] * self.n_freq_masks
        tm = [(0, 0)] * self.n_time_masks
        if random.random() > self.prob:
            return fm, tm
        fmax = int(self.freq_mask_ratio * n_bins)
        for i in range(self.n_freq_masks):
            f = int(random.uniform(0, fmax))
            fm[i] = (int(random.uniform(0, n_bins - f)), f)
        tmax = int(self.time_mask_ratio * n_frames)
        for i in range(self.n_time_masks):
            t = int(random.uniform(0, tmax))
            tm[i] = (int(random.uniform(0, n_frames - t)), t)
        return fm, tm

    def batch(self, features, lengths=None):
        """features (B, T, F) f32 / bf16 on the GPU, masked in place (a copy when inplace=False)."""
        if not features.is_cuda:
            raise N.VpmiError('SpecAugmentor needs a GPU tensor: the engine has no CPU fallback')
        x = features if self.inplace else features.clone()
        x = x.contiguous()
        B, T, F = x.shape
        fms, tms = [], []
        for b in range(B):
            fm, tm = self._draw(int(lengths[b]) if lengths is not None else T, F)
            fms.append(fm)
            tms.append(tm)
        fmask = torch.tensor(fms, dtype=torch.int32).reshape(B, self.n_freq_masks, 2).to(x.device)
        tmask = torch.tensor(tms, dtype=torch.int32).reshape(B, self.n_time_masks, 2).to(x.device)
        ctx = N.ctx(x.device)
        N.check(N.lib().vp_spec_augment(ctx, N.dtype_id(x.dtype), x.data_ptr(), B, T, F, fmask.data_ptr(), self.n_freq_masks,
                                        tmask.data_ptr(), self.n_time_masks, int(self.replace_with_zero), N.stream_ptr()), ctx)
        return x

    def __call__(self, x):
        """One (T, F) feature.  A NumPy array (the reference's call, reader.py:106) is uploaded, masked on the GPU
        and returned as NumPy; a GPU tensor stays on the GPU."""
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(N.default_device())
            return self.batch(t.unsqueeze(0))[0].cpu().numpy()
        return self.batch(torch.as_tensor(x).unsqueeze(0))[0]
