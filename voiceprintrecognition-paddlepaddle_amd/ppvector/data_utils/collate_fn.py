"""collate_fn (ppvector/data_utils/collate_fn.py:5-23): zero-pad a list of (feature (T_i, F), label) to the longest
feature; returns (features (B, Tmax, F) f32, labels int64, input_lens int64).  Features already on the GPU are packed
by one kernel (csrc/augment.hip: vp_pad_batch) instead of B slice assignments."""
import torch

from ppvector import _native as N


def collate_fn(batch):
    feats = [torch.as_tensor(f) for f, _ in batch]
    if not all(f.is_cuda for f in feats):
        raise N.VpmiError('collate_fn packs GPU features (AudioFeaturizer output): the engine has no CPU fallback')
    feats = [f.contiguous().float() for f in feats]
    B, F = len(feats), feats[0].shape[1]
    lens = [int(f.shape[0]) for f in feats]
    tmax = max(lens)
    dev = feats[0].device
    out = torch.empty((B, tmax, F), dtype=torch.float32, device=dev)
    ptrs = torch.tensor([f.data_ptr() for f in feats], dtype=torch.int64).to(dev)
    lens_d = torch.tensor(lens, dtype=torch.int32).to(dev)
    ctx = N.ctx(dev)
    N.check(N.lib().vp_pad_batch(ctx, N.VP_F32, ptrs.data_ptr(), lens_d.data_ptr(), B, tmax, F, out.data_ptr(), N.stream_ptr()), ctx)
    labels = torch.tensor([int(l) for _, l in batch], dtype=torch.int64, device=dev)
    input_lens = torch.tensor(lens, dtype=torch.int64, device=dev)
    return out, labels, input_lens
