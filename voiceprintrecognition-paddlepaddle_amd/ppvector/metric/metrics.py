"""EER / minDCF on the host (NumPy), semantics of ppvector/metric/metrics.py:4-37, plus the GPU
trial scorer that replaces the per-trial sklearn loop of trainer.py:416-423.
"""
import numpy as np


def compute_fnr_fpr(scores, labels, weights=None):
    """False-negative / false-positive rates at every score threshold (ascending)."""
    order = np.argsort(scores)
    thresholds = scores[order]
    labels = labels[order]
    weights = np.ones(labels.shape, dtype='f8') if weights is None else weights[order]
    tgt = weights * (labels == 1).astype('f8')
    imp = weights * (labels == 0).astype('f8')
    fnr = np.cumsum(tgt) / np.sum(tgt)
    fpr = 1 - np.cumsum(imp) / np.sum(imp)
    return fnr, fpr, thresholds


def compute_eer(fnr, fpr, scores=None):
    """Equal error rate by linear interpolation between the two points around fnr == fpr."""
    gap = fnr - fpr
    hi = np.flatnonzero(gap >= 0)[0]
    lo = np.flatnonzero(gap < 0)[-1]
    a = (fnr[hi] - fpr[hi]) / (fpr[lo] - fpr[hi] - (fnr[lo] - fnr[hi]))
    rate = fnr[hi] + a * (fnr[lo] - fnr[hi])
    if scores is not None:
        return rate, np.sort(scores)[hi]
    return rate


def compute_dcf(fnr, fpr, p_target=0.01, c_miss=1, c_fa=1):
    """Minimum normalised detection cost."""
    c_det = min(c_miss * fnr * p_target + c_fa * fpr * (1 - p_target))
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return c_det / c_def


def cosine_score_matrix(trials, enroll):
    """(Nt, D), (Ne, D) GPU tensors -> (Nt, Ne) cosine scores on the engine (csrc/head.hip)."""
    import torch
    from ppvector import _native as N
    if not trials.is_cuda:
        raise N.VpmiError('cosine_score_matrix needs GPU tensors: the engine has no CPU fallback')
    a = trials.contiguous().float()
    b = enroll.to(a.device).contiguous().float()
    lib, ctx = N.lib(), N.ctx(a.device)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    ws = torch.empty(max(lib.vp_cosine_scores_workspace_bytes(a.shape[0], b.shape[0], a.shape[1]), 256),
                     dtype=torch.uint8, device=a.device)
    N.check(lib.vp_cosine_scores_f32(ctx, a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], a.shape[1],
                                     out.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
    return out


def evaluate_trials(enroll_features, enroll_labels, trials_features, trials_labels):
    """The scoring half of PPVectorTrainer.evaluate (ppvector/trainer.py:412-431): every trial embedding against every enrolled
    one (one cosine GEMM on the GPU instead of a Python loop of sklearn calls), target / non-target labels in the same
    trial-major order, then the reference's fnr/fpr -> EER / minDCF arithmetic on the host.  Returns (eer, min_dcf, threshold)."""
    import numpy as np
    import torch
    scores = cosine_score_matrix(torch.as_tensor(trials_features), torch.as_tensor(enroll_features)).cpu().numpy()
    el = np.asarray(enroll_labels).astype(np.int32)
    tl = np.asarray(trials_labels).astype(np.int32)
    all_score = scores.astype(np.float32).reshape(-1)
    all_labels = (el[None, :] == tl[:, None]).astype(np.int32).reshape(-1)
    fnr, fpr, thresholds = compute_fnr_fpr(all_score, all_labels)
    eer, threshold = compute_eer(fnr, fpr, all_score)
    min_dcf = compute_dcf(fnr, fpr)
    return float(eer), float(min_dcf), float(threshold)
