"""Oracle (CPU, NumPy): trial scoring and EER / minDCF.  TEST INFRASTRUCTURE ONLY.

Follows ppvector/trainer.py:416-431 (per-trial cosine against every enrol embedding, label =
same speaker) and ppvector/metric/metrics.py:4-37 (fnr/fpr by sorted cumulative sums, EER by
linear interpolation at the crossing, minDCF with p_target 0.01), predict.py:282 (cosine).
"""
import numpy as np


def cosine_matrix(trials, enroll):
    t = trials / np.linalg.norm(trials, axis=1, keepdims=True)
    e = enroll / np.linalg.norm(enroll, axis=1, keepdims=True)
    return t @ e.T


def trial_scores(trials, trial_labels, enroll, enroll_labels):
    """trainer.py:416-423: all (trial, enrol) pairs, row-major over trials."""
    s = cosine_matrix(np.asarray(trials, np.float64), np.asarray(enroll, np.float64))
    y = (np.asarray(trial_labels)[:, None] == np.asarray(enroll_labels)[None, :])
    return s.reshape(-1), y.reshape(-1).astype(np.int64)


def fnr_fpr(scores, labels):
    order = np.argsort(scores)
    thr = scores[order]
    lab = labels[order]
    tgt = (lab == 1).astype('f8')
    imp = (lab == 0).astype('f8')
    fnr = np.cumsum(tgt) / tgt.sum()
    fpr = 1.0 - np.cumsum(imp) / imp.sum()
    return fnr, fpr, thr


def eer(fnr, fpr, scores=None):
    d = fnr - fpr
    i1 = np.flatnonzero(d >= 0)[0]
    i2 = np.flatnonzero(d < 0)[-1]
    a = (fnr[i1] - fpr[i1]) / (fpr[i2] - fpr[i1] - (fnr[i2] - fnr[i1]))
    val = fnr[i1] + a * (fnr[i2] - fnr[i1])
    if scores is not None:
        return val, np.sort(scores)[i1]
    return val


def min_dcf(fnr, fpr, p_target=0.01, c_miss=1, c_fa=1):
    det = np.min(c_miss * fnr * p_target + c_fa * fpr * (1 - p_target))
    return det / min(c_miss * p_target, c_fa * (1 - p_target))
