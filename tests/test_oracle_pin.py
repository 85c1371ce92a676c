"""CPU tests: the oracle against the committed golden fixtures (tests/golden/*.npz).

The fixtures hold outputs of the REFERENCE's own model files (ppvector/models/ecapa_tdnn.py,
tdnn.py, fc.py, loss/aamloss.py) executed through oracle/paddle_shim by oracle/gen_golden.py.
Where /root/reference is present (build container) the generator's --check mode is re-run too.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import fbank as ofb
from oracle import models as om
from oracle import scoring as osc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_ecapa_oracle_matches_reference_golden(golden_dir):
    g = _load(golden_dir, 'ecapa_ref_small.npz')
    p = om.ecapa_params(input_size=80, seed=int(g['param_seed']))
    n_train, n_buf = om.count_params(p)
    assert (n_train, n_buf) == (int(g['n_train']), int(g['n_buf'])) == (6194048, 19328)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        emb = om.ecapa_forward(p, x).numpy()
        emb_tr = om.ecapa_forward(p, x, training=True).numpy()
    assert np.max(np.abs(emb - g['emb_eval'])) < 1e-4 * np.max(np.abs(g['emb_eval']))
    assert np.max(np.abs(emb_tr - g['emb_train'])) < 1e-3 * np.max(np.abs(g['emb_train']))
    W = om.head_params(192, 2796, seed=int(g['head_seed']))
    logits = om.cosine_head(torch.from_numpy(g['emb_eval']), W)
    assert np.max(np.abs(logits.numpy() - g['logits'])) < 1e-6
    for (margin, ls, easy), ref in zip(g['loss_cfg'], g['losses']):
        l = om.aam_loss(logits, torch.from_numpy(g['labels']), float(margin), 32.0, bool(easy), float(ls))
        assert abs(float(l) - ref) < 1e-4


def test_readme_parameter_inventory():
    # README.md:341-345 (paddle.summary at F=64, 9726 classes): total 8,039,808; non-trainable 19,328
    p = om.ecapa_params(input_size=64, seed=0)
    t, b = om.count_params(p)
    assert t + b + 192 * 9726 == 8039808
    assert b == 19328


def test_tdnn_oracle_matches_reference_golden(golden_dir):
    g = _load(golden_dir, 'tdnn_ref_small.npz')
    p = om.tdnn_params(input_size=80, seed=int(g['param_seed']))
    with torch.no_grad():
        emb = om.tdnn_forward(p, torch.from_numpy(g['x'])).numpy()
    assert np.max(np.abs(emb - g['emb_eval'])) < 1e-4 * np.max(np.abs(g['emb_eval']))


def test_2d_backbone_oracles_match_reference_golden(golden_dir):
    from oracle import campplus as oc, eres2net as oer, resnet_se as orse
    for name, params, fwd in (('campplus_ref_small.npz', oc.campplus_params, oc.campplus_forward),
                              ('resnetse_ref_small.npz', orse.resnetse_params, orse.resnetse_forward),
                              ('eres2net_ref_small.npz', oer.eres2net_params, oer.eres2net_forward),
                              ('eres2netv2_ref_small.npz', lambda f, e, seed: oer.eres2net_params(f, e, base_width=26, seed=seed, v2=True),
                               oer.eres2netv2_forward),
                              # BASELINE configs[4]: the 55 M-parameter ERes2Net (m_channels 64, expansion 4, base_width 24, scale 3)
                              ('eres2net_large_ref_small.npz',
                               lambda f, e, seed: oer.eres2net_params(f, e, seed=seed, m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3),
                               lambda p, x: oer.eres2net_forward(p, x, m_channels=64, expansion=4, base_width=24, scale=3))):
        g = _load(golden_dir, name)
        p = params(80, 192, seed=int(g['param_seed']))
        with torch.no_grad():
            emb = fwd(p, torch.from_numpy(g['x'])).numpy()
        assert np.max(np.abs(emb - g['emb_eval'])) < 1e-4 * np.max(np.abs(g['emb_eval'])), name


def test_real_speech_fixture(golden_dir):
    g = _load(golden_dir, 'wavs_3s.npz')
    wav = g['pcm'].astype(np.float32) / 32768.0
    feats = ofb.featurize(wav, method_args=dict(sr=16000, n_mels=80))
    assert feats.shape == (4, 298, 80)
    assert np.max(np.abs(feats - g['feats'])) < 1e-4
    p = om.ecapa_params(input_size=80, seed=1000)
    with torch.no_grad():
        emb = om.ecapa_forward(p, torch.from_numpy(g['feats'])).numpy()
    cos = osc.cosine_matrix(emb.astype(np.float64), emb.astype(np.float64))
    assert np.max(np.abs(cos - g['cos'])) < 1e-5


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference sources not on this box')
def test_regenerate_against_reference_sources():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'gen_golden.py'), '--check'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_margin_schedule_endpoints():
    spe, me = 10, 60
    assert om.margin_schedule(0, spe, me) == 0.0
    assert om.margin_schedule(int(me * 0.3) * spe - 1, spe, me) == 0.0
    assert om.margin_schedule(int(me * 0.7) * spe, spe, me) == 0.3
    mid = om.margin_schedule(int(me * 0.5) * spe, spe, me)
    assert 0.0 < mid < 0.3
    m = om.aam_margins(0.2)
    assert abs(m['th'] - np.cos(np.pi - 0.2)) < 1e-12 and abs(m['mmm'] - (1 + np.cos(np.pi - 0.2))) < 1e-12


def test_eer_known_answer():
    # perfectly separable scores -> EER 0; fully overlapping symmetric -> 0.5
    s = np.asarray([0.9, 0.8, 0.7, 0.2, 0.1, 0.0])
    y = np.asarray([1, 1, 1, 0, 0, 0])
    fnr, fpr, _ = osc.fnr_fpr(s, y)
    assert osc.eer(fnr, fpr) == pytest.approx(0.0, abs=1e-12)
    assert osc.min_dcf(fnr, fpr) == pytest.approx(0.0, abs=1e-12)
    s2 = np.asarray([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8])
    y2 = np.asarray([1, 0, 1, 0, 1, 0, 1, 0])
    fnr, fpr, _ = osc.fnr_fpr(s2, y2)
    e, thr = osc.eer(fnr, fpr, s2)
    assert 0.3 < e < 0.7
    sc, lab = osc.trial_scores(np.eye(3), [0, 1, 2], np.eye(3), [0, 1, 2])
    assert sc.shape == (9,) and lab.sum() == 3


def test_spec_augment_restatement_properties():
    """The SpecAugment restatement (parity unpinned, oracle/augment.py): mask geometry bounds of configs/augmentation.yml,
    prob gating, fill value = current mean, and collation."""
    import random
    from oracle import augment as oa
    conf = dict(prob=0.5, freq_mask_ratio=0.1, n_freq_masks=1, time_mask_ratio=0.05, n_time_masks=1)
    random.seed(0)
    applied = 0
    for _ in range(400):
        ok, fm, tm = oa.draw_masks(298, 80, **conf)
        applied += ok
        (f0, f), (t0, t) = fm[0], tm[0]
        assert 0 <= f < 8 and 0 <= f0 <= 80 - f and 0 <= t < 14 and 0 <= t0 <= 298 - t
    assert 150 < applied < 250
    x = np.arange(12, dtype=np.float32).reshape(4, 3)
    y = oa.apply_masks(x, [(1, 1)], [(2, 1)])
    m1 = x.mean()
    assert np.allclose(y[:, 1][[0, 1, 3]], m1)
    x1 = x.copy(); x1[:, 1] = m1
    assert np.allclose(y[2], x1.mean())
    assert np.array_equal(oa.apply_masks(x, [(0, 0)], [(0, 0)]), x)
    f, l, n = oa.collate([(np.ones((3, 2), np.float32), 5), (np.ones((1, 2), np.float32), 1)])
    assert f.shape == (2, 3, 2) and f[1, 1:].sum() == 0 and list(l) == [5, 1] and list(n) == [3, 1]


def _loss_cases():
    from oracle import losses as ol
    z = torch.zeros(())
    return {'AMLoss': (lambda l, y: ol.am_loss(l, y, 0.2, 30.0, 0.0), False),
            'AMLoss_ls': (lambda l, y: ol.am_loss(l, y, 0.35, 30.0, 0.1), False),
            'ARMLoss': (lambda l, y: ol.arm_loss(l, y, 0.2, 30.0, 0.0), False),
            'ARMLoss_ls': (lambda l, y: ol.arm_loss(l, y, 0.1, 20.0, 0.1), False),
            'CELoss': (lambda l, y: ol.ce_loss(l, y, 0.0), False),
            'CELoss_ls': (lambda l, y: ol.ce_loss(l, y, 0.2), False),
            'SubCenterLoss': (lambda l, y: ol.subcenter_loss(l, y, 0.2, 32.0, False, 3, 0.0), True),
            'SubCenterLoss_easy_ls': (lambda l, y: ol.subcenter_loss(l, y, 0.3, 32.0, True, 3, 0.1), True),
            'SphereFace2_C': (lambda l, y: ol.sphereface2_loss(l, y, z, 0.2, 32.0, 0.7, 3, 'C'), False),
            'SphereFace2_A': (lambda l, y: ol.sphereface2_loss(l, y, z, 0.15, 32.0, 0.7, 3, 'A'), False)}


def test_loss_family_oracle_matches_reference_golden(golden_dir):
    """loss/{amloss,armloss,celoss,subcenterloss,sphereface2}.py run through the shim: value and d loss / d logits."""
    g = _load(golden_dir, 'losses_ref.npz')
    y = torch.from_numpy(g['labels'])
    cases = _loss_cases()
    assert sorted(cases) == sorted(str(n) for n in g['names'])
    for name, (fn, use_k) in cases.items():
        l = torch.from_numpy(g['logits_k' if use_k else 'logits']).clone().requires_grad_(True)
        loss = fn(l, y)
        grad, = torch.autograd.grad(loss, l)
        assert abs(float(loss.detach()) - float(g[name + '_loss'])) < 1e-5 * max(1.0, abs(float(g[name + '_loss']))), name
        assert np.max(np.abs(grad.numpy() - g[name + '_dlogits'])) < 1e-5 * max(1.0, np.max(np.abs(g[name + '_dlogits']))), name


def test_head_variants_oracle_matches_reference_golden(golden_dir):
    """fc.py:6-87 run through the shim: DenseLayer('batchnorm') blocks in front of the cosine head, 'Linear' output."""
    g = _load(golden_dir, 'head_variants_ref.npz')
    emb = torch.from_numpy(g['emb'])
    for tag, ctype, nb in (('cos_b2', 'Cosine', 2), ('lin_b0', 'Linear', 0), ('lin_b1', 'Linear', 1)):
        p = om.classifier_params(24, 9, ctype, 1, nb, 16, seed=1002)
        with torch.no_grad():
            ev = om.classifier_head(emb, p, ctype, nb).numpy()
            tr = om.classifier_head(emb, p, ctype, nb, training=True).numpy()
        assert np.max(np.abs(ev - g[tag + '_eval'])) < 1e-5 * max(1.0, np.max(np.abs(g[tag + '_eval']))), tag
        assert np.max(np.abs(tr - g[tag + '_train'])) < 1e-5 * max(1.0, np.max(np.abs(g[tag + '_train']))), tag
