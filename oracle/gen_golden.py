"""Generate / verify the golden fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY.

Runs ONLY where /root/reference exists (the build container).  It executes the reference's own
model source files (ppvector/models/{ecapa_tdnn,tdnn,pooling,utils,fc}.py, ppvector/loss/aamloss.py)
unmodified through ``oracle/paddle_shim`` with weights from ``oracle.models.*_params(seed)`` and

  1. asserts that the oracle restatement (oracle/models.py) reproduces the reference graph's
     outputs (eval AND train-mode BN) to float32 round-off, and
  2. freezes inputs + reference outputs as small .npz fixtures that travel to the GPU box.

Usage:  python oracle/gen_golden.py            # check + (re)write fixtures
        python oracle/gen_golden.py --check    # check only (used by tests/test_oracle_pin.py)
"""
import argparse
import importlib
import os
import sys
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference'


def read_wav_16k_mono(path, n=48000):
    w = wave.open(path)
    assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
    x = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    out = np.zeros(n, dtype=np.int16)
    m = min(n, x.shape[0])
    out[:m] = x[:m]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print('reference not present; nothing to do')
        return 0

    from oracle import paddle_shim
    from oracle import models as om
    from oracle import fbank as ofb
    paddle_shim.install()
    ref_ecapa = importlib.import_module('ppvector.models.ecapa_tdnn')
    ref_tdnn = importlib.import_module('ppvector.models.tdnn')
    ref_fc = importlib.import_module('ppvector.models.fc')
    ref_aam = importlib.import_module('ppvector.loss.aamloss')
    torch.manual_seed(0)
    out = {}
    worst = 0.0

    def cmp(name, a, b, tol):
        nonlocal worst
        a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
        b = b.detach().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
        err = float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))
        worst = max(worst, err)
        print(f'  {name:34s} rel-max-err {err:.3e}  (tol {tol:.0e})')
        assert err <= tol, f'{name}: oracle deviates from the reference graph ({err})'

    rng = np.random.RandomState(7)
    # ---------------- ECAPA-TDNN (configs/ecapa_tdnn.yml model_args), F=80
    F_, C_ = 80, 2796
    p = om.ecapa_params(input_size=F_, seed=1000)
    model = ref_ecapa.EcapaTdnn(input_size=F_, embd_dim=192, pooling_type='ASP',
                                channels=[512, 512, 512, 512, 1536])
    sd = model.state_dict()
    assert set(sd.keys()) == set(p.keys()), (set(sd.keys()) ^ set(p.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(p[k].shape), k
    model.load_state_dict(p)
    n_train, n_buf = om.count_params(p)
    print(f'ECAPA F=80 params: trainable {n_train}, buffers {n_buf}')
    p64 = om.ecapa_params(input_size=64, seed=1)
    t64, b64 = om.count_params(p64)
    # README.md:341-345: total 8,039,808 incl. classifier 192*9726; non-trainable 19,328
    assert t64 + b64 + 192 * 9726 == 8039808 and b64 == 19328, (t64, b64)
    x = rng.standard_normal((2, 64, F_)).astype(np.float32) * 3.0
    xt = paddle_shim.to_tensor(x)
    model.eval()
    with torch.no_grad():
        emb_ref = model(xt)
        emb_or = om.ecapa_forward(p, torch.from_numpy(x))
    cmp('ecapa eval emb', emb_or, emb_ref, 2e-5)
    model.train()
    with torch.no_grad():
        emb_ref_tr = model(xt)
        emb_or_tr = om.ecapa_forward(p, torch.from_numpy(x), training=True)
    cmp('ecapa train-BN emb', emb_or_tr, emb_ref_tr, 2e-4)
    model.load_state_dict(p)            # the train-mode forward moved the running stats: restore
    # head + AAM loss
    W = om.head_params(192, C_, seed=1001)
    head = ref_fc.SpeakerIdentification(input_dim=192, num_speakers=C_, classifier_type='Cosine')
    head.load_state_dict({'weight': W})
    labels = rng.randint(0, C_, size=(2,)).astype(np.int64)
    with torch.no_grad():
        o = head(emb_ref)
        logits_or = om.cosine_head(emb_ref.as_subclass(torch.Tensor), W)
    cmp('cosine head logits', logits_or, o['logits'], 1e-6)
    losses_ref = []
    for margin, ls, easy in ((0.2, 0.0, False), (0.0, 0.0, False), (0.3, 0.1, False), (0.2, 0.0, True)):
        crit = ref_aam.AAMLoss(margin=0.2, scale=32, easy_margin=easy, label_smoothing=ls)
        crit.update(margin=margin)
        with torch.no_grad():
            l_ref = crit(o, paddle_shim.to_tensor(labels))
            l_or = om.aam_loss(logits_or, torch.from_numpy(labels), margin, 32.0, easy, ls)
        cmp(f'aam loss m={margin} ls={ls} easy={easy}', l_or, l_ref, 1e-5)
        losses_ref.append(float(l_ref))
    out['ecapa_ref_small.npz'] = dict(
        x=x, labels=labels, emb_eval=emb_ref.numpy(), emb_train=emb_ref_tr.numpy(),
        logits=o['logits'].numpy(), losses=np.asarray(losses_ref, np.float64),
        loss_cfg=np.asarray([[0.2, 0.0, 0], [0.0, 0.0, 0], [0.3, 0.1, 0], [0.2, 0.0, 1]], np.float64),
        param_seed=np.int64(1000), head_seed=np.int64(1001), n_train=np.int64(n_train), n_buf=np.int64(n_buf))

    # ---------------- TDNN (configs/tdnn.yml), F=80
    pt = om.tdnn_params(input_size=F_, seed=1000)
    tm = ref_tdnn.TDNN(input_size=F_, channels=512, embd_dim=192, pooling_type='ASP')
    sdt = tm.state_dict()
    assert set(sdt.keys()) == set(pt.keys()), (set(sdt.keys()) ^ set(pt.keys()))
    tm.load_state_dict(pt)
    tm.eval()
    with torch.no_grad():
        e_ref = tm(xt)
        e_or = om.tdnn_forward(pt, torch.from_numpy(x))
    cmp('tdnn eval emb', e_or, e_ref, 2e-5)
    out['tdnn_ref_small.npz'] = dict(x=x, emb_eval=e_ref.numpy(), param_seed=np.int64(1000))

    # ---------------- CAM++ (configs/cam++.yml: embd_dim 192), F=80
    from oracle import campplus as oc
    ref_cam = importlib.import_module('ppvector.models.campplus')
    pc = oc.campplus_params(input_size=F_, embd_dim=192, seed=1000)
    cm = ref_cam.CAMPPlus(input_size=F_, embd_dim=192)
    sdc = cm.state_dict()
    assert set(sdc.keys()) == set(pc.keys()), sorted(set(sdc.keys()) ^ set(pc.keys()))[:10]
    for k in sdc:
        assert tuple(sdc[k].shape) == tuple(pc[k].shape), k
    cm.load_state_dict(pc)
    cm.eval()
    nt, nb_ = om.count_params(pc)
    print(f'CAM++ F=80 embd 192 params: trainable {nt}, buffers {nb_}')
    xc = rng.standard_normal((2, 230, F_)).astype(np.float32) * 3.0       # 230 frames -> 115 after stride 2: 2 segments
    with torch.no_grad():
        ec_ref = cm(paddle_shim.to_tensor(xc))
        ec_or = oc.campplus_forward(pc, torch.from_numpy(xc))
    cmp('cam++ eval emb', ec_or, ec_ref, 2e-5)
    out['campplus_ref_small.npz'] = dict(x=xc, emb_eval=ec_ref.numpy(), param_seed=np.int64(1000))

    # ---------------- ResNetSE (configs/resnet_se.yml), F=80
    from oracle import resnet_se as orn
    ref_rn = importlib.import_module('ppvector.models.resnet_se')
    pr = orn.resnetse_params(input_size=F_, embd_dim=192, seed=1000)
    rm = ref_rn.ResNetSE(input_size=F_, embd_dim=192, pooling_type='ASP')
    sdr = rm.state_dict()
    assert set(sdr.keys()) == set(pr.keys()), sorted(set(sdr.keys()) ^ set(pr.keys()))[:10]
    for k in sdr:
        assert tuple(sdr[k].shape) == tuple(pr[k].shape), k
    rm.load_state_dict(pr)
    rm.eval()
    nt, nb_ = om.count_params(pr)
    print(f'ResNetSE F=80 params: trainable {nt}, buffers {nb_}')
    xr = rng.standard_normal((2, 66, F_)).astype(np.float32) * 3.0
    with torch.no_grad():
        er_ref = rm(paddle_shim.to_tensor(xr))
        er_or = orn.resnetse_forward(pr, torch.from_numpy(xr))
    cmp('resnetse eval emb', er_or, er_ref, 2e-5)
    out['resnetse_ref_small.npz'] = dict(x=xr, emb_eval=er_ref.numpy(), param_seed=np.int64(1000))

    # ---------------- ERes2Net (configs/eres2net.yml: m_channels 32, embd 192), F=80
    from oracle import eres2net as oer
    ref_er = importlib.import_module('ppvector.models.eres2net')
    pe = oer.eres2net_params(input_size=F_, embd_dim=192, seed=1000)
    em = ref_er.ERes2Net(input_size=F_, embd_dim=192, m_channels=32)
    sde = em.state_dict()
    assert set(sde.keys()) == set(pe.keys()), sorted(set(sde.keys()) ^ set(pe.keys()))[:10]
    for k in sde:
        assert tuple(sde[k].shape) == tuple(pe[k].shape), k
    em.load_state_dict(pe)
    em.eval()
    nt, nb_ = om.count_params(pe)
    print(f'ERes2Net F=80 params: trainable {nt}, buffers {nb_}')
    xe = rng.standard_normal((2, 70, F_)).astype(np.float32) * 3.0
    with torch.no_grad():
        ee_ref = em(paddle_shim.to_tensor(xe))
        ee_or = oer.eres2net_forward(pe, torch.from_numpy(xe))
    cmp('eres2net eval emb', ee_or, ee_ref, 2e-5)
    out['eres2net_ref_small.npz'] = dict(x=xe, emb_eval=ee_ref.numpy(), param_seed=np.int64(1000))

    # ---------------- ERes2NetV2 (base_width 26), F=80
    pv = oer.eres2net_params(input_size=F_, embd_dim=192, base_width=26, seed=1000, v2=True)
    vm = ref_er.ERes2NetV2(input_size=F_, embd_dim=192, m_channels=32)
    sdv = vm.state_dict()
    assert set(sdv.keys()) == set(pv.keys()), sorted(set(sdv.keys()) ^ set(pv.keys()))[:10]
    for k in sdv:
        assert tuple(sdv[k].shape) == tuple(pv[k].shape), k
    vm.load_state_dict(pv)
    vm.eval()
    nt, nb_ = om.count_params(pv)
    print(f'ERes2NetV2 F=80 params: trainable {nt}, buffers {nb_}')
    xv2 = rng.standard_normal((2, 70, F_)).astype(np.float32) * 3.0
    with torch.no_grad():
        ev_ref = vm(paddle_shim.to_tensor(xv2))
        ev_or = oer.eres2netv2_forward(pv, torch.from_numpy(xv2))
    cmp('eres2netv2 eval emb', ev_or, ev_ref, 2e-5)
    out['eres2netv2_ref_small.npz'] = dict(x=xv2, emb_eval=ev_ref.numpy(), param_seed=np.int64(1000))

    # ---------------- ERes2Net-large (BASELINE configs[4]: the 55 M-parameter shape -- m_channels 64, expansion 4, base_width 24,
    # scale 3, mul_channel 2; README.md:80), F=80, short T to keep the fixture small
    LARGE = dict(m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3)
    pl = oer.eres2net_params(input_size=F_, embd_dim=192, seed=1001, **LARGE)
    lm = ref_er.ERes2Net(input_size=F_, embd_dim=192, **LARGE)
    sdl = lm.state_dict()
    assert set(sdl.keys()) == set(pl.keys()), sorted(set(sdl.keys()) ^ set(pl.keys()))[:10]
    for k in sdl:
        assert tuple(sdl[k].shape) == tuple(pl[k].shape), k
    lm.load_state_dict(pl)
    lm.eval()
    nt, nb_ = om.count_params(pl)
    print(f'ERes2Net-large F=80 params: trainable {nt}, buffers {nb_}')
    assert 54.5e6 < nt < 56e6, nt                                   # "55 M"
    xl = rng.standard_normal((2, 43, F_)).astype(np.float32) * 3.0
    with torch.no_grad():
        el_ref = lm(paddle_shim.to_tensor(xl))
        el_or = oer.eres2net_forward(pl, torch.from_numpy(xl), **{k: v for k, v in LARGE.items() if k != 'mul_channel'})
    cmp('eres2net-large eval emb', el_or, el_ref, 2e-5)
    out['eres2net_large_ref_small.npz'] = dict(x=xl, emb_eval=el_ref.numpy(), param_seed=np.int64(1001))

    # ---------------- the other classification losses (loss/{amloss,armloss,celoss,subcenterloss,sphereface2}.py)
    from oracle import losses as ol
    ref_l = {n: importlib.import_module('ppvector.loss.' + n) for n in ('amloss', 'armloss', 'celoss', 'subcenterloss', 'sphereface2')}
    lr_ = np.random.RandomState(77)
    Bq, Dq, Cq, Kq = 6, 32, 12, 3
    embq = torch.from_numpy(lr_.standard_normal((Bq, Dq)).astype(np.float32))
    labq = lr_.randint(0, Cq, size=(Bq,)).astype(np.int64)
    lg1 = om.cosine_head(embq, torch.from_numpy(lr_.standard_normal((Dq, Cq)).astype(np.float32)))
    lgk = om.cosine_head(embq, torch.from_numpy(lr_.standard_normal((Dq, Cq * Kq)).astype(np.float32)))
    lg1[0, labq[0]] = -0.995                 # a target below th = cos(pi - m): the (cos - mmm) branch
    lgk[0, labq[0] * Kq:(labq[0] + 1) * Kq] = torch.tensor([-0.997, -0.995, -0.996])   # no ties: max-backward picks one
    cases = [('AMLoss', lambda: ref_l['amloss'].AMLoss(margin=0.2, scale=30, label_smoothing=0.0), lambda l, y: ol.am_loss(l, y, 0.2, 30.0, 0.0), lg1),
             ('AMLoss_ls', lambda: ref_l['amloss'].AMLoss(margin=0.35, scale=30, label_smoothing=0.1), lambda l, y: ol.am_loss(l, y, 0.35, 30.0, 0.1), lg1),
             ('ARMLoss', lambda: ref_l['armloss'].ARMLoss(margin=0.2, scale=30, label_smoothing=0.0), lambda l, y: ol.arm_loss(l, y, 0.2, 30.0, 0.0), lg1),
             ('ARMLoss_ls', lambda: ref_l['armloss'].ARMLoss(margin=0.1, scale=20, label_smoothing=0.1), lambda l, y: ol.arm_loss(l, y, 0.1, 20.0, 0.1), lg1),
             ('CELoss', lambda: ref_l['celoss'].CELoss(label_smoothing=0.0), lambda l, y: ol.ce_loss(l, y, 0.0), lg1),
             ('CELoss_ls', lambda: ref_l['celoss'].CELoss(label_smoothing=0.2), lambda l, y: ol.ce_loss(l, y, 0.2), lg1),
             ('SubCenterLoss', lambda: ref_l['subcenterloss'].SubCenterLoss(margin=0.2, scale=32, K=3), lambda l, y: ol.subcenter_loss(l, y, 0.2, 32.0, False, 3, 0.0), lgk),
             ('SubCenterLoss_easy_ls', lambda: ref_l['subcenterloss'].SubCenterLoss(margin=0.3, scale=32, easy_margin=True, K=3, label_smoothing=0.1),
              lambda l, y: ol.subcenter_loss(l, y, 0.3, 32.0, True, 3, 0.1), lgk),
             ('SphereFace2_C', lambda: ref_l['sphereface2'].SphereFace2(margin=0.2, scale=32.0, lanbuda=0.7, t=3, margin_type='C'),
              lambda l, y: ol.sphereface2_loss(l, y, torch.zeros(()), 0.2, 32.0, 0.7, 3, 'C'), lg1),
             ('SphereFace2_A', lambda: ref_l['sphereface2'].SphereFace2(margin=0.15, scale=32.0, lanbuda=0.7, t=3, margin_type='A'),
              lambda l, y: ol.sphereface2_loss(l, y, torch.zeros(()), 0.15, 32.0, 0.7, 3, 'A'), lg1)]
    gl = dict(labels=labq, logits=lg1.numpy(), logits_k=lgk.numpy(), names=np.asarray([c[0] for c in cases]))
    for name, mk, orc, lg in cases:
        a = lg.clone().requires_grad_(True)
        l_ref = mk()({'features': embq, 'logits': paddle_shim._wrap(a)}, paddle_shim.to_tensor(labq))
        g_ref, = torch.autograd.grad(l_ref, a)
        b = lg.clone().requires_grad_(True)
        l_or = orc(b, torch.from_numpy(labq))
        g_or, = torch.autograd.grad(l_or, b)
        cmp(f'{name} loss', l_or, l_ref, 1e-5)
        cmp(f'{name} dlogits', g_or, g_ref, 1e-5)
        gl[name + '_loss'] = np.float64(float(l_ref))
        gl[name + '_dlogits'] = g_ref.numpy()
    out['losses_ref.npz'] = gl

    # ---------------- classifier head variants (fc.py:6-87): DenseLayer('batchnorm') blocks, 'Linear' output
    hr = np.random.RandomState(91)
    embh = torch.from_numpy(hr.standard_normal((5, 24)).astype(np.float32))
    gh = dict(emb=embh.numpy())
    for tag, ctype, nb in (('cos_b2', 'Cosine', 2), ('lin_b0', 'Linear', 0), ('lin_b1', 'Linear', 1)):
        ph = om.classifier_params(24, 9, ctype, 1, nb, 16, seed=1002)
        hm = ref_fc.SpeakerIdentification(input_dim=24, num_speakers=9, classifier_type=ctype, num_blocks=nb, inter_dim=16)
        sdh = hm.state_dict()
        assert set(sdh) == set(ph), (sorted(set(sdh) ^ set(ph)))
        hm.load_state_dict(ph)
        hm.eval()
        with torch.no_grad():
            lr_ev = hm(paddle_shim.to_tensor(embh.numpy()))['logits']
            lo_ev = om.classifier_head(embh, ph, ctype, nb)
        cmp(f'head {tag} eval logits', lo_ev, lr_ev, 1e-5)
        hm.train()
        with torch.no_grad():
            lr_tr = hm(paddle_shim.to_tensor(embh.numpy()))['logits']
            lo_tr = om.classifier_head(embh, ph, ctype, nb, training=True)
        cmp(f'head {tag} train logits', lo_tr, lr_tr, 1e-5)
        gh[tag + '_eval'], gh[tag + '_train'] = lr_ev.numpy(), lr_tr.numpy()
    out['head_variants_ref.npz'] = gh

    # ---------------- real speech: 4 reference WAVs (3 s crops) -> oracle Fbank -> reference ECAPA graph
    names = ['a_1', 'a_2', 'b_1', 'b_2']
    pcm = np.stack([read_wav_16k_mono(f'{REF}/dataset/{n}.wav') for n in names])
    wav = (pcm.astype(np.float32) / 32768.0)
    feats = ofb.featurize(wav, feature_method='Fbank', method_args=dict(sr=16000, n_mels=80))
    model.eval()
    with torch.no_grad():
        emb_w = model(paddle_shim.to_tensor(feats)).numpy()
    en = emb_w / np.linalg.norm(emb_w, axis=1, keepdims=True)
    out['wavs_3s.npz'] = dict(pcm=pcm, names=np.asarray(names), feats=feats, emb_eval=emb_w, cos=en @ en.T)
    print(f'worst rel-max-err {worst:.3e}')

    if not args.check:
        os.makedirs(GOLD, exist_ok=True)
        for fn, d in out.items():
            np.savez_compressed(os.path.join(GOLD, fn), **d)
            print('wrote', fn, os.path.getsize(os.path.join(GOLD, fn)), 'bytes')
    return 0


if __name__ == '__main__':
    sys.exit(main())
