"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/vpmi.h declares (no compute calls: there is no GPU here), host-side logic of the Python
mirror (registries, config loading, schedules, metrics, parameter naming)."""
import os
import re

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ppvector import _native as N
    hdr = open(os.path.join(ROOT, 'include', 'vpmi.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(vp_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 25
    lib = N.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in vpmi.h but not exported'
    assert declared == set(N.EXPORTED_SYMBOLS), declared ^ set(N.EXPORTED_SYMBOLS)
    assert lib.vp_version() == 100
    import ctypes as C
    o = N.FbankOpts()
    lib.vp_fbank_default_opts(C.byref(o))
    assert (o.sample_rate, o.n_mels, o.remove_dc) == (16000, 23, 1)
    assert lib.vp_fbank_num_frames(C.byref(o), 48000) == 298
    assert lib.vp_fbank_num_frames(C.byref(o), 399) == 0
    assert lib.vp_conv1d_tiles_m(256, 298) == 596 and lib.vp_conv1d_nseg(298) == 2 and lib.vp_conv1d_nseg(64) == 3


def test_struct_sizes_match_header_layout():
    import ctypes as C
    from ppvector import _native as N
    assert C.sizeof(N.TdnnLayer) == 4 * 8 + 4 * 4 + 8              # four pointers, four ints, w_hl
    assert C.sizeof(N.FbankOpts) == 9 * 4
    assert C.sizeof(N.SeRes2Block) == 17 * C.sizeof(N.TdnnLayer) + 4 * 8
    assert C.sizeof(N.AspWeights) == C.sizeof(N.TdnnLayer) + 3 * 8 + 8


def test_hl32_packing_matches_the_header_definition():
    """VP_HL32 (include/vpmi.h): a row of C channels = C / 32 groups of 128 bytes, each [32 x bf16 hi | 32 x bf16 lo], hi = bf16(v) round
    to nearest even, lo = bf16(v - hi), 4 bytes per element.  ppvector.models.utils.pack_hl32 / unpack_hl32 against a NumPy restatement
    of that definition, byte for byte -- the layout the split-precision kernels and every `w_hl` weight buffer rely on."""
    import numpy as np
    from ppvector.models.utils import pack_hl32, unpack_hl32

    def bf16_rne(a):                                   # f32 -> bf16 bits (uint16), round to nearest even
        u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)

    def bf16_val(b):
        return (b.astype(np.uint32) << 16).view(np.float32)

    rng = np.random.RandomState(3)
    x = (rng.randn(7, 96) * np.exp(rng.randn(7, 96) * 3)).astype(np.float32)
    hi = bf16_rne(x)
    lo = bf16_rne(x - bf16_val(hi))
    want = np.concatenate([hi.reshape(7, 3, 32), lo.reshape(7, 3, 32)], axis=-1).reshape(7, 3 * 64)      # uint16 view of the rows
    got = pack_hl32(torch.from_numpy(x))
    assert got.shape == (7, 96) and got.dtype == torch.float32 and got.numel() * 4 == want.size * 2
    assert np.array_equal(got.numpy().view(np.uint16).reshape(7, 192), want)
    back = unpack_hl32(got).numpy()
    assert np.array_equal(back, bf16_val(hi) + bf16_val(lo))
    rel = np.abs(back - x) / np.abs(x)
    assert rel.max() < 2.0 ** -15, rel.max()             # two 8-bit pieces: at most 2^-16 (1 + 2^-8) of the value
    with pytest.raises(ValueError):
        pack_hl32(torch.zeros(2, 40))


def test_no_cpu_fallback():
    from ppvector import _native as N
    from ppvector.data_utils.featurizer import AudioFeaturizer
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    fz = AudioFeaturizer('Fbank', {'sr': 16000, 'n_mels': 80})
    with pytest.raises(N.VpmiError):
        fz(torch.zeros(2, 16000))
    m = EcapaTdnn(80).eval()
    with pytest.raises(N.VpmiError):
        m(torch.zeros(1, 50, 80))


@pytest.mark.parametrize('cfg', ['ecapa_tdnn.yml', 'tdnn.yml', 'cam++.yml', 'resnet_se.yml', 'eres2net.yml', 'eres2netv2.yml'])
def test_reference_configs_build(cfg):
    from ppvector.loss import build_loss
    from ppvector.models import build_model
    from ppvector.data_utils.featurizer import AudioFeaturizer
    from ppvector.utils.utils import dict_to_object
    path = os.path.join('/root/reference/configs', cfg)
    if os.path.exists(path):        # the reference's own YAML, unmodified, where it is present
        raw = yaml.safe_load(open(path, encoding='utf-8'))
    else:                           # same keys, restated (the reference tree does not travel)
        model = {'ecapa_tdnn.yml': dict(model='EcapaTdnn', model_args=dict(embd_dim=192, pooling_type='ASP',
                                                                           channels=[512, 512, 512, 512, 1536])),
                 'tdnn.yml': dict(model='TDNN', model_args=dict(embd_dim=192, pooling_type='ASP')),
                 'cam++.yml': dict(model='CAMPPlus', model_args=dict(embd_dim=192)),
                 'resnet_se.yml': dict(model='ResNetSE', model_args=dict(embd_dim=192, pooling_type='ASP')),
                 'eres2net.yml': dict(model='ERes2Net', model_args=dict(embd_dim=192, m_channels=32)),
                 'eres2netv2.yml': dict(model='ERes2NetV2', model_args=dict(embd_dim=192, m_channels=32))}[cfg]
        model['classifier'] = dict(classifier_type='Cosine', num_speakers=2796, num_blocks=0)
        raw = dict(preprocess_conf=dict(feature_method='Fbank', method_args=dict(sr=16000, n_mels=80)),
                   model_conf=model,
                   loss_conf=dict(loss='AAMLoss', loss_args=dict(margin=0.2, scale=32, easy_margin=False,
                                                                 label_smoothing=0.0)))
    configs = dict_to_object(raw)
    fz = AudioFeaturizer(feature_method=configs.preprocess_conf.feature_method,
                         method_args=configs.preprocess_conf.get('method_args', {}))
    assert fz.feature_dim == 80
    model = build_model(input_size=fz.feature_dim, configs=configs)
    assert model.embd_dim == 192
    loss = build_loss(configs)
    assert hasattr(loss, 'update')
    loss.update(margin=0.1)
    assert abs(loss.cos_m - np.cos(0.1)) < 1e-12


def test_state_dict_names_match_reference_scheme():
    from oracle import models as om
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.tdnn import TDNN
    m = EcapaTdnn(80)
    p = om.ecapa_params(80)
    assert set(m.state_dict().keys()) == set(p.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(p[k].shape), k
    m.load_state_dict(p)
    t = TDNN(80)
    pt = om.tdnn_params(80)
    assert set(t.state_dict().keys()) == set(pt.keys())
    t.load_state_dict(pt)
    from oracle import campplus as oc, eres2net as oer, resnet_se as orse
    from ppvector.models.campplus import CAMPPlus
    from ppvector.models.eres2net import ERes2Net, ERes2NetV2
    from ppvector.models.resnet_se import ResNetSE
    for mod, pp in ((CAMPPlus(80, embd_dim=192), oc.campplus_params(80, 192)), (ResNetSE(80), orse.resnetse_params(80, 192)),
                    (ERes2Net(80), oer.eres2net_params(80, 192)), (ERes2NetV2(80), oer.eres2net_params(80, 192, base_width=26, v2=True))):
        assert set(mod.state_dict().keys()) == set(pp.keys())
        mod.load_state_dict(pp)
    # nn.Sequential(backbone, classifier) key scheme of the reference checkpoints: "0.<...>", "1.weight"
    from ppvector.models.fc import SpeakerIdentification
    seq = torch.nn.Sequential(m, SpeakerIdentification(192, 10))
    assert '0.blocks.0.conv.conv.weight' in seq.state_dict() and '1.weight' in seq.state_dict()
    assert tuple(seq.state_dict()['1.weight'].shape) == (192, 10)


def test_unknown_names_raise_like_reference():
    from ppvector.data_utils.featurizer import AudioFeaturizer
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    with pytest.raises(Exception):
        AudioFeaturizer('Nope')
    with pytest.raises(Exception):
        EcapaTdnn(80, pooling_type='XYZ')
    with pytest.raises(ValueError):
        SpeakerIdentification(192, 4, classifier_type='Nope')
    assert AudioFeaturizer('MelSpectrogram', {}).feature_dim == 64
    assert AudioFeaturizer('Spectrogram', {}).feature_dim == 257


def test_margin_scheduler_matches_oracle():
    from oracle import models as om
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.optimizer.scheduler import MarginScheduler, cosine_decay_with_warmup
    crit = AAMLoss()
    spe, me = 7, 20
    ms = MarginScheduler(crit, increase_start_epoch=int(me * 0.3), fix_epoch=int(me * 0.7), step_per_epoch=spe)
    for step in range(spe * me):
        ms.step()
        assert abs(ms.get_margin() - om.margin_schedule(step, spe, me)) < 1e-12
        assert abs(crit.margin - ms.get_margin()) < 1e-15
    lr = cosine_decay_with_warmup(0.001, spe, fix_epoch=me, warmup_epoch=5, min_lr=1e-5)
    assert lr[0] == 0.0 and abs(lr[5 * spe] - 0.001) < 1e-12 and lr[-1] > 1e-5 and len(lr) == spe * me


def test_lr_table_matches_the_reference_schedule(golden_dir):
    """lr at every scheduler step == the reference's own cosine_decay_with_warmup run through PiecewiseDecay
    (oracle/gen_lr_golden.py: the duplicated boundary at the end of warm-up, the hold past the table's end)."""
    import numpy as np
    from ppvector.optimizer import build_lr_scheduler
    from ppvector.utils.utils import dict_to_object
    g = np.load(os.path.join(golden_dir, 'lr_table_ref.npz'))
    for name in ('ecapa_yaml', 'short_warm'):
        lr0, spe, fix, warm, min_lr = g[name + '_args']
        cfg = dict_to_object(dict(optimizer_conf=dict(scheduler='WarmupCosineSchedulerLR',
                                                      scheduler_args=dict(learning_rate=float(lr0), min_lr=float(min_lr), warmup_epoch=int(warm))),
                                  train_conf=dict(max_epoch=int(fix))))
        sch = build_lr_scheduler(step_per_epoch=int(spe), configs=cfg)
        for k, ref in enumerate(g[name]):
            assert abs(sch.get_lr() - ref) < 1e-15, (name, k, sch.get_lr(), ref)
            sch.step()


def test_metrics_match_oracle():
    from oracle import scoring as osc
    from ppvector.metric.metrics import compute_dcf, compute_eer, compute_fnr_fpr
    rng = np.random.RandomState(0)
    scores = np.concatenate([rng.normal(0.6, 0.2, 300), rng.normal(0.1, 0.2, 3000)])
    labels = np.concatenate([np.ones(300, int), np.zeros(3000, int)])
    fnr, fpr, thr = compute_fnr_fpr(scores, labels)
    f2, p2, t2 = osc.fnr_fpr(scores, labels)
    assert np.array_equal(fnr, f2) and np.array_equal(fpr, p2) and np.array_equal(thr, t2)
    e, th = compute_eer(fnr, fpr, scores)
    e2, th2 = osc.eer(f2, p2, scores)
    assert e == e2 and th == th2 and 0.05 < e < 0.2
    assert compute_dcf(fnr, fpr) == osc.min_dcf(f2, p2)


def test_ctx_path_does_not_deadlock(monkeypatch):
    """ctx() nests lib()/load_library() under one lock: must fail fast without a device, not hang."""
    import threading
    from ppvector import _native as N
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    res = []

    def run():
        try:
            N.ctx(0)
            res.append('ok')
        except N.VpmiError as e:
            res.append(str(e))

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(30)
    assert res and 'vp_create' in res[0]


def test_audio_segment_front_end(tmp_path):
    """Host audio front end of the predictor (yeaudio semantics restated): WAV decode, stereo down-mix,
    resample, dB normalisation."""
    import wave
    from ppvector.predict import AudioSegment
    sr = 44100
    t = np.arange(sr) / sr
    x = (0.25 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    st = np.stack([x, 0.5 * x], axis=1)
    p = str(tmp_path / 's.wav')
    with wave.open(p, 'wb') as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((st * 32767).astype(np.int16).tobytes())
    seg = AudioSegment.from_file(p)
    assert seg.sample_rate == sr and seg.samples.ndim == 1 and abs(seg.duration - 1.0) < 1e-3
    assert np.max(np.abs(seg.samples - 0.75 * x)) < 1e-3
    seg.resample(16000)
    assert seg.sample_rate == 16000 and abs(len(seg.samples) - 16000) <= 1
    seg.normalize(target_db=-20)
    assert abs(seg.rms_db - (-20.0)) < 1e-3


def test_optimizer_options_that_change_the_update_raise():
    """paddle.optimizer.Adam keyword arguments: the ones that change the update rule and are not built must raise, not be
    dropped (a config naming amsgrad=True would silently train with another optimiser); the no-ops are accepted."""
    from ppvector.optimizer.adam import _no_unbuilt_options
    _no_unbuilt_options('Adam', dict(name='x', lazy_mode=False, multi_precision=True, use_multi_tensor=False, amsgrad=False, rescale_grad=1.0))
    for bad in (dict(amsgrad=True), dict(rescale_grad=0.5), dict(grad_clip=object())):
        with pytest.raises(NotImplementedError):
            _no_unbuilt_options('Adam', dict(bad))
    with pytest.raises(TypeError):
        _no_unbuilt_options('Adam', dict(no_such_option=1))
