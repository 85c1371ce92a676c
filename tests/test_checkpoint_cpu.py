"""CPU tests of the checkpoint importer / exporter (ppvector/utils/checkpoint.py: the reference's directory layout and
paddle.save's pickle layout).  No Paddle binary is available: the files below are hand-built in that layout."""
import collections
import json
import os
import pickle
import types

import numpy as np
import pytest
import torch

from ppvector.utils import checkpoint as ck
from ppvector.utils.utils import dict_to_object


def _cfg():
    return dict_to_object({'model_conf': {'model': 'TDNN'}, 'preprocess_conf': {'feature_method': 'Fbank'}, 'loss_conf': {}})


def _model():
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.models.tdnn import TDNN
    torch.manual_seed(0)
    return torch.nn.Sequential(TDNN(80, channels=64, embd_dim=32), SpeakerIdentification(32, 10))


def test_pdparams_round_trip_keeps_keys_layout_and_values(tmp_path):
    m = _model()
    sd = m.state_dict()
    assert any(k.startswith('0.') for k in sd) and '1.weight' in sd
    path = str(tmp_path / 'model.pdparams')
    ck.save_pdparams(sd, path)
    with open(path, 'rb') as f:                                         # the file is a plain pickle of ndarrays + the name table
        raw = pickle.load(f)
    assert isinstance(raw[ck.NAME_TABLE], dict) and set(raw[ck.NAME_TABLE]) == set(sd)
    assert all(isinstance(v, np.ndarray) for k, v in raw.items() if k != ck.NAME_TABLE)
    back, names = ck.load_pdparams(path, with_names=True)
    assert list(back) == list(sd) and names == {k: k for k in sd}
    for k in sd:
        assert back[k].dtype == sd[k].dtype and torch.equal(back[k], sd[k]), k
    m2 = _model()
    for p in m2.parameters():
        p.data.add_(1.0)
    ck.load_pretrained(m2, str(tmp_path))                               # a directory resolves to <dir>/model.pdparams
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_reads_paddle_style_files(tmp_path):
    """paddle.save layouts: name table with Paddle's internal parameter names, (name, ndarray) tuples for tensors pickled
    outside a state dict, arrays split for the 4 GB limit of pickle protocol 2."""
    rng = np.random.RandomState(0)
    w, b, big = rng.standard_normal((4, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32), rng.standard_normal((6, 5)).astype(np.float32)
    obj = {'0.linear.weight': w, '0.linear.bias': ('linear_0.b_0', b),
           '0.big@@.0': big.reshape(-1)[:16], '0.big@@.1': big.reshape(-1)[16:],
           ck.BIG_PARAMS: {'0.big': {'OriginShape': (6, 5), 'slices': ['0.big@@.0', '0.big@@.1']}},
           ck.NAME_TABLE: {'0.linear.weight': 'linear_0.w_0', '0.linear.bias': 'linear_0.b_0', '0.big': 'big_0.w_0'}}
    for proto in (2, 4):
        path = str(tmp_path / f'p{proto}.pdparams')
        with open(path, 'wb') as f:
            pickle.dump(obj, f, protocol=proto)
        state, names = ck.load_pdparams(path, with_names=True)
        assert set(state) == {'0.linear.weight', '0.linear.bias', '0.big'} and names['0.big'] == 'big_0.w_0'
        assert np.array_equal(state['0.linear.weight'].numpy(), w) and np.array_equal(state['0.linear.bias'].numpy(), b)
        assert np.array_equal(state['0.big'].numpy(), big)


def test_refuses_pickles_that_would_run_code(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ('echo pwned > /dev/null',))
    path = str(tmp_path / 'evil.pdparams')
    with open(path, 'wb') as f:
        pickle.dump({'0.w': Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        ck.load_pdparams(path)


def test_checkpoint_directory_save_resume_rotation(tmp_path):
    from ppvector.optimizer import Adam
    cfg, root = _cfg(), str(tmp_path / 'models')
    m = _model()
    opt = Adam(m.parameters(), learning_rate=1e-3, weight_decay=1e-6)
    opt.m.copy_(torch.randn(opt.m.numel())); opt.v.copy_(torch.rand(opt.v.numel())); opt.t = 7
    margin = types.SimpleNamespace(get_margin=lambda: 0.15, stepped=None)
    margin.step = lambda current_step=None: setattr(margin, 'stepped', current_step)
    for epoch in (1, 2, 3, 4, 5):
        ck.save_checkpoint(cfg, m, opt, None, margin, root, epoch, eer=0.05, min_dcf=0.3, threshold=0.4)
    ck.save_checkpoint(cfg, m, opt, None, margin, root, 5, eer=0.05, min_dcf=0.3, threshold=0.4, best_model=True)
    fam = os.path.join(root, 'TDNN_Fbank')
    assert sorted(os.listdir(fam)) == ['best_model', 'epoch_3', 'epoch_4', 'epoch_5', 'last_model']     # epoch_{n-3} dropped
    assert sorted(os.listdir(os.path.join(fam, 'last_model'))) == ['model.pdparams', 'model.state', 'optimizer.pdopt']
    with open(os.path.join(fam, 'last_model', 'model.state'), encoding='utf-8') as f:
        st = json.load(f)
    assert st == {'last_epoch': 5, 'version': st['version'], 'model_conf.model': 'TDNN', 'feature_method': 'Fbank', 'loss': 'AAMLoss',
                  'threshold': 0.4, 'eer': 0.05, 'min_dcf': 0.3, 'margin': 0.15}
    pdopt = ck.read_pd(os.path.join(fam, 'last_model', 'optimizer.pdopt'))
    k0 = next(k for k, _ in m.named_parameters())
    assert f'{k0}_moment1_0' in pdopt and abs(float(pdopt[f'{k0}_beta1_pow_acc_0'][0]) - 0.9 ** 8) < 1e-7
    assert pdopt['LR_Scheduler']['last_lr'] == pytest.approx(1e-3)
    # resume into a fresh model / optimizer: auto-pick last_model, fast-forward the schedules by last_epoch * step_epoch
    m2 = _model()
    for p in m2.parameters():
        p.data.mul_(0.5)
    opt2 = Adam(m2.parameters(), learning_rate=1e-3, weight_decay=1e-6)
    sched = types.SimpleNamespace(n=0)
    sched.step = lambda: setattr(sched, 'n', sched.n + 1)
    *_, last_epoch, best_eer = ck.load_checkpoint(cfg, m2, opt2, None, sched, margin, 11, root, None)
    assert (last_epoch, best_eer) == (5, 0.05) and sched.n == 55 and margin.stepped == 55
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert opt2.t == 7
    by_name = dict(m2.named_parameters())
    for k, p in m.named_parameters():
        o1, o2 = opt._offset(p), opt2._offset(by_name[k])
        assert torch.equal(opt.m[o1:o1 + p.numel()], opt2.m[o2:o2 + p.numel()]) and torch.equal(opt.v[o1:o1 + p.numel()], opt2.v[o2:o2 + p.numel()])
    # a mismatching model is refused on explicit resume, as in the reference
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.models.tdnn import TDNN
    other = torch.nn.Sequential(TDNN(80, channels=32, embd_dim=32), SpeakerIdentification(32, 10))
    with pytest.raises(Exception):
        ck.load_checkpoint(cfg, other, Adam(other.parameters()), None, sched, None, 11, root, os.path.join(fam, 'best_model'))


def test_momentum_state_round_trips_through_pdopt():
    """Momentum's state leaves and returns under paddle's '<param>_velocity_0' keys; Adam's moments are not mixed up with it."""
    from ppvector.optimizer import Momentum
    m = _model()
    opt = Momentum(m.parameters(), learning_rate=0.1, momentum=0.8, use_nesterov=True, weight_decay=1e-4)
    opt.velocity.copy_(torch.randn(opt.velocity.numel())); opt.t = 3
    pdopt = ck.optimizer_to_pdopt(opt, m)
    k0 = next(k for k, _ in m.named_parameters())
    assert f'{k0}_velocity_0' in pdopt and f'{k0}_moment1_0' not in pdopt and f'{k0}_beta1_pow_acc_0' not in pdopt
    m2 = _model()
    opt2 = Momentum(m2.parameters(), learning_rate=0.1)
    assert ck.pdopt_to_optimizer(pdopt, opt2, m2) == []
    assert torch.equal(opt.velocity, opt2.velocity)


def test_optimizer_and_scheduler_factories():
    """build_optimizer / build_lr_scheduler resolve names like the reference's getattr on its module
    (ppvector/optimizer/__init__.py:12-33): the built classes, AttributeError for an unknown name, the filled-in defaults."""
    import math
    from ppvector import optimizer as O
    m = _model()

    def cfg(**oc):
        return dict_to_object({'optimizer_conf': oc, 'train_conf': {'max_epoch': 10}})

    for name, cls, args in (('Adam', O.Adam, {'weight_decay': 1e-6}), ('AdamW', O.AdamW, {}), ('SGD', O.SGD, {'weight_decay': 1e-4}),
                            ('Momentum', O.Momentum, {'momentum': 0.95, 'use_nesterov': True})):
        opt = O.build_optimizer(_model().parameters(), 1e-3, cfg(optimizer=name, optimizer_args=args))
        assert type(opt) is cls
    assert O.build_optimizer(_model().parameters(), 1e-3, cfg(optimizer='AdamW')).wd == pytest.approx(0.01)        # paddle's default
    with pytest.raises(AttributeError):
        O.build_optimizer(m.parameters(), 1e-3, cfg(optimizer='NoSuchOptimizer'))
    with pytest.raises(NotImplementedError):
        O.build_optimizer(m.parameters(), 1e-3, cfg(optimizer='Lamb'))
    with pytest.raises(NotImplementedError):
        O.build_optimizer(m.parameters(), 1e-3, cfg(optimizer='Adam', optimizer_args={'grad_clip': object()}))
    s = O.build_lr_scheduler(7, cfg(scheduler='CosineAnnealingDecay', scheduler_args={'learning_rate': 0.01}))
    assert s.T_max == int(10 * 1.2) * 7                                      # optimizer/__init__.py:24-25
    lrs = []
    for _ in range(s.T_max + 1):
        lrs.append(s.get_lr()); s.step()
    assert lrs[0] == pytest.approx(0.01) and lrs[-1] == pytest.approx(0.0, abs=1e-12) and lrs[s.T_max // 2] == pytest.approx(0.005)
    # paddle steps it recursively: lr_t = eta + (1 + cos(pi t / T)) / (1 + cos(pi (t - 1) / T)) * (lr_{t-1} - eta); same values
    rec, T = [0.01], s.T_max
    for t in range(1, T):
        rec.append((1 + math.cos(math.pi * t / T)) / (1 + math.cos(math.pi * (t - 1) / T)) * rec[-1])
    assert max(abs(a - b) for a, b in zip(rec, lrs)) < 1e-12
    w = O.build_lr_scheduler(7, cfg(scheduler='WarmupCosineSchedulerLR', scheduler_args={'learning_rate': 0.001, 'min_lr': 1e-5, 'warmup_epoch': 2}))
    assert len(w.table) >= 70
    with pytest.raises(NotImplementedError):
        O.build_lr_scheduler(7, cfg(scheduler='StepDecay'))
