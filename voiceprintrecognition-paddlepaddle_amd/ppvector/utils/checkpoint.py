"""Checkpoints in the reference's on-disk layout (ppvector/utils/checkpoint.py:11-165): a directory holding
``model.pdparams`` (state dict of ``nn.Sequential(backbone, classifier)``: keys ``0.<backbone...>`` / ``1.weight``),
``optimizer.pdopt`` and the ``model.state`` JSON (last_epoch, version, model_conf.model, feature_method, loss, threshold,
eer, min_dcf, margin -- :137-147), with ``epoch_N`` / ``last_model`` / ``best_model`` rotation (:118-163).

``.pdparams`` / ``.pdopt`` are what ``paddle.save`` writes for a state dict: a pickle (protocol 4) of
``{structured_key: numpy.ndarray, ..., 'StructuredToParameterName@@': {structured_key: parameter_name}}``; values saved
outside a state dict appear as ``(name, ndarray)`` tuples, and arrays over 4 GB written with protocol < 4 are split under
``'UnpackBigParamInfor@@'``.  PaddlePaddle is a third-party dependency absent from /root/reference and not installable
here: the layout is restated from Paddle 2.6's published behaviour and is UNPINNED against a Paddle binary -- the tests
cover the round trip and hand-built files of that layout.  The parameter layouts need no conversion: this package's modules
already keep Paddle's conventions ([in, out] Linear weights, BN ``_mean`` / ``_variance``, [D, C] cosine head).
Files are read through an allow-list unpickler (numpy array reconstruction + plain containers only): a checkpoint from an
untrusted source cannot run code on load.
"""
import collections
import io
import json
import logging
import math
import os
import pickle
import shutil

import numpy as np
import torch

from ppvector import __version__

_LOG = logging.getLogger('ppvector')
NAME_TABLE = 'StructuredToParameterName@@'
BIG_PARAMS = 'UnpackBigParamInfor@@'
_ALLOWED = {
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
    ('numpy', 'ndarray'), ('numpy', 'dtype'), ('collections', 'OrderedDict'),
    ('builtins', 'tuple'), ('builtins', 'list'), ('builtins', 'dict'), ('builtins', 'set'), ('builtins', 'frozenset'),
    ('builtins', 'int'), ('builtins', 'float'), ('builtins', 'bool'), ('builtins', 'str'), ('builtins', 'bytes'),
    ('builtins', 'complex'), ('builtins', 'bytearray'), ('_codecs', 'encode'),
}


class _ArraysOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'checkpoint refers to {module}.{name}: only arrays and plain containers are loaded')


def _collapse(obj):
    """(name, ndarray) tuples (tensors pickled outside a state dict) -> ndarray; containers recursively."""
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and isinstance(obj[1], np.ndarray):
        return obj[1]
    if isinstance(obj, dict):
        return type(obj)((k, _collapse(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_collapse(v) for v in obj)
    return obj


def read_pd(path):
    """One paddle.save file -> python object with ndarrays (name table kept under NAME_TABLE if present)."""
    with open(path, 'rb') as f:
        obj = _ArraysOnlyUnpickler(io.BytesIO(f.read()), encoding='latin1').load()
    obj = _collapse(obj)
    if isinstance(obj, dict) and BIG_PARAMS in obj:                       # re-join arrays split for the 4 GB pickle limit
        for key, info in obj.pop(BIG_PARAMS).items():
            parts = [np.asarray(obj.pop(s)).reshape(-1) for s in info['slices']]
            obj[key] = np.concatenate(parts).reshape(info['OriginShape'])
    return obj


def write_pd(obj, path, protocol=4):
    with open(path, 'wb') as f:
        pickle.dump(obj, f, protocol=protocol)


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def load_pdparams(path, with_names=False):
    """model.pdparams -> OrderedDict{structured key: float tensor} (and the structured -> parameter-name table)."""
    raw = read_pd(path)
    names = dict(raw.pop(NAME_TABLE, {}) or {})
    state = collections.OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in raw.items() if isinstance(v, np.ndarray))
    return (state, names) if with_names else state


def save_pdparams(state_dict, path, names=None):
    out = collections.OrderedDict((k, _np(v)) for k, v in state_dict.items())
    out[NAME_TABLE] = dict(names) if names is not None else {k: k for k in state_dict}
    write_pd(out, path)


# ------------------------------------------------------------------------------------------------ optimizer state
def optimizer_to_pdopt(optimizer, model, names=None):
    """Optimiser state in paddle's state_dict() form: Adam / AdamW '<param name>_moment1_0', '_moment2_0', '_beta1_pow_acc_0',
    '_beta2_pow_acc_0' (the pow accumulators hold beta^(t+1) after t steps), Momentum '<param name>_velocity_0'; and 'LR_Scheduler'."""
    names = names or {}
    own = {id(p): k for k, p in model.named_parameters()}
    out = collections.OrderedDict()
    for p in optimizer.params:
        key = own[id(p)]
        pn = names.get(key, key)
        off = optimizer._offset(p)
        for suffix, buf in optimizer.state_slots():          # Adam / AdamW: moment1_0, moment2_0; Momentum: velocity_0
            out[f'{pn}_{suffix}'] = _np(buf[off:off + p.numel()].view_as(p))
        if hasattr(optimizer, 'beta1'):
            out[f'{pn}_beta1_pow_acc_0'] = np.asarray([optimizer.beta1 ** (optimizer.t + 1)], np.float32)
            out[f'{pn}_beta2_pow_acc_0'] = np.asarray([optimizer.beta2 ** (optimizer.t + 1)], np.float32)
    sched = optimizer.lr
    out['LR_Scheduler'] = {'last_epoch': int(getattr(sched, 'i', optimizer.t)), 'last_lr': float(optimizer.get_lr())}
    return out


def pdopt_to_optimizer(state, optimizer, model, names=None):
    """Inverse of optimizer_to_pdopt.  Returns the keys it could not place (empty when the file matches the model)."""
    names = names or {}
    own = {id(p): k for k, p in model.named_parameters()}
    missing, t = [], None
    for p in optimizer.params:
        key = own[id(p)]
        pn = names.get(key, key)
        slots = [(state.get(f'{pn}_{suffix}'), buf) for suffix, buf in optimizer.state_slots()]
        if any(a is None or tuple(np.shape(a)) != tuple(p.shape) for a, _ in slots):
            missing.append(key)
            continue
        off = optimizer._offset(p)
        for a, buf in slots:
            buf[off:off + p.numel()].copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(-1))
        b1 = state.get(f'{pn}_beta1_pow_acc_0')
        if t is None and b1 is not None and hasattr(optimizer, 'beta1') and 0.0 < float(np.reshape(b1, -1)[0]) < 1.0:
            t = int(round(math.log(float(np.reshape(b1, -1)[0])) / math.log(optimizer.beta1))) - 1
    if t is None:
        t = int((state.get('LR_Scheduler') or {}).get('last_epoch', 0))
    optimizer.t = max(t, 0)
    return missing


# ------------------------------------------------------------------------------------------------ reference entry points
def _model_file(path):
    return os.path.join(path, 'model.pdparams') if os.path.isdir(path) else path


def load_pretrained(model, pretrained_model):
    """utils/checkpoint.py:11-42: take what matches by name and shape, warn about the rest."""
    if pretrained_model is None:
        return model
    pretrained_model = _model_file(pretrained_model)
    assert os.path.exists(pretrained_model), f"{pretrained_model} 模型不存在！"
    own = model.state_dict()
    loaded = load_pdparams(pretrained_model)
    for name, weight in own.items():
        if name in loaded:
            if list(weight.shape) != list(loaded[name].shape):
                _LOG.warning('%s not used, shape %s unmatched with %s in model.', name, list(loaded[name].shape), list(weight.shape))
                loaded.pop(name, None)
        else:
            _LOG.warning('Lack weight: %s', name)
    res = model.load_state_dict(loaded, strict=False)
    if res.unexpected_keys:
        _LOG.warning('Unexpected key(s) in state_dict: %s. ', ', '.join(f'"{k}"' for k in res.unexpected_keys))
    if res.missing_keys:
        _LOG.warning('Missing key(s) in state_dict: %s. ', ', '.join(f'"{k}"' for k in res.missing_keys))
    _LOG.info('成功加载预训练模型：%s', pretrained_model)
    return model


def _family_dir(configs, save_model_path):
    return os.path.join(save_model_path, f'{configs.model_conf.model}_{configs.preprocess_conf.feature_method}')


def load_checkpoint(configs, model, optimizer, amp_scaler, scheduler, margin_scheduler, step_epoch, save_model_path, resume_model):
    """utils/checkpoint.py:45-106: resume from resume_model, else from <save_model_path>/<model>_<feature>/last_model."""
    last_epoch1, best_eer1 = 0, 1

    def load_model(model_path):
        assert os.path.exists(os.path.join(model_path, 'model.pdparams')), "模型参数文件不存在！"
        assert os.path.exists(os.path.join(model_path, 'optimizer.pdopt')), "优化方法参数文件不存在！"
        state, names = load_pdparams(os.path.join(model_path, 'model.pdparams'), with_names=True)
        res = model.load_state_dict(state, strict=False)
        assert len(res.missing_keys) == len(res.unexpected_keys) == 0, "模型参数加载失败，参数权重不匹配，请可以考虑当做预训练模型！"
        missing = pdopt_to_optimizer(read_pd(os.path.join(model_path, 'optimizer.pdopt')), optimizer, model, names)
        assert not missing, f'optimizer.pdopt lacks the moments of {missing[:3]}...'
        with open(os.path.join(model_path, 'model.state'), 'r', encoding='utf-8') as f:
            json_data = json.load(f)
        last_epoch = json_data['last_epoch']
        best_eer = json_data.get('eer', 1)
        _LOG.info('成功恢复模型参数和优化方法参数：%s', model_path)
        for _ in range(last_epoch * step_epoch):
            scheduler.step()
        if margin_scheduler is not None:
            margin_scheduler.step(current_step=last_epoch * step_epoch)
        return last_epoch, best_eer

    last_model_dir = os.path.join(_family_dir(configs, save_model_path), 'last_model')
    have_last = all(os.path.exists(os.path.join(last_model_dir, f)) for f in ('model.pdparams', 'optimizer.pdopt'))
    if resume_model is not None:
        last_epoch1, best_eer1 = load_model(resume_model)
    elif have_last:
        try:
            last_epoch1, best_eer1 = load_model(last_model_dir)
        except Exception as e:                                                 # as the reference: a broken last_model is not fatal
            _LOG.warning('尝试自动恢复最新模型失败，错误信息：%s', e)
    return model, optimizer, amp_scaler, scheduler, margin_scheduler, last_epoch1, best_eer1


def save_checkpoint(configs, model, optimizer, amp_scaler, margin_scheduler, save_model_path, epoch_id, eer=None, min_dcf=None,
                    threshold=None, best_model=False):
    """utils/checkpoint.py:110-165."""
    family = _family_dir(configs, save_model_path)
    model_path = os.path.join(family, 'best_model' if best_model else f'epoch_{epoch_id}')
    if os.path.exists(model_path):
        shutil.rmtree(model_path)
    os.makedirs(model_path, exist_ok=True)
    write_pd(optimizer_to_pdopt(optimizer, model), os.path.join(model_path, 'optimizer.pdopt'))
    save_pdparams(model.state_dict(), os.path.join(model_path, 'model.pdparams'))
    data = {"last_epoch": epoch_id, "version": __version__, "model_conf.model": configs.model_conf.model,
            "feature_method": configs.preprocess_conf.feature_method, "loss": configs.loss_conf.get('use_loss', 'AAMLoss')}
    if eer is not None:
        data.update(threshold=threshold, eer=eer, min_dcf=min_dcf)
    if margin_scheduler:
        data['margin'] = margin_scheduler.get_margin()
    with open(os.path.join(model_path, 'model.state'), 'w', encoding='utf-8') as f:
        f.write(json.dumps(data, indent=4, ensure_ascii=False))
    if not best_model:
        last_model_path = os.path.join(family, 'last_model')
        shutil.rmtree(last_model_path, ignore_errors=True)
        shutil.copytree(model_path, last_model_path)
        shutil.rmtree(os.path.join(family, f'epoch_{epoch_id - 3}'), ignore_errors=True)
    _LOG.info('已保存模型：%s', model_path)
