"""Backbone factory.  Same contract as the reference's ``build_model(input_size, configs)`` (ppvector/models/__init__.py:15-21):
the class is named by ``configs.model_conf.model`` (default 'CAMPPlus') and receives ``input_size`` plus
``configs.model_conf.model_args``; the result exposes ``.embd_dim`` and maps (B, T, F) features to (B, embd_dim)."""
import logging

from .campplus import CAMPPlus
from .ecapa_tdnn import EcapaTdnn
from .eres2net import ERes2Net, ERes2NetV2
from .resnet_se import ResNetSE
from .tdnn import TDNN

__all__ = ['build_model']

_LOG = logging.getLogger('ppvector')
_BUILT = {cls.__name__: cls for cls in (CAMPPlus, EcapaTdnn, ERes2Net, ERes2NetV2, ResNetSE, TDNN)}
_REFERENCE_ONLY = frozenset(('Res2Net',))


def build_model(input_size, configs):
    conf = configs.model_conf
    name, kwargs = conf.get('model', 'CAMPPlus'), dict(conf.get('model_args', {}) or {})
    cls = _BUILT.get(name)
    if cls is None:
        if name in _REFERENCE_ONLY:
            raise NotImplementedError(f'{name} is not built on the HIP engine yet ({", ".join(sorted(_BUILT))} are)')
        raise AttributeError(f"module '{__name__}' has no attribute '{name}'")
    backbone = cls(input_size=input_size, **kwargs)
    _LOG.info('成功创建模型：%s，参数为：%s', name, kwargs)
    return backbone
