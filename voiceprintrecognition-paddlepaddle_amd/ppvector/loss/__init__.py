"""Loss factory.  Same contract as the reference's ``build_loss(configs)`` (ppvector/loss/__init__.py:16-22): the class is
named by ``configs.loss_conf.loss`` and built with ``configs.loss_conf.loss_args``; an unknown name is an AttributeError on
this module, a known-but-unbuilt one says so."""
import logging

from .aamloss import AAMLoss
from .amloss import AMLoss
from .armloss import ARMLoss
from .celoss import CELoss
from .sphereface2 import SphereFace2
from .subcenterloss import SubCenterLoss

__all__ = ['build_loss']

_LOG = logging.getLogger('ppvector')
_BUILT = {cls.__name__: cls for cls in (AAMLoss, AMLoss, ARMLoss, CELoss, SphereFace2, SubCenterLoss)}
_REFERENCE_ONLY = frozenset(('TripletAngularMarginLoss',))


def build_loss(configs):
    conf = configs.loss_conf
    name, kwargs = conf.get('loss', 'AAMLoss'), dict(conf.get('loss_args', {}) or {})
    cls = _BUILT.get(name)
    if cls is None:
        if name in _REFERENCE_ONLY:
            raise NotImplementedError(f'{name} is not built on the HIP engine yet ({", ".join(sorted(_BUILT))} are)')
        raise AttributeError(f"module '{__name__}' has no attribute '{name}'")
    criterion = cls(**kwargs)
    _LOG.info('成功创建损失函数：%s，参数为：%s', name, kwargs)
    return criterion
