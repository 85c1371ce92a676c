"""build_loss registry (ppvector/loss/__init__.py:16-22): class by ``configs.loss_conf.loss``,
kwargs from ``configs.loss_conf.loss_args``."""
import importlib
import logging

from .aamloss import AAMLoss

logger = logging.getLogger('ppvector')

__all__ = ['build_loss']

_NOT_BUILT = ('AMLoss', 'ARMLoss', 'CELoss', 'SphereFace2', 'SubCenterLoss', 'TripletAngularMarginLoss')


def build_loss(configs):
    use_loss = configs.loss_conf.get('loss', 'AAMLoss')
    loss_args = configs.loss_conf.get('loss_args', {})
    los = importlib.import_module(__name__)
    if not hasattr(los, use_loss):
        if use_loss in _NOT_BUILT:
            raise NotImplementedError(f'{use_loss} is not built on the HIP engine yet (AAMLoss is)')
        raise AttributeError(f"module '{__name__}' has no attribute '{use_loss}'")
    loss = getattr(los, use_loss)(**loss_args)
    logger.info(f'成功创建损失函数：{use_loss}，参数为：{loss_args}')
    return loss
