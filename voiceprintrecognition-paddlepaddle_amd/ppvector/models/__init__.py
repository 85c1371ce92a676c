"""build_model registry (ppvector/models/__init__.py:15-21): class by name from
``configs.model_conf.model``, kwargs from ``configs.model_conf.model_args``."""
import importlib
import logging

from .campplus import CAMPPlus
from .ecapa_tdnn import EcapaTdnn
from .eres2net import ERes2Net
from .resnet_se import ResNetSE
from .tdnn import TDNN

logger = logging.getLogger('ppvector')

__all__ = ['build_model']

_NOT_BUILT = ('ERes2NetV2', 'Res2Net')


def build_model(input_size, configs):
    use_model = configs.model_conf.get('model', 'CAMPPlus')
    model_args = configs.model_conf.get('model_args', {})
    mod = importlib.import_module(__name__)
    if not hasattr(mod, use_model):
        if use_model in _NOT_BUILT:
            raise NotImplementedError(f'{use_model} is not built on the HIP engine yet (EcapaTdnn, TDNN, CAMPPlus, ResNetSE and ERes2Net are)')
        raise AttributeError(f"module '{__name__}' has no attribute '{use_model}'")
    model = getattr(mod, use_model)(input_size=input_size, **model_args)
    logger.info(f'成功创建模型：{use_model}，参数为：{model_args}')
    return model
