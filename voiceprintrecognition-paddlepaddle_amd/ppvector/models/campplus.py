"""CAM++ backbone on the MI355X engine.

Same constructor surface, ``embd_dim`` attribute and state-dict keys as ppvector/models/campplus.py
(CAMPPlus :284-335, FCM :246-281, BasicResBlock :211-243, TDNNLayer :38-64, CAMLayer :67-106,
CAMDenseTDNNLayer/Block :109-173, TransitLayer :176-189, DenseLayer :192-208).  As in the reference,
every conv carries a bias (its ``bias=`` arguments are never forwarded) and 'batchnorm_' is a plain
affine BatchNorm.  The modules are parameter containers; ``forward`` runs the whole graph through
libvpmi (csrc/campplus.hip: vp_campplus_fwd).
"""
import math

import torch
from torch import nn

from ppvector.models.engine import CamppEngine, EngineMixin
from ppvector.models.utils import _BNParams


class _ConvNd(nn.Module):
    """nn.Conv1D / nn.Conv2D stand-in: weight (out, in, *kernel) + bias."""

    def __init__(self, in_channels, out_channels, *kernel):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *kernel))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = in_channels * int(torch.tensor(kernel).prod()) if kernel else in_channels
        bound = 1.0 / math.sqrt(fan_in)
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))


def get_nonlinear(config_str, channels):
    nonlinear = nn.Sequential()
    for name in config_str.split('-'):
        if name == 'relu':
            nonlinear.add_module('relu', nn.ReLU())
        elif name in ('batchnorm', 'batchnorm_'):
            nonlinear.add_module('batchnorm', _BNParams(channels))
        elif name == 'prelu':
            raise NotImplementedError('prelu is not fused on the HIP engine')
        else:
            raise ValueError('Unexpected module ({}).'.format(name))
    return nonlinear


class TDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False,
                 config_str='batchnorm-relu'):
        super().__init__()
        if padding < 0:
            assert kernel_size % 2 == 1, 'Expect equal paddings, but got even kernel size ({})'.format(kernel_size)
            padding = (kernel_size - 1) // 2 * dilation
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.linear = _ConvNd(in_channels, out_channels, kernel_size)
        self.nonlinear = get_nonlinear(config_str, out_channels)


class CAMLayer(nn.Module):
    def __init__(self, bn_channels, out_channels, kernel_size, stride, padding, dilation, bias, reduction=2):
        super().__init__()
        self.kernel_size, self.dilation = kernel_size, dilation
        self.linear_local = _ConvNd(bn_channels, out_channels, kernel_size)
        self.linear1 = _ConvNd(bn_channels, bn_channels // reduction, 1)
        self.relu = nn.ReLU()
        self.linear2 = _ConvNd(bn_channels // reduction, out_channels, 1)
        self.sigmoid = nn.Sigmoid()


class CAMDenseTDNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1, bias=False,
                 config_str='batchnorm-relu', memory_efficient=False):
        super().__init__()
        assert kernel_size % 2 == 1, 'Expect equal paddings, but got even kernel size ({})'.format(kernel_size)
        padding = (kernel_size - 1) // 2 * dilation
        self.nonlinear1 = get_nonlinear(config_str, in_channels)
        self.linear1 = _ConvNd(in_channels, bn_channels, 1)
        self.nonlinear2 = get_nonlinear(config_str, bn_channels)
        self.cam_layer = CAMLayer(bn_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, bias=bias)


class CAMDenseTDNNBlock(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, bn_channels, kernel_size, stride=1, dilation=1,
                 bias=False, config_str='batchnorm-relu', memory_efficient=False):
        super().__init__()
        self.num_layers = num_layers
        for i in range(num_layers):
            self.add_module('tdnnd%d' % (i + 1),
                            CAMDenseTDNNLayer(in_channels=in_channels + i * out_channels, out_channels=out_channels,
                                              bn_channels=bn_channels, kernel_size=kernel_size, stride=stride,
                                              dilation=dilation, bias=bias, config_str=config_str))


class TransitLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, config_str='batchnorm-relu'):
        super().__init__()
        self.nonlinear = get_nonlinear(config_str, in_channels)
        self.linear = _ConvNd(in_channels, out_channels, 1)


class DenseLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=False, config_str='batchnorm-relu'):
        super().__init__()
        self.linear = _ConvNd(in_channels, out_channels, 1)
        self.nonlinear = get_nonlinear(config_str, out_channels)


class StatsPool(nn.Module):
    pass


class BasicResBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1 = _ConvNd(in_planes, planes, 3, 3)
        self.bn1 = _BNParams(planes)
        self.conv2 = _ConvNd(planes, planes, 3, 3)
        self.bn2 = _BNParams(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(_ConvNd(in_planes, self.expansion * planes, 1, 1),
                                          _BNParams(self.expansion * planes))


class FCM(nn.Module):
    def __init__(self, block=BasicResBlock, num_blocks=[2, 2], m_channels=32, feat_dim=80):
        super().__init__()
        self.in_planes = m_channels
        self.conv1 = _ConvNd(1, m_channels, 3, 3)
        self.bn1 = _BNParams(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], stride=2)
        self.layer2 = self._make_layer(block, m_channels, num_blocks[0], stride=2)
        self.conv2 = _ConvNd(m_channels, m_channels, 3, 3)
        self.bn2 = _BNParams(m_channels)
        self.out_channels = m_channels * (math.ceil(feat_dim / 8))

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)


class CAMPPlus(EngineMixin, nn.Module):
    _bf16_trained_score_err = '1.3e-2'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = CamppEngine

    def __init__(self, input_size, embd_dim=512, growth_rate=32, bn_size=4, init_channels=128,
                 config_str='batchnorm-relu', memory_efficient=True):
        super().__init__()
        if config_str != 'batchnorm-relu':
            raise NotImplementedError("only config_str='batchnorm-relu' is built on the HIP engine")
        self.input_size, self.embd_dim = input_size, embd_dim
        self.growth_rate, self.bn_size, self.init_channels = growth_rate, bn_size, init_channels
        self.head = FCM(feat_dim=input_size)
        channels = self.head.out_channels
        self.xvector = nn.Sequential()
        self.xvector.add_module('tdnn', TDNNLayer(channels, init_channels, 5, stride=2, dilation=1, padding=-1,
                                                  config_str=config_str))
        channels = init_channels
        self.block_cfg = tuple(zip((12, 24, 16), (3, 3, 3), (1, 2, 2)))
        for i, (num_layers, kernel_size, dilation) in enumerate(self.block_cfg):
            self.xvector.add_module('block%d' % (i + 1),
                                    CAMDenseTDNNBlock(num_layers=num_layers, in_channels=channels,
                                                      out_channels=growth_rate, bn_channels=bn_size * growth_rate,
                                                      kernel_size=kernel_size, dilation=dilation, config_str=config_str))
            channels = channels + num_layers * growth_rate
            self.xvector.add_module('transit%d' % (i + 1), TransitLayer(channels, channels // 2, bias=False,
                                                                         config_str=config_str))
            channels //= 2
        self.xvector.add_module('out_nonlinear', get_nonlinear(config_str, channels))
        self.xvector.add_module('stats', StatsPool())
        self.xvector.add_module('dense', DenseLayer(channels * 2, embd_dim, config_str='batchnorm_'))

    def _train_forward(self, x):
        """Training mode: batch-statistics BatchNorm, autograd through libvpmi's backward entry points (f32 engine)."""
        from ppvector import _native as N
        from ppvector.train.campplus_train import campplus_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return campplus_forward_train(self, x.float().contiguous())
