"""ResNetSE backbone on the MI355X engine.

Same constructor surface, ``embd_dim`` attribute and state-dict keys as ppvector/models/resnet_se.py
(SEBottleneck :8-45, SELayer :48-63, ResNetSE :66-139): ``conv1``/``bn1``, ``layer{1..4}.{j}.conv{1,2,3}`` /
``bn{1,2,3}`` / ``se.fc.{0,2}`` (Paddle Linear, weight [in, out]) / ``downsample.{0,1}``, ``pooling.*``,
``bn2.norm``, ``linear``, ``bn3.norm``.  The modules are parameter containers; ``forward`` runs the whole
graph through libvpmi (csrc/resnet_se.hip: vp_resnetse_fwd).
"""
import math

import torch
from torch import nn

from ppvector.models.campplus import _ConvNd
from ppvector.models.engine import EngineMixin, ResNetSEEngine
from ppvector.models.pooling import (AttentiveStatisticsPooling, SelfAttentivePooling, TemporalAveragePooling,
                                     TemporalStatisticsPooling)
from ppvector.models.utils import BatchNorm1d, _BNParams


class _LinearParams(nn.Module):
    """paddle.nn.Linear stand-in: weight [in, out] + bias."""

    def __init__(self, in_features, out_features):
        super().__init__()
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(in_features, out_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))


class SELayer(nn.Module):
    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(_LinearParams(channel, channel // reduction), nn.ReLU(),
                                _LinearParams(channel // reduction, channel), nn.Sigmoid())


class SEBottleneck(nn.Module):
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, downsample=None, reduction=8):
        super().__init__()
        self.conv1 = _ConvNd(inplanes, planes, 1, 1)
        self.bn1 = _BNParams(planes)
        self.conv2 = _ConvNd(planes, planes, 3, 3)
        self.bn2 = _BNParams(planes)
        self.conv3 = _ConvNd(planes, planes * self.expansion, 1, 1)
        self.bn3 = _BNParams(planes * self.expansion)
        self.relu = nn.ReLU()
        self.se = SELayer(planes * self.expansion, reduction)
        self.downsample = downsample
        self.stride = stride


class ResNetSE(EngineMixin, nn.Module):
    _bf16_trained_score_err = '4.0e-2'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = ResNetSEEngine

    def __init__(self, input_size, layers=[3, 4, 6, 3], num_filters=[32, 64, 128, 256], embd_dim=192,
                 pooling_type="ASP"):
        super().__init__()
        self.input_size = input_size
        self.inplanes = num_filters[0]
        self.embd_dim = embd_dim
        self.conv1 = _ConvNd(1, num_filters[0], 3, 3)
        self.bn1 = _BNParams(num_filters[0])
        self.relu = nn.ReLU()
        self.layer1 = self._make_layer(SEBottleneck, num_filters[0], layers[0])
        self.layer2 = self._make_layer(SEBottleneck, num_filters[1], layers[1], stride=(2, 2))
        self.layer3 = self._make_layer(SEBottleneck, num_filters[2], layers[2], stride=(2, 2))
        self.layer4 = self._make_layer(SEBottleneck, num_filters[3], layers[3], stride=(2, 2))
        cat_channels = num_filters[3] * SEBottleneck.expansion * (input_size // 8)
        if pooling_type == "ASP":
            self.pooling = AttentiveStatisticsPooling(cat_channels, attention_channels=128)
            self.bn2 = BatchNorm1d(cat_channels * 2)
            self.linear = _LinearParams(cat_channels * 2, embd_dim)
            self.bn3 = BatchNorm1d(embd_dim)
        elif pooling_type == "SAP":
            self.pooling = SelfAttentivePooling(cat_channels, 128)
        elif pooling_type == "TAP":
            self.pooling = TemporalAveragePooling()
        elif pooling_type == "TSP":
            self.pooling = TemporalStatisticsPooling()
        else:
            raise Exception(f'没有{pooling_type}池化层！')

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_ConvNd(self.inplanes, planes * block.expansion, 1, 1),
                                       _BNParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _train_forward(self, x):
        """Training mode: batch-statistics BatchNorm, autograd through libvpmi's backward entry points (f32 engine)."""
        from ppvector import _native as N
        from ppvector.train.resnetse_train import resnetse_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return resnetse_forward_train(self, x.float().contiguous())
