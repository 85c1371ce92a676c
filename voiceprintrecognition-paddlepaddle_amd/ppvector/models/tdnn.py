"""x-vector TDNN backbone on the MI355X engine (ppvector/models/tdnn.py:9-68).

Five un-padded Conv1D (k5 d1, k3 d2, k3 d3, k1, k1) with ReLU then BN, ASP with global context,
BN, Linear (Paddle layout [in, out]), BN.  Parameter containers + one fused launch graph
(csrc/ecapa.hip: vp_tdnn_fwd).
"""
import math

import torch
from torch import nn

from ppvector.models.engine import EngineMixin, TdnnEngine
from ppvector.models.pooling import AttentiveStatisticsPooling
from ppvector.models.utils import BatchNorm1d, _BNParams, _ConvParams


class _LinearParams(nn.Module):
    """paddle.nn.Linear stand-in: weight [in, out]."""

    def __init__(self, in_features, out_features):
        super().__init__()
        w = torch.empty(in_features, out_features)
        nn.init.xavier_uniform_(w)
        self.weight = nn.Parameter(w)
        bound = 1.0 / math.sqrt(in_features)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))


class TDNN(EngineMixin, nn.Module):
    _bf16_trained_score_err = '1.7e-3'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = TdnnEngine

    def __init__(self, input_size, channels=512, embd_dim=192, pooling_type="ASP"):
        super().__init__()
        self.input_size, self.channels, self.embd_dim = input_size, channels, embd_dim
        self.td_layer1 = _ConvParams(input_size, channels, 5)
        self.bn1 = _BNParams(channels)
        self.td_layer2 = _ConvParams(channels, channels, 3)
        self.bn2 = _BNParams(channels)
        self.td_layer3 = _ConvParams(channels, channels, 3)
        self.bn3 = _BNParams(channels)
        self.td_layer4 = _ConvParams(channels, channels, 1)
        self.bn4 = _BNParams(channels)
        self.td_layer5 = _ConvParams(channels, channels, 1)
        if pooling_type == "ASP":
            self.pooling = AttentiveStatisticsPooling(channels, attention_channels=128)
            self.bn5 = BatchNorm1d(channels * 2)
            self.linear = _LinearParams(channels * 2, embd_dim)
            self.bn6 = BatchNorm1d(embd_dim)
        elif pooling_type in ("SAP", "TAP", "TSP"):
            raise NotImplementedError(f'pooling_type {pooling_type} is not built on the HIP engine (ASP is)')
        else:
            raise Exception(f'没有{pooling_type}池化层！')

    def _train_forward(self, x):
        """Training mode: batch-statistics BatchNorm, autograd through libvpmi's backward entry points (f32 engine)."""
        from ppvector.train.tdnn_train import tdnn_forward_train
        if not x.is_cuda:
            from ppvector import _native as N
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return tdnn_forward_train(self, x.float().contiguous())
