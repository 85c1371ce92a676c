"""Classifier head on the MI355X engine (ppvector/models/fc.py:6-87).

'Cosine': logits = normalize(x) @ normalize(W, axis=0) with W [input_dim, num_speakers*K]
(Xavier-uniform, fc.py:31-34), computed by vp_cosine_logits_f32 (csrc/head.hip) in exact f32 on
the f32 matrix cores.  'Linear': logits = x @ output.weight + output.bias (Paddle Linear, weight [in, out]).
``num_blocks`` DenseLayer(config 'batchnorm') stages in front (fc.py:27-29, :56-71: Conv1D(k=1) on (B, C) + BatchNorm1D),
state-dict keys ``blocks.<i>.linear.{weight,bias}`` / ``blocks.<i>.nonlinear.batchnorm.{weight,bias,_mean,_variance}``.
Returns the reference's dict {"features", "logits"}.  Eval runs vp_dense_f32 (BatchNorm folded on the fly); training runs
the dense / BatchNorm / activation functions of ppvector.train.functions (batch statistics, running-stat update, backward).
"""

import torch
from torch import nn

from ppvector import _native as N
from ppvector.models.utils import _BNParams, _ConvParams


def _training(mod, *tensors):
    return mod.training and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _dense(x, w, w_is_kn, bias, n_out, act=N.VP_ACT_NONE):
    x = x.contiguous().float()
    out = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=x.device)
    ctx = N.ctx(x.device)
    N.check(N.lib().vp_dense_f32(ctx, x.data_ptr(), x.shape[1], w.data_ptr(), int(w_is_kn), None if bias is None else bias.data_ptr(),
                                 x.shape[0], n_out, x.shape[1], act, out.data_ptr(), n_out, N.stream_ptr()), ctx)
    return out


def get_nonlinear(config_str, channels):
    """fc.py:74-87: the named stages after a DenseLayer's linear map ('batchnorm' / 'batchnorm_' / 'relu')."""
    nonlinear = nn.Sequential()
    for name in config_str.split('-'):
        if name == 'relu':
            nonlinear.add_module('relu', nn.ReLU())
        elif name in ('batchnorm', 'batchnorm_'):
            nonlinear.add_module('batchnorm', _BNParams(channels))
        elif name == 'prelu':
            raise NotImplementedError('prelu is not built on the HIP engine (no shipped config uses it)')
        else:
            raise ValueError('Unexpected module ({}).'.format(name))
    return nonlinear


class DenseLayer(nn.Module):
    def __init__(self, in_channels, out_channels, config_str='batchnorm-relu'):
        super().__init__()
        self.linear = _ConvParams(in_channels, out_channels, 1)
        self.nonlinear = get_nonlinear(config_str, out_channels)

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError('DenseLayer on the HIP engine takes (B, C) embeddings (its use in fc.py:44-45)')
        if not x.is_cuda:
            raise N.VpmiError('DenseLayer needs GPU tensors: the engine has no CPU fallback')
        w, b = self.linear.weight, self.linear.bias
        if _training(self, x, w):
            from ppvector.train.functions import Act, BNRows, Dense
            y = Dense.apply(x.float(), w.reshape(w.shape[0], -1).t(), b)
            for name, stage in self.nonlinear.named_children():
                y = Act.apply(y, 'relu') if name == 'relu' else BNRows.apply(y, stage.weight, stage.bias, stage._mean, stage._variance,
                                                                             stage.momentum, stage.eps)
            return y
        y = _dense(x, w.detach().float().reshape(w.shape[0], -1).contiguous(), 0, None if b is None else b.detach().float(), w.shape[0])
        lib, ctx = N.lib(), N.ctx(x.device)
        for name, stage in self.nonlinear.named_children():
            if name == 'relu':
                N.check(lib.vp_act_f32(ctx, N.VP_ACT_RELU, y.data_ptr(), y.numel(), y.data_ptr(), N.stream_ptr()), ctx)
            else:
                scale, shift = stage.folded()
                N.check(lib.vp_affine_rows_f32(ctx, y.data_ptr(), y.shape[1], scale.data_ptr(), shift.data_ptr(), y.shape[0], y.shape[1],
                                               y.data_ptr(), y.shape[1], 0, N.stream_ptr()), ctx)
        return y


class _LinearParams(nn.Module):
    """paddle.nn.Linear stand-in: weight [in, out], bias [out]."""

    def __init__(self, in_features, out_features):
        super().__init__()
        w = torch.empty(in_features, out_features)
        nn.init.xavier_uniform_(w)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(out_features))


class CosineHeadOutputs(dict):
    """The reference's {"features", "logits"} dict (fc.py:53) whose (B, C) cosine logits exist only if somebody reads them: the
    evaluation-mode forward of the 'Cosine' head defers the GEMM, so that AAMLoss can run head + loss class-tiled in one pass over the
    weights (csrc/head_tiled.hip: no (B, C) tensor -- 102 MB for 200 000 classes x 128 utterances) while `outputs["logits"]` still
    works for every other consumer."""

    def __init__(self, features, x, W, ws, training=False):
        super().__init__(features=features)
        self.x, self.W, self._ws, self.training = x, W, ws, training
        self.pred = None                  # (B,) int32 predictions, set by a criterion that ran head + loss class-tiled

    def _logits(self):
        if not dict.__contains__(self, 'logits') and self.training:
            from ppvector.train.functions import CosineLogits       # training: logits with their backward (csrc/head.hip)
            dict.__setitem__(self, 'logits', CosineLogits.apply(self.x, self.W))
        if not dict.__contains__(self, 'logits'):
            x, W = self.x, self.W
            B, D = x.shape
            Cn = W.shape[1]
            lib, ctx = N.lib(), N.ctx(x.device)
            logits = torch.empty((B, Cn), dtype=torch.float32, device=x.device)
            ws = self._ws.get(lib.vp_cosine_logits_workspace_bytes(B, D, Cn), x.device)
            N.check(lib.vp_cosine_logits_f32(ctx, x.data_ptr(), W.data_ptr(), B, D, Cn, logits.data_ptr(),
                                             ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
            dict.__setitem__(self, 'logits', logits)
        return dict.__getitem__(self, 'logits')

    def __getitem__(self, k):
        return self._logits() if k == 'logits' else dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self._logits() if k == 'logits' else dict.get(self, k, default)

    def __contains__(self, k):
        return k == 'logits' or dict.__contains__(self, k)

    def keys(self):
        return ['features', 'logits']

    def items(self):
        return [('features', dict.__getitem__(self, 'features')), ('logits', self._logits())]

    def values(self):
        return [v for _, v in self.items()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 2


class SpeakerIdentification(nn.Module):
    def __init__(self, input_dim, num_speakers, classifier_type='Cosine', K=1, num_blocks=0, inter_dim=512):
        super().__init__()
        self.classifier_type = classifier_type
        self.blocks = nn.ModuleList()
        for _ in range(num_blocks):
            self.blocks.append(DenseLayer(input_dim, inter_dim, config_str='batchnorm'))
            input_dim = inter_dim
        if self.classifier_type == 'Cosine':
            w = torch.empty(input_dim, num_speakers * K)
            nn.init.xavier_uniform_(w)
            self.weight = nn.Parameter(w)
        elif self.classifier_type == 'Linear':
            self.output = _LinearParams(input_dim, num_speakers)
        else:
            raise ValueError(f'不支持该输出层：{self.classifier_type}')
        self._ws = N.Workspace()

    def forward(self, features):
        x = features
        if not x.is_cuda:
            raise N.VpmiError('SpeakerIdentification needs GPU tensors: the engine has no CPU fallback')
        for layer in self.blocks:
            x = layer(x)
        if self.classifier_type == 'Linear':
            w, b = self.output.weight, self.output.bias
            if _training(self, x, w):
                from ppvector.train.functions import Dense
                logits = Dense.apply(x.float(), w, b)
            else:
                logits = _dense(x, w.detach().float().contiguous(), 1, b.detach().float(), w.shape[1])
            return {"features": features, "logits": logits}
        if _training(self, x, self.weight):
            # training: the logits (CosineLogits, with their backward) are formed only if somebody reads outputs["logits"]; AAMLoss
            # takes embeddings + weights instead and runs head + loss + both gradients class-tiled (functions.HeadLoss)
            return CosineHeadOutputs(features, x.float(), self.weight, self._ws, training=True)
        x = x.contiguous().float()
        W = self.weight.detach().contiguous().float()
        return CosineHeadOutputs(features, x, W, self._ws)
