"""Oracle (CPU, PyTorch fp32/fp64): ResNetSE forward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement over the reference's Paddle parameter names of
  ppvector/models/resnet_se.py:8-45    SEBottleneck (1x1 -> 3x3(stride) -> 1x1, BN each, SE, residual, ReLU)
  ppvector/models/resnet_se.py:48-63   SELayer (global average over (F, T); Linear C -> C/8 -> C; sigmoid)
  ppvector/models/resnet_se.py:66-139  ResNetSE (conv3x3(1->32), layers [3,4,6,3], strides 1,2,2,2 on BOTH axes,
                                       reshape (B, C*F/8, T/8), ASP, BN, Linear, BN)
Paddle Linear weights are [in, out]; every conv has a bias.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle.campplus import _bn, _conv
from oracle.models import _bn_keys, asp, batchnorm

LAYERS = [3, 4, 6, 3]
FILTERS = [32, 64, 128, 256]


def _block(x, p, pre, stride, training=False):
    out = F.relu(_bn(F.conv2d(x, p[pre + 'conv1.weight'], p[pre + 'conv1.bias']), p, pre + 'bn1.', training))
    out = F.relu(_bn(F.conv2d(out, p[pre + 'conv2.weight'], p[pre + 'conv2.bias'], stride=stride, padding=1), p, pre + 'bn2.', training))
    out = _bn(F.conv2d(out, p[pre + 'conv3.weight'], p[pre + 'conv3.bias']), p, pre + 'bn3.', training)
    y = out.mean(dim=(2, 3))
    y = F.relu(y @ p[pre + 'se.fc.0.weight'] + p[pre + 'se.fc.0.bias'])
    y = torch.sigmoid(y @ p[pre + 'se.fc.2.weight'] + p[pre + 'se.fc.2.bias'])
    out = out * y[:, :, None, None]
    if (pre + 'downsample.0.weight') in p:
        res = _bn(F.conv2d(x, p[pre + 'downsample.0.weight'], p[pre + 'downsample.0.bias'], stride=stride), p, pre + 'downsample.1.', training)
    else:
        res = x
    return F.relu(out + res)


def resnetse_forward(p, x, prefix='', layers=LAYERS, taps=None, training=False):
    """ResNetSE.forward (resnet_se.py:121-139), pooling_type ASP, eval mode.  x (B, T, F) -> (B, embd)."""
    x = x.transpose(1, 2).unsqueeze(1)
    x = F.relu(_bn(F.conv2d(x, p[prefix + 'conv1.weight'], p[prefix + 'conv1.bias'], padding=1), p, prefix + 'bn1.', training))
    for li, n in enumerate(layers, start=1):
        for bi in range(n):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _block(x, p, f'{prefix}layer{li}.{bi}.', stride, training)
    if taps is not None:
        taps['layer4'] = x
    x = x.reshape(x.shape[0], -1, x.shape[-1])
    x = asp(x, p, prefix + 'pooling.', True, training)
    x = batchnorm(x, p, prefix + 'bn2.norm.', training)
    x = x @ p[prefix + 'linear.weight'] + p[prefix + 'linear.bias']
    return batchnorm(x, p, prefix + 'bn3.norm.', training)


def resnetse_params(input_size=80, embd_dim=192, layers=LAYERS, filters=FILTERS, seed=1000, randomize_stats=True,
                    dtype=torch.float32):
    rng = np.random.RandomState(seed)
    p = {}
    p.update(_conv('conv1.', (filters[0], 1, 3, 3), rng)); p.update(_bn_keys('bn1.', filters[0], rng, randomize_stats))
    inpl = filters[0]
    for li, (n, planes) in enumerate(zip(layers, filters), start=1):
        for bi in range(n):
            pre = f'layer{li}.{bi}.'
            stride = 2 if (li > 1 and bi == 0) else 1
            p.update(_conv(pre + 'conv1.', (planes, inpl, 1, 1), rng)); p.update(_bn_keys(pre + 'bn1.', planes, rng, randomize_stats))
            p.update(_conv(pre + 'conv2.', (planes, planes, 3, 3), rng)); p.update(_bn_keys(pre + 'bn2.', planes, rng, randomize_stats))
            p.update(_conv(pre + 'conv3.', (planes * 2, planes, 1, 1), rng)); p.update(_bn_keys(pre + 'bn3.', planes * 2, rng, randomize_stats))
            c = planes * 2
            b1, b2 = 1.0 / math.sqrt(c), 1.0 / math.sqrt(c // 8)
            p[pre + 'se.fc.0.weight'] = rng.uniform(-b1, b1, (c, c // 8)) * math.sqrt(3.0)
            p[pre + 'se.fc.0.bias'] = rng.uniform(-b1, b1, c // 8)
            p[pre + 'se.fc.2.weight'] = rng.uniform(-b2, b2, (c // 8, c)) * math.sqrt(3.0)
            p[pre + 'se.fc.2.bias'] = rng.uniform(-b2, b2, c)
            if bi == 0 and (stride != 1 or inpl != planes * 2):
                p.update(_conv(pre + 'downsample.0.', (planes * 2, inpl, 1, 1), rng))
                p.update(_bn_keys(pre + 'downsample.1.', planes * 2, rng, randomize_stats))
            inpl = planes * 2
    C = filters[3] * 2 * (input_size // 8)
    bound = 1.0 / math.sqrt(3 * C)
    p['pooling.tdnn.conv.conv.weight'] = rng.uniform(-bound, bound, (128, 3 * C, 1)) * math.sqrt(3.0)
    p['pooling.tdnn.conv.conv.bias'] = rng.uniform(-bound, bound, 128)
    p.update(_bn_keys('pooling.tdnn.norm.norm.', 128, rng, randomize_stats))
    b2 = 1.0 / math.sqrt(128)
    p['pooling.conv.conv.weight'] = rng.uniform(-b2, b2, (C, 128, 1)) * math.sqrt(3.0)
    p['pooling.conv.conv.bias'] = rng.uniform(-b2, b2, C)
    p.update(_bn_keys('bn2.norm.', 2 * C, rng, randomize_stats))
    b3 = 1.0 / math.sqrt(2 * C)
    p['linear.weight'] = rng.uniform(-b3, b3, (2 * C, embd_dim)) * math.sqrt(3.0)
    p['linear.bias'] = rng.uniform(-b3, b3, embd_dim)
    p.update(_bn_keys('bn3.norm.', embd_dim, rng, randomize_stats))
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}
