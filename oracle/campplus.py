"""Oracle (CPU, PyTorch fp32/fp64): CAM++ forward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement over the reference's Paddle parameter names of
  ppvector/models/campplus.py:246-281  FCM (2-D ResBlocks on (B,1,F,T), stride on F only)
  ppvector/models/campplus.py:211-243  BasicResBlock
  ppvector/models/campplus.py:38-64    TDNNLayer (Conv1D k5 s2 pad 2 -> BN -> ReLU)
  ppvector/models/campplus.py:67-106   CAMLayer (+ seg_pooling: 100-frame ceil-mode average, expanded back)
  ppvector/models/campplus.py:109-173  CAMDenseTDNNLayer / Block (BN-ReLU-1x1, BN-ReLU-CAM, concat)
  ppvector/models/campplus.py:176-208  TransitLayer, DenseLayer
  ppvector/models/campplus.py:24-30    statistics_pooling (mean, unbiased std)
  ppvector/models/campplus.py:284-335  CAMPPlus
As in the reference every conv has a bias (the bias= arguments are never forwarded) and 'batchnorm_'
is an ordinary affine BatchNorm1D (campplus.py:17-18).  [3P-memory] F.avg_pool1d(ceil_mode=True) with
Paddle's default exclusive=True averages the last partial segment over its valid frames only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle.models import BN_EPS, _bn_keys

BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))      # (num_layers, kernel, dilation), campplus.py:307-308


def _bn(x, p, pre, training=False):
    w, b = p[pre + 'weight'], p[pre + 'bias']
    if training:
        dims = [0] + list(range(2, x.dim()))
        m, v = x.mean(dim=dims), x.var(dim=dims, unbiased=False)
    else:
        m, v = p[pre + '_mean'], p[pre + '_variance']
    sh = [1, -1] + [1] * (x.dim() - 2)
    return (x - m.view(sh)) / torch.sqrt(v.view(sh) + BN_EPS) * w.view(sh) + b.view(sh)


def _resblock(x, p, pre, stride, training=False):
    out = F.relu(_bn(F.conv2d(x, p[pre + 'conv1.weight'], p[pre + 'conv1.bias'], stride=(stride, 1), padding=1),
                     p, pre + 'bn1.', training))
    out = _bn(F.conv2d(out, p[pre + 'conv2.weight'], p[pre + 'conv2.bias'], padding=1), p, pre + 'bn2.', training)
    if (pre + 'shortcut.0.weight') in p:
        sc = _bn(F.conv2d(x, p[pre + 'shortcut.0.weight'], p[pre + 'shortcut.0.bias'], stride=(stride, 1)),
                 p, pre + 'shortcut.1.', training)
    else:
        sc = x
    return F.relu(out + sc)


def fcm(x, p, pre='head.', training=False):
    """x (B, F, T) -> (B, 32 * ceil(F/8), T)."""
    out = F.relu(_bn(F.conv2d(x.unsqueeze(1), p[pre + 'conv1.weight'], p[pre + 'conv1.bias'], padding=1), p, pre + 'bn1.', training))
    for layer in ('layer1.', 'layer2.'):
        out = _resblock(out, p, pre + layer + '0.', 2, training)
        out = _resblock(out, p, pre + layer + '1.', 1, training)
    out = F.relu(_bn(F.conv2d(out, p[pre + 'conv2.weight'], p[pre + 'conv2.bias'], stride=(2, 1), padding=1), p, pre + 'bn2.', training))
    B, C, Fq, T = out.shape
    return out.reshape(B, C * Fq, T)


def seg_pooling(x, seg_len=100):
    """campplus.py:96-106 ('avg'): ceil-mode, exclusive average, expanded back to T."""
    T = x.shape[-1]
    nseg = (T + seg_len - 1) // seg_len
    segs = [x[..., i * seg_len:min((i + 1) * seg_len, T)].mean(dim=-1, keepdim=True) for i in range(nseg)]
    seg = torch.cat(segs, dim=-1)
    return seg.unsqueeze(-1).expand(*seg.shape, seg_len).reshape(*seg.shape[:-1], -1)[..., :T]


def cam_layer(x, p, pre, dilation):
    y = F.conv1d(x, p[pre + 'linear_local.weight'], p[pre + 'linear_local.bias'], padding=dilation, dilation=dilation)
    ctx = x.mean(-1, keepdim=True) + seg_pooling(x)
    ctx = F.relu(F.conv1d(ctx, p[pre + 'linear1.weight'], p[pre + 'linear1.bias']))
    m = torch.sigmoid(F.conv1d(ctx, p[pre + 'linear2.weight'], p[pre + 'linear2.bias']))
    return y * m


def campplus_forward(p, x, prefix='', taps=None, training=False):
    """CAMPPlus.forward (campplus.py:331-335), eval mode.  x (B, T, F) -> (B, embd)."""
    x = fcm(x.transpose(1, 2), p, prefix + 'head.', training)
    if taps is not None:
        taps['fcm'] = x
    xv = prefix + 'xvector.'
    x = F.conv1d(x, p[xv + 'tdnn.linear.weight'], p[xv + 'tdnn.linear.bias'], stride=2, padding=2)
    x = F.relu(_bn(x, p, xv + 'tdnn.nonlinear.batchnorm.', training))
    if taps is not None:
        taps['tdnn'] = x
    for bi, (nl, k, d) in enumerate(BLOCKS, start=1):
        for li in range(1, nl + 1):
            lp = f'{xv}block{bi}.tdnnd{li}.'
            h = F.relu(_bn(x, p, lp + 'nonlinear1.batchnorm.', training))
            h = F.conv1d(h, p[lp + 'linear1.weight'], p[lp + 'linear1.bias'])
            h = F.relu(_bn(h, p, lp + 'nonlinear2.batchnorm.', training))
            x = torch.cat([x, cam_layer(h, p, lp + 'cam_layer.', d)], dim=1)
        tp = f'{xv}transit{bi}.'
        x = F.conv1d(F.relu(_bn(x, p, tp + 'nonlinear.batchnorm.', training)), p[tp + 'linear.weight'], p[tp + 'linear.bias'])
        if taps is not None:
            taps[f'transit{bi}'] = x
    x = F.relu(_bn(x, p, xv + 'out_nonlinear.batchnorm.', training))
    stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
    y = F.conv1d(stats.unsqueeze(-1), p[xv + 'dense.linear.weight'], p[xv + 'dense.linear.bias']).squeeze(-1)
    return _bn(y, p, xv + 'dense.nonlinear.batchnorm.', training)


def _conv(prefix, shape, rng):
    fan_in = int(np.prod(shape[1:]))
    bound = 1.0 / math.sqrt(fan_in)
    return {prefix + 'weight': rng.uniform(-bound, bound, shape) * math.sqrt(3.0), prefix + 'bias': rng.uniform(-bound, bound, shape[0])}


def campplus_params(input_size=80, embd_dim=192, growth_rate=32, bn_size=4, init_channels=128, seed=1000,
                    randomize_stats=True, dtype=torch.float32):
    """Random CAM++ parameters keyed with the reference's Paddle names (configs/cam++.yml: embd_dim 192)."""
    rng = np.random.RandomState(seed)
    p = {}
    m = 32
    p.update(_conv('head.conv1.', (m, 1, 3, 3), rng)); p.update(_bn_keys('head.bn1.', m, rng, randomize_stats))
    for layer in ('layer1', 'layer2'):
        for bi, stride in ((0, 2), (1, 1)):
            pre = f'head.{layer}.{bi}.'
            p.update(_conv(pre + 'conv1.', (m, m, 3, 3), rng)); p.update(_bn_keys(pre + 'bn1.', m, rng, randomize_stats))
            p.update(_conv(pre + 'conv2.', (m, m, 3, 3), rng)); p.update(_bn_keys(pre + 'bn2.', m, rng, randomize_stats))
            if stride != 1:
                p.update(_conv(pre + 'shortcut.0.', (m, m, 1, 1), rng)); p.update(_bn_keys(pre + 'shortcut.1.', m, rng, randomize_stats))
    p.update(_conv('head.conv2.', (m, m, 3, 3), rng)); p.update(_bn_keys('head.bn2.', m, rng, randomize_stats))
    ch = m * math.ceil(input_size / 8)
    p.update(_conv('xvector.tdnn.linear.', (init_channels, ch, 5), rng))
    p.update(_bn_keys('xvector.tdnn.nonlinear.batchnorm.', init_channels, rng, randomize_stats))
    ch = init_channels
    bnc = bn_size * growth_rate
    for bi, (nl, k, d) in enumerate(BLOCKS, start=1):
        for li in range(1, nl + 1):
            lp = f'xvector.block{bi}.tdnnd{li}.'
            cin = ch + (li - 1) * growth_rate
            p.update(_bn_keys(lp + 'nonlinear1.batchnorm.', cin, rng, randomize_stats))
            p.update(_conv(lp + 'linear1.', (bnc, cin, 1), rng))
            p.update(_bn_keys(lp + 'nonlinear2.batchnorm.', bnc, rng, randomize_stats))
            p.update(_conv(lp + 'cam_layer.linear_local.', (growth_rate, bnc, k), rng))
            p.update(_conv(lp + 'cam_layer.linear1.', (bnc // 2, bnc, 1), rng))
            p.update(_conv(lp + 'cam_layer.linear2.', (growth_rate, bnc // 2, 1), rng))
        ch = ch + nl * growth_rate
        p.update(_bn_keys(f'xvector.transit{bi}.nonlinear.batchnorm.', ch, rng, randomize_stats))
        p.update(_conv(f'xvector.transit{bi}.linear.', (ch // 2, ch, 1), rng))
        ch //= 2
    p.update(_bn_keys('xvector.out_nonlinear.batchnorm.', ch, rng, randomize_stats))
    p.update(_conv('xvector.dense.linear.', (embd_dim, ch * 2, 1), rng))
    p.update(_bn_keys('xvector.dense.nonlinear.batchnorm.', embd_dim, rng, randomize_stats))
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}
