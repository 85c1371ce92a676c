"""Oracle (CPU, PyTorch fp32/fp64): ERes2Net forward.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional restatement over the reference's Paddle parameter names of
  ppvector/models/eres2net.py:12-20    ReLU = Hardtanh(0, 20)
  ppvector/models/eres2net.py:33-53    AFF: 1x1(2C -> C/4) -> BN -> SiLU -> 1x1(C/4 -> C) -> BN; att = 1 + tanh; x*att + y*(2 - att)
  ppvector/models/eres2net.py:56-108   BasicBlockERes2Net (1x1(stride) -> [3x3 on width-chunks, sp += next chunk] -> 1x1, + shortcut)
  ppvector/models/eres2net.py:111-169  BasicBlockERes2Net_diff_AFF (the chunk hand-off is an AFF instead of a sum)
  ppvector/models/eres2net.py:172-263  ERes2Net: conv3x3(1->m) + BN + ReLU (plain F.relu :242), 4 stages (strides 1,2,2,2),
                                       3x3 stride-2 down-sampling convs + AFF fusion of the stage outputs, TSTP, Linear
  ppvector/models/pooling.py:128-146   TemporalStatsPool: mean, sqrt(var_unbiased + 1e-8), flattened (C, F), concatenated
nn.Linear weights are [in, out]; every conv has a bias.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle.campplus import _bn, _conv
from oracle.models import _bn_keys


def _ht(x):
    return torch.clamp(x, 0.0, 20.0)


def _c2d(x, p, pre, stride=1, padding=0):
    return F.conv2d(x, p[pre + 'weight'], p[pre + 'bias'], stride=stride, padding=padding)


def aff(x, y, p, pre, training=False):
    a = torch.cat((x, y), dim=1)
    a = _bn(_c2d(a, p, pre + 'local_att.0.'), p, pre + 'local_att.1.', training)
    a = a * torch.sigmoid(a)
    a = _bn(_c2d(a, p, pre + 'local_att.3.'), p, pre + 'local_att.4.', training)
    att = 1.0 + torch.tanh(a)
    return x * att + y * (2.0 - att)


def block(x, p, pre, stride, width, scale, fuse, training=False):
    out = _ht(_bn(_c2d(x, p, pre + 'conv1.', stride=stride), p, pre + 'bn1.', training))
    spx = torch.split(out, width, dim=1)
    outs = []
    sp = None
    for i in range(scale):
        if i == 0:
            sp = spx[0]
        elif fuse:
            sp = aff(sp, spx[i], p, f'{pre}fuse_models.{i - 1}.', training)
        else:
            sp = sp + spx[i]
        sp = _ht(_bn(_c2d(sp, p, f'{pre}convs.{i}.', padding=1), p, f'{pre}bns.{i}.', training))
        outs.append(sp)
    out = _bn(_c2d(torch.cat(outs, dim=1), p, pre + 'conv3.'), p, pre + 'bn3.', training)
    if (pre + 'shortcut.0.weight') in p:
        res = _bn(_c2d(x, p, pre + 'shortcut.0.', stride=stride), p, pre + 'shortcut.1.', training)
    else:
        res = x
    return _ht(out + res)


def tstp(x):
    mean = x.mean(dim=-1)
    std = torch.sqrt(x.var(dim=-1, unbiased=True) + 1e-8)
    return torch.cat((mean.flatten(1), std.flatten(1)), dim=1)


def second_embedding(p, embed_a, training=False):
    """two_emb_layer=True (eres2net.py:255-260, :455-460): relu -> seg_bn_1 (BatchNorm1D) -> seg_2 (Linear); the parameter set
    carries 'seg_2.weight' only for such a model."""
    if 'seg_2.weight' not in p:
        return embed_a
    return _bn(F.relu(embed_a), p, 'seg_bn_1.', training) @ p['seg_2.weight'] + p['seg_2.bias']


def eres2net_forward(p, x, num_blocks=(3, 4, 6, 3), m_channels=32, expansion=2, base_width=32, scale=2, taps=None, training=False):
    """ERes2Net.forward (eres2net.py:239-263), TSTP; two_emb_layer when p holds seg_2 (second_embedding).  x (B, T, F) -> (B, embd)."""
    x = x.transpose(1, 2).unsqueeze(1)
    out = F.relu(_bn(_c2d(x, p, 'conv1.', padding=1), p, 'bn1.', training))
    stage = []
    for li, (n, mult) in enumerate(zip(num_blocks, (1, 2, 4, 8)), start=1):
        planes = m_channels * mult
        width = int(math.floor(planes * (base_width / 64.0)))
        for bi in range(n):
            out = block(out, p, f'layer{li}.{bi}.', (1 if li == 1 else 2) if bi == 0 else 1, width, scale, li >= 3, training)
        stage.append(out)
    o1, o2, o3, o4 = stage
    f12 = aff(o2, _c2d(o1, p, 'layer1_downsample.', stride=2, padding=1), p, 'fuse_mode12.', training)
    f123 = aff(o3, _c2d(f12, p, 'layer2_downsample.', stride=2, padding=1), p, 'fuse_mode123.', training)
    f1234 = aff(o4, _c2d(f123, p, 'layer3_downsample.', stride=2, padding=1), p, 'fuse_mode1234.', training)
    if taps is not None:
        taps.update(o1=o1, o2=o2, o3=o3, o4=o4, f12=f12, f123=f123, f1234=f1234)
    stats = tstp(f1234)
    return second_embedding(p, stats @ p['seg_1.weight'] + p['seg_1.bias'], training)


def eres2netv2_forward(p, x, num_blocks=(3, 4, 6, 3), m_channels=32, expansion=2, base_width=26, scale=2, training=False):
    """ERes2NetV2.forward (eres2net.py:441-462): same blocks with base_width 26 (chunk widths 13 / 26 / 52 / 104), ONE bottom-up
    fusion (layer3_ds + fuse34), TSTP, Linear."""
    x = x.transpose(1, 2).unsqueeze(1)
    out = F.relu(_bn(_c2d(x, p, 'conv1.', padding=1), p, 'bn1.', training))
    stage = []
    for li, (n, mult) in enumerate(zip(num_blocks, (1, 2, 4, 8)), start=1):
        planes = m_channels * mult
        width = int(math.floor(planes * (base_width / 64.0)))
        for bi in range(n):
            out = block(out, p, f'layer{li}.{bi}.', (1 if li == 1 else 2) if bi == 0 else 1, width, scale, li >= 3, training)
        stage.append(out)
    f34 = aff(stage[3], _c2d(stage[2], p, 'layer3_ds.', stride=2, padding=1), p, 'fuse34.', training)
    return second_embedding(p, tstp(f34) @ p['seg_1.weight'] + p['seg_1.bias'], training)


def _aff_params(p, pre, channels, rng, randomize_stats, r=4):
    inter = channels // r
    p.update(_conv(pre + 'local_att.0.', (inter, channels * 2, 1, 1), rng)); p.update(_bn_keys(pre + 'local_att.1.', inter, rng, randomize_stats))
    p.update(_conv(pre + 'local_att.3.', (channels, inter, 1, 1), rng)); p.update(_bn_keys(pre + 'local_att.4.', channels, rng, randomize_stats))


def eres2net_params(input_size=80, embd_dim=192, num_blocks=(3, 4, 6, 3), m_channels=32, mul_channel=1, expansion=2, base_width=32,
                    scale=2, seed=1000, randomize_stats=True, dtype=torch.float32, v2=False, two_emb_layer=False):
    """Random ERes2Net parameters keyed with the reference's Paddle names (configs/eres2net.yml: m_channels 32, embd 192)."""
    rng = np.random.RandomState(seed)
    p = {}
    p.update(_conv('conv1.', (m_channels, 1, 3, 3), rng)); p.update(_bn_keys('bn1.', m_channels, rng, randomize_stats))
    inpl = m_channels
    for li, (n, mult) in enumerate(zip(num_blocks, (1, 2, 4, 8)), start=1):
        planes = m_channels * mult
        width = int(math.floor(planes * (base_width / 64.0)))
        for bi in range(n):
            pre = f'layer{li}.{bi}.'
            stride = (1 if li == 1 else 2) if bi == 0 else 1
            p.update(_conv(pre + 'conv1.', (width * scale, inpl, 1, 1), rng)); p.update(_bn_keys(pre + 'bn1.', width * scale, rng, randomize_stats))
            for i in range(scale):
                p.update(_conv(f'{pre}convs.{i}.', (width, width, 3, 3), rng)); p.update(_bn_keys(f'{pre}bns.{i}.', width, rng, randomize_stats))
            if li >= 3:
                for j in range(scale - 1):
                    _aff_params(p, f'{pre}fuse_models.{j}.', width, rng, randomize_stats)
            p.update(_conv(pre + 'conv3.', (planes * expansion, width * scale, 1, 1), rng)); p.update(_bn_keys(pre + 'bn3.', planes * expansion, rng, randomize_stats))
            if stride != 1 or inpl != planes * expansion:
                p.update(_conv(pre + 'shortcut.0.', (planes * expansion, inpl, 1, 1), rng))
                p.update(_bn_keys(pre + 'shortcut.1.', planes * expansion, rng, randomize_stats))
            inpl = planes * expansion
    m = m_channels * mul_channel
    if v2:
        p.update(_conv('layer3_ds.', (m * 16, m * 8, 3, 3), rng))
        _aff_params(p, 'fuse34.', m * 16, rng, randomize_stats)
    else:
      p.update(_conv('layer1_downsample.', (m * 4, m * 2, 3, 3), rng))
      p.update(_conv('layer2_downsample.', (m * 8, m * 4, 3, 3), rng))
      p.update(_conv('layer3_downsample.', (m * 16, m * 8, 3, 3), rng))
      _aff_params(p, 'fuse_mode12.', m * 4, rng, randomize_stats)
      _aff_params(p, 'fuse_mode123.', m * 8, rng, randomize_stats)
      _aff_params(p, 'fuse_mode1234.', m * 16, rng, randomize_stats)
    stats_dim = int(input_size / 8) * m_channels * 8
    fin = stats_dim * expansion * 2
    b = 1.0 / math.sqrt(fin)
    p['seg_1.weight'] = rng.uniform(-b, b, (fin, embd_dim)) * math.sqrt(3.0)
    p['seg_1.bias'] = rng.uniform(-b, b, embd_dim)
    if two_emb_layer:                                   # drawn after everything else: the other keys keep their values
        p.update(_bn_keys('seg_bn_1.', embd_dim, rng, randomize_stats))
        b2 = 1.0 / math.sqrt(embd_dim)
        p['seg_2.weight'] = rng.uniform(-b2, b2, (embd_dim, embd_dim)) * math.sqrt(3.0)
        p['seg_2.bias'] = rng.uniform(-b2, b2, embd_dim)
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}
