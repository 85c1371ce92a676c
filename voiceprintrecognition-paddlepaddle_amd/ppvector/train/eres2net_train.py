"""Training-mode forward of ERes2Net (ppvector/models/eres2net.py:239-263) through the autograd functions of functions.py.
Activations are (B*T*F, C) position-major.  Chunk splitting / concatenation, the residual and hand-off adds and the final
(C, F) flatten are tensor slicing, torch.cat, `+` and a permute; every conv / BatchNorm / activation / AFF / pooling runs in
libvpmi.  Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim)."""
import torch

from ppvector.train.functions import Act, AffCombine, Conv2dBlock, ConvBlock, TimeStats


def _bn(p):
    return p.weight, p.bias, p._mean, p._variance


def _cb(x, conv, bn, B, T, F, act=None, stride=1):
    args = _bn(bn) if bn is not None else (None, None, None, None)
    cfg = dict(B=B, T=T, F=F, act=act, stride=stride)
    if bn is not None:
        cfg.update(momentum=bn.momentum, eps=bn.eps)
    return Conv2dBlock.apply(x, conv.weight, conv.bias, *args, cfg)


def aff(m, x, y, B, T, F):
    la = m.local_att
    a = _cb(torch.cat((x, y), dim=1), la[0], la[1], B, T, F, act='silu')
    t = _cb(a, la[3], la[4], B, T, F, act='tanh')
    return AffCombine.apply(t, x, y)


def block(b, x, B, T, F):
    s = b.stride
    out = _cb(x, b.conv1, b.bn1, B, T, F, act='hardtanh', stride=s)
    To, Fo = (T - 1) // s + 1, (F - 1) // s + 1
    spx = torch.split(out, b.width, dim=1)
    outs, sp = [], None
    for i in range(b.nums):
        if i == 0:
            sp = spx[0]
        elif b.use_aff:
            sp = aff(b.fuse_models[i - 1], sp, spx[i], B, To, Fo)
        else:
            sp = sp + spx[i]
        sp = _cb(sp, b.convs[i], b.bns[i], B, To, Fo, act='hardtanh')
        outs.append(sp)
    out = _cb(torch.cat(outs, dim=1), b.conv3, b.bn3, B, To, Fo)
    res = x
    if len(b.shortcut) > 0:
        res = _cb(x, b.shortcut[0], b.shortcut[1], B, T, F, stride=s)
    return Act.apply(out + res, 'hardtanh'), To, Fo


def eres2net_forward_train(m, feats):
    B, T, F = feats.shape
    x = torch.zeros((B * T * F, 4), dtype=torch.float32, device=feats.device)       # single input channel padded to 4
    x[:, 0] = feats.reshape(-1)
    w = m.conv1.weight
    w4 = torch.cat([w, torch.zeros((w.shape[0], 3, 3, 3), dtype=w.dtype, device=w.device)], dim=1)
    x = Conv2dBlock.apply(x, w4, m.conv1.bias, *_bn(m.bn1), dict(B=B, T=T, F=F, act='relu', momentum=m.bn1.momentum, eps=m.bn1.eps))
    stages, dims = [], []
    for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
        for b in layer:
            x, T, F = block(b, x, B, T, F)
        stages.append(x)
        dims.append((T, F))
    low = stages[0]
    for k, (dn, fm) in enumerate(((m.layer1_downsample, m.fuse_mode12), (m.layer2_downsample, m.fuse_mode123),
                                  (m.layer3_downsample, m.fuse_mode1234))):
        Tl, Fl = dims[k]
        ds = _cb(low, dn, None, B, Tl, Fl, stride=2)
        Th, Fh = dims[k + 1]
        low = aff(fm, stages[k + 1], ds, B, Th, Fh)
    T, F = dims[3]
    Cc = low.shape[1]
    v = low.reshape(B, T, F, Cc).permute(0, 1, 3, 2).reshape(B * T, Cc * F)          # TSTP flattens (C, F): index c*F + f
    stats = TimeStats.apply(v, B, T, True)
    return ConvBlock.apply(stats, m.seg_1.weight.t().unsqueeze(2), m.seg_1.bias, None, None, None, None, None, dict(B=B, T=1))
