"""Training-mode forward of ERes2Net / ERes2NetV2 (ppvector/models/eres2net.py:239-263, :441-462) through the autograd functions of
functions.py.
Activations are (B*T*F, C) position-major.  Chunk splitting / concatenation, the residual and hand-off adds and the final
(C, F) flatten are tensor slicing, torch.cat, `+` and a permute; every conv / BatchNorm / activation / AFF / pooling runs in
libvpmi.  Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim).
ERes2NetV2's chunk widths (13, 26 in the first two stages) are not multiples of 4, which the f32 conv / weight-gradient
kernels require: those blocks run on per-chunk zero-padded weights (13 -> 16, 26 -> 28; padded channels have zero weights,
gamma 1, beta 0 and therefore stay exactly 0 through BatchNorm and Hardtanh), built with tensor ops the tape sees, so the
gradients land in the reference-shaped parameters."""
import torch
import torch.nn.functional as TF

from ppvector.train.functions import Act, AffCombine, Conv2dBlock, ConvBlock, TimeStats
from ppvector.train.segments import cut


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
    inter = la[0].weight.shape[0]
    if inter % 4:                                              # ERes2NetV2: channels // 4 = 13, 26 -> zero-padded bottleneck
        ip = (inter + 3) // 4 * 4
        a = _cb_padded(torch.cat((x, y), dim=1), la[0], la[1], B, T, F, 'silu', 1, 1, inter, ip, False)
        t = _cb_padded(a, la[3], la[4], B, T, F, 'tanh', 1, 0, inter, ip, True)
    else:
        a = _cb(torch.cat((x, y), dim=1), la[0], la[1], B, T, F, act='silu')
        t = _cb(a, la[3], la[4], B, T, F, act='tanh')
    return AffCombine.apply(t, x, y)


def _pad_vec(v, n_chunks, w, wp, fill):
    return TF.pad(v.reshape(n_chunks, w), (0, wp - w), value=fill).reshape(-1)


def _cb_padded(x, conv, bn, B, T, F, act, stride, n_chunks, w, wp, pad_in):
    """Conv2dBlock on a weight whose OUTPUT channels are n_chunks chunks of w (padded to wp each) and, when pad_in, whose INPUT
    channels are laid out the same way.  Running statistics are updated on padded scratch buffers and copied back."""
    weight, bias = conv.weight, conv.bias
    co, ci, kf, kt = weight.shape
    wt = weight
    if pad_in:
        wt = TF.pad(wt.reshape(co, ci // w, w, kf, kt), (0, 0, 0, 0, 0, wp - w)).reshape(co, (ci // w) * wp, kf, kt)
    if n_chunks:
        wt = TF.pad(wt.reshape(n_chunks, w, wt.shape[1], kf, kt), (0, 0, 0, 0, 0, 0, 0, wp - w)).reshape(n_chunks * wp, wt.shape[1], kf, kt)
    cfg = dict(B=B, T=T, F=F, act=act, stride=stride, momentum=bn.momentum, eps=bn.eps)
    if not n_chunks:
        return Conv2dBlock.apply(x, wt, bias, bn.weight, bn.bias, bn._mean, bn._variance, cfg)
    rm, rv = _pad_vec(bn._mean, n_chunks, w, wp, 0.0).contiguous(), _pad_vec(bn._variance, n_chunks, w, wp, 1.0).contiguous()
    bp = None if bias is None else _pad_vec(bias, n_chunks, w, wp, 0.0)
    y = Conv2dBlock.apply(x, wt, bp, _pad_vec(bn.weight, n_chunks, w, wp, 1.0), _pad_vec(bn.bias, n_chunks, w, wp, 0.0), rm, rv, cfg)
    with torch.no_grad():
        bn._mean.copy_(rm.reshape(n_chunks, wp)[:, :w].reshape(-1))
        bn._variance.copy_(rv.reshape(n_chunks, wp)[:, :w].reshape(-1))
    return y


def block_padded(b, x, B, T, F):
    """block() for chunk widths that are not multiples of 4 (ERes2NetV2 stages 1-2: no AFF there)."""
    s, w, n = b.stride, b.width, b.nums
    wp = (w + 3) // 4 * 4
    if b.use_aff:
        raise NotImplementedError('padded-chunk training path: AFF blocks are not expected at these widths')
    out = _cb_padded(x, b.conv1, b.bn1, B, T, F, 'hardtanh', s, n, w, wp, False)
    To, Fo = (T - 1) // s + 1, (F - 1) // s + 1
    spx = torch.split(out, wp, dim=1)
    outs, sp = [], None
    for i in range(n):
        sp = spx[0] if i == 0 else sp + spx[i]
        sp = _cb_padded(sp, b.convs[i], b.bns[i], B, To, Fo, 'hardtanh', 1, 1, w, wp, True)
        outs.append(sp)
    out = _cb_padded(torch.cat(outs, dim=1), b.conv3, b.bn3, B, To, Fo, None, 1, 0, w, wp, True)
    res = x
    if len(b.shortcut) > 0:
        res = _cb(x, b.shortcut[0], b.shortcut[1], B, T, F, stride=s)
    return Act.apply(out + res, 'hardtanh'), To, Fo


def block(b, x, B, T, F):
    if b.width % 4:
        return block_padded(b, x, B, T, F)
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


def _stem_and_stages(m, feats):
    B, T, F = feats.shape
    x = torch.zeros((B * T * F, 4), dtype=torch.float32, device=feats.device)       # single input channel padded to 4
    x[:, 0] = feats.reshape(-1)
    w = m.conv1.weight
    w4 = torch.cat([w, torch.zeros((w.shape[0], 3, 3, 3), dtype=w.dtype, device=w.device)], dim=1)
    x = Conv2dBlock.apply(x, w4, m.conv1.bias, *_bn(m.bn1), dict(B=B, T=T, F=F, act='relu', momentum=m.bn1.momentum, eps=m.bn1.eps))
    stages, dims = [], []
    for li, layer in enumerate((m.layer1, m.layer2, m.layer3, m.layer4)):
        for b in layer:
            x, T, F = block(b, x, B, T, F)
        stages.append(x)
        dims.append((T, F))
        if li < 3:
            # backward stage boundary (train/segments.py): the stage outputs so far are all live across it -- the next stage reads the
            # last one, the bottom-up fusion reads every one.  Stage 0 of the backward then holds layer4 + the fusion + seg_1 + the head
            # (85 % of ERes2Net-large's 93.6 M gradients with the 200 k-class head): their all-reduce chunks travel while the three
            # expensive full-resolution stages are still being differentiated (BASELINE configs[4]: "grad all-reduce overlapped with backward")
            stages = list(cut(*stages))
            x = stages[-1]
    return stages, dims


def _pool_and_embed(m, low, B, T, F):
    Cc = low.shape[1]
    v = low.reshape(B, T, F, Cc).permute(0, 1, 3, 2).reshape(B * T, Cc * F)          # TSTP flattens (C, F): index c*F + f
    stats = TimeStats.apply(v, B, T, True)
    return ConvBlock.apply(stats, m.seg_1.weight.t().unsqueeze(2), m.seg_1.bias, None, None, None, None, None, dict(B=B, T=1))


def eres2netv2_forward_train(m, feats):
    """ERes2NetV2.forward (eres2net.py:441-462): ONE bottom-up fusion, layer3_ds (3x3 stride 2) + fuse34."""
    B = feats.shape[0]
    stages, dims = _stem_and_stages(m, feats)
    ds = _cb(stages[2], m.layer3_ds, None, B, *dims[2], stride=2)
    low = aff(m.fuse34, stages[3], ds, B, *dims[3])
    return _pool_and_embed(m, low, B, *dims[3])


def eres2net_forward_train(m, feats):
    B = feats.shape[0]
    stages, dims = _stem_and_stages(m, feats)
    low = stages[0]
    for k, (dn, fm) in enumerate(((m.layer1_downsample, m.fuse_mode12), (m.layer2_downsample, m.fuse_mode123),
                                  (m.layer3_downsample, m.fuse_mode1234))):
        Tl, Fl = dims[k]
        ds = _cb(low, dn, None, B, Tl, Fl, stride=2)
        Th, Fh = dims[k + 1]
        low = aff(fm, stages[k + 1], ds, B, Th, Fh)
    return _pool_and_embed(m, low, B, *dims[3])
