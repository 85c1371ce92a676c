"""Training-mode forward of CAM++ (ppvector/models/campplus.py:331-335) through the autograd functions of functions.py.
Activations are position-major: (B*T*F, C) in the FCM head, (B*T, C) in the D-TDNN.  The DenseNet concatenations are
ONE buffer per dense block (functions.py: CamDenseBlockFn -- no torch.cat; VPMI_CAM_BLOCK_UNFUSED=1: torch.cat per layer); every conv /
BatchNorm / activation / context gate / pooling runs in libvpmi.  A CAMLayer (local conv + context gate) is ONE tape entry
(CamLayerFn; VPMI_CAM_LAYER_UNFUSED=1 keeps the per-op form the tests compare it with).
Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim)."""
import torch

from ppvector.train.functions import Act, BNRows, CamDenseBlockFn, CamLayerFn, Conv2dBlock, ConvBlock, SegCtx, SegScale, TimeStats

SEG_LEN = 100       # CAMLayer.seg_pooling default (campplus.py:96)


def _bn(p):
    return p.weight, p.bias, p._mean, p._variance


def _c2(x, conv, bn, B, T, F, act=None, stride_f=1):
    args = _bn(bn) if bn is not None else (None, None, None, None)
    cfg = dict(B=B, T=T, F=F, act=act, stride_t=1, stride_f=stride_f)
    if bn is not None:
        cfg.update(momentum=bn.momentum, eps=bn.eps)
    return Conv2dBlock.apply(x, conv.weight, conv.bias, *args, cfg)


def _bnrelu(x, bn, relu=True):
    return BNRows.apply(x, bn.weight, bn.bias, bn._mean, bn._variance, bn.momentum, bn.eps, relu)


def _c1x1(x, conv, B, T, relu=False, sigmoid=False):
    return ConvBlock.apply(x, conv.weight, conv.bias, None, None, None, None, None, dict(B=B, T=T, relu=relu, sigmoid=sigmoid))


def res_block(b, x, B, T, F):
    s = b.stride
    out = _c2(x, b.conv1, b.bn1, B, T, F, act='relu', stride_f=s)
    Fo = (F - 1) // s + 1
    out = _c2(out, b.conv2, b.bn2, B, T, Fo)
    sc = x
    if len(b.shortcut) > 0:
        sc = _c2(x, b.shortcut[0], b.shortcut[1], B, T, F, stride_f=s)
    return Act.apply(out + sc, 'relu'), Fo


def fcm(head, feats):
    B, T, F = feats.shape
    x = torch.zeros((B * T * F, 4), dtype=torch.float32, device=feats.device)       # single input channel padded to 4
    x[:, 0] = feats.reshape(-1)
    w = head.conv1.weight
    w4 = torch.cat([w, torch.zeros((w.shape[0], 3, 3, 3), dtype=w.dtype, device=w.device)], dim=1)
    x = Conv2dBlock.apply(x, w4, head.conv1.bias, *_bn(head.bn1), dict(B=B, T=T, F=F, act='relu', momentum=head.bn1.momentum, eps=head.bn1.eps))
    for layer in (head.layer1, head.layer2):
        for b in layer:
            x, F = res_block(b, x, B, T, F)
    x = _c2(x, head.conv2, head.bn2, B, T, F, act='relu', stride_f=2)
    F = (F - 1) // 2 + 1
    Cc = x.shape[1]
    return x.reshape(B, T, F, Cc).permute(0, 1, 3, 2).reshape(B * T, Cc * F), T     # channel index c*F' + f (campplus.py:279-280)


def cam_dense_layer(lay, x, B, T):
    h = _bnrelu(x, lay.nonlinear1.batchnorm)
    h = _c1x1(h, lay.linear1, B, T)
    h = _bnrelu(h, lay.nonlinear2.batchnorm)
    cl = lay.cam_layer
    wl = cl.linear_local.weight                                          # (out, bn, k)
    if CamLayerFn.usable(h, wl, cl.linear1.weight, cl.linear2.weight, T, SEG_LEN):
        # the local conv + the whole context gate as one tape entry: 3 launches forward, 5 backward (per-op form below: 6 and ~25)
        return CamLayerFn.apply(h, wl, cl.linear_local.bias, cl.linear1.weight, cl.linear1.bias, cl.linear2.weight, cl.linear2.bias,
                                dict(B=B, T=T, seg_len=SEG_LEN, dilation=cl.dilation))
    y = Conv2dBlock.apply(h, wl.unsqueeze(2), cl.linear_local.bias, None, None, None, None,
                          dict(B=B, T=T, F=1, dilation=cl.dilation, act=None))
    nseg = (T + SEG_LEN - 1) // SEG_LEN
    ctx = SegCtx.apply(h, B, T, SEG_LEN)                                 # (B*nseg, bn)
    ctx = _c1x1(ctx, cl.linear1, B * nseg, 1, relu=True)
    m = _c1x1(ctx, cl.linear2, B * nseg, 1, sigmoid=True)
    return SegScale.apply(y, m, B, T, SEG_LEN)


def dense_block(layers, x, B, T):
    """CAMDenseTDNNBlock (campplus.py:137-171) as ONE tape entry on one preallocated buffer (functions.CamDenseBlockFn)."""
    params, bufs = [], []
    for lay in layers:
        bn1, bn2, cl = lay.nonlinear1.batchnorm, lay.nonlinear2.batchnorm, lay.cam_layer
        params += [bn1.weight, bn1.bias, lay.linear1.weight, lay.linear1.bias, bn2.weight, bn2.bias, cl.linear_local.weight, cl.linear_local.bias,
                   cl.linear1.weight, cl.linear1.bias, cl.linear2.weight, cl.linear2.bias]
        bufs.append((bn1._mean, bn1._variance, bn1.momentum, bn1.eps, bn2._mean, bn2._variance, bn2.momentum, bn2.eps, cl.dilation))
    return CamDenseBlockFn.apply(x, dict(B=B, T=T, seg_len=SEG_LEN, bufs=bufs), *params)


def campplus_forward_train(m, feats):
    B = feats.shape[0]
    x, T = fcm(m.head, feats)
    xv = m.xvector
    td = xv.tdnn
    x = Conv2dBlock.apply(x, td.linear.weight.unsqueeze(2), td.linear.bias, *_bn(td.nonlinear.batchnorm),
                          dict(B=B, T=T, F=1, stride_t=td.stride, stride_f=1, act='relu', momentum=td.nonlinear.batchnorm.momentum,
                               eps=td.nonlinear.batchnorm.eps))
    T = (T + 2 * 2 - 4 - 1) // td.stride + 1
    for bi, (nl, _, _) in enumerate(m.block_cfg, start=1):
        blk = getattr(xv, f'block{bi}')
        layers = [getattr(blk, f'tdnnd{l}') for l in range(1, nl + 1)]
        if CamDenseBlockFn.usable(x, layers, T, SEG_LEN):
            x = dense_block(layers, x, B, T)         # the whole block on one buffer (no torch.cat, one gradient buffer backward)
        else:
            for lay in layers:
                x = torch.cat([x, cam_dense_layer(lay, x, B, T)], dim=1)
        tr = getattr(xv, f'transit{bi}')
        x = _c1x1(_bnrelu(x, tr.nonlinear.batchnorm), tr.linear, B, T)
    x = _bnrelu(x, xv.out_nonlinear.batchnorm)
    stats = TimeStats.apply(x, B, T, True, 0.0)                          # mean | unbiased std (campplus.py:24-30)
    y = _c1x1(stats, xv.dense.linear, B, 1)
    return _bnrelu(y, xv.dense.nonlinear.batchnorm, relu=False)
