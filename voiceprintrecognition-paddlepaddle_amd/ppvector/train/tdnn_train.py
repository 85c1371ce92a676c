"""Training-mode forward of the x-vector TDNN (ppvector/models/tdnn.py:46-68) through the autograd functions of
functions.py: batch-statistics BatchNorm, running statistics updated in place (momentum 0.9), gradients for every
parameter.  Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim)."""
import torch

from ppvector.train.functions import AspFn, BNRows, ConvBlock


def _bn(p):
    return p.weight, p.bias, p._mean, p._variance


def asp_forward(asp, x, B, T):
    """AttentiveStatisticsPooling.forward with lengths=None (pooling.py:86-125): x (B*T, C) -> (B, 2C), one tape entry."""
    conv, norm, c2 = asp.tdnn.conv.conv, asp.tdnn.norm.norm, asp.conv.conv
    g, b, rm, rv = _bn(norm)
    return AspFn.apply(x, conv.weight, conv.bias, g, b, rm, rv, c2.weight, c2.bias,
                       dict(B=B, T=T, global_context=bool(asp.global_context), momentum=norm.momentum, eps=norm.eps))


def tdnn_forward_train(m, feats):
    B, T, F = feats.shape
    x = feats.reshape(B * T, F)
    for i, d in zip(range(1, 5), (1, 2, 3, 1)):
        conv, bn = getattr(m, f'td_layer{i}'), getattr(m, f'bn{i}')
        g, b, rm, rv = _bn(bn)
        KW = conv.weight.shape[2]
        x = ConvBlock.apply(x, conv.weight, conv.bias, None, g, b, rm, rv, dict(B=B, T=T, dilation=d, relu=True, momentum=bn.momentum, eps=bn.eps))
        T = T - d * (KW - 1)
    c5 = m.td_layer5
    x = ConvBlock.apply(x, c5.weight, c5.bias, None, None, None, None, None, dict(B=B, T=T, relu=True))
    p = asp_forward(m.pooling, x, B, T)
    n5, n6 = m.bn5.norm, m.bn6.norm
    p = BNRows.apply(p, n5.weight, n5.bias, n5._mean, n5._variance, n5.momentum, n5.eps)
    # paddle Linear ([in, out]) as a 1x1 conv over B "frames"
    lw = m.linear.weight.t().unsqueeze(2)
    y = ConvBlock.apply(p, lw, m.linear.bias, None, None, None, None, None, dict(B=B, T=1))
    return BNRows.apply(y, n6.weight, n6.bias, n6._mean, n6._variance, n6.momentum, n6.eps)
