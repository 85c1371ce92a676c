"""Training-mode forward of ResNetSE (ppvector/models/resnet_se.py:121-139) through the autograd functions of functions.py.
Activations are (B*T*F, C) position-major; the reference's reshape (B, C*F/8, T/8) before the pooling is a permute + reshape
of a small tensor (data movement).  Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim)."""
import torch

from ppvector.train.functions import Act, BNRows, Conv2dBlock, ConvBlock, SEDenseFn, SEScale, TimeStats
from ppvector.train.tdnn_train import asp_forward
from ppvector.train.segments import cut


def _bn(p):
    return p.weight, p.bias, p._mean, p._variance


def _cfg(B, T, F, bn, **kw):
    return dict(B=B, T=T, F=F, momentum=bn.momentum if bn is not None else 0.9, eps=bn.eps if bn is not None else 1e-5, **kw)


def bottleneck(b, x, B, T, F):
    """SEBottleneck (resnet_se.py:8-45)."""
    s = b.stride[0] if isinstance(b.stride, (tuple, list)) else b.stride
    out = Conv2dBlock.apply(x, b.conv1.weight, b.conv1.bias, *_bn(b.bn1), _cfg(B, T, F, b.bn1, relu=True))
    out = Conv2dBlock.apply(out, b.conv2.weight, b.conv2.bias, *_bn(b.bn2), _cfg(B, T, F, b.bn2, relu=True, stride=s))
    To, Fo = (T - 1) // s + 1, (F - 1) // s + 1
    out = Conv2dBlock.apply(out, b.conv3.weight, b.conv3.bias, *_bn(b.bn3), _cfg(B, To, Fo, b.bn3))
    Cc = out.shape[1]
    # SELayer (:48-63): global average over (F, T) -> Linear -> ReLU -> Linear -> sigmoid (paddle Linear weights are [in, out])
    y = TimeStats.apply(out, B, To * Fo)[:, :Cc]
    fc0, fc2 = b.se.fc[0], b.se.fc[2]
    w1, w2 = fc0.weight.t().unsqueeze(2), fc2.weight.t().unsqueeze(2)         # (H, C, 1), (C, H, 1)
    if SEDenseFn.usable(y, w1, fc0.bias, w2, fc2.bias):
        y = SEDenseFn.apply(y.contiguous(), w1.contiguous(), fc0.bias, w2.contiguous(), fc2.bias)      # one launch forward, two backward
    else:
        y = ConvBlock.apply(y, w1, fc0.bias, None, None, None, None, None, dict(B=B, T=1, relu=True))
        y = ConvBlock.apply(y, w2, fc2.bias, None, None, None, None, None, dict(B=B, T=1, sigmoid=True))
    res = x
    if b.downsample is not None:
        dconv, dbn = b.downsample[0], b.downsample[1]
        res = Conv2dBlock.apply(x, dconv.weight, dconv.bias, *_bn(dbn), _cfg(B, T, F, dbn, stride=s))
    out = SEScale.apply(out, y, res, B, To * Fo)
    return Act.apply(out, 'relu'), To, Fo


def resnetse_forward_train(m, feats):
    B, T, F = feats.shape
    # stem: conv3x3(1 -> C1) -> BN -> ReLU; the single input channel is zero-padded to 4 (16-byte channel chunks)
    x = torch.zeros((B * T * F, 4), dtype=torch.float32, device=feats.device)
    x[:, 0] = feats.reshape(-1)
    w = m.conv1.weight
    w4 = torch.cat([w, torch.zeros((w.shape[0], 3, 3, 3), dtype=w.dtype, device=w.device)], dim=1)
    x = Conv2dBlock.apply(x, w4, m.conv1.bias, *_bn(m.bn1), _cfg(B, T, F, m.bn1, relu=True))
    for li, layer in enumerate((m.layer1, m.layer2, m.layer3, m.layer4)):
        for b in layer:
            x, T, F = bottleneck(b, x, B, T, F)
        if li < 3:
            (x,) = cut(x)          # backward stage boundary (train/segments.py): a plain chain, one live tensor
    Cc = x.shape[1]
    # (B, T', F', C) -> the reference's (B, C*F', T') channel order c*F' + f, frame-major for the pooling: (B*T', C*F')
    x = x.reshape(B, T, F, Cc).permute(0, 1, 3, 2).reshape(B * T, Cc * F)
    p = asp_forward(m.pooling, x, B, T)
    n2, n3 = m.bn2.norm, m.bn3.norm
    p = BNRows.apply(p, n2.weight, n2.bias, n2._mean, n2._variance, n2.momentum, n2.eps)
    y = ConvBlock.apply(p, m.linear.weight.t().unsqueeze(2), m.linear.bias, None, None, None, None, None, dict(B=B, T=1))
    return BNRows.apply(y, n3.weight, n3.bias, n3._mean, n3._variance, n3.momentum, n3.eps)
