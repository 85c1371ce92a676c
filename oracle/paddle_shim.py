"""A minimal ``paddle`` API restated on PyTorch-CPU.  TEST INFRASTRUCTURE ONLY.

Purpose: PaddlePaddle cannot be installed here, so the reference's arithmetic cannot run as
shipped.  Its *model graphs*, however, are plain Python under /root/reference/ppvector/models
and /root/reference/ppvector/loss.  ``install()`` registers this module as ``paddle`` (plus
``paddle.nn`` / ``paddle.nn.functional``) so those files can be imported UNMODIFIED, from where
they lie, and executed on the CPU.  ``oracle/gen_golden.py`` uses that to (a) check the oracle's
restatement (oracle/models.py) against the reference's own graph code and (b) freeze golden
vectors under tests/golden/.

What this pins and what it does not: the reference files supply the graph (op order, shapes,
paddings, concatenations, masks); this shim supplies the semantics of each Paddle op as recalled
from the public PaddlePaddle 2.5/2.6 API ([3P-memory] in SURVEY.md): Conv1D weight (Cout,Cin,k);
Linear weight [in,out]; BatchNorm names weight/bias/_mean/_variance, momentum convention
running = m*running + (1-m)*batch, train-mode normalisation by the biased batch variance;
paddle.var/std unbiased; F.normalize eps 1e-12; F.pad reflect on NCL; Tensor.transpose(perm).
Never imported by the product package.
"""
import sys
import types

import numpy as np
import torch
import torch.nn.functional as TF

_DT = {'float32': torch.float32, 'float64': torch.float64, 'int32': torch.int32, 'int64': torch.int64,
       'int': torch.int64, 'bool': torch.bool, 'float': torch.float32}
float32, float64, int32, int64 = torch.float32, torch.float64, torch.int32, torch.int64


def _dtype(d):
    if d is None or isinstance(d, torch.dtype):
        return d
    return _DT[str(d)]


class Tensor(torch.Tensor):
    """torch.Tensor with the handful of Paddle-flavoured methods the reference files call."""

    def transpose(self, *perm):
        if len(perm) == 1 and isinstance(perm[0], (list, tuple)):
            return self.permute(*perm[0])
        return super().transpose(*perm)

    def astype(self, d):
        return self.to(_dtype(d))

    def tile(self, reps):
        return super().tile(tuple(reps))

    def expand(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = tuple(shape[0])
        return super().expand(*shape)

    def clip(self, min=None, max=None):
        return super().clamp(min=min, max=max)

    def std(self, axis=None, unbiased=True, keepdim=False):
        return super().std(dim=axis, unbiased=unbiased, keepdim=keepdim)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = tuple(shape[0])
        return super().reshape(*shape)

    def flatten(self, start_axis=0, stop_axis=-1):
        return super().flatten(start_axis, stop_axis)

    def unsqueeze(self, axis=None, dim=None):
        return super().unsqueeze(axis if dim is None else dim)

    def unsqueeze_(self, axis):
        return _wrap(torch.Tensor.unsqueeze(self, axis))

    def squeeze(self, axis=None, dim=None):
        a = axis if dim is None else dim
        return super().squeeze() if a is None else super().squeeze(a)


def _wrap(t):
    return t.as_subclass(Tensor) if isinstance(t, torch.Tensor) and not isinstance(t, Tensor) else t


def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    if isinstance(data, torch.Tensor):
        t = data.detach() if dtype is None else data.detach().to(_dtype(dtype))
    else:
        t = torch.as_tensor(np.asarray(data))
        if dtype is not None:
            t = t.to(_dtype(dtype))
        elif t.dtype == torch.float64:
            t = t.to(torch.float32)
    return _wrap(t)


def ones(shape, dtype='float32'):
    return _wrap(torch.ones(list(shape), dtype=_dtype(dtype)))


def zeros(shape, dtype='float32'):
    return _wrap(torch.zeros(list(shape), dtype=_dtype(dtype)))


def ones_like(x):
    return _wrap(torch.ones_like(x))


def zeros_like(x):
    return _wrap(torch.zeros_like(x))


def arange(*args, dtype=None):
    return _wrap(torch.arange(*args, dtype=_dtype(dtype)))


def concat(xs, axis=0):
    return _wrap(torch.cat(list(xs), dim=axis))


def stack(xs, axis=0):
    return _wrap(torch.stack(list(xs), dim=axis))


def chunk(x, chunks, axis=0):
    return [_wrap(c) for c in torch.chunk(x, chunks, dim=axis)]


def split(x, num_or_sections, axis=0):
    n = num_or_sections if isinstance(num_or_sections, int) else len(num_or_sections)
    return [_wrap(t) for t in torch.chunk(x, n, dim=axis)]


def multiply(a, b):
    return _wrap(a * b)


def where(cond, a, b):
    return _wrap(torch.where(cond, a, b))


def sqrt(x):
    return _wrap(torch.sqrt(x))


def pow(x, y):
    return _wrap(torch.pow(x, y))


def tanh(x):
    return _wrap(torch.tanh(x))


def exp(x):
    return _wrap(torch.exp(x))


def log(x):
    return _wrap(torch.log(x))


def mean(x, axis=None, keepdim=False):
    return _wrap(torch.mean(x) if axis is None else torch.mean(x, dim=axis, keepdim=keepdim))


def sum(x, axis=None, keepdim=False):
    return _wrap(torch.sum(x) if axis is None else torch.sum(x, dim=axis, keepdim=keepdim))


def var(x, axis=None, unbiased=True, keepdim=False):
    return _wrap(torch.var(x, dim=axis, unbiased=unbiased, keepdim=keepdim))


def std(x, axis=None, unbiased=True, keepdim=False):
    return _wrap(torch.std(x, dim=axis, unbiased=unbiased, keepdim=keepdim))


def max(x, axis=None, keepdim=False):
    return _wrap(torch.max(x) if axis is None else torch.max(x, dim=axis, keepdim=keepdim)[0])


def reshape(x, shape):
    return _wrap(torch.reshape(x, tuple(shape)))


class ParamAttr:
    def __init__(self, initializer=None, **kw):
        self.initializer = initializer


def create_parameter(shape, dtype='float32', attr=None, **kw):
    p = torch.nn.Parameter(torch.zeros(list(shape), dtype=_dtype(dtype)))
    if not kw.get('is_bias', False):                     # Paddle: bias parameters start at zero
        torch.nn.init.xavier_uniform_(p)
    return p


# ----------------------------------------------------------------------------- paddle.nn
nn = types.ModuleType('paddle.nn')


class Layer(torch.nn.Module):
    def __call__(self, *a, **k):
        return _wrap(super().__call__(*a, **k))

    def add_sublayer(self, name, layer):
        self.add_module(name, layer)
        return layer


class _Conv(Layer):
    nd = 1

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode='zeros', weight_attr=None, bias_attr=None, data_format=None):
        super().__init__()
        tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * self.nd
        self.stride, self.padding, self.dilation, self.groups = tup(stride), tup(padding), tup(dilation), groups
        ks = tup(kernel_size)
        w = torch.empty(out_channels, in_channels // groups, *ks)
        torch.nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = torch.nn.Parameter(w)
        if bias_attr is False:
            self.bias = None
        else:
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))

    def forward(self, x):
        f = TF.conv1d if self.nd == 1 else TF.conv2d
        return f(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class Conv1D(_Conv):
    nd = 1


class Conv2D(_Conv):
    nd = 2


class _BatchNorm(Layer):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None, bias_attr=None,
                 data_format=None, use_global_stats=None, name=None):
        super().__init__()
        self.momentum, self.epsilon = momentum, epsilon
        self.weight = torch.nn.Parameter(torch.ones(num_features))
        self.bias = torch.nn.Parameter(torch.zeros(num_features))
        self.register_buffer('_mean', torch.zeros(num_features))
        self.register_buffer('_variance', torch.ones(num_features))

    def forward(self, x):
        shape = [1, -1] + [1] * (x.dim() - 2)
        if self.training:
            dims = [0] + list(range(2, x.dim()))
            m = x.mean(dim=dims)
            v = x.var(dim=dims, unbiased=False)
            with torch.no_grad():
                self._mean.mul_(self.momentum).add_((1 - self.momentum) * m)
                self._variance.mul_(self.momentum).add_((1 - self.momentum) * v)
        else:
            m, v = self._mean, self._variance
        return (x - m.view(shape)) / torch.sqrt(v.view(shape) + self.epsilon) * self.weight.view(shape) \
            + self.bias.view(shape)


class BatchNorm1D(_BatchNorm):
    pass


class BatchNorm2D(_BatchNorm):
    pass


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        w = torch.empty(in_features, out_features)
        torch.nn.init.xavier_uniform_(w)
        self.weight = torch.nn.Parameter(w)                       # Paddle layout: [in, out]
        self.bias = torch.nn.Parameter(torch.zeros(out_features))

    def forward(self, x):
        return x @ self.weight + self.bias


def _act(fn):
    class _A(Layer):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return fn(x)
    return _A


class Hardtanh(Layer):
    def __init__(self, min=-1.0, max=1.0, name=None):
        super().__init__()
        self.min, self.max = min, max

    def forward(self, x):
        return _wrap(torch.clamp(x, self.min, self.max))


class CrossEntropyLoss(Layer):
    def __init__(self, label_smoothing=0.0, reduction='mean', **kw):
        super().__init__()
        self.label_smoothing, self.reduction = label_smoothing, reduction

    def forward(self, logits, labels):
        return _wrap(TF.cross_entropy(logits, labels.reshape(-1).long(), label_smoothing=self.label_smoothing,
                                      reduction=self.reduction))


class AdaptiveAvgPool2D(Layer):
    def __init__(self, output_size, **kw):
        super().__init__()
        assert output_size == 1

    def forward(self, x):
        return x.mean(dim=(2, 3), keepdim=True)


class LayerList(torch.nn.ModuleList):
    def forward(self, *a, **k):
        raise NotImplementedError

    def add_sublayer(self, name, layer):
        self.add_module(name, layer)
        return layer

    def __call__(self, *a, **k):
        return _wrap(super().__call__(*a, **k))


class Sequential(torch.nn.Sequential):
    def __init__(self, *layers):
        if layers and isinstance(layers[0], (tuple, list)) and isinstance(layers[0][0], str):
            super().__init__()
            for name, layer in layers:
                self.add_module(name, layer)
        else:
            super().__init__(*layers)

    def add_sublayer(self, name, layer):
        self.add_module(name, layer)
        return layer

    def __call__(self, *a, **k):
        return _wrap(super().__call__(*a, **k))


_init = types.ModuleType('paddle.nn.initializer')
_init.XavierUniform = lambda *a, **k: None
_init.KaimingNormal = lambda *a, **k: None
_init.Constant = lambda *a, **k: None

for _n, _v in dict(Layer=Layer, Conv1D=Conv1D, Conv2D=Conv2D, BatchNorm1D=BatchNorm1D, BatchNorm2D=BatchNorm2D,
                   Linear=Linear, AdaptiveAvgPool2D=AdaptiveAvgPool2D, ReLU=_act(torch.relu), Sigmoid=_act(torch.sigmoid), Tanh=_act(torch.tanh), Silu=_act(TF.silu), Identity=_act(lambda x: x),
                   Hardtanh=Hardtanh,
                   CrossEntropyLoss=CrossEntropyLoss, LayerList=LayerList, Sequential=Sequential,
                   initializer=_init).items():
    setattr(nn, _n, _v)

# ----------------------------------------------------------------------------- paddle.nn.functional
F = types.ModuleType('paddle.nn.functional')


def _pad(x, pad, mode='constant', value=0.0, data_format='NCL'):
    return _wrap(TF.pad(x, tuple(pad), mode=mode) if mode != 'constant' else TF.pad(x, tuple(pad), value=value))


def _normalize(x, p=2, axis=1, epsilon=1e-12):
    return _wrap(x / x.norm(p=p, dim=axis, keepdim=True).clamp(min=epsilon))


def _avg_pool1d(x, kernel_size, stride=None, padding=0, exclusive=True, ceil_mode=False):
    # Paddle default exclusive=True: a ceil-mode partial window averages over its valid elements only
    assert padding == 0 and exclusive
    stride = stride or kernel_size
    T = x.shape[-1]
    n = (T - kernel_size + stride - 1) // stride + 1 if ceil_mode else (T - kernel_size) // stride + 1
    outs = [x[..., i * stride:min(i * stride + kernel_size, T)].mean(dim=-1, keepdim=True) for i in range(n)]
    return _wrap(torch.cat(outs, dim=-1))


F.avg_pool1d = _avg_pool1d
F.pad = _pad
F.normalize = _normalize
F.relu = lambda x: _wrap(torch.relu(x))
F.softmax = lambda x, axis=-1: _wrap(torch.softmax(x, dim=axis))
F.linear = lambda x, w, b=None: _wrap(x @ w if b is None else x @ w + b)
F.one_hot = lambda x, n: _wrap(TF.one_hot(x.long(), n).to(torch.float32))
F.sigmoid = lambda x: _wrap(torch.sigmoid(x))
nn.functional = F


def install():
    """Register the shim as ``paddle`` and make ``ppvector.*`` resolve to the REFERENCE's files
    (without running its package __init__s, which import loguru / every model family)."""
    me = sys.modules[__name__]
    sys.modules['paddle'] = me
    sys.modules['paddle.nn'] = nn
    sys.modules['paddle.nn.functional'] = F
    sys.modules['paddle.nn.initializer'] = _init
    ref = '/root/reference/ppvector'
    for name, path in (('ppvector', ref), ('ppvector.models', ref + '/models'), ('ppvector.loss', ref + '/loss')):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
