"""Parameter containers with the reference's layer / state-dict names.

Mirrors ppvector/models/utils.py:22-148 (Conv1d, BatchNorm1d, TDNNBlock, length_to_mask) as far as
*parameters and naming* go -- ``conv.weight`` (Cout, Cin, k), ``norm.weight/bias/_mean/_variance`` --
so checkpoints keyed like the reference's load unchanged.  The arithmetic does not live here: the
owning model packs these tensors once (BN folded to scale/shift, conv weights re-laid as
[Cout][k*Cin]) and runs the whole graph through libvpmi (csrc/ecapa.hip).
"""
import math

import torch
from torch import nn

BN_EPS = 1e-5


class _ConvParams(nn.Module):
    """nn.Conv1D stand-in: parameters only."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)


class _BNParams(nn.Module):
    """nn.BatchNorm1D stand-in (Paddle names: weight, bias, _mean, _variance)."""

    def __init__(self, num_features, eps=BN_EPS, momentum=0.9):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('_mean', torch.zeros(num_features))
        self.register_buffer('_variance', torch.ones(num_features))

    def folded(self):
        """Eval-mode affine: y = x * scale + shift."""
        scale = self.weight.detach().float() / torch.sqrt(self._variance.float() + self.eps)
        shift = self.bias.detach().float() - self._mean.float() * scale
        return scale.contiguous(), shift.contiguous()


class Conv1d(nn.Module):
    """models/utils.py:22-93: 'same' reflect-padded conv; holds ``conv.{weight,bias}``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding="same", dilation=1, groups=1,
                 bias=True, padding_mode="reflect"):
        super().__init__()
        if padding != "same":
            raise ValueError(f"Padding must be 'same'. Got {padding}")
        if stride != 1 or groups != 1 or padding_mode != "reflect":
            raise NotImplementedError('only stride 1, groups 1, reflect padding are built on the HIP engine')
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.padding, self.padding_mode = padding, padding_mode
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, bias)


class BatchNorm1d(nn.Module):
    """models/utils.py:96-119; holds ``norm.{weight,bias,_mean,_variance}``."""

    def __init__(self, input_size, eps=1e-05, momentum=0.9):
        super().__init__()
        self.norm = _BNParams(input_size, eps, momentum)


class TDNNBlock(nn.Module):
    """models/utils.py:122-148: BN(ReLU(Conv1d(x)))."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation, activation=nn.ReLU):
        super().__init__()
        if activation is not nn.ReLU:
            raise NotImplementedError('only ReLU is fused in the HIP conv epilogue')
        self.conv = Conv1d(in_channels, out_channels, kernel_size, dilation=dilation)
        self.activation = activation()
        self.norm = BatchNorm1d(out_channels)


def length_to_mask(length, max_len=None, dtype=None):
    """models/utils.py:8-19."""
    assert len(length.shape) == 1
    if max_len is None:
        max_len = int(length.max().item())
    mask = torch.arange(max_len, dtype=length.dtype, device=length.device).expand(len(length), max_len) \
        < length.unsqueeze(1)
    return mask.to(dtype if dtype is not None else length.dtype)


# ------------------------------------------------------------------ packing helpers (host side)
def pack_conv_weight(weight, dtype):
    """(Cout, Cin, k) -> [Cout][k*Cin] with k-index = tap*Cin + channel, in the network dtype."""
    cout, cin, k = weight.shape
    return weight.detach().permute(0, 2, 1).reshape(cout, k * cin).to(dtype).contiguous()


def f32(t):
    return None if t is None else t.detach().float().contiguous()


def pack_hl32(t):
    """(..., C) float tensor, C % 32 == 0  ->  the same shape as torch.float32 whose BYTES are the split bf16 planes of include/vpmi.h's
    VP_HL32: per 32-channel group [32 x bf16 hi | 32 x bf16 lo], hi = bf16(v) (round to nearest even), lo = bf16(v - hi)."""
    t = t.detach().float().contiguous()
    C = t.shape[-1]
    if C % 32:
        raise ValueError(f'hl32 needs a multiple of 32 channels, got {C}')
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    g = torch.cat([hi.reshape(*t.shape[:-1], C // 32, 32), lo.reshape(*t.shape[:-1], C // 32, 32)], dim=-1)      # (..., C/32, 64) bf16
    return g.contiguous().view(torch.float32).reshape(t.shape)


def unpack_hl32(t):
    """Inverse view of pack_hl32: the values hi + lo as float32."""
    C = t.shape[-1]
    g = t.contiguous().view(torch.bfloat16).reshape(*t.shape[:-1], C // 32, 64).float()
    return (g[..., :32] + g[..., 32:]).reshape(t.shape)
