"""ERes2Net backbone on the MI355X engine.

Same constructor surface, ``embd_dim`` attribute and state-dict keys as ppvector/models/eres2net.py
(ReLU = Hardtanh(0, 20) :12-20, AFF :33-53, BasicBlockERes2Net :56-108, BasicBlockERes2Net_diff_AFF :111-169,
ERes2Net :172-263).  The modules are parameter containers; ``forward`` runs the whole graph through libvpmi
(csrc/eres2net.hip: vp_eres2net_fwd).  ERes2NetV2 (:266-462) shares the launch graph: its chunk widths (13 / 26 / 52 / 104) are
zero-padded to multiples of 8 at pack time.  Training mode is built for ERes2Net only.
"""
import math

import torch
from torch import nn

from ppvector.models.campplus import _ConvNd
from ppvector.models.engine import EngineMixin, Eres2netEngine
from ppvector.models.pooling import TemporalStatsPool
from ppvector.models.resnet_se import _LinearParams
from ppvector.models.utils import _BNParams

__all__ = ['ERes2Net', 'ERes2NetV2']


class ReLU(nn.Hardtanh):
    def __init__(self, inplace=False):
        super().__init__(0, 20, inplace)


def conv1x1(in_planes, out_planes, stride=1):
    c = _ConvNd(in_planes, out_planes, 1, 1)
    c.stride = stride
    return c


def conv3x3(in_planes, out_planes, stride=1):
    c = _ConvNd(in_planes, out_planes, 3, 3)
    c.stride = stride
    return c


class AFF(nn.Module):
    def __init__(self, channels=64, r=4):
        super().__init__()
        inter_channels = int(channels // r)
        self.local_att = nn.Sequential(_ConvNd(channels * 2, inter_channels, 1, 1), _BNParams(inter_channels), nn.SiLU(),
                                       _ConvNd(inter_channels, channels, 1, 1), _BNParams(channels))


class BasicBlockERes2Net(nn.Module):
    use_aff = False

    def __init__(self, expansion, in_planes, planes, stride=1, base_width=32, scale=2):
        super().__init__()
        self.expansion = expansion
        width = int(math.floor(planes * (base_width / 64.0)))
        self.conv1 = conv1x1(in_planes, width * scale, stride)
        self.bn1 = _BNParams(width * scale)
        self.nums = scale
        self.convs = nn.ModuleList([conv3x3(width, width) for _ in range(self.nums)])
        self.bns = nn.ModuleList([_BNParams(width) for _ in range(self.nums)])
        if self.use_aff:
            self.fuse_models = nn.ModuleList([AFF(channels=width) for _ in range(self.nums - 1)])
        self.relu = ReLU(inplace=True)
        self.conv3 = conv1x1(width * scale, planes * self.expansion)
        self.bn3 = _BNParams(planes * self.expansion)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(_ConvNd(in_planes, self.expansion * planes, 1, 1), _BNParams(self.expansion * planes))
        self.stride = stride
        self.width = width
        self.scale = scale


class BasicBlockERes2Net_diff_AFF(BasicBlockERes2Net):
    use_aff = True


def _second_embedding(m, embed_a):
    """two_emb_layer=True (eres2net.py:255-260 / :455-460): embed_b = seg_2(seg_bn_1(relu(embed_a))) on the (B, embd) output of the
    engine (eval: ReLU, folded BatchNorm and the dense kernel; training: the Act / BNRows / Dense functions)."""
    bn, lin = m.seg_bn_1, m.seg_2
    if m.training and torch.is_grad_enabled() and embed_a.requires_grad:
        from ppvector.train.functions import Act, BNRows, Dense
        y = BNRows.apply(Act.apply(embed_a, 'relu'), bn.weight, bn.bias, bn._mean, bn._variance, bn.momentum, bn.eps)
        return Dense.apply(y, lin.weight, lin.bias)
    from ppvector import _native as N
    with torch.no_grad():
        y = embed_a.float().clone()
        lib, ctx = N.lib(), N.ctx(y.device)
        scale, shift = bn.folded()
        N.check(lib.vp_act_f32(ctx, N.VP_ACT_RELU, y.data_ptr(), y.numel(), y.data_ptr(), N.stream_ptr()), ctx)
        N.check(lib.vp_affine_rows_f32(ctx, y.data_ptr(), y.shape[1], scale.data_ptr(), shift.data_ptr(), y.shape[0], y.shape[1],
                                       y.data_ptr(), y.shape[1], 0, N.stream_ptr()), ctx)
        out = torch.empty((y.shape[0], lin.weight.shape[1]), dtype=torch.float32, device=y.device)
        w = lin.weight.detach().float().contiguous()                # paddle Linear: [in, out]
        N.check(lib.vp_dense_f32(ctx, y.data_ptr(), y.shape[1], w.data_ptr(), 1, lin.bias.detach().float().data_ptr(), y.shape[0],
                                 out.shape[1], y.shape[1], N.VP_ACT_NONE, out.data_ptr(), out.shape[1], N.stream_ptr()), ctx)
        return out


def _forward_with_second_embedding(self, x, lengths=None):
    emb = EngineMixin.forward(self, x, lengths)
    return _second_embedding(self, emb) if self.two_emb_layer else emb


class ERes2Net(EngineMixin, nn.Module):
    _bf16_trained_score_err = '1.2e-2'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = Eres2netEngine

    def __init__(self, input_size, block=BasicBlockERes2Net, block_fuse=BasicBlockERes2Net_diff_AFF, num_blocks=[3, 4, 6, 3],
                 m_channels=32, mul_channel=1, expansion=2, base_width=32, scale=2, embd_dim=192, pooling_type='TSTP',
                 two_emb_layer=False):
        super().__init__()
        self.in_planes = m_channels
        self.expansion = expansion
        self.feat_dim = input_size
        self.input_size = input_size
        self.embd_dim = embd_dim
        self.stats_dim = int(input_size / 8) * m_channels * 8
        self.two_emb_layer = two_emb_layer
        self.m_channels, self.num_blocks = m_channels, list(num_blocks)
        self.conv1 = _ConvNd(1, m_channels, 3, 3)
        self.bn1 = _BNParams(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], 1, base_width, scale)
        self.layer2 = self._make_layer(block, m_channels * 2, num_blocks[1], 2, base_width, scale)
        self.layer3 = self._make_layer(block_fuse, m_channels * 4, num_blocks[2], 2, base_width, scale)
        self.layer4 = self._make_layer(block_fuse, m_channels * 8, num_blocks[3], 2, base_width, scale)
        self.layer1_downsample = _ConvNd(m_channels * 2 * mul_channel, m_channels * 4 * mul_channel, 3, 3)
        self.layer2_downsample = _ConvNd(m_channels * 4 * mul_channel, m_channels * 8 * mul_channel, 3, 3)
        self.layer3_downsample = _ConvNd(m_channels * 8 * mul_channel, m_channels * 16 * mul_channel, 3, 3)
        self.fuse_mode12 = AFF(channels=m_channels * 4 * mul_channel)
        self.fuse_mode123 = AFF(channels=m_channels * 8 * mul_channel)
        self.fuse_mode1234 = AFF(channels=m_channels * 16 * mul_channel)
        self.n_stats = 1 if pooling_type == 'TAP' else 2
        if pooling_type == "TSTP":
            self.pooling = TemporalStatsPool()
        else:
            raise Exception(f'没有{pooling_type}池化层！')
        self.seg_1 = _LinearParams(self.stats_dim * self.expansion * self.n_stats, embd_dim)
        if self.two_emb_layer:
            self.seg_bn_1 = _BNParams(embd_dim)
            self.seg_2 = _LinearParams(embd_dim, embd_dim)
        else:
            self.seg_bn_1 = nn.Identity()
            self.seg_2 = nn.Identity()

    def _make_layer(self, block, planes, num_blocks, stride, base_width, scale):
        strides = [stride] + [1] * (num_blocks - 1)
        layers = []
        for stride in strides:
            layers.append(block(self.expansion, self.in_planes, planes, stride, base_width, scale))
            self.in_planes = planes * self.expansion
        return nn.Sequential(*layers)


    forward = _forward_with_second_embedding

    def _train_forward(self, x):
        """Training mode: batch-statistics BatchNorm, autograd through libvpmi's backward entry points (f32 engine)."""
        from ppvector import _native as N
        from ppvector.train.eres2net_train import eres2net_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return eres2net_forward_train(self, x.float().contiguous())


class BasicBlockERes2NetV2(BasicBlockERes2Net):
    """models/eres2net.py:266-318: the V1 block with base_width 26 by default."""

    def __init__(self, expansion, in_planes, planes, stride=1, base_width=26, scale=2):
        super().__init__(expansion, in_planes, planes, stride, base_width, scale)


class BasicBlockERes2NetV2_AFF(BasicBlockERes2NetV2):
    use_aff = True


class ERes2NetV2(EngineMixin, nn.Module):
    """models/eres2net.py:376-462: four stages of V2 blocks, ONE bottom-up fusion (layer3_ds + fuse34), TSTP, Linear."""
    _bf16_trained_score_err = '1.2e-2'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = Eres2netEngine
    v2 = True

    def __init__(self, input_size, block=BasicBlockERes2NetV2, block_fuse=BasicBlockERes2NetV2_AFF, num_blocks=[3, 4, 6, 3],
                 m_channels=32, expansion=2, base_width=26, scale=2, embd_dim=192, pooling_type='TSTP', two_emb_layer=False):
        super().__init__()
        self.in_planes = m_channels
        self.expansion = expansion
        self.input_size = input_size
        self.embd_dim = embd_dim
        self.stats_dim = int(input_size / 8) * m_channels * 8
        self.two_emb_layer = two_emb_layer
        self.m_channels, self.num_blocks = m_channels, list(num_blocks)
        self.conv1 = _ConvNd(1, m_channels, 3, 3)
        self.bn1 = _BNParams(m_channels)
        self.layer1 = self._make_layer(block, m_channels, num_blocks[0], 1, base_width, scale)
        self.layer2 = self._make_layer(block, m_channels * 2, num_blocks[1], 2, base_width, scale)
        self.layer3 = self._make_layer(block_fuse, m_channels * 4, num_blocks[2], 2, base_width, scale)
        self.layer4 = self._make_layer(block_fuse, m_channels * 8, num_blocks[3], 2, base_width, scale)
        self.layer3_ds = _ConvNd(m_channels * 8, m_channels * 16, 3, 3)
        self.fuse34 = AFF(channels=m_channels * 16, r=4)
        self.n_stats = 1 if pooling_type == 'TAP' else 2
        if pooling_type == "TSTP":
            self.pooling = TemporalStatsPool()
        else:
            raise Exception(f'没有{pooling_type}池化层！')
        self.seg_1 = _LinearParams(self.stats_dim * self.expansion * self.n_stats, embd_dim)
        if self.two_emb_layer:
            self.seg_bn_1 = _BNParams(embd_dim)
            self.seg_2 = _LinearParams(embd_dim, embd_dim)
        else:
            self.seg_bn_1 = nn.Identity()
            self.seg_2 = nn.Identity()

    _make_layer = ERes2Net._make_layer
    forward = _forward_with_second_embedding

    def _train_forward(self, x):
        """Training mode (f32 engine); stages 1-2 run on zero-padded chunk widths (train/eres2net_train.py)."""
        from ppvector import _native as N
        from ppvector.train.eres2net_train import eres2netv2_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return eres2netv2_forward_train(self, x.float().contiguous())
