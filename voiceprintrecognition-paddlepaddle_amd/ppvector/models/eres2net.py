"""ERes2Net backbone on the MI355X engine.

Same constructor surface, ``embd_dim`` attribute and state-dict keys as ppvector/models/eres2net.py
(ReLU = Hardtanh(0, 20) :12-20, AFF :33-53, BasicBlockERes2Net :56-108, BasicBlockERes2Net_diff_AFF :111-169,
ERes2Net :172-263).  The modules are parameter containers; ``forward`` runs the whole graph through libvpmi
(csrc/eres2net.hip: vp_eres2net_fwd).  ERes2NetV2 (:266-462) shares the launch graph: its chunk widths (13 / 26 / 52 / 104) are
zero-padded to multiples of 8 at pack time.  Training mode is built for ERes2Net only.
"""
import math

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


class ERes2Net(EngineMixin, nn.Module):
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
            raise NotImplementedError('two_emb_layer=True (ReLU -> BatchNorm -> second Linear) is not built on the HIP engine')
        self.seg_bn_1 = nn.Identity()
        self.seg_2 = nn.Identity()

    def _make_layer(self, block, planes, num_blocks, stride, base_width, scale):
        strides = [stride] + [1] * (num_blocks - 1)
        layers = []
        for stride in strides:
            layers.append(block(self.expansion, self.in_planes, planes, stride, base_width, scale))
            self.in_planes = planes * self.expansion
        return nn.Sequential(*layers)


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
            raise NotImplementedError('two_emb_layer=True (ReLU -> BatchNorm -> second Linear) is not built on the HIP engine')
        self.seg_bn_1 = nn.Identity()
        self.seg_2 = nn.Identity()

    _make_layer = ERes2Net._make_layer

    def _train_forward(self, x):
        """Training mode (f32 engine); stages 1-2 run on zero-padded chunk widths (train/eres2net_train.py)."""
        from ppvector import _native as N
        from ppvector.train.eres2net_train import eres2netv2_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return eres2netv2_forward_train(self, x.float().contiguous())
