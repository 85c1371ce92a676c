"""ECAPA-TDNN backbone on the MI355X engine.

Same constructor surface, attribute names (``embd_dim``) and state-dict keys as
ppvector/models/ecapa_tdnn.py:145-276 (EcapaTdnn), :85-142 (SERes2NetBlock), :50-82 (SEBlock),
:11-47 (Res2NetBlock).  The modules below are parameter containers; ``EcapaTdnn.forward`` runs
the whole graph through libvpmi (csrc/ecapa.hip: vp_ecapa_fwd).
"""
from torch import nn

from ppvector.models.engine import EcapaEngine, EngineMixin
from ppvector.models.pooling import AttentiveStatisticsPooling
from ppvector.models.utils import BatchNorm1d, Conv1d, TDNNBlock

__all__ = ['EcapaTdnn']


class Res2NetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, scale=8, dilation=1):
        super().__init__()
        assert in_channels % scale == 0
        assert out_channels % scale == 0
        self.blocks = nn.ModuleList([TDNNBlock(in_channels // scale, out_channels // scale, kernel_size=3,
                                               dilation=dilation) for _ in range(scale - 1)])
        self.scale = scale


class SEBlock(nn.Module):
    def __init__(self, in_channels, se_channels, out_channels):
        super().__init__()
        self.conv1 = Conv1d(in_channels=in_channels, out_channels=se_channels, kernel_size=1)
        self.relu = nn.ReLU()
        self.conv2 = Conv1d(in_channels=se_channels, out_channels=out_channels, kernel_size=1)
        self.sigmoid = nn.Sigmoid()


class SERes2NetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, res2net_scale=8, se_channels=128, kernel_size=1, dilation=1,
                 activation=nn.ReLU):
        super().__init__()
        self.out_channels = out_channels
        self.tdnn1 = TDNNBlock(in_channels, out_channels, kernel_size=1, dilation=1, activation=activation)
        self.res2net_block = Res2NetBlock(out_channels, out_channels, res2net_scale, dilation)
        self.tdnn2 = TDNNBlock(out_channels, out_channels, kernel_size=1, dilation=1, activation=activation)
        self.se_block = SEBlock(out_channels, se_channels, out_channels)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=1)


class EcapaTdnn(EngineMixin, nn.Module):
    _bf16_trained_score_err = '1.9e-3'      # quoted by engine('bfloat16')'s warning (models/engine.py; profiles/r05_trained_weights_parity.log)
    _engine_cls = EcapaEngine

    def __init__(self, input_size, embd_dim=192, pooling_type="ASP", activation=nn.ReLU,
                 channels=[512, 512, 512, 512, 1536], kernel_sizes=[5, 3, 3, 3, 1], dilations=[1, 2, 3, 4, 1],
                 attention_channels=128, res2net_scale=8, se_channels=128, global_context=True):
        super().__init__()
        assert len(channels) == len(kernel_sizes)
        assert len(channels) == len(dilations)
        self.input_size = input_size
        self.channels = channels
        self.embd_dim = embd_dim
        self.res2net_scale, self.se_channels = res2net_scale, se_channels
        self.blocks = nn.ModuleList()
        self.blocks.append(TDNNBlock(input_size, channels[0], kernel_sizes[0], dilations[0], activation))
        for i in range(1, len(channels) - 1):
            self.blocks.append(SERes2NetBlock(channels[i - 1], channels[i], res2net_scale=res2net_scale,
                                              se_channels=se_channels, kernel_size=kernel_sizes[i],
                                              dilation=dilations[i], activation=activation))
        self.mfa = TDNNBlock(channels[-1], channels[-1], kernel_sizes[-1], dilations[-1], activation)
        if pooling_type == "ASP":
            self.asp = AttentiveStatisticsPooling(channels[-1], attention_channels=attention_channels,
                                                  global_context=global_context)
            self.asp_bn = BatchNorm1d(input_size=channels[-1] * 2)
            self.fc = Conv1d(in_channels=channels[-1] * 2, out_channels=self.embd_dim, kernel_size=1)
        elif pooling_type in ("SAP", "TAP", "TSP"):
            raise NotImplementedError(f'pooling_type {pooling_type} is not built on the HIP engine (ASP is); the '
                                      'reference wires SAP/TAP/TSP shape-inconsistently (SURVEY.md section 2 row 4)')
        else:
            raise Exception(f'没有{pooling_type}池化层！')

    def _train_forward(self, x):
        """Training mode: batch-statistics BatchNorm, autograd through libvpmi's backward entry points (f32 engine)."""
        from ppvector import _native as N
        from ppvector.train.ecapa_train import ecapa_forward_train
        if not x.is_cuda:
            raise N.VpmiError('model input must be a GPU tensor: the engine has no CPU fallback')
        return ecapa_forward_train(self, x.float().contiguous())
