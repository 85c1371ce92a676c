"""Pooling layers (parameters + names of ppvector/models/pooling.py).

AttentiveStatisticsPooling (pooling.py:69-125) is the one on the hot path; it runs inside the
model's fused forward (csrc/ecapa.hip run_asp): the global-context mean/std come from sums fused
into the producing conv, their 2C columns of the attention TDNN collapse to a per-utterance bias,
and softmax + weighted mean/std run in one pass (csrc/small_ops.hip).
"""
from torch import nn

from ppvector.models.utils import Conv1d, TDNNBlock


class AttentiveStatisticsPooling(nn.Module):
    def __init__(self, channels, attention_channels=128, global_context=True):
        super().__init__()
        self.eps = 1e-12
        self.global_context = global_context
        self.channels, self.attention_channels = channels, attention_channels
        self.tdnn = TDNNBlock(channels * 3 if global_context else channels, attention_channels, 1, 1)
        self.tanh = nn.Tanh()
        self.conv = Conv1d(in_channels=attention_channels, out_channels=channels, kernel_size=1)


class _NotBuilt(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(f'{type(self).__name__} is not built on the HIP engine (ASP is); '
                                  'the reference wires it shape-inconsistently anyway (SURVEY.md section 2 row 4)')


class TemporalAveragePooling(_NotBuilt):
    pass


class TemporalStatisticsPooling(_NotBuilt):
    pass


class SelfAttentivePooling(_NotBuilt):
    pass


class TemporalStatsPool(nn.Module):
    """TSTP (pooling.py:128-146): parameter-free; runs inside the ERes2Net launch graph (csrc/eres2net.hip:
    vp_time_moments with the unbiased variance + 1e-8)."""
