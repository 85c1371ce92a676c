"""Shared body of the margin-softmax criteria (AM / ARM / CE / SubCenter): argument checks, the eval-time forward through
vp_margin_ce_fwd and the training-time forward + backward through ppvector.train.functions.MarginCe (csrc/losses.hip)."""
import math

import torch
from torch import nn

from ppvector import _native as N


class MarginTable:
    """The criterion's margin as five device floats [m, cos m, sin m, cos(pi - m), 1 + cos(pi - m)] (vp_set_margin_table): while
    the table is installed (`with table:`) every loss launch reads the margin from it when the kernel RUNS, so a captured HIP
    graph follows MarginScheduler (optimizer/scheduler.py:69,76) without a re-capture.  `sync()` uploads the criterion's current
    margin when it changed; the values are the same double -> float roundings the launch-scalar path makes."""

    def __init__(self, criterion, device):
        self.criterion, self.device = criterion, torch.device(device)
        self.has_margin = hasattr(criterion, 'margin')
        self.buf = torch.zeros(5, dtype=torch.float32, device=self.device) if self.has_margin else None
        self._last = None
        self.sync()

    def sync(self):
        if not self.has_margin:
            return
        m = float(self.criterion.margin)
        if m != self._last:
            vals = [m, math.cos(m), math.sin(m), math.cos(math.pi - m), 1.0 + math.cos(math.pi - m)]
            self.buf.copy_(torch.tensor(vals, dtype=torch.float64).to(torch.float32))
            self._last = m

    def __enter__(self):
        if self.has_margin:
            self.sync()
            c = N.ctx(self.device)
            N.check(N.lib().vp_set_margin_table(c, self.buf.data_ptr()), c)
        return self

    def __exit__(self, *exc):
        if self.has_margin:
            c = N.ctx(self.device)
            N.lib().vp_set_margin_table(c, None)
        return False


class MarginSoftmax(nn.Module):
    kind = None                                   # one of N.VP_LOSS_*
    K = 1

    def _loss(self, inputs, labels, margin, scale, label_smoothing, easy_margin=False):
        logits = inputs['logits']
        if not logits.is_cuda:
            raise N.VpmiError(f'{type(self).__name__} needs GPU tensors: the engine has no CPU fallback')
        if logits.shape[1] % self.K:
            raise ValueError(f'logits have {logits.shape[1]} columns, not a multiple of K={self.K}')
        if torch.is_grad_enabled() and logits.requires_grad:
            from ppvector.train.functions import MarginCe
            return MarginCe.apply(logits, labels, self.kind, self.K, margin, scale, label_smoothing, easy_margin)
        logits = logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        B, CK = logits.shape
        lib, ctx = N.lib(), N.ctx(logits.device)
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        row = torch.empty((B,), dtype=torch.float32, device=logits.device)
        N.check(lib.vp_margin_ce_fwd(ctx, logits.data_ptr(), labels.data_ptr(), B, CK // self.K, self.K, self.kind, float(margin),
                                     float(scale), float(label_smoothing), int(bool(easy_margin)), loss.data_ptr(), row.data_ptr(),
                                     N.stream_ptr()), ctx)
        self.row_loss = row
        return loss[0]
