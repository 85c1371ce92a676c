"""SphereFace2 on the MI355X engine (ppvector/loss/sphereface2.py:9-77): one binary loss per speaker, margin type 'A'
(arc) or 'C' (cosine), with the module's learnable (1, 1) bias.  One launch gives the value and both gradients
(csrc/losses.hip sphereface2_rows_kernel)."""
import math

import torch
from torch import nn

from ppvector import _native as N


class SphereFace2(nn.Module):
    def __init__(self, margin=0.2, scale=32.0, lanbuda=0.7, t=3, margin_type='C'):
        super().__init__()
        self.scale, self.t, self.lanbuda, self.margin_type = scale, t, lanbuda, margin_type
        self.bias = nn.Parameter(torch.zeros(1, 1))
        self.update(margin)

    def forward(self, inputs, labels):
        logits = inputs['logits']
        if not logits.is_cuda:
            raise N.VpmiError('SphereFace2 needs GPU tensors: the engine has no CPU fallback')
        bias = self.bias.to(logits.device)
        type_a = self.margin_type == 'A'
        if torch.is_grad_enabled() and (logits.requires_grad or bias.requires_grad):
            from ppvector.train.functions import SphereFace2Fn
            return SphereFace2Fn.apply(logits, labels, bias, self.margin, self.scale, self.lanbuda, self.t, type_a)
        logits = logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        B, Cn = logits.shape
        lib, ctx = N.lib(), N.ctx(logits.device)
        out = torch.empty((1 + B,), dtype=torch.float32, device=logits.device)
        N.check(lib.vp_sphereface2(ctx, logits.data_ptr(), labels.data_ptr(), bias.detach().float().contiguous().data_ptr(), B, Cn,
                                   float(self.margin), float(self.scale), float(self.lanbuda), int(self.t), int(type_a), 0.0,
                                   out.data_ptr(), out[1:].data_ptr(), None, None, None, N.stream_ptr()), ctx)
        self.row_loss = out[1:]
        return out[0]

    def update(self, margin=0.2):
        self.margin = margin
        self.cos_m, self.sin_m = math.cos(margin), math.sin(margin)
        self.th, self.mmm = math.cos(math.pi - margin), 1.0 + math.cos(math.pi - margin)
