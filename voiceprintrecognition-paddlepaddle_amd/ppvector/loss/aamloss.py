"""AAMLoss on the MI355X engine (ppvector/loss/aamloss.py:8-53).

forward(inputs, labels) takes the classifier's dict and returns the scalar mean loss;
update(margin) is what MarginScheduler calls every step (optimizer/scheduler.py:69,76).  The
margin / one-hot mix / scale / softmax-CE(label_smoothing) chain is one kernel over the logits
(csrc/head.hip aam_ce_rows_kernel) -- no one-hot tensor, no (B, C) temporaries.
"""
import math

import torch
from torch import nn

from ppvector import _native as N


class AAMLoss(nn.Module):
    def __init__(self, margin=0.2, scale=32, easy_margin=False, label_smoothing=0.0):
        super().__init__()
        self.scale = scale
        self.easy_margin = easy_margin
        self.label_smoothing = label_smoothing
        self.update(margin)

    def forward(self, inputs, labels):
        from ppvector.models.fc import CosineHeadOutputs
        if isinstance(inputs, CosineHeadOutputs) and not dict.__contains__(inputs, 'logits'):
            if inputs.training:
                if torch.is_grad_enabled():             # training: head + loss + d emb + d W class-tiled when the shape allows it
                    from ppvector.train.functions import HeadLoss
                    loss, pred = HeadLoss.apply(inputs.x, inputs.W, labels, self.margin, self.scale, self.label_smoothing, self.easy_margin)
                    inputs.pred = pred if pred.numel() else None
                    self.row_loss = None                # (the fused training head reports the mean only; the evaluation paths set row losses)
                    return loss
            elif inputs.W.shape[0] % 4 == 0 and inputs.W.shape[0] <= 256:
                return self._tiled(inputs, labels)      # evaluation-mode head + loss in one pass over the class weights
        features, logits = inputs['features'], inputs['logits']
        if not logits.is_cuda:
            raise N.VpmiError('AAMLoss needs GPU tensors: the engine has no CPU fallback')
        if torch.is_grad_enabled() and logits.requires_grad:            # training: loss with its backward (csrc/head.hip)
            from ppvector.train.functions import AamCe
            return AamCe.apply(logits, labels, self.margin, self.scale, self.label_smoothing, self.easy_margin)
        logits = logits.contiguous().float()
        labels = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        B, Cn = logits.shape
        lib, ctx = N.lib(), N.ctx(logits.device)
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        row = torch.empty((B,), dtype=torch.float32, device=logits.device)
        N.check(lib.vp_aam_ce_fwd(ctx, logits.data_ptr(), labels.data_ptr(), B, Cn, float(self.margin),
                                  float(self.scale), float(self.label_smoothing), int(bool(self.easy_margin)),
                                  loss.data_ptr(), row.data_ptr(), N.stream_ptr()), ctx)
        self.row_loss = row
        return loss[0]

    def _tiled(self, inputs, labels):
        """csrc/head_tiled.hip: cosine logits (exact f32 matrix cores) -> margin -> online log-sum-exp per 64-class tile, merged per
        row; the (B, C) logits never exist."""
        x, W = inputs.x, inputs.W
        if not x.is_cuda:
            raise N.VpmiError('AAMLoss needs GPU tensors: the engine has no CPU fallback')
        labels = labels.to(device=x.device, dtype=torch.int64).reshape(-1).contiguous()
        B, D = x.shape
        Cn = W.shape[1]
        lib, ctx = N.lib(), N.ctx(x.device)
        out = torch.empty((1 + 2 * B,), dtype=torch.float32, device=x.device)            # loss | row losses | row log-sum-exps
        pred = torch.empty((B,), dtype=torch.int32, device=x.device)
        ws = inputs._ws.get(lib.vp_cosine_aam_tiled_workspace_bytes(B, D, Cn), x.device)
        N.check(lib.vp_cosine_aam_tiled_fwd(ctx, x.data_ptr(), W.data_ptr(), labels.data_ptr(), B, D, Cn, float(self.margin), float(self.scale),
                                            float(self.label_smoothing), int(bool(self.easy_margin)), out.data_ptr(), out[1:].data_ptr(),
                                            out[1 + B:].data_ptr(), None, pred.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
        self.row_loss = out[1:1 + B]
        inputs.pred = pred
        return out[0]

    def update(self, margin=0.2):
        self.margin = margin
        self.cos_m = math.cos(margin)
        self.sin_m = math.sin(margin)
        self.th = math.cos(math.pi - margin)
        self.mmm = 1.0 + math.cos(math.pi - margin)
