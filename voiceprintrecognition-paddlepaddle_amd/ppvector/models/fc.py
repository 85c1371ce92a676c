"""Classifier head on the MI355X engine (ppvector/models/fc.py:6-53).

'Cosine': logits = normalize(x) @ normalize(W, axis=0) with W [input_dim, num_speakers*K]
(Xavier-uniform, fc.py:31-34), computed by vp_cosine_logits_f32 (csrc/head.hip) in exact f32 on
the f32 matrix cores.  Returns the reference's dict {"features", "logits"}.
"""
import torch
from torch import nn

from ppvector import _native as N


class SpeakerIdentification(nn.Module):
    def __init__(self, input_dim, num_speakers, classifier_type='Cosine', K=1, num_blocks=0, inter_dim=512):
        super().__init__()
        self.classifier_type = classifier_type
        self.blocks = nn.ModuleList()
        if num_blocks != 0:
            raise NotImplementedError('num_blocks > 0 (DenseLayer stack) is not built on the HIP engine; '
                                      'every shipped config uses num_blocks: 0')
        if self.classifier_type == 'Cosine':
            w = torch.empty(input_dim, num_speakers * K)
            nn.init.xavier_uniform_(w)
            self.weight = nn.Parameter(w)
        elif self.classifier_type == 'Linear':
            raise NotImplementedError("classifier_type 'Linear' is not built on the HIP engine ('Cosine' is)")
        else:
            raise ValueError(f'不支持该输出层：{self.classifier_type}')
        self._ws = N.Workspace()

    def forward(self, features):
        x = features
        if not x.is_cuda:
            raise N.VpmiError('SpeakerIdentification needs GPU tensors: the engine has no CPU fallback')
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad) and self.training:
            from ppvector.train.functions import CosineLogits       # training: logits with their backward (csrc/head.hip)
            return {"features": features, "logits": CosineLogits.apply(x.float(), self.weight)}
        x = x.contiguous().float()
        W = self.weight.detach().contiguous().float()
        B, D = x.shape
        Cn = W.shape[1]
        lib, ctx = N.lib(), N.ctx(x.device)
        logits = torch.empty((B, Cn), dtype=torch.float32, device=x.device)
        ws = self._ws.get(lib.vp_cosine_logits_workspace_bytes(B, D, Cn), x.device)
        N.check(lib.vp_cosine_logits_f32(ctx, x.data_ptr(), W.data_ptr(), B, D, Cn, logits.data_ptr(),
                                         ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
        return {"features": features, "logits": logits}
