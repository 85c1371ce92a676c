"""SubCenterLoss on the MI355X engine (ppvector/loss/subcenterloss.py:8-61): the head carries K sub-centres per speaker
(SpeakerIdentification(K=...), columns c*K .. c*K+K-1); a class scores the best of its sub-centres, then the AAM margin."""
import math

from ppvector import _native as N
from ppvector.loss._margin import MarginSoftmax


class SubCenterLoss(MarginSoftmax):
    kind = N.VP_LOSS_SUBCENTER

    def __init__(self, margin=0.2, scale=32, easy_margin=False, K=3, label_smoothing=0.0):
        super().__init__()
        self.scale, self.K, self.easy_margin, self.label_smoothing = scale, K, easy_margin, label_smoothing
        self.update(margin)

    def forward(self, inputs, labels):
        return self._loss(inputs, labels, self.margin, self.scale, self.label_smoothing, self.easy_margin)

    def update(self, margin=0.2):
        self.margin = margin
        self.cos_m, self.sin_m = math.cos(margin), math.sin(margin)
        self.th, self.mmm = math.cos(math.pi - margin), 1.0 + math.cos(math.pi - margin)
