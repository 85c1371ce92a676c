"""ARMLoss on the MI355X engine (ppvector/loss/armloss.py:6-35): the AM logits with every entry below the target's zeroed
before the cross-entropy."""
from ppvector import _native as N
from ppvector.loss._margin import MarginSoftmax


class ARMLoss(MarginSoftmax):
    kind = N.VP_LOSS_ARM

    def __init__(self, margin=0.2, scale=30, label_smoothing=0.0):
        super().__init__()
        self.margin, self.scale, self.label_smoothing = margin, scale, label_smoothing

    def forward(self, inputs, labels):
        return self._loss(inputs, labels, self.margin, self.scale, self.label_smoothing)

    def update(self, margin=0.2):
        self.margin = margin
