"""AMLoss on the MI355X engine (ppvector/loss/amloss.py:6-28): CE(sum) / B over scale * (cos - margin * onehot)."""
from ppvector import _native as N
from ppvector.loss._margin import MarginSoftmax


class AMLoss(MarginSoftmax):
    kind = N.VP_LOSS_AM

    def __init__(self, margin=0.2, scale=30, label_smoothing=0.0):
        super().__init__()
        self.margin, self.scale, self.label_smoothing = margin, scale, label_smoothing

    def forward(self, inputs, labels):
        return self._loss(inputs, labels, self.margin, self.scale, self.label_smoothing)

    def update(self, margin=0.2):
        self.margin = margin
