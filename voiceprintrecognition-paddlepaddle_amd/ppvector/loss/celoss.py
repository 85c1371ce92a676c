"""CELoss on the MI355X engine (ppvector/loss/celoss.py:6-22): plain cross-entropy on the head's logits; update() is a
no-op so MarginScheduler can drive it like the margin losses."""
from ppvector import _native as N
from ppvector.loss._margin import MarginSoftmax


class CELoss(MarginSoftmax):
    kind = N.VP_LOSS_CE

    def __init__(self, label_smoothing=0.0):
        super().__init__()
        self.label_smoothing = label_smoothing

    def forward(self, inputs, labels):
        return self._loss(inputs, labels, 0.0, 1.0, self.label_smoothing)

    def update(self, margin=0.2):
        pass
