"""Host-side schedules (ppvector/optimizer/scheduler.py): the per-step LR table of
cosine_decay_with_warmup (:6-40) and MarginScheduler (:43-102).  They produce scalars that become
kernel arguments (the AAM margin) -- nothing to accelerate.
"""
import math


def cosine_decay_with_warmup(learning_rate, step_per_epoch, fix_epoch=1000, warmup_epoch=5, min_lr=0.0):
    """Returns the list lr[step] the reference's schedule yields at scheduler step `step` (scheduler.py:6-40): it builds
    paddle PiecewiseDecay(boundaries, values) with values = [lr i / warm for i = 0..warm] + [cos(i) for i = warm..total-1] and
    boundaries = [1..warm] + [warm..total-1] -- the boundary `warm` appears TWICE, so PiecewiseDecay (first boundary > step picks
    the value) skips the warm-up's last entry: lr[step] = lr step / warm below `warm` and cos(step) from `warm` on, where
    cos(i) = min_lr + (lr - min_lr) (1 + cos((i - warm) pi / (total - warm))) / 2; past the table the last value holds."""
    warm = warmup_epoch * step_per_epoch
    total = fix_epoch * int(step_per_epoch)
    table = [learning_rate * i / warm for i in range(min(warm, total))]
    for i in range(max(warm, 0), total):
        table.append(min_lr + (learning_rate - min_lr) * 0.5 * (math.cos((i - warm) * math.pi / (total - warm)) + 1))
    return table


class MarginScheduler:
    def __init__(self, criterion, increase_start_epoch, fix_epoch, step_per_epoch, initial_margin=0.0,
                 final_margin=0.3, increase_type='exp'):
        assert hasattr(criterion, 'update'), "Loss function not has 'update()' attributes."
        self.criterion = criterion
        self.increase_start_step = increase_start_epoch * step_per_epoch
        self.fix_step = fix_epoch * step_per_epoch
        self.initial_margin, self.final_margin = initial_margin, final_margin
        self.increase_type = increase_type
        self.margin = initial_margin
        self.current_step = 0
        self.increase_step = self.fix_step - self.increase_start_step
        self.init_margin()

    def init_margin(self):
        self.criterion.update(margin=self.initial_margin)

    def step(self, current_step=None):
        if current_step is not None:
            self.current_step = current_step
        self.margin = self.iter_margin()
        self.criterion.update(margin=self.margin)
        self.current_step += 1

    def iter_margin(self):
        if self.current_step < self.increase_start_step:
            return self.initial_margin
        if self.current_step >= self.fix_step:
            return self.final_margin
        cur = self.current_step - self.increase_start_step
        if self.increase_type == 'exp':
            ratio = 1.0 - math.exp((cur / self.increase_step) * math.log(1e-3 / (1.0 + 1e-6))) * 1.0
        else:
            ratio = 1.0 * cur / self.increase_step
        return self.initial_margin + (self.final_margin - self.initial_margin) * ratio

    def get_margin(self):
        return self.margin
