from .scheduler import MarginScheduler, cosine_decay_with_warmup  # noqa: F401
from .adam import Adam  # noqa: F401,E402

__all__ = ['build_optimizer', 'build_lr_scheduler', 'Adam', 'MarginScheduler', 'cosine_decay_with_warmup']


def build_optimizer(parameters, learning_rate, configs):
    """ppvector/optimizer/__init__.py:12-18: class by name from configs.optimizer_conf.optimizer (every shipped YAML: 'Adam'),
    kwargs from optimizer_args (weight_decay 1e-6)."""
    use_optimizer = configs.optimizer_conf.get('optimizer', 'Adam')
    optimizer_args = configs.optimizer_conf.get('optimizer_args', {})
    if use_optimizer != 'Adam':
        raise NotImplementedError(f'optimizer {use_optimizer} is not built on the HIP engine (Adam is)')
    return Adam(parameters=parameters, learning_rate=learning_rate, **dict(optimizer_args))


class _WarmupCosine:
    """The per-step LR table of cosine_decay_with_warmup behind the scheduler interface the trainer uses (.step(), .get_lr())."""

    def __init__(self, **kw):
        self.table = cosine_decay_with_warmup(**kw)
        self.i = 0

    def step(self):
        self.i += 1

    def get_lr(self):
        return float(self.table[min(self.i, len(self.table) - 1)])

    __call__ = get_lr


def build_lr_scheduler(step_per_epoch, configs):
    """ppvector/optimizer/__init__.py:21-33 for the default 'WarmupCosineSchedulerLR'."""
    use_scheduler = configs.optimizer_conf.get('scheduler', 'WarmupCosineSchedulerLR')
    scheduler_args = dict(configs.optimizer_conf.get('scheduler_args', {}))
    if use_scheduler != 'WarmupCosineSchedulerLR':
        raise NotImplementedError(f'scheduler {use_scheduler} is not built (WarmupCosineSchedulerLR is)')
    scheduler_args.setdefault('fix_epoch', configs.train_conf.max_epoch)
    scheduler_args.setdefault('step_per_epoch', step_per_epoch)
    return _WarmupCosine(**scheduler_args)
