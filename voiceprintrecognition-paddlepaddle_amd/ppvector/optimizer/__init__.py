from .scheduler import MarginScheduler, cosine_decay_with_warmup  # noqa: F401
import math

from .adam import SGD, Adam, AdamW, Momentum  # noqa: F401,E402

__all__ = ['build_optimizer', 'build_lr_scheduler', 'Adam', 'AdamW', 'Momentum', 'SGD', 'MarginScheduler', 'cosine_decay_with_warmup',
           'CosineAnnealingDecay']

_OPTIMIZERS = {c.__name__: c for c in (Adam, AdamW, Momentum, SGD)}
_PADDLE_ONLY = frozenset(('Adadelta', 'Adagrad', 'Adamax', 'Lamb', 'LBFGS', 'RMSProp', 'ASGD', 'NAdam', 'RAdam', 'Rprop'))


def build_optimizer(parameters, learning_rate, configs):
    """ppvector/optimizer/__init__.py:12-18: class by name from configs.optimizer_conf.optimizer (every shipped YAML: 'Adam'),
    kwargs from optimizer_args (weight_decay 1e-6).  An unknown name is an AttributeError like the reference's getattr on its
    module; a paddle.optimizer member that is not built says so."""
    use_optimizer = configs.optimizer_conf.get('optimizer', 'Adam')
    optimizer_args = configs.optimizer_conf.get('optimizer_args', {})
    cls = _OPTIMIZERS.get(use_optimizer)
    if cls is None:
        if use_optimizer in _PADDLE_ONLY:
            raise NotImplementedError(f'optimizer {use_optimizer} is not built on the HIP engine ({", ".join(sorted(_OPTIMIZERS))} are)')
        raise AttributeError(f"module '{__name__}' has no attribute '{use_optimizer}'")
    return cls(parameters=parameters, learning_rate=learning_rate, **dict(optimizer_args or {}))


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


class CosineAnnealingDecay:
    """paddle.optimizer.lr.CosineAnnealingDecay(learning_rate, T_max, eta_min=0): the other scheduler the reference's factory
    knows by name (optimizer/__init__.py:24-25 fills in T_max = int(max_epoch * 1.2) * step_per_epoch).  [3P-memory] paddle steps it
    recursively; in closed form lr(t) = eta_min + (lr0 - eta_min) * (1 + cos(pi * t / T_max)) / 2, which the recursion reproduces
    for t <= T_max (beyond T_max the closed form continues the cosine, as the recursion does)."""

    def __init__(self, learning_rate, T_max, eta_min=0.0, last_epoch=-1, verbose=False):
        if int(T_max) <= 0:
            raise ValueError('CosineAnnealingDecay: T_max must be a positive integer')
        self.base_lr, self.T_max, self.eta_min = float(learning_rate), int(T_max), float(eta_min)
        self.i = max(int(last_epoch), -1) + 1

    def step(self):
        self.i += 1

    def get_lr(self):
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * self.i / self.T_max)) / 2.0

    __call__ = get_lr


def build_lr_scheduler(step_per_epoch, configs):
    """ppvector/optimizer/__init__.py:21-33: 'WarmupCosineSchedulerLR' (the default and every shipped YAML) or
    'CosineAnnealingDecay', with the same defaults filled in."""
    use_scheduler = configs.optimizer_conf.get('scheduler', 'WarmupCosineSchedulerLR')
    scheduler_args = dict(configs.optimizer_conf.get('scheduler_args', {}) or {})
    if use_scheduler == 'CosineAnnealingDecay':
        scheduler_args.setdefault('T_max', int(configs.train_conf.max_epoch * 1.2) * step_per_epoch)
        return CosineAnnealingDecay(**scheduler_args)
    if use_scheduler != 'WarmupCosineSchedulerLR':
        raise NotImplementedError(f'scheduler {use_scheduler} is not built (WarmupCosineSchedulerLR and CosineAnnealingDecay are)')
    scheduler_args.setdefault('fix_epoch', configs.train_conf.max_epoch)
    scheduler_args.setdefault('step_per_epoch', step_per_epoch)
    return _WarmupCosine(**scheduler_args)
