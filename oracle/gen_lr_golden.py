"""Golden LR tables from the reference's OWN cosine_decay_with_warmup (ppvector/optimizer/scheduler.py:6-40).  TEST INFRASTRUCTURE.

Runs only where /root/reference exists.  The reference function is executed unmodified; the one Paddle class it needs,
paddle.optimizer.lr.PiecewiseDecay, is restated here from its published behaviour [3P-memory]: get_lr() returns values[i]
for the first i with last_epoch < boundaries[i], else values[-1]; last_epoch starts at 0 and step() adds one.
Usage: python oracle/gen_lr_golden.py   ->  tests/golden/lr_table_ref.npz
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/ppvector/optimizer/scheduler.py'
CONFIGS = {'ecapa_yaml': dict(learning_rate=0.001, step_per_epoch=7, fix_epoch=20, warmup_epoch=5, min_lr=1e-5),
           'short_warm': dict(learning_rate=0.01, step_per_epoch=3, fix_epoch=6, warmup_epoch=1, min_lr=0.0)}


class PiecewiseDecay:
    def __init__(self, boundaries, values, last_epoch=-1, verbose=False):
        self.boundaries, self.values, self.last_epoch = list(boundaries), list(values), 0

    def step(self):
        self.last_epoch += 1

    def get_lr(self):
        for i, b in enumerate(self.boundaries):
            if self.last_epoch < b:
                return self.values[i]
        return self.values[len(self.values) - 1]


def main():
    if not os.path.exists(REF):
        print('reference not present; nothing to do')
        return 0
    paddle = types.ModuleType('paddle')
    paddle.optimizer = types.ModuleType('paddle.optimizer')
    paddle.optimizer.lr = types.ModuleType('paddle.optimizer.lr')
    paddle.optimizer.lr.PiecewiseDecay = PiecewiseDecay
    sys.modules.update({'paddle': paddle, 'paddle.optimizer': paddle.optimizer, 'paddle.optimizer.lr': paddle.optimizer.lr})
    ns = {}
    exec(compile(open(REF).read(), REF, 'exec'), ns)
    out = {}
    for name, kw in CONFIGS.items():
        sch = ns['cosine_decay_with_warmup'](**kw)
        n = kw['fix_epoch'] * kw['step_per_epoch'] + 3              # three steps past the table's end
        lrs = []
        for _ in range(n):
            lrs.append(sch.get_lr())
            sch.step()
        out[name] = np.asarray(lrs, dtype=np.float64)
        out[name + '_args'] = np.asarray([kw['learning_rate'], kw['step_per_epoch'], kw['fix_epoch'], kw['warmup_epoch'], kw['min_lr']])
    np.savez(os.path.join(ROOT, 'tests', 'golden', 'lr_table_ref.npz'), **out)
    print('wrote lr_table_ref.npz', {k: v.shape for k, v in out.items()})
    return 0


if __name__ == '__main__':
    sys.exit(main())
