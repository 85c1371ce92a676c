"""The optimisers build_optimizer can name, on the MI355X engine (ppvector/optimizer/__init__.py:12-18 resolves
configs.optimizer_conf.optimizer against paddle.optimizer; every shipped YAML: Adam with weight_decay 1e-6; AdamW, Momentum and SGD
are the other members a config is likely to name -- [3P-memory] update rules, stated per class).

All parameters live in ONE flat f32 buffer (their .data are views into it) with flat gradient / moment buffers beside
it: one kernel launch per step (csrc/train_ops.hip: vp_adam_step_f32), and the data-parallel gradient average is one
all-reduce over the flat gradient buffer (ppvector/train/ddp.py).

Gradients reach the flat buffer by `pack_grads` (one or two launches of vp_pack_segments_f32 over a pointer table), not
through .grad views: with .grad bound to views autograd ACCUMULATES into them, one add_ launch per parameter -- 148 per
ECAPA step, a seventh of the step's launches.  `clear_grad` therefore drops the .grad tensors (autograd then hands each
gradient over without a kernel), `pack_grads` gathers them (zeros for a parameter that received none) and `step` runs on the
packed buffer."""
import torch

from ppvector import _native as N


def _no_unbuilt_options(name, kwargs):
    """paddle.optimizer.* keyword arguments that change the update and are not built raise; the ones that do not are dropped."""
    for k in ('grad_clip', 'lr_ratio', 'apply_decay_param_fun'):
        if kwargs.pop(k, None) is not None:
            raise NotImplementedError(f'{name}({k}=...) is not built on the HIP engine')
    if kwargs.pop('amsgrad', False):
        raise NotImplementedError(f'{name}(amsgrad=True) changes the update rule and is not built on the HIP engine')
    if kwargs.pop('rescale_grad', None) not in (None, 1, 1.0):
        raise NotImplementedError(f'{name}(rescale_grad != 1) changes the update rule and is not built on the HIP engine')
    for k in ('name', 'lazy_mode', 'multi_precision', 'use_multi_tensor'):       # no effect on the arithmetic of an f32 flat buffer
        kwargs.pop(k, None)
    if kwargs:
        raise TypeError(f'{name}: unexpected keyword arguments {sorted(kwargs)}')


class FlatOptimizer:
    """Flat parameter / gradient buffers, gradient packing, learning-rate lookup: what every optimiser of the engine shares.
    Subclasses add their state buffers (named like the slots of the paddle optimiser's state_dict, `state_slots`) and `_update`."""

    def __init__(self, parameters, learning_rate):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError(f'{type(self).__name__}: no trainable parameters')
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.lr = learning_rate
        self.t = 0
        self._packed = False

    def _zeros(self):
        return torch.zeros_like(self.flat)

    def get_lr(self):
        return self.lr() if callable(self.lr) else (self.lr.get_lr() if hasattr(self.lr, 'get_lr') else float(self.lr))

    def clear_grad(self):
        for p in self.params:
            p.grad = None
        self._packed = False

    def pack_range(self, params):
        """Gather the .grad tensors of `params` into their slots of the flat gradient buffer (zeros where .grad is None)."""
        import ctypes as C
        todo = [p for p in params if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self._offset(p)]
        if not todo:
            return
        keep = [None if p.grad is None else p.grad.contiguous().float() for p in todo]
        if not self.grad.is_cuda:                  # host tensors (the CPU tests of the data-parallel plumbing): plain copies
            for p, g in zip(todo, keep):
                o = self._offset(p)
                self.grad[o:o + p.numel()] = 0.0 if g is None else g.reshape(-1)
            return
        n = len(todo)
        srcs = (C.c_void_p * n)(*[None if g is None else g.data_ptr() for g in keep])
        offs = (C.c_longlong * n)(*[self._offset(p) for p in todo])
        sizes = (C.c_longlong * n)(*[p.numel() for p in todo])
        ctx = N.ctx(self.flat.device)
        N.check(N.lib().vp_pack_segments_f32(ctx, srcs, offs, sizes, n, self.grad.data_ptr(), N.stream_ptr()), ctx)

    def pack_grads(self):
        """All gradients into the flat buffer (idempotent until the next clear_grad): call before reading `self.grad`."""
        if not self._packed:
            self.pack_range(self.params)
            self._packed = True

    def step(self, grad_scale=1.0):
        self.pack_grads()
        self.t += 1
        if self.flat.device.type != 'cuda':
            raise N.VpmiError(f'{type(self).__name__}.step runs on the GPU: the engine has no CPU fallback')
        ctx = N.ctx(self.flat.device)
        N.check(self._update(ctx, float(self.get_lr()), float(grad_scale)), ctx)
        N.bump_weights_epoch()                     # parameters changed behind torch's version counters: packed engines are stale

    def _offset(self, p):
        return (p.data.data_ptr() - self.flat.data_ptr()) // 4

    def state_slots(self):
        """[(paddle state_dict suffix, flat buffer)], e.g. ('moment1_0', self.m): what the .pdopt exporter walks."""
        raise NotImplementedError

    def state_dict(self):
        return {**{k: v for k, v in self.state_slots()}, 't': self.t}

    def set_state_dict(self, sd):
        legacy = {'moment1_0': 'm', 'moment2_0': 'v'}
        for k, buf in self.state_slots():
            buf.copy_(sd[k] if k in sd else sd[legacy[k]])
        self.t = int(sd['t'])


class Adam(FlatOptimizer):
    """paddle.optimizer.Adam(learning_rate, beta1, beta2, epsilon, parameters, weight_decay): weight_decay (a float = L2Decay) is
    COUPLED -- added to the gradient before the moments."""

    def __init__(self, parameters, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0, **kwargs):
        _no_unbuilt_options(type(self).__name__, kwargs)
        super().__init__(parameters, learning_rate)
        self.m, self.v = self._zeros(), self._zeros()
        self.beta1, self.beta2, self.eps, self.wd = beta1, beta2, epsilon, float(weight_decay or 0.0)

    def state_slots(self):
        return [('moment1_0', self.m), ('moment2_0', self.v)]

    def _update(self, ctx, lr, grad_scale):
        return N.lib().vp_adam_step_f32(ctx, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                        self.flat.numel(), lr, self.beta1, self.beta2, self.eps, self.wd, self.t, grad_scale,
                                        N.stream_ptr())


class AdamW(Adam):
    """paddle.optimizer.AdamW(..., weight_decay=0.01): DECOUPLED decay, p *= 1 - lr * weight_decay before the Adam update; the
    moments never see it ([3P-memory] paddle/phi adamw kernel with lr_ratio = 1 and no apply_decay_param_fun)."""

    def __init__(self, parameters, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.01, **kwargs):
        super().__init__(parameters, learning_rate, beta1, beta2, epsilon, weight_decay, **kwargs)

    def _update(self, ctx, lr, grad_scale):
        return N.lib().vp_adamw_step_f32(ctx, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.flat.numel(), lr, self.beta1, self.beta2, self.eps, self.wd, self.t, grad_scale,
                                         N.stream_ptr())


class Momentum(FlatOptimizer):
    """paddle.optimizer.Momentum(learning_rate, momentum=0.9, parameters, use_nesterov=False, weight_decay=None):
    g += wd * p; velocity = momentum * velocity + g; p -= lr * velocity (use_nesterov: p -= lr * (g + momentum * velocity))."""

    def __init__(self, parameters, learning_rate=1e-3, momentum=0.9, use_nesterov=False, weight_decay=None, **kwargs):
        _no_unbuilt_options(type(self).__name__, kwargs)
        super().__init__(parameters, learning_rate)
        self.velocity = self._zeros()
        self.momentum, self.nesterov, self.wd = float(momentum), bool(use_nesterov), float(weight_decay or 0.0)

    def state_slots(self):
        return [('velocity_0', self.velocity)]

    def _update(self, ctx, lr, grad_scale):
        return N.lib().vp_momentum_step_f32(ctx, self.flat.data_ptr(), self.grad.data_ptr(), self.velocity.data_ptr(), self.flat.numel(),
                                            lr, self.momentum, self.wd, int(self.nesterov), grad_scale, N.stream_ptr())


class SGD(Momentum):
    """paddle.optimizer.SGD(learning_rate, parameters, weight_decay=None): p -= lr * (g + wd * p) -- Momentum with momentum 0."""

    def __init__(self, parameters, learning_rate=1e-3, weight_decay=None, **kwargs):
        super().__init__(parameters, learning_rate, 0.0, False, weight_decay, **kwargs)

    def state_slots(self):
        return []
