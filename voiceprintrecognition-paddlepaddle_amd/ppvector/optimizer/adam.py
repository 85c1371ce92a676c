"""Adam with coupled L2 on the MI355X engine (paddle.optimizer.Adam(parameters, learning_rate, weight_decay) as built by
ppvector/optimizer/__init__.py:12-18; configs/*.yml: weight_decay 1e-6).

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


class Adam:
    def __init__(self, parameters, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, weight_decay=0.0):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError('Adam: no trainable parameters')
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.lr = learning_rate
        self.beta1, self.beta2, self.eps, self.wd = beta1, beta2, epsilon, float(weight_decay or 0.0)
        self.t = 0
        self._packed = False

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
            raise N.VpmiError('Adam.step runs on the GPU: the engine has no CPU fallback')
        ctx = N.ctx(self.flat.device)
        N.check(N.lib().vp_adam_step_f32(ctx, self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.flat.numel(), float(self.get_lr()), self.beta1, self.beta2, self.eps, self.wd, self.t,
                                         float(grad_scale), N.stream_ptr()), ctx)
        N.bump_weights_epoch()                     # parameters changed behind torch's version counters: packed engines are stale

    def _offset(self, p):
        return (p.data.data_ptr() - self.flat.data_ptr()) // 4

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 't': self.t}

    def set_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v']); self.t = int(sd['t'])
