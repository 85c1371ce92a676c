"""Backward in segments: where a training forward may be cut so that the backward pass runs as a chain of separately
captured HIP graphs with a gradient all-reduce between them (ppvector/train/step.py: GraphedTrainStep).

The reference gets "gradient all-reduce overlapped with backward" from fleet.distributed_model's reducer hooks
(ppvector/trainer.py:318-320); an eager autograd hook cannot fire inside a graph replay, so the overlap is built from the
other side: the train-mode forward of a backbone calls

    a, b = cut(a, b)            # every tensor that is live across this point of the network

at a few layer boundaries.  Without an active recorder `cut` returns its arguments unchanged (the eager step, the tests'
float64 comparisons, inference: nothing changes).  With one, the tensors are replaced by detached leaves, which splits the
autograd tape into a chain of stages; `Recorder.backward` then differentiates stage by stage, last stage first:

    stage K:   loss.backward()                               -> gradients of the late parameters and of the K-th cut's leaves
    stage k:   backward(originals of cut k+1, their leaves' .grad)

and calls `between(k)` after every stage -- that is where GraphedTrainStep ends one graph capture, packs the stage's parameter
gradients and starts their all-reduce while the next (earlier) stage is still to be replayed.  Passing ALL live tensors at a
cut is what makes the stages a chain: a tensor consumed on both sides of a later cut (ECAPA's block outputs feed the next block
AND the MFA concatenation) must go through every cut in between, so that its two gradient contributions meet in one leaf.
"""
import torch

_ACTIVE = None


class Recorder:
    def __init__(self):
        self.cuts = []            # [(originals, leaves)] in forward order

    def __enter__(self):
        global _ACTIVE
        if _ACTIVE is not None:
            raise RuntimeError('nested cut recorders')
        _ACTIVE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = None
        return False

    def _cut(self, tensors):
        keep = [t for t in tensors if t.requires_grad]
        if not keep:
            return tensors
        leaves = [t.detach().requires_grad_(True) for t in keep]
        for t, l in zip(keep, leaves):                    # (a producer's bf16 twin of the tensor travels with it: train/functions.py)
            if getattr(t, '_vp_bf16', None) is not None:
                l._vp_bf16 = t._vp_bf16
                if getattr(t, '_vp_bf16_only', False):        # ... which may be the ONLY copy (t is a memory-less placeholder)
                    l._vp_bf16_only = True
        self.cuts.append((keep, leaves))
        it = iter(leaves)
        return tuple(next(it) if t.requires_grad else t for t in tensors)

    def backward(self, loss, between=None):
        """Differentiate `loss` stage by stage (last stage first); between(i) runs after stage i (0 = the stage that holds
        the loss), including the last one."""
        loss.backward()
        if between is not None:
            between(0)
        for i, (origs, leaves) in enumerate(reversed(self.cuts), start=1):
            pairs = []
            for o, l in zip(origs, leaves):
                g = l.grad
                if g is None:
                    continue
                l.grad = None
                if o.grad_fn is None and o.is_leaf and o.grad is None:
                    # a tensor that only travels through this cut (a leaf of the previous one, consumed further down): its gradient so
                    # far IS the later leaf's -- handed on as the same tensor (autograd.backward on a leaf would clone it: 60 us per
                    # (B*T, 512) tensor); what the stage adds to it is accumulated in place by autograd
                    o.grad = g
                else:
                    pairs.append((o, g))
            if pairs:
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            if between is not None:
                between(i)

    @property
    def n_stages(self):
        return len(self.cuts) + 1


def cut(*tensors):
    """Identity outside a Recorder; inside one, the given tensors (ALL tensors live across this point) become the leaves of the
    next stage.  Always returns a tuple of the same length."""
    if _ACTIVE is None or not torch.is_grad_enabled():
        return tensors
    return _ACTIVE._cut(tensors)
