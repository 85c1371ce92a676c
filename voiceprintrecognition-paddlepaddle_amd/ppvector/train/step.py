"""One optimisation step, laid out like the body of PPVectorTrainer.__train_epoch (ppvector/trainer.py:206-274):

    outputs = model(features); los = loss(outputs, label); los.backward(); optimizer.step(); optimizer.clear_grad();
    scheduler.step(); margin_scheduler.step()

with the data-parallel gradient average (fleet.distributed_model in the reference, trainer.py:318-320) made explicit:
one all-reduce over the optimiser's flat gradient buffer between backward and the Adam kernel."""
import torch

from ppvector.train.ddp import OverlappedReducer, allreduce_mean_


class TrainStep:
    def __init__(self, model, criterion, optimizer, scheduler=None, margin_scheduler=None, featurizer=None, spec_augment=None,
                 overlap_allreduce=True):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.scheduler, self.margin_scheduler = scheduler, margin_scheduler
        self.featurizer, self.spec_augment = featurizer, spec_augment
        self.step_id = 0
        # data-parallel gradient average: bucketed all-reduce launched from autograd hooks while backward is still running
        self.reducer = OverlappedReducer(optimizer) if overlap_allreduce else None

    def __call__(self, inputs, labels):
        """inputs: waveforms (B, L) when a featurizer was given, else features (B, T, F).  Returns (loss, accuracy) tensors."""
        self.model.train()
        feats = inputs
        if self.featurizer is not None:
            with torch.no_grad():
                feats = self.featurizer(inputs)
                if self.spec_augment is not None:
                    feats = self.spec_augment.batch(feats)
        outputs = self.model(feats)
        loss = self.criterion(outputs, labels)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        else:
            self.optimizer.pack_grads()
            allreduce_mean_(self.optimizer.grad)
        self.optimizer.step()
        self.optimizer.clear_grad()
        with torch.no_grad():
            acc = (outputs['logits'].argmax(dim=1) == labels.to(outputs['logits'].device)).float().mean()
        if self.scheduler is not None:
            self.scheduler.step()
        if self.margin_scheduler is not None:
            self.margin_scheduler.step()
        self.step_id += 1
        return loss.detach(), acc


class GraphedTrainStep(TrainStep):
    """The same step with forward + backward replayed from ONE captured HIP graph.

    The step is ~1000 kernel launches whose count does not depend on the batch.  At the 32 utterances per GPU of the
    strong-scaled configuration (global batch 256 over 8 GPUs) the eager step is bound by the host issuing them (9.4 ms where
    the GPU needs a fraction of that): data-parallel scaling would stall at ~2x.  Here the first `warm` calls run eagerly (the
    plain step: identical semantics, and every lazy table / kernel attribute gets set up outside a capture), then the featurizer
    output is copied into a static buffer and model forward -> criterion -> backward are captured once and replayed.  What stays
    outside the graph, eager, every step: the featurizer (+ SpecAugment, whose masks the host draws), the gradient all-reduce
    (bucketed over the flat buffer; a collective inside a capture is not attempted), flat Adam (the learning rate is a launch
    scalar) and the schedulers.  The loss margin is a launch scalar too: a changed margin (MarginScheduler's ramp) re-captures.
    BatchNorm running statistics are updated by the replayed kernels in place, as in the eager step."""

    def __init__(self, *a, warm=3, bucket_bytes=16 << 20, **kw):
        kw['overlap_allreduce'] = False
        super().__init__(*a, **kw)
        self.warm, self.bucket_bytes = warm, bucket_bytes
        self._graph, self._key = None, None
        self._static = {}
        self.capture_error = None
        self.skip_allreduce = False

    def _features(self, inputs):
        feats = inputs
        if self.featurizer is not None:
            with torch.no_grad():
                feats = self.featurizer(inputs)
                if self.spec_augment is not None:
                    feats = self.spec_augment.batch(feats)
        return feats

    def _capture(self, feats, labels):
        st = self._static
        st['feats'], st['labels'] = feats.clone(), labels.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread-local error mode: another thread of the process (the collective library's watchdog polling its events) must not
        # invalidate the capture
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            outputs = self.model(st['feats'])
            loss = self.criterion(outputs, st['labels'])
            loss.backward()
            self.optimizer.pack_grads()                     # gradients -> flat buffer: part of the replayed sequence
            st['loss'] = loss.detach()
            st['acc'] = (outputs['logits'].detach().argmax(dim=1) == st['labels']).float().mean()
        self._graph = g

    def __call__(self, inputs, labels):
        from ppvector import _native as N
        if self.capture_error is not None or self.step_id < self.warm:
            return super().__call__(inputs, labels)
        self.model.train()
        feats = self._features(inputs)
        labels = labels.to(feats.device)
        key = (tuple(feats.shape), feats.dtype, tuple(labels.shape), float(getattr(self.criterion, 'margin', 0.0)))
        if self._graph is None or key != self._key:
            try:
                self.optimizer.clear_grad()
                self._capture(feats, labels)
                self._key = key
                self.optimizer.clear_grad()                     # a capture executes nothing, but keep the buffer defined
            except Exception as e:                              # noqa: BLE001 -- fall back to the eager step for good
                self.capture_error = f'{type(e).__name__}: {e}'[:300]
                self._graph = None
                return super().__call__(inputs, labels)
        else:
            self._static['feats'].copy_(feats)
            self._static['labels'].copy_(labels)
        self._graph.replay()
        self.optimizer._packed = True                           # the replay gathered the gradients
        N.bump_weights_epoch()                                  # the replayed forward rewrote the BatchNorm running statistics
        if not self.skip_allreduce:
            allreduce_mean_(self.optimizer.grad, bucket_bytes=self.bucket_bytes)
        self.optimizer.step()
        self.optimizer.clear_grad()
        if self.scheduler is not None:
            self.scheduler.step()
        if self.margin_scheduler is not None:
            self.margin_scheduler.step()
        self.step_id += 1
        return self._static['loss'], self._static['acc']
