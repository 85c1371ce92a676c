"""One optimisation step, laid out like the body of PPVectorTrainer.__train_epoch (ppvector/trainer.py:206-274):

    outputs = model(features); los = loss(outputs, label); los.backward(); optimizer.step(); optimizer.clear_grad();
    scheduler.step(); margin_scheduler.step()

with the data-parallel gradient average (fleet.distributed_model in the reference, trainer.py:318-320) made explicit:
a sum-all-reduce over the optimiser's flat gradient buffer between backward and the optimiser kernel, whose 1 / world scale
rides on the optimiser's `grad_scale` launch scalar (no pass over the buffer for the division).

`TrainStep` is the eager step (autograd hooks launch each bucket's all-reduce while backward runs).  `GraphedTrainStep` is what
PPVectorTrainer and bench.py run: forward + backward replayed from captured HIP graphs, cut into stages
(ppvector/train/segments.py) so that stage k's gradients are all-reduced while stage k + 1 replays."""
import torch
import torch.distributed as dist

from ppvector.train.ddp import OverlappedReducer, all_reduce_sum_, world_size
from ppvector.train.segments import Recorder

MIN_CHUNK = 1 << 20            # 4 MB of f32 gradients: below this a ring all-reduce over xGMI is latency-bound
MAX_CHUNKS = 16


def reduce_chunks(n, chunk=None):
    """THE collective schedule of a data-parallel step: the flat gradient buffer [0, n) cut into equal chunks (4 MB, or n / 16 rounded
    up to whole 4 MB units for the large models), all-reduced from the LAST chunk to the first (backward produces the late layers'
    gradients first).  It depends on the parameter count alone -- never on the batch shape, on whether this rank replays graphs or
    runs eagerly, or on where the backward stages were cut -- so every rank issues the same collectives in the same order whatever
    its local state (a rank whose padded length differs, or whose capture failed, must not desynchronise the job: ADVICE r03).
    ECAPA-TDNN (6.7 M parameters): seven chunks, the one that waits for the last backward stage is 15 % of the buffer."""
    if chunk is None:
        per = (n + MAX_CHUNKS - 1) // MAX_CHUNKS
        chunk = max(MIN_CHUNK, (per + MIN_CHUNK - 1) // MIN_CHUNK * MIN_CHUNK)
    k = max(1, (n + chunk - 1) // chunk)
    return [(i * chunk, min(n, (i + 1) * chunk)) for i in reversed(range(k))]


def batch_accuracy(outputs, labels, K=1):
    """trainer.py:233-236: argmax of the logits against the labels; SubCenter heads score a class by its best sub-centre."""
    pred = getattr(outputs, 'pred', None)
    if pred is not None and K == 1:                   # the class-tiled head already holds the argmax of the cosines
        return (pred.to(labels.device) == labels.to(torch.int32)).float().mean()
    logits = outputs['logits'].detach()
    if K > 1:
        logits = logits.reshape(logits.shape[0], -1, K).max(dim=2)[0]
    return (logits.argmax(dim=1) == labels.to(logits.device)).float().mean()


class TrainStep:
    def __init__(self, model, criterion, optimizer, scheduler=None, margin_scheduler=None, featurizer=None, spec_augment=None,
                 overlap_allreduce=True):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.scheduler, self.margin_scheduler = scheduler, margin_scheduler
        self.featurizer, self.spec_augment = featurizer, spec_augment
        self.step_id = 0
        self.K = int(getattr(criterion, 'K', 1) or 1)
        # data-parallel gradient average: bucketed all-reduce launched from autograd hooks while backward is still running
        self.reducer = OverlappedReducer(optimizer) if overlap_allreduce else None
        self.skip_allreduce = False            # measurement switch (bench.py: the same step without the collective)

    def _features(self, inputs):
        feats = inputs
        if self.featurizer is not None:
            with torch.no_grad():
                feats = self.featurizer(inputs)
                if self.spec_augment is not None:
                    feats = self.spec_augment.batch(feats)
        return feats

    def _after(self):
        if self.scheduler is not None:
            self.scheduler.step()
        if self.margin_scheduler is not None:
            self.margin_scheduler.step()
        self.step_id += 1

    def _eager(self, feats, labels):
        outputs = self.model(feats)
        loss = self.criterion(outputs, labels)
        loss.backward()
        world = 1 if self.skip_allreduce else world_size()
        if self.reducer is not None:
            self.reducer.finish()
        else:
            self.optimizer.pack_grads()
            if world > 1:                                 # the same chunks, in the same order, as the graphed step of a peer
                works = [all_reduce_sum_(self.optimizer.grad[lo:hi], async_op=True) for lo, hi in reduce_chunks(self.optimizer.grad.numel())]
                for w in works:
                    w.wait()
        self.optimizer.step(grad_scale=1.0 / world)
        self.optimizer.clear_grad()
        with torch.no_grad():
            acc = batch_accuracy(outputs, labels, self.K)
        self._after()
        return loss.detach(), acc

    def __call__(self, inputs, labels):
        """inputs: waveforms (B, L) when a featurizer was given, else features (B, T, F).  Returns (loss, accuracy) tensors."""
        self.model.train()
        return self._eager(self._features(inputs), labels)


class GraphedTrainStep(TrainStep):
    """The same step with forward + backward replayed from captured HIP graphs, one per backward stage.

    The step is several hundred kernel launches whose count does not depend on the batch; at the 32 utterances per GPU of the
    strong-scaled configuration (global batch 256 over 8 GPUs) the eager step is bound by the host issuing them.  A batch
    shape's first `warm` sightings run the eager step (identical semantics; every lazy table / kernel attribute gets set up
    outside a capture); then the features are copied into a static buffer and

        graph 0: model forward -> criterion -> accuracy -> backward of the last stage -> its gradients into the flat buffer
        graph k: backward of the k-th stage from the end -> its gradients into the flat buffer

    are captured once (stages = the backbone's `cut` points, ppvector/train/segments.py; a model without cuts is one graph).
    Per step: replay graph 0, START the all-reduce of its slice of the flat gradient buffer (async, the collective library's
    stream), replay graph 1 underneath it, ... wait for the collectives, one optimiser launch (1 / world folded into it).
    Outside the graphs, eager, every step: the featurizer (+ SpecAugment, whose masks the host draws), the collectives, the
    optimiser (the learning rate is a launch scalar) and the schedulers.  The loss margin is DEVICE data while captured
    (vp_set_margin_table): MarginScheduler's ramp moves it without a re-capture.  BatchNorm running statistics are updated by the
    replayed kernels in place, as in the eager step.  Up to `max_graphs` batch shapes keep their graphs (a ragged training set
    yields a few distinct padded lengths); others run eagerly."""

    def __init__(self, *a, warm=3, max_graphs=6, fault_every=25, **kw):
        kw['overlap_allreduce'] = False          # an autograd hook cannot launch a collective from inside a capture
        super().__init__(*a, **kw)
        self.warm, self.max_graphs = warm, max_graphs
        self._plans, self._seen = {}, {}
        self.capture_error = None
        self._margin = None
        self.fault_every = fault_every           # steps between polls of the grid-barrier bail-out word (a host sync each)
        self.faults = 0

    def check_faults(self):
        """Did a grid barrier of the fused training kernels give up since the last poll?  The device has already protected the
        weights (the optimiser kernels and the running-statistics update test the same word); here the host finds out, switches the
        fused kernels off for the rest of the process, re-arms the barrier words and drops the captured graphs (they replay the
        fused kernels).  Called every `fault_every` steps, and by the trainer before every checkpoint / at the end of an epoch.
        Returns True when a fault was found (the steps since the last poll did not update the model)."""
        from ppvector import _native as N
        import ppvector
        if not torch.cuda.is_available():
            return False
        ctx = N.ctx(torch.device('cuda', torch.cuda.current_device()))
        if N.lib().vp_grid_barrier_status(ctx) <= 0:
            return False
        self.faults += 1
        ppvector.set_fused_grid_kernels(False)
        N.check(N.lib().vp_grid_barrier_reset(ctx, N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        self._plans.clear()
        self._seen.clear()
        import warnings
        warnings.warn('a grid barrier of the fused Res2Net training kernels timed out (is another process using this GPU?): the '
                      'optimiser dropped the affected steps on the device; continuing on the per-chunk kernels', RuntimeWarning)
        return True

    # ------------------------------------------------------------------------------------------------ capture
    def _spans(self, params):
        """Contiguous [begin, end) element ranges of the flat gradient buffer covered by `params`."""
        opt = self.optimizer
        iv = sorted((opt._offset(p), opt._offset(p) + p.numel()) for p in params)
        out = []
        for lo, hi in iv:
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return [tuple(v) for v in out]

    def _capture(self, feats, labels):
        from ppvector.loss._margin import MarginTable
        opt = self.optimizer
        st = {'feats': feats.clone(), 'labels': labels.clone()}
        if self._margin is None:
            self._margin = MarginTable(self.criterion, feats.device)
        opt.clear_grad()
        torch.cuda.synchronize()
        graphs, spans, done = [], [], set()
        packed = {}
        cur = {}

        def open_graph():
            g = torch.cuda.CUDAGraph()
            # thread-local error mode: another thread of the process (the collective library's watchdog polling its events) must
            # not invalidate the capture; later graphs read tensors the earlier ones saved: one memory pool
            cm = torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None, capture_error_mode='thread_local')
            cm.__enter__()
            cur['g'], cur['cm'] = g, cm

        def close_graph(last):
            new = [p for p in opt.params if id(p) not in done and (p.grad is not None or last)]
            done.update(id(p) for p in new)
            opt.pack_range(new)                           # the stage's gradients -> flat buffer: part of the replayed sequence
            cm = cur.pop('cm')
            cm.__exit__(None, None, None)
            graphs.append(cur.pop('g'))
            spans.append(self._spans(new))
            packed.update((id(p), p.grad._version) for p in new if p.grad is not None)

        rec = Recorder()
        try:
            with rec, self._margin:
                open_graph()
                outputs = self.model(st['feats'])
                loss = self.criterion(outputs, st['labels'])
                st['loss'] = loss.detach()
                st['acc'] = batch_accuracy(outputs, st['labels'], self.K)

                def between(i):
                    close_graph(last=(i == rec.n_stages - 1))
                    if i < rec.n_stages - 1:
                        open_graph()

                rec.backward(loss, between)
        except BaseException as e:
            if 'cm' in cur:                               # leave the open capture before anything else touches the stream
                try:
                    cur['cm'].__exit__(type(e), e, e.__traceback__)
                except Exception:                         # noqa: BLE001
                    pass
            raise
        # a parameter whose gradient was accumulated into AFTER its stage packed it (tied weights, a parameter used on both sides of a
        # cut) would lose the later contributions under replay: refuse the capture (the eager step has no such limit)
        late = [p for p in opt.params if p.grad is not None and id(p) in packed and p.grad._version != packed[id(p)]]
        if late:
            raise RuntimeError(f'{len(late)} parameter(s) receive gradient in more than one backward stage: not capturable')
        opt.clear_grad()                                  # a capture executes nothing: the .grad tensors hold no data
        # chunk c of the fixed collective schedule may start once every element of it has been packed: after stage ready[c]
        chunks = reduce_chunks(opt.grad.numel())
        ready = []
        for lo, hi in chunks:
            r = 0
            for k, sp in enumerate(spans):
                if any(a < hi and b > lo for a, b in sp):
                    r = k
            ready.append(r)
        # launch order = the schedule's order: a chunk also waits for the chunks ahead of it (identical sequence on every rank)
        for c in range(1, len(ready)):
            ready[c] = max(ready[c], ready[c - 1])
        return {'graphs': graphs, 'spans': spans, 'static': st, 'chunks': chunks, 'ready': ready}

    # ------------------------------------------------------------------------------------------------ step
    def __call__(self, inputs, labels):
        from ppvector import _native as N
        self.model.train()
        feats = self._features(inputs)
        labels = labels.to(feats.device)
        key = (tuple(feats.shape), feats.dtype, tuple(labels.shape))
        plan = self._plans.get(key)
        if plan is None:
            seen = self._seen[key] = self._seen.get(key, 0) + 1
            if self.capture_error is not None or seen <= self.warm or len(self._plans) >= self.max_graphs:
                return self._eager(feats, labels)
            try:
                plan = self._plans[key] = self._capture(feats, labels)
            except Exception as e:                        # noqa: BLE001 -- fall back to the eager step for good
                self.capture_error = f'{type(e).__name__}: {e}'[:300]
                # the aborted capture left .grad tensors that point into its private pool and were never written: drop them
                self.optimizer.clear_grad()
                torch.cuda.synchronize()
                return self._eager(feats, labels)
        else:
            plan['static']['feats'].copy_(feats)
            plan['static']['labels'].copy_(labels)
        self._margin.sync()                               # the margin the criterion holds NOW (MarginScheduler stepped it)
        world = 1 if self.skip_allreduce else world_size()
        works = []
        nxt = 0
        for k, g in enumerate(plan['graphs']):
            g.replay()
            while world > 1 and nxt < len(plan['chunks']) and plan['ready'][nxt] <= k:
                lo, hi = plan['chunks'][nxt]              # complete after this stage: travels while the next stage replays
                works.append(all_reduce_sum_(self.optimizer.grad[lo:hi], async_op=True))
                nxt += 1
        for w in works:
            if w is not None:
                w.wait()
        self.optimizer._packed = True                     # the replays gathered every gradient
        N.bump_weights_epoch()                            # the replayed forward rewrote the BatchNorm running statistics
        self.optimizer.step(grad_scale=1.0 / world)
        self.optimizer.clear_grad()
        self._after()
        out = plan['static']['loss'].clone(), plan['static']['acc'].clone()
        if self.fault_every and self.step_id % self.fault_every == 0:
            self.check_faults()
        return out

    @property
    def n_stages(self):
        """Backward stages of the captured step(s) (1 = no cut points: the all-reduce follows the whole backward)."""
        return max((len(p['graphs']) for p in self._plans.values()), default=0)
