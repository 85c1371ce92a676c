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

from ppvector.train.ddp import OverlappedReducer, all_reduce_max_, all_reduce_sum_, world_size
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


class _FaultWork:
    """Handle of the step's fault-word collective: wait() = the all-reduce's wait, then the reduced word back into the rank's own
    barrier words (when it has any), so that every rank's optimiser kernels test the same value."""

    def __init__(self, work, buf, word):
        self.work, self.buf, self.word = work, buf, word

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.word is not None:
            self.word.copy_(self.buf.view(self.word.dtype) if self.word.dtype != self.buf.dtype else self.buf)


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
        self.faults = 0                        # grid-barrier bail-outs noticed so far (check_faults)
        self._poll = None                      # (pinned host word, event, side stream) of the asynchronous poll in flight
        self._reserve_set = False

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
        self._poll_faults()                    # every step, eager or replayed: no host stall (the copy issued a step ago is read)

    # ------------------------------------------------------------------------------------------------ grid-barrier bail-out
    # The fused Res2Net training kernels (csrc/res2_train.hip) meet at an in-kernel grid barrier that GIVES UP instead of hanging the
    # device when its workgroups cannot all be resident.  Device side, immediately: the bail-out word makes every writer of persistent
    # state (optimiser kernels, BatchNorm running statistics) a no-op, so the step is dropped.  Data-parallel: the word is
    # MAX-all-reduced with the gradients (`_share_fault`), so every rank drops the SAME steps and the replicas stay identical (the
    # faulted rank's gradient is garbage on every rank; nobody applies it).  Host side, one step later, without a stall: `_poll_faults`
    # reads the word through an asynchronous 4-byte copy on a side stream and `_on_fault` switches the process to the per-chunk kernels.
    def _device(self):
        return self.optimizer.flat.device

    def _fault_word(self):
        from ppvector import _native as N
        dev = self._device()
        if dev.type != 'cuda':
            return None
        w = N.grid_words(dev)
        return None if w is None else w[N.FAULT_WORD:N.FAULT_WORD + 1]

    def _reserve_cus(self, world):
        """A collective runs beside the step when there are peers: the grid-barrier kernels leave it CUs (VPMI_GRID_RESERVE_CUS, default
        64 of 256) so that their own workgroups are co-resident whatever the collective holds."""
        if self._reserve_set or self._device().type != 'cuda':
            return
        import os
        from ppvector import _native as N
        ctx = N.ctx(self._device())
        n = int(os.environ.get('VPMI_GRID_RESERVE_CUS', '64')) if world > 1 else 0
        N.check(N.lib().vp_set_grid_reserve_cus(ctx, max(0, n)), ctx)
        self._reserve_set = True

    def _share_fault(self, world):
        """The bail-out word as every rank will see it: maximum over the ranks, asynchronous like the gradient chunks.
        The collective is issued by EVERY rank on EVERY step whatever its local state (a rank without barrier words -- its
        context was created inside a capture, vp_set_grid_barrier_words failed -- must not issue one collective fewer than its peers:
        the next gradient chunk would pair with their 1-element MAX).  The word travels through a dedicated one-element int32 tensor:
        copied in (0 when this rank has no words), MAX-reduced, copied back behind the collective by `_FaultWork.wait`."""
        if world <= 1:
            return None
        w = self._fault_word()
        dev = self._device()
        buf = getattr(self, '_fault_buf', None)
        if buf is None or buf.device != dev:
            buf = self._fault_buf = torch.zeros(1, dtype=torch.int32, device=dev)
        if w is not None:
            buf.copy_(w.view(torch.int32) if w.dtype != torch.int32 else w)
        else:
            buf.zero_()
        return _FaultWork(all_reduce_max_(buf, async_op=True), buf, w)

    def _poll_faults(self):
        if self._device().type != 'cuda' or torch.cuda.is_current_stream_capturing():
            return False
        found = False
        if self._poll is not None:
            host, ev, side = self._poll
            if ev.query():                                 # issued a step ago: done unless the GPU is several steps behind the host
                found = int(host[0]) != 0
                self._poll = None
            else:
                return False
        if found:
            self._on_fault()
            return True
        w = self._fault_word()
        if w is None:
            return False
        cur = torch.cuda.current_stream(w.device)
        side = getattr(self, '_poll_stream', None)
        if side is None:
            side = self._poll_stream = torch.cuda.Stream(w.device)
            self._poll_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._poll_host.copy_(w, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        self._poll = (self._poll_host, ev, side)
        return False

    def check_faults(self):
        """Synchronous form (checkpoints, the end of an epoch): did a grid barrier give up since the last look?  Returns True when a
        fault was found -- the steps since it happened did not update the model on ANY rank."""
        from ppvector import _native as N
        if self._device().type != 'cuda':
            return False
        self._poll = None
        ctx = N.ctx(self._device())
        if N.lib().vp_grid_barrier_status(ctx) <= 0:
            return False
        self._on_fault()
        return True

    def _on_fault(self):
        import warnings
        import ppvector
        from ppvector import _native as N
        self.faults += 1
        ppvector.set_fused_grid_kernels(False)
        ctx = N.ctx(self._device())
        N.check(N.lib().vp_grid_barrier_reset(ctx, N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        self._drop_captures()
        # the device dropped every update since the fault, running statistics included; what it cannot undo is state written by kernels
        # that ran BEFORE the faulting one inside the same step -- valid statistics of valid activations.  Anything non-finite here
        # would mean the guard has a hole: stop rather than checkpoint it
        bad = [n for n, b in self.model.named_buffers() if b.is_floating_point() and not bool(torch.isfinite(b).all())]
        if bad or not bool(torch.isfinite(self.optimizer.flat).all()):
            raise N.VpmiError(f'grid-barrier bail-out left non-finite state behind ({bad[:3]}): refusing to continue')
        warnings.warn('a grid barrier of the fused Res2Net training kernels timed out (is another process using this GPU?): every rank '
                      'dropped the affected steps on the device; continuing on the per-chunk kernels', RuntimeWarning)

    def _drop_captures(self):
        pass

    def _eager(self, feats, labels):
        outputs = self.model(feats)
        loss = self.criterion(outputs, labels)
        loss.backward()
        world = 1 if self.skip_allreduce else world_size()
        if self.reducer is not None:
            self.reducer.finish()
            fw = self._share_fault(world)
            if fw is not None:
                fw.wait()
        else:
            self.optimizer.pack_grads()
            if world > 1:                                 # the same chunks, in the same order, as the graphed step of a peer
                works = [all_reduce_sum_(self.optimizer.grad[lo:hi], async_op=True) for lo, hi in reduce_chunks(self.optimizer.grad.numel())]
                works.append(self._share_fault(world))    # ... and the bail-out word behind them, as there
                for w in works:
                    if w is not None:
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
        self._reserve_cus(1 if self.skip_allreduce else world_size())
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
        self.fault_every = fault_every           # (kept for callers that pass it: the word is now polled EVERY step, asynchronously)

    def _drop_captures(self):
        from ppvector.train.functions import drop_weight_panels
        self._plans.clear()                    # the captured graphs replay the fused kernels
        self._seen.clear()
        drop_weight_panels()                   # bf16 weight panels of a captured step point into that graph's private pool

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
        self._reserve_cus(1 if self.skip_allreduce else world_size())
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
        works.append(self._share_fault(world))            # the bail-out word behind the last chunk: every rank drops the same steps
        for w in works:
            if w is not None:
                w.wait()
        self.optimizer._packed = True                     # the replays gathered every gradient
        N.bump_weights_epoch()                            # the replayed forward rewrote the BatchNorm running statistics
        self.optimizer.step(grad_scale=1.0 / world)
        self.optimizer.clear_grad()
        self._after()
        return plan['static']['loss'].clone(), plan['static']['acc'].clone()

    @property
    def n_stages(self):
        """Backward stages of the captured step(s) (1 = no cut points: the all-reduce follows the whole backward)."""
        return max((len(p['graphs']) for p in self._plans.values()), default=0)
