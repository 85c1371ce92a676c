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
