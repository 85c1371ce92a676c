"""PPVectorTrainer on the MI355X engine -- the caller of the hot path (ppvector/trainer.py:33-480), same constructor,
``train`` / ``evaluate`` / ``extract_features`` signatures, return values, checkpoint layout and ``stop_train`` /
``stop_eval`` flags.

What moved: the reference builds features one utterance at a time on CPU DataLoader workers (reader.py:72-109) and feeds
(B, T, F) tensors to the device; here worker THREADS only decode audio, and one step is

    decoded waves --H2D--> speed_perturb -> assemble_waves (dB normalise, crop, pad: 1 launch) -> AudioFeaturizer (Fbank / Mel + CMN)
      -> SpecAugmentor.batch -> nn.Sequential(backbone, classifier) train-mode forward -> criterion -> backward
      -> bucketed gradient all-reduce over RCCL (one process per GPU, torch.distributed) -> flat Adam step

(`__train_epoch`, trainer.py:202-274).  Evaluation embeds the enrol / trial lists in eval mode and scores all trials with one
cosine GEMM + the reference's EER / minDCF arithmetic (trainer.py:367-447 -> metric/metrics.py).  VisualDL logging, the
parameter summary table and ``export`` (paddle.jit static graphs) have no counterpart: scalars go to the ``ppvector``
logger, ``export`` raises.  AMP (``enable_amp``, trainer.py:209-229) selects the bf16 matrix cores with f32 master weights and statistics.
"""
import logging
import os
import random
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import timedelta

import numpy as np
import torch
import torch.distributed as dist
import yaml

import ppvector
from torch import nn

from ppvector.data_utils.featurizer import AudioFeaturizer
from ppvector.data_utils.reader import PPVectorDataset
from ppvector.data_utils.wave_batch import assemble_waves, speed_perturb
from ppvector.loss import build_loss
from ppvector.metric.metrics import evaluate_trials
from ppvector.models import build_model
from ppvector.models.fc import SpeakerIdentification
from ppvector.optimizer import MarginScheduler, build_lr_scheduler, build_optimizer
from ppvector.train.step import GraphedTrainStep
from ppvector.utils.checkpoint import load_checkpoint, load_pretrained, save_checkpoint
from ppvector.utils.utils import dict_to_object

_LOG = logging.getLogger('ppvector')


class _BatchLoader:
    """Batches of decoded utterances: index batches like paddle.io.BatchSampler / DistributedBatchSampler (shuffle per
    epoch, drop_last, contiguous per-rank shards of the shuffled order), items decoded by a thread pool one batch ahead."""

    def __init__(self, dataset, batch_size=64, shuffle=False, drop_last=False, num_workers=0, rank=0, world=1, seed=1000):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(num_workers or 1)))

    def _indices(self):
        idx = list(range(len(self.dataset)))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(idx)
        if self.world > 1:
            per = (len(idx) + self.world - 1) // self.world
            idx = (idx + idx[:per * self.world - len(idx)])[self.rank * per:(self.rank + 1) * per]
        return idx

    def __len__(self):
        n = len(self.dataset)
        if self.world > 1:
            n = (n + self.world - 1) // self.world                # every rank's shard has the same (wrap-padded) size
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        idx = self._indices()
        self.epoch += 1
        batches = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        pending = None
        for b in batches + [None]:
            nxt = [self.pool.submit(self.dataset.__getitem__, i) for i in b] if b is not None else None
            if pending is not None:
                yield [f.result() for f in pending]
            pending = nxt


def outputs_classes(model, K=1):
    """Number of classes the labels may name: logit columns of nn.Sequential(backbone, SpeakerIdentification) / K sub-centres."""
    head = model[1]
    w = head.weight if hasattr(head, 'weight') else head.output.weight          # Paddle layout: [in, out]
    return int(w.shape[1]) // max(1, int(K))


class PPVectorTrainer(object):
    def __init__(self, configs, use_gpu=True, data_augment_configs=None):
        if not use_gpu:
            raise RuntimeError('the MI355X engine has no CPU path (use_gpu=False is not available)')
        assert torch.cuda.is_available(), 'GPU不可用'
        self.use_gpu = use_gpu
        self.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device('cuda', self.local_rank)
        if isinstance(configs, str):
            with open(configs, 'r', encoding='utf-8') as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.configs = dict_to_object(configs)
        if isinstance(data_augment_configs, str):
            with open(data_augment_configs, 'r', encoding='utf-8') as f:
                data_augment_configs = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.data_augment_configs = dict_to_object(data_augment_configs)
        self.model = self.backbone = self.optimizer = self.scheduler = self.audio_featurizer = None
        self.train_dataset = self.train_loader = None
        self.enroll_dataset = self.enroll_loader = self.trials_dataset = self.trials_loader = None
        self.margin_scheduler = self.amp_scaler = self.loss = self.train_step_fn = None
        self.max_step, self.train_step = None, None
        self.train_loss, self.train_acc = None, None
        self.train_eta_sec = None
        self.eval_eer, self.eval_min_dcf, self.eval_threshold = None, None, None
        self.test_log_step, self.train_log_step = 0, 0
        self.stop_train, self.stop_eval = False, False

    # ------------------------------------------------------------------------------------------------ data
    def __setup_dataloader(self, is_train=False):
        conf = self.configs.dataset_conf
        self.audio_featurizer = AudioFeaturizer(feature_method=self.configs.preprocess_conf.feature_method,
                                                method_args=self.configs.preprocess_conf.get('method_args', {}))
        dataset_args = dict(conf.get('dataset', {}))
        sampler_args = dict(conf.get('sampler', {}))
        workers = int(dict(conf.get('dataLoader', {})).get('num_workers', 0))
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        if is_train:
            if conf.get('is_use_pksampler', False):
                raise NotImplementedError('PKSampler is not built (it only serves TripletAngularMarginLoss, which is not built either)')
            self.train_dataset = PPVectorDataset(data_list_path=conf.train_list, audio_featurizer=self.audio_featurizer,
                                                 aug_conf=self.data_augment_configs,
                                                 num_speakers=self.configs.model_conf.classifier.num_speakers, mode='train',
                                                 **dataset_args)
            self.train_loader = _BatchLoader(self.train_dataset, num_workers=workers, rank=rank, world=world, **sampler_args)
        dataset_args['max_duration'] = conf.eval_conf.max_duration
        self.enroll_dataset = PPVectorDataset(data_list_path=conf.enroll_list, audio_featurizer=self.audio_featurizer, mode='eval',
                                              **dataset_args)
        self.enroll_loader = _BatchLoader(self.enroll_dataset, batch_size=conf.eval_conf.batch_size, num_workers=workers)
        self.trials_dataset = PPVectorDataset(data_list_path=conf.trials_list, audio_featurizer=self.audio_featurizer, mode='eval',
                                              **dataset_args)
        self.trials_loader = _BatchLoader(self.trials_dataset, batch_size=conf.eval_conf.batch_size, num_workers=workers)

    def _features(self, items, dataset):
        """One decoded batch -> (features (B, T, F) f32 on the GPU, labels int64 on the GPU)."""
        labels = torch.tensor([int(it['label']) for it in items], dtype=torch.int64).to(self.device)
        self._last_frames = None
        if 'feature' in items[0]:                                           # pre-extracted .npy features: the collate_fn path
            from ppvector.data_utils.collate_fn import collate_fn
            feats, _, _ = collate_fn([(torch.from_numpy(it['feature']).to(self.device), it['label']) for it in items])
            return feats, labels
        waves = self._upload([it['samples'] for it in items])
        waves = speed_perturb(waves, [it.get('speed', 1.0) for it in items])
        longest = max(min(int(w.numel()) - int(it['start']), dataset.max_samples) if dataset.mode != 'extract_feature'
                      else int(w.numel()) for w, it in zip(waves, items))
        batch, _, n_valid = assemble_waves(waves, max_len=longest, starts=[it['start'] for it in items],
                                           use_dB_normalization=dataset._use_dB_normalization, target_dB=dataset._target_dB,
                                           gains_dB=[it['gain_dB'] for it in items], with_valid=True)
        with torch.no_grad():                                   # per-utterance featurisation + zero-padded collate, batched
            if int(n_valid.min()) == batch.shape[1]:
                feats = self.audio_featurizer(batch)
            else:
                feats, nf = self.audio_featurizer.forward_ragged(batch, n_valid)
                self._last_frames = nf.tolist()                 # frames of each utterance before the zero padding
        return feats, labels

    def _upload(self, arrays):
        """The batch's decoded utterances -> ONE pinned host buffer -> ONE host-to-device copy; returns per-utterance views.
        (B separate pageable copies cost more than a TDNN training step at B = 64.)"""
        lens = [int(a.shape[0]) for a in arrays]
        offs = np.concatenate(([0], np.cumsum([(n + 3) // 4 * 4 for n in lens]))).astype(np.int64)     # 16-byte aligned starts
        total = int(offs[-1])
        if getattr(self, '_pinned', None) is None or self._pinned.numel() < total:
            self._pinned = torch.empty(max(total, 1), dtype=torch.float32).pin_memory()
        host = self._pinned[:total]
        hv = host.numpy()
        for a, o, n in zip(arrays, offs[:-1], lens):
            hv[o:o + n] = a
        dev = host.to(self.device, non_blocking=True)
        torch.cuda.current_stream().synchronize()                 # the pinned buffer is reused by the next batch
        return [dev[o:o + n] for o, n in zip(offs[:-1].tolist(), lens)]

    def extract_features(self, save_dir='dataset/features', max_duration=100):
        """trainer.py:134-160: dump every list's features to .npy and write '<list>_features.txt'."""
        self.audio_featurizer = AudioFeaturizer(feature_method=self.configs.preprocess_conf.feature_method,
                                                method_args=self.configs.preprocess_conf.get('method_args', {}))
        conf = self.configs.dataset_conf
        for data_list in [conf.train_list, conf.enroll_list, conf.trials_list]:
            dataset_args = dict(conf.get('dataset', {}))
            dataset_args['max_duration'] = max_duration
            ds = PPVectorDataset(data_list_path=data_list, audio_featurizer=self.audio_featurizer, mode='extract_feature', **dataset_args)
            save_data_list = data_list.replace('.txt', '_features.txt')
            with open(save_data_list, 'w', encoding='utf-8') as f:
                for i in range(len(ds)):
                    item = ds[i]
                    feats, _ = self._features([item], ds)
                    label = int(item['label'])
                    save_path = os.path.join(save_dir, str(label), f'{int(time.time() * 1000)}_{i}.npy').replace('\\', '/')
                    os.makedirs(os.path.dirname(save_path), exist_ok=True)
                    np.save(save_path, feats[0].cpu().numpy())
                    f.write(f'{save_path}\t{label}\n')
            _LOG.info('%s列表中的数据已提取特征完成，新列表为：%s', data_list, save_data_list)

    # ------------------------------------------------------------------------------------------------ model
    def __setup_model(self, input_size, is_train=False):
        self.backbone = build_model(input_size=input_size, configs=self.configs)
        if is_train:
            # enable_amp (trainer.py:209-229: auto_cast O1 + GradScaler): conv GEMMs on the bf16 matrix cores over f32 tensors;
            # bf16 keeps f32's exponent range, so there is no loss scale to maintain (amp_scaler stays None)
            ppvector.set_train_amp(bool(self.configs.train_conf.get('enable_amp', False)))
            # an extension key (absent from the reference's YAMLs = off): the f32 step with its conv GEMMs in split precision
            # (ppvector.set_train_x3); ignored under enable_amp
            ppvector.set_train_x3(bool(self.configs.train_conf.get('split_precision', False)))
            num_class = self.configs.model_conf.classifier.num_speakers
            spd = self.data_augment_configs.get('speed') if self.data_augment_configs is not None else None
            if spd is not None and spd.get('prob', 0.0) > 0 and spd.get('speed_perturb_3_class', False):
                self.configs.model_conf.classifier.num_speakers = num_class * 3       # trainer.py:171-173
            classifier = SpeakerIdentification(input_dim=self.backbone.embd_dim, **dict(self.configs.model_conf.classifier))
            self.model = nn.Sequential(self.backbone, classifier).to(self.device)
            self.loss = build_loss(configs=self.configs)
            if isinstance(self.loss, nn.Module):
                self.loss.to(self.device)
            if self.configs.loss_conf.get('use_margin_scheduler', False):
                args = dict(increase_start_epoch=int(self.configs.train_conf.max_epoch * 0.3),
                            fix_epoch=int(self.configs.train_conf.max_epoch * 0.7), initial_margin=0.0, final_margin=0.3)
                args.update(self.configs.loss_conf.get('margin_scheduler_args', {}))
                self.margin_scheduler = MarginScheduler(criterion=self.loss, step_per_epoch=len(self.train_loader), **args)
            self.scheduler = build_lr_scheduler(step_per_epoch=len(self.train_loader), configs=self.configs)
            self.optimizer = build_optimizer(parameters=self.model.parameters(), learning_rate=self.scheduler, configs=self.configs)
        else:
            self.model = nn.Sequential(self.backbone).to(self.device)

    # ------------------------------------------------------------------------------------------------ training
    def __train_epoch(self, epoch_id, save_model_path, local_rank):
        train_times, accuracies, loss_sum = [], [], []
        start = time.time()
        K = int(self.configs.loss_conf.get('loss_args', {}).get('K', 1)) if self.configs.loss_conf.get('loss') == 'SubCenterLoss' else 1
        spec = self.train_dataset.spec_augment
        for batch_id, items in enumerate(self.train_loader):
            if self.stop_train:
                break
            features, label = self._features(items, self.train_dataset)
            if spec is not None:                                # the reference masks each utterance's OWN feature before the
                features = spec.batch(features, lengths=self._last_frames)      # collate (reader.py:105-107): T = its frames
            if self._label_checks < 3:                          # a wrong num_speakers / speed-perturb label offset reads outside
                self._label_checks += 1                         # the logits row in the loss kernels; paddle raises here too.
                lo, hi = int(label.min()), int(label.max())     # First batches of every run (also after a resume).
                if lo < 0 or hi >= outputs_classes(self.model, K):
                    raise ValueError(f'label range [{lo}, {hi}] outside the classifier\'s {outputs_classes(self.model, K)} classes')
            # forward -> loss -> backward -> data-parallel gradient all-reduce -> optimiser (trainer.py:206-231):
            # ONE implementation, ppvector/train/step.py -- the step bench.py times
            los, acc = self.train_step_fn(features, label)
            if local_rank == 0:                                 # device scalars, read back at the log interval only (the reference
                accuracies.append(acc)                          # syncs twice per step, trainer.py:237-238); other ranks never log
                loss_sum.append(los)
            train_times.append((time.time() - start) * 1000)
            self.train_step += 1
            if batch_id % self.configs.train_conf.log_interval == 0:
                if local_rank == 0:
                    per = sum(train_times) / len(train_times)
                    world = dist.get_world_size() if dist.is_initialized() else 1
                    train_speed = len(items) * world / (per / 1000)          # GLOBAL utterances per second
                    self.train_eta_sec = per * (self.max_step - self.train_step) / 1000
                    self.train_loss = float(torch.stack(loss_sum).mean())
                    self.train_acc = float(torch.stack(accuracies).mean())
                    margin_str = f'margin: {self.margin_scheduler.get_margin()}' if self.margin_scheduler else ''
                    _LOG.info('Train epoch: [%d/%d], batch: [%d/%d], loss: %.5f, accuracy: %.5f, learning rate: %.8f, %s speed: %.2f data/sec, '
                              'eta: %s', epoch_id, self.configs.train_conf.max_epoch, batch_id, len(self.train_loader), self.train_loss,
                              self.train_acc, self.scheduler.get_lr(), margin_str, train_speed, timedelta(seconds=int(self.train_eta_sec)))
                    self.train_log_step += 1
                train_times, accuracies, loss_sum = [], [], []
            if batch_id % 10000 == 0 and batch_id != 0:
                self.train_step_fn.check_faults()          # never checkpoint across an unnoticed grid-barrier bail-out
            if batch_id % 10000 == 0 and batch_id != 0 and local_rank == 0:
                save_checkpoint(configs=self.configs, model=self.model, optimizer=self.optimizer, amp_scaler=self.amp_scaler,
                                margin_scheduler=self.margin_scheduler, save_model_path=save_model_path, epoch_id=epoch_id)
            start = time.time()
            self.scheduler.step()
            if self.margin_scheduler:
                self.margin_scheduler.step()
        # the fused Res2Net training kernels meet at an in-kernel grid barrier that gives up instead of hanging the device when its
        # workgroups are not co-resident (two training processes on one GPU).  The device drops the update of such a step itself
        # (csrc/train_ops.hip: adam_kernel's fault word); GraphedTrainStep polls the word every 25 steps, here before the epoch's
        # evaluation / checkpoint, and continues on the per-chunk kernels after a warning.
        self.train_step_fn.check_faults()

    def train(self, save_model_path='models/', log_dir='log/', resume_model=None, pretrained_model=None, do_eval=True):
        torch.manual_seed(1000)
        world = int(os.environ.get('WORLD_SIZE', 1))
        # weights: the same seed on every rank (replicas start identical); crop / speed / SpecAugment draws: a seed per rank,
        # or every rank would augment its shard with the same sequence
        random.seed(1000 + int(os.environ.get('RANK', 0)))
        np.random.seed(1000 + int(os.environ.get('RANK', 0)))
        if world > 1 and not dist.is_initialized():                       # one process per GPU over RCCL (torchrun env)
            dist.init_process_group('nccl', device_id=self.device)
        local_rank = dist.get_rank() if dist.is_initialized() else 0
        self.__setup_dataloader(is_train=True)
        self.__setup_model(input_size=self.audio_featurizer.feature_dim, is_train=True)
        self.model = load_pretrained(model=self.model, pretrained_model=pretrained_model)
        self.model, self.optimizer, self.amp_scaler, self.scheduler, self.margin_scheduler, last_epoch, best_eer = \
            load_checkpoint(configs=self.configs, model=self.model, optimizer=self.optimizer, amp_scaler=self.amp_scaler,
                            scheduler=self.scheduler, margin_scheduler=self.margin_scheduler, step_epoch=len(self.train_loader),
                            save_model_path=save_model_path, resume_model=resume_model)
        # the optimisation step: HIP-graph replays of forward + backward in stages, each stage's gradient all-reduce (RCCL) under
        # the next stage's replay, one optimiser launch; shapes seen fewer than four times run the same step eagerly
        # (the LR / margin schedulers stay in the epoch loop below, stepped after the log line as in the reference)
        self.train_step_fn = GraphedTrainStep(self.model, self.loss, self.optimizer)
        self._label_checks = 0
        _LOG.info('训练数据：%d', len(self.train_dataset))
        self.train_loss, self.train_acc = None, None
        self.test_log_step, self.train_log_step = 0, 0
        self.eval_eer, self.eval_min_dcf, self.eval_threshold = None, None, None
        self.max_step = len(self.train_loader) * self.configs.train_conf.max_epoch
        self.train_step = max(last_epoch, 0) * len(self.train_loader)
        self.train_loader.epoch = max(last_epoch, 0)
        self.model.train()
        for epoch_id in range(last_epoch, self.configs.train_conf.max_epoch):
            if self.stop_train:
                break
            epoch_id += 1
            start_epoch = time.time()
            self.__train_epoch(epoch_id=epoch_id, save_model_path=save_model_path, local_rank=local_rank)
            if local_rank == 0 and do_eval and not self.stop_eval:
                # (the reference `continue`s on stop_eval, trainer.py:343; here that would skip the barrier the other ranks wait in)
                self.eval_eer, self.eval_min_dcf, self.eval_threshold = self.evaluate()
                _LOG.info('Test epoch: %d, time/epoch: %s, threshold: %.2f, EER: %.5f, MinDCF: %.5f', epoch_id,
                          timedelta(seconds=(time.time() - start_epoch)), self.eval_threshold, self.eval_eer, self.eval_min_dcf)
                self.test_log_step += 1
                self.model.train()
                if self.eval_eer <= best_eer:
                    best_eer = self.eval_eer
                    save_checkpoint(configs=self.configs, model=self.model, optimizer=self.optimizer, amp_scaler=self.amp_scaler,
                                    margin_scheduler=self.margin_scheduler, save_model_path=save_model_path, epoch_id=epoch_id,
                                    eer=self.eval_eer, min_dcf=self.eval_min_dcf, threshold=self.eval_threshold, best_model=True)
            if local_rank == 0:
                save_checkpoint(configs=self.configs, model=self.model, optimizer=self.optimizer, amp_scaler=self.amp_scaler,
                                margin_scheduler=self.margin_scheduler, save_model_path=save_model_path, epoch_id=epoch_id,
                                eer=self.eval_eer, min_dcf=self.eval_min_dcf, threshold=self.eval_threshold)
            if dist.is_initialized():
                dist.barrier()

    # ------------------------------------------------------------------------------------------------ evaluation
    def _embed(self, loader, dataset, eval_model):
        feats, labels = [], []
        with torch.no_grad():
            for items in loader:
                if self.stop_eval:
                    break
                x, y = self._features(items, dataset)
                feats.append(eval_model(x).float())
                labels.append(y)
        return (torch.cat(feats), torch.cat(labels)) if feats else (None, None)

    def evaluate(self, resume_model=None, save_image_path=None):
        """-> (eer, min_dcf, threshold) floats, or (-1, -1, -1) when stop_eval was raised (trainer.py:367-447)."""
        if self.enroll_loader is None or self.trials_loader is None:
            self.__setup_dataloader()
        if self.model is None:
            self.__setup_model(input_size=self.audio_featurizer.feature_dim)
        if resume_model is not None:
            if os.path.isdir(resume_model):
                resume_model = os.path.join(resume_model, 'model.pdparams')
            assert os.path.exists(resume_model), f"{resume_model} 模型不存在！"
            self.model = load_pretrained(self.model, resume_model)
        self.model.eval()
        eval_model = self.model if len(self.model) == 1 else self.model[0]
        enroll_features, enroll_labels = self._embed(self.enroll_loader, self.enroll_dataset, eval_model)
        trials_features, trials_labels = self._embed(self.trials_loader, self.trials_dataset, eval_model)
        self.model.train()
        if self.stop_eval:
            return -1, -1, -1
        eer, min_dcf, threshold = evaluate_trials(enroll_features, enroll_labels.cpu().numpy(), trials_features,
                                                  trials_labels.cpu().numpy())
        if save_image_path:
            _LOG.warning('save_image_path: the fnr/fpr plot (matplotlib) is not produced by this build')
        return float(eer), float(min_dcf), float(threshold)

    def export(self, save_model_path='models/', resume_model='models/CAMPPlus_Fbank/best_model/'):
        raise NotImplementedError('export writes a paddle.jit static graph (trainer.py:449-480); the MI355X engine loads '
                                  'model.pdparams directly (PPVectorPredictor(model_path=<checkpoint dir>))')
