"""Dataset of the training / evaluation lists (ppvector/data_utils/reader.py:16-163), re-cut for a GPU front end.

The reference's __getitem__ runs, per utterance on a CPU worker: decode -> resample -> waveform augmentation -> dB
normalisation -> crop -> Fbank -> SpecAugment (reader.py:72-109).  Here the worker threads only decode / resample and draw
the random numbers (crop start, volume gain); everything from the dB normalisation on runs batched on the MI355X
(data_utils/wave_batch.py -> AudioFeaturizer -> SpecAugmentor.batch).  __getitem__ therefore returns the RAW utterance:

    audio list entry  -> dict(samples float32 (n,), speed float, start int, gain_dB float, label int)
    '.npy' list entry -> dict(feature float32 (T, F) cropped to max_feature_len (:78-83), label int)

List format, min_duration skipping (:89-91), the eval-mode length sort (:121-139), train-mode random crop (yeaudio
AudioSegment.crop: a uniform random start, 0 otherwise) follow the reference.  Speed perturbation (SpeedPerturbAugmentor:
rate drawn from {1.0, 0.9, 1.1}, optional 3-class label offset) is drawn here and applied on the GPU; the crop start is drawn on
the PERTURBED length, as the reference crops after it augments.  Noise / reverb perturbation (need external audio
libraries, yeaudio DSP on CPU) are not built: a configured prob > 0 is reported once and skipped, never silently emulated.
"""
import logging
import os
import random

import numpy as np

from ppvector.data_utils.wave_batch import SPEEDS
from ppvector.predict import AudioSegment

_LOG = logging.getLogger('ppvector')


class PPVectorDataset:
    def __init__(self, data_list_path, audio_featurizer, max_duration=3, min_duration=0.5, mode='train', sample_rate=16000,
                 aug_conf=None, num_speakers=None, use_dB_normalization=True, target_dB=-20):
        assert mode in ['train', 'eval', 'extract_feature']
        self.data_list_path, self.mode = data_list_path, mode
        self.max_duration, self.min_duration = max_duration, min_duration
        self._target_sample_rate = sample_rate
        self._use_dB_normalization, self._target_dB = use_dB_normalization, target_dB
        self.num_speakers = num_speakers
        self.audio_featurizer = audio_featurizer
        self.volume_conf = self.spec_augment = self.speed_conf = None
        self.max_samples = int(self.max_duration * self._target_sample_rate)
        self.max_feature_len = self.get_crop_feature_len()
        with open(self.data_list_path, 'r', encoding='utf-8') as f:
            self.lines = [l for l in f.readlines() if l.strip()]
        self.labels = [np.int64(line.strip().split('\t')[1]) for line in self.lines]
        if mode == 'train' and aug_conf is not None:
            self.get_augmentor(aug_conf)
        if self.mode == 'eval':
            self.sort_list()

    def _decode(self, path):
        seg = AudioSegment.from_file(path)
        if seg.sample_rate != self._target_sample_rate:
            seg.resample(self._target_sample_rate)
        return seg

    def __getitem__(self, idx):
        data_path, spk_id = self.lines[idx].strip().split('\t')
        spk_id = int(spk_id)
        if data_path.endswith('.npy'):
            feature = np.load(data_path)
            if feature.shape[0] > self.max_feature_len:
                crop_start = random.randint(0, feature.shape[0] - self.max_feature_len) if self.mode == 'train' else 0
                feature = feature[crop_start:crop_start + self.max_feature_len, :]
            return dict(feature=np.ascontiguousarray(feature, dtype=np.float32), label=spk_id)
        seg = self._decode(data_path)
        if self.mode in ('train', 'extract_feature') and seg.duration < self.min_duration:
            return self.__getitem__(idx + 1 if idx < len(self.lines) - 1 else 0)
        speed, gain = 1.0, 0.0
        if self.mode == 'train' and self.speed_conf is not None and random.random() < self.speed_conf['prob']:
            speed_idx = random.randint(0, 2)
            speed = SPEEDS[speed_idx]
            if self.speed_conf['speed_perturb_3_class']:
                spk_id = spk_id + self.num_speakers * speed_idx
        if self.mode == 'train' and self.volume_conf is not None and random.random() < self.volume_conf['prob']:
            gain = random.uniform(self.volume_conf['min_gain_dBFS'], self.volume_conf['max_gain_dBFS'])
        start = 0
        n = seg.samples.shape[0] if speed == 1.0 else int(seg.samples.shape[0] / speed)      # length after the speed change
        if self.mode == 'train' and n > self.max_samples:
            start = int(random.uniform(0.0, n / float(self._target_sample_rate) - self.max_duration) * self._target_sample_rate)
        return dict(samples=seg.samples, speed=speed, start=start, gain_dB=gain, label=spk_id)

    def __len__(self):
        return len(self.lines)

    def get_crop_feature_len(self):
        """Frames of a max_duration utterance (reader.py:115-119), from the featurizer's framing arithmetic."""
        return int(self.audio_featurizer.num_frames(self.max_samples))

    def sort_list(self):
        lengths = []
        for line in self.lines:
            data_path, _ = line.split('\t')
            if data_path.endswith('.npy'):
                lengths.append(np.load(data_path, mmap_mode='r').shape[0])
            else:
                lengths.append(self._decode(data_path).duration)
        self.lines = [self.lines[i] for i in np.argsort(lengths, kind='stable')]
        self.labels = [np.int64(line.strip().split('\t')[1]) for line in self.lines]

    def get_augmentor(self, aug_conf):
        from ppvector.data_utils.spec_aug import SpecAugmentor
        spd = aug_conf.get('speed')
        if spd is not None and float(spd.get('prob', 0.0)) > 0:
            self.speed_conf = dict(prob=float(spd['prob']), speed_perturb_3_class=bool(spd.get('speed_perturb_3_class', False)))
        for name in ('noise', 'reverb'):
            c = aug_conf.get(name) if hasattr(aug_conf, 'get') else None
            if c is None or float(c.get('prob', 0.0)) <= 0:
                continue
            lib_dir = c.get(f'{name}_dir', '')
            if not lib_dir or not os.path.isdir(lib_dir) or not os.listdir(lib_dir):
                continue      # as upstream: without its library of noise / impulse-response files the augmentor is a no-op
            _LOG.warning('%s perturbation (prob %s, %s) is not built on the MI355X engine: skipped', name, c.get('prob'), lib_dir)
        vol = aug_conf.get('volume')
        if vol is not None and float(vol.get('prob', 0.0)) > 0:
            self.volume_conf = dict(prob=float(vol['prob']), min_gain_dBFS=float(vol.get('min_gain_dBFS', -15)),
                                    max_gain_dBFS=float(vol.get('max_gain_dBFS', 15)))
        if aug_conf.get('spec_aug') is not None:
            self.spec_augment = SpecAugmentor(**aug_conf['spec_aug'])
