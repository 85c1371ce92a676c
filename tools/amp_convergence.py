"""Does the enable_amp step TRAIN?  ("parity first" item 1.)

PPVectorTrainer (the reference-shaped trainer, ppvector/trainer.py:202-274 upstream) is run twice on the same synthetic, separable
speakers from the same seed and the same list files: train_conf.enable_amp False (the reference's default, ecapa_tdnn.yml:100 -- the
f32 engine) and True (bf16 matrix cores, bf16-only activations between the GEMMs).  Everything else is the reference's
configs/ecapa_tdnn.yml: ECAPA-TDNN (512, 512, 512, 512, 1536), Fbank 80, cosine head, AAMLoss (margin 0.2, scale 32) with the margin
scheduler, Adam (weight decay 1e-6), WarmupCosineSchedulerLR (1e-3 -> 1e-5), SpecAugment, 3 s crops.

    python tools/amp_convergence.py [--steps 300] [--batch 256] [--speakers 64] [--out profiles/r05_amp_convergence.log]

prints the two loss / accuracy curves (mean over each log interval), the final EER / minDCF on held-out utterances of the training
speakers' voices, and the ratio of the curves.  tests/test_gpu_train.py::test_enable_amp_trains_like_f32 runs a shorter version of
the same functions and asserts the band.

Synthetic speakers: a voiced source (harmonics of a speaker-specific pitch with per-utterance jitter and vibrato) through three
speaker-specific formant resonances, syllable-rate amplitude modulation, plus white noise at 15 dB SNR -- separable by spectral
envelope and pitch, not by level (the loader normalises to -20 dBFS) and not by a fixed waveform (phases, jitter and the noise are
drawn per utterance)."""
import argparse
import json
import logging
import os
import re
import sys
import tempfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

SR = 16000


def speaker_table(n_spk, seed=1000):
    rng = np.random.RandomState(seed)
    f0 = rng.uniform(90.0, 260.0, n_spk)
    formants = np.stack([rng.uniform(300, 900, n_spk), rng.uniform(1000, 2400, n_spk), rng.uniform(2600, 3800, n_spk)], axis=1)
    tilt = rng.uniform(0.6, 1.4, n_spk)
    return f0, formants, tilt


def synth_utterance(table, spk, n, rng):
    """n samples of speaker `spk`; the generator `rng` supplies everything that differs between that speaker's utterances."""
    f0s, formants, tilts = table
    t = np.arange(n, dtype=np.float64) / SR
    f0 = f0s[spk] * (1.0 + rng.uniform(-0.03, 0.03))
    vib = 1.0 + 0.01 * np.sin(2 * np.pi * rng.uniform(4.0, 7.0) * t + rng.uniform(0, 2 * np.pi))
    phase = 2 * np.pi * np.cumsum(f0 * vib) / SR
    x = np.zeros(n, dtype=np.float64)
    fm = formants[spk] * (1.0 + rng.uniform(-0.02, 0.02, 3))
    bw = np.array([90.0, 130.0, 180.0])
    k = 1
    while k * f0 < 0.45 * SR and k <= 40:
        fk = k * f0
        gain = sum(1.0 / (1.0 + ((fk - c) / b) ** 2) for c, b in zip(fm, bw)) / k ** tilts[spk] + 0.02 / k
        x += gain * np.sin(k * phase + rng.uniform(0, 2 * np.pi))
        k += 1
    env = 0.55 + 0.45 * np.sin(2 * np.pi * rng.uniform(2.5, 5.0) * t + rng.uniform(0, 2 * np.pi))
    x *= env
    x /= np.sqrt(np.mean(x * x)) + 1e-12
    x += 10.0 ** (-15.0 / 20.0) * rng.standard_normal(n)
    x *= 0.1 / (np.sqrt(np.mean(x * x)) + 1e-12)
    return np.clip(x, -1.0, 1.0)


def _write_wav(path, x):
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    with wave.open(path, 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(SR)
        w.writeframes(pcm.tobytes())


def build_dataset(root, n_spk=64, train_files=8, seconds=4.0, steps_per_epoch=30, batch=256, seed=1000, ragged=False):
    """WAV files + the three list files of the reference's layout under `root`.  Every training file is listed often enough for
    `steps_per_epoch` batches; the loader crops a random 3 s window out of the 4 s file each time it is drawn."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(root, exist_ok=True)
    table = speaker_table(n_spk, seed)
    n = int(seconds * SR)
    jobs, files = [], []
    for s in range(n_spk):
        for u in range(train_files):
            p = f'{root}/tr_{s}_{u}.wav'
            # ragged: half of the files are SHORTER than the 3 s crop (1.2 .. 2.8 s): batches then mix lengths and take the ragged Fbank +
            # zero-padded collate path (reader.py:72-109, collate_fn.py:5-23)
            nu = int((1.2 + 1.6 * ((s * 7 + u * 3) % 10) / 9.0) * SR) if (ragged and u % 2) else n
            jobs.append((p, s, nu, seed * 7919 + s * 131 + u))
            files.append((p, s))
    lists = {}
    for name, per, base in (('enroll', 1, 500), ('trials', 3, 600)):
        lists[name] = []
        for s in range(n_spk):
            for u in range(per):
                p = f'{root}/{name}_{s}_{u}.wav'
                jobs.append((p, s, 3 * SR, seed * 104729 + s * 131 + base + u))
                lists[name].append((p, s))
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as pool:        # (NumPy releases the GIL inside its loops)
        list(pool.map(lambda j: _write_wav(j[0], synth_utterance(table, j[1], j[2], np.random.RandomState(j[3]))), jobs))
    rep = (steps_per_epoch * batch + len(files) - 1) // len(files)
    with open(f'{root}/train_list.txt', 'w') as f:
        for _ in range(rep):
            for p, s in files:
                f.write(f'{p}\t{s}\n')
    for name, rows in lists.items():
        with open(f'{root}/{name}_list.txt', 'w') as f:
            for p, s in rows:
                f.write(f'{p}\t{s}\n')
    return len(files) * rep // batch


def configs(root, n_spk, batch, max_epoch, enable_amp, model='EcapaTdnn'):
    # model_args of the reference's shipped YAMLs (configs/{ecapa_tdnn,tdnn,cam++,resnet_se,eres2net}.yml)
    margs = {'EcapaTdnn': dict(embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536]),
             'TDNN': dict(embd_dim=192, channels=512, pooling_type='ASP'), 'CAMPPlus': dict(embd_dim=192),
             'ResNetSE': dict(embd_dim=192, pooling_type='ASP'), 'ERes2Net': dict(embd_dim=192, m_channels=32),
             'ERes2NetV2': dict(embd_dim=192, m_channels=32)}[model]
    return dict(
        dataset_conf=dict(dataset=dict(min_duration=0.3, max_duration=3, sample_rate=SR, use_dB_normalization=True, target_dB=-20),
                          sampler=dict(batch_size=batch, shuffle=True, drop_last=True), dataLoader=dict(num_workers=8),
                          eval_conf=dict(batch_size=64, max_duration=20),
                          train_list=f'{root}/train_list.txt', enroll_list=f'{root}/enroll_list.txt', trials_list=f'{root}/trials_list.txt',
                          is_use_pksampler=False, sample_per_id=4),
        preprocess_conf=dict(feature_method='Fbank', method_args=dict(sr=SR, n_mels=80)),
        model_conf=dict(model=model, model_args=margs,
                        classifier=dict(classifier_type='Cosine', num_speakers=n_spk, num_blocks=0)),
        loss_conf=dict(loss='AAMLoss', loss_args=dict(margin=0.2, scale=32, easy_margin=False, label_smoothing=0.0),
                       use_margin_scheduler=True, margin_scheduler_args=dict(initial_margin=0.0, final_margin=0.3)),
        optimizer_conf=dict(optimizer='Adam', optimizer_args=dict(weight_decay=1e-6), scheduler='WarmupCosineSchedulerLR',
                            scheduler_args=dict(learning_rate=1e-3, min_lr=1e-5, warmup_epoch=1)),
        train_conf=dict(enable_amp=bool(enable_amp), max_epoch=max_epoch, log_interval=10))


AUG = dict(speed=dict(prob=0.0), volume=dict(prob=0.0, min_gain_dBFS=-15, max_gain_dBFS=15), noise=dict(prob=0.0), reverb=dict(prob=0.0),
           spec_aug=dict(prob=0.5, freq_mask_ratio=0.1, n_freq_masks=1, time_mask_ratio=0.05, n_time_masks=1, max_time_warp=0))


class _Curve(logging.Handler):
    """The trainer's own log lines ('Train epoch: [e/E], batch: [b/N], loss: x, accuracy: y, ...') -> [(step, loss, accuracy)]."""
    PAT = re.compile(r'Train epoch: \[(\d+)/\d+\], batch: \[(\d+)/(\d+)\], loss: ([0-9.eE+-]+|nan|inf), accuracy: ([0-9.eE+-]+|nan)')

    def __init__(self):
        super().__init__(level=logging.INFO)
        self.points = []

    def emit(self, record):
        m = self.PAT.search(record.getMessage())
        if m:
            e, b, n = int(m.group(1)), int(m.group(2)), int(m.group(3))
            self.points.append(((e - 1) * n + b, float(m.group(4)), float(m.group(5))))


def run(root, n_spk, batch, max_epoch, enable_amp, model='EcapaTdnn', save=None, aug=None):
    """One training run through PPVectorTrainer.train + .evaluate -> dict(curve, eer, min_dcf, threshold, seconds, steps)."""
    import random
    import torch
    from ppvector.trainer import PPVectorTrainer
    log = logging.getLogger('ppvector')
    old_level = log.level
    log.setLevel(logging.INFO)
    h = _Curve()
    log.addHandler(h)
    try:
        random.seed(1000)
        np.random.seed(1000)
        tr = PPVectorTrainer(configs(root, n_spk, batch, max_epoch, enable_amp, model), use_gpu=True, data_augment_configs=aug or AUG)
        t0 = time.time()
        tr.train(save_model_path=save or f'{root}/models_{"amp" if enable_amp else "f32"}', do_eval=False)
        torch.cuda.synchronize()
        dt = time.time() - t0
        eer, min_dcf, thr = tr.evaluate()
        faults = int(getattr(tr.train_step_fn, 'faults', 0))
        err = getattr(tr.train_step_fn, 'capture_error', None)
    finally:
        log.removeHandler(h)
        log.setLevel(old_level)
    return dict(enable_amp=bool(enable_amp), curve=h.points, eer=eer, min_dcf=min_dcf, threshold=thr, seconds=dt, steps=tr.train_step,
                barrier_faults=faults, capture_error=err)


def compare(f32, amp):
    """Summary numbers of an A/B: mean loss over the last fifth of the run, final accuracy, EERs."""
    def tail(c):
        k = max(1, len(c) // 5)
        return float(np.mean([p[1] for p in c[-k:]])), float(np.mean([p[2] for p in c[-k:]]))
    lf, af = tail(f32['curve'])
    la, aa = tail(amp['curve'])
    return dict(tail_loss_f32=lf, tail_loss_amp=la, tail_acc_f32=af, tail_acc_amp=aa, eer_f32=f32['eer'], eer_amp=amp['eer'],
                min_dcf_f32=f32['min_dcf'], min_dcf_amp=amp['min_dcf'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--speakers', type=int, default=64)
    ap.add_argument('--epochs', type=int, default=10)
    ap.add_argument('--model', default='EcapaTdnn')
    ap.add_argument('--out', default=None)
    ap.add_argument('--ragged', action='store_true', help='half of the training files shorter than the crop: the ragged Fbank / collate path')
    ap.add_argument('--speed', action='store_true', help='the reference default: speed perturbation on every utterance (augmentation.yml:1-6)')
    ap.add_argument('--speed3', action='store_true', help='... with the 3-class label offset (trainer.py:171-173)')
    ap.add_argument('--volume', action='store_true', help='volume perturbation, prob 0.5')
    a = ap.parse_args()
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    with tempfile.TemporaryDirectory(prefix='vp_ampconv_') as root:
        spe = build_dataset(root, a.speakers, steps_per_epoch=max(1, a.steps // a.epochs), batch=a.batch, ragged=a.ragged)
        aug = dict(AUG)
        if a.speed or a.speed3:
            aug['speed'] = dict(prob=1.0, speed_perturb_3_class=bool(a.speed3))
        if a.volume:
            aug['volume'] = dict(prob=0.5, min_gain_dBFS=-15, max_gain_dBFS=15)
        say(f'# options: ragged={a.ragged} speed={a.speed or a.speed3} speed_perturb_3_class={a.speed3} volume={a.volume}')
        say(f'# enable_amp vs f32 through PPVectorTrainer: {a.model}, {a.speakers} synthetic speakers, batch {a.batch}, '
            f'{a.epochs} epochs x {spe} steps, Fbank 80, AAMLoss + margin scheduler, Adam 1e-3 warm-up 1 epoch -> cosine, SpecAugment 0.5')
        res = {}
        for amp in (False, True):
            r = res[amp] = run(root, a.speakers, a.batch, a.epochs, amp, a.model, aug=aug)
            say(f'## enable_amp={amp}: {r["steps"]} steps in {r["seconds"]:.1f} s (data loading included), EER {r["eer"]:.5f}, '
                f'minDCF {r["min_dcf"]:.5f}, threshold {r["threshold"]:.2f}, grid-barrier faults {r["barrier_faults"]}, '
                f'capture_error {r["capture_error"]}')
            say('step loss accuracy')
            for s, l, ac in r['curve']:
                say(f'{s:5d} {l:9.5f} {ac:7.5f}')
        c = compare(res[False], res[True])
        say('## summary ' + json.dumps(c))
    if a.out:
        with open(a.out, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
