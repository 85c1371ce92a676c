"""GPU integration test of the reference-shaped trainer (ppvector/trainer.py mirror): list files -> decode threads ->
GPU batch assembly -> Fbank -> SpecAugment -> training step -> evaluation -> checkpoints in the reference's directory layout
-> resume -> PPVectorPredictor on the saved best_model.  Real speech: the four reference WAVs (tests/golden/wavs_3s.npz)."""
import json
import os
import wave

import numpy as np
import pytest
import torch

from oracle import fbank as ofb
from oracle import models as om

pytestmark = pytest.mark.gpu


def _write_wav(path, pcm):
    with wave.open(path, 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(np.asarray(pcm, np.int16).tobytes())


def _configs(root, max_epoch):
    return dict(
        dataset_conf=dict(dataset=dict(min_duration=0.3, max_duration=2, sample_rate=16000, use_dB_normalization=True, target_dB=-20),
                          sampler=dict(batch_size=4, shuffle=True, drop_last=True), dataLoader=dict(num_workers=2),
                          eval_conf=dict(batch_size=2, max_duration=20),
                          train_list=f'{root}/train_list.txt', enroll_list=f'{root}/enroll_list.txt', trials_list=f'{root}/trials_list.txt',
                          is_use_pksampler=False, sample_per_id=4),
        preprocess_conf=dict(feature_method='Fbank', method_args=dict(sr=16000, n_mels=80)),
        model_conf=dict(model='TDNN', model_args=dict(embd_dim=192, pooling_type='ASP'),
                        classifier=dict(classifier_type='Cosine', num_speakers=2, num_blocks=0)),
        loss_conf=dict(loss='AAMLoss', loss_args=dict(margin=0.2, scale=32, easy_margin=False, label_smoothing=0.0),
                       use_margin_scheduler=True, margin_scheduler_args=dict(initial_margin=0.0, final_margin=0.3)),
        optimizer_conf=dict(optimizer='Adam', optimizer_args=dict(weight_decay=1e-6), scheduler='WarmupCosineSchedulerLR',
                            scheduler_args=dict(learning_rate=2e-3, min_lr=1e-5, warmup_epoch=1)),
        train_conf=dict(enable_amp=False, max_epoch=max_epoch, log_interval=1))


def test_trainer_end_to_end(golden_dir, tmp_path):
    from ppvector.predict import PPVectorPredictor
    from ppvector.trainer import PPVectorTrainer
    root = str(tmp_path)
    pcm = np.load(f'{golden_dir}/wavs_3s.npz')['pcm']                       # a_1, a_2, b_1, b_2 (3 s each)
    train = []
    for spk, rows in ((0, (0, 1)), (1, (2, 3))):
        for r in rows:
            for k, (a, b) in enumerate(((0, 48000), (4000, 44000), (8000, 30000), (0, 3000))):   # the last one is < min_duration
                p = f'{root}/s{spk}_{r}_{k}.wav'
                _write_wav(p, pcm[r, a:b])
                train.append(f'{p}\t{spk}')
    for name, rows in (('enroll', ((0, 0), (2, 1))), ('trials', ((1, 0), (3, 1)))):
        lines = []
        for r, spk in rows:
            p = f'{root}/{name}_{r}.wav'
            _write_wav(p, pcm[r])
            lines.append(f'{p}\t{spk}')
        open(f'{root}/{name}_list.txt', 'w').write('\n'.join(lines) + '\n')
    open(f'{root}/train_list.txt', 'w').write('\n'.join(train) + '\n')
    aug = dict(speed=dict(prob=0.0), volume=dict(prob=0.0, min_gain_dBFS=-15, max_gain_dBFS=15), noise=dict(prob=0.0),
               reverb=dict(prob=0.0), spec_aug=dict(prob=0.5, freq_mask_ratio=0.1, n_freq_masks=1, time_mask_ratio=0.05, n_time_masks=1,
                                                   max_time_warp=0))
    save = f'{root}/models'
    tr = PPVectorTrainer(_configs(root, 3), use_gpu=True, data_augment_configs=aug)
    tr.train(save_model_path=save, resume_model=None, pretrained_model=None, do_eval=True)
    fam = f'{save}/TDNN_Fbank'
    assert sorted(os.listdir(fam)) == ['best_model', 'epoch_1', 'epoch_2', 'epoch_3', 'last_model']
    st = json.load(open(f'{fam}/last_model/model.state', encoding='utf-8'))
    assert st['last_epoch'] == 3 and st['model_conf.model'] == 'TDNN' and st['feature_method'] == 'Fbank' and 0.0 <= st['eer'] <= 1.0
    assert abs(st['margin'] - tr.margin_scheduler.get_margin()) < 1e-12
    assert tr.train_step == 3 * len(tr.train_loader) == 12 and tr.train_loss is not None and np.isfinite(tr.train_loss)
    eer, min_dcf, thr = tr.evaluate()
    assert (eer, min_dcf, thr) == (tr.eval_eer, tr.eval_min_dcf, tr.eval_threshold) and 0.0 <= eer <= 1.0
    # the saved best_model serves PPVectorPredictor; its embedding equals the trainer's backbone in eval mode on the same input
    cfg = _configs(root, 3)
    pred = PPVectorPredictor(cfg, model_path=f'{fam}/last_model')
    e_pred = pred.predict(f'{root}/enroll_0.wav')
    x = pcm[0].astype(np.float32) / 32768.0
    x = x * 10.0 ** ((-20.0 - 10.0 * np.log10(np.mean(x.astype(np.float64) ** 2))) / 20.0)
    feats = ofb.featurize(x[None].astype(np.float32), feature_method='Fbank', method_args=dict(sr=16000, n_mels=80))
    sd = {k[2:]: v.detach().cpu() for k, v in tr.model.state_dict().items() if k.startswith('0.')}
    with torch.no_grad():
        e_or = om.tdnn_forward(sd, torch.from_numpy(feats))[0].numpy()
    cos = float(np.dot(e_pred, e_or) / (np.linalg.norm(e_pred) * np.linalg.norm(e_or)))
    assert cos > 1 - 1e-4, cos
    # resume: a new trainer with one more epoch picks last_model up, fast-forwards the schedules and trains exactly one epoch
    tr2 = PPVectorTrainer(_configs(root, 4), use_gpu=True, data_augment_configs=aug)
    tr2.train(save_model_path=save, do_eval=False)
    assert tr2.train_step == 16 and tr2.optimizer.t == 16 and tr2.scheduler.i == 16
    assert sorted(os.listdir(fam)) == ['best_model', 'epoch_2', 'epoch_3', 'epoch_4', 'last_model']
    # extract_features (trainer.py:134-160) dumps per-utterance features + '<list>_features.txt'; a trainer on those lists takes
    # the .npy route (reader.py:78-83: crop to the max_duration frame count) through collate_fn's zero padding
    tr2.extract_features(save_dir=f'{root}/features', max_duration=100)
    lines = open(f'{root}/train_list_features.txt').read().strip().split('\n')
    assert len(lines) == 16 and all(l.split('\t')[0].endswith('.npy') for l in lines)
    f0 = np.load(lines[0].split('\t')[0])
    x0 = pcm[0, 0:48000].astype(np.float32) / 32768.0
    x0 = x0 * 10.0 ** ((-20.0 - 10.0 * np.log10(np.mean(x0.astype(np.float64) ** 2))) / 20.0)
    ref0 = ofb.featurize(x0[None].astype(np.float32), feature_method='Fbank', method_args=dict(sr=16000, n_mels=80))[0]
    assert f0.shape == ref0.shape == (298, 80) and np.max(np.abs(f0 - ref0)) < 5e-3
    cfg_f = _configs(root, 1)
    for k in ('train', 'enroll', 'trials'):
        cfg_f['dataset_conf'][f'{k}_list'] = f'{root}/{k}_list_features.txt'
    tr3 = PPVectorTrainer(cfg_f, use_gpu=True, data_augment_configs=aug)
    tr3.train(save_model_path=f'{root}/models_f', do_eval=True)
    assert tr3.train_step == 4 and np.isfinite(tr3.train_loss) and 0.0 <= tr3.eval_eer <= 1.0
    assert tr3.train_dataset.max_feature_len == 198                          # frames of a 2 s crop
    # the reference's default augmentation: speed perturbation on every utterance, here with the 3-class label offset
    # (trainer.py:171-173: the classifier grows to 3 x num_speakers)
    aug_s = dict(aug, speed=dict(prob=1.0, speed_perturb_3_class=True))
    tr4 = PPVectorTrainer(_configs(root, 1), use_gpu=True, data_augment_configs=aug_s)
    tr4.train(save_model_path=f'{root}/models_s', do_eval=False)
    assert tr4.model[1].weight.shape == (192, 6) and tr4.train_step == 4 and np.isfinite(tr4.train_loss)
    # cooperative stop flags (trainer.py:81, :424)
    tr2.stop_eval = True
    assert tr2.evaluate() == (-1, -1, -1)
    with pytest.raises(NotImplementedError):
        tr2.export()
