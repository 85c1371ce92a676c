"""CPU tests of the dataset mirror (ppvector/data_utils/reader.py vs ppvector/data_utils/reader.py:16-163 of the reference): list
parsing, min_duration skipping, train-mode crop starts, eval-mode length sort, the .npy route.  No GPU: items are raw host data."""
import random
import wave

import numpy as np
import pytest

from ppvector.data_utils.featurizer import AudioFeaturizer
from ppvector.data_utils.reader import PPVectorDataset
from ppvector.utils.utils import dict_to_object


def _wav(path, n, sr=16000, seed=0):
    rng = np.random.RandomState(seed)
    pcm = (rng.standard_normal(n) * 3000).astype(np.int16)
    with wave.open(str(path), 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
    return pcm


def test_dataset_items_crops_sort_and_npy(tmp_path):
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    lens = [48000, 3000, 70000, 20000, 8000]                       # the 2nd one is below min_duration (0.3 s = 4800 samples)
    rows = []
    for i, n in enumerate(lens):
        _wav(tmp_path / f'u{i}.wav', n, seed=i)
        rows.append(f'{tmp_path}/u{i}.wav\t{i % 3}')
    lst = tmp_path / 'list.txt'
    lst.write_text('\n'.join(rows) + '\n\n')
    aug = dict_to_object(dict(speed=dict(prob=0.0), volume=dict(prob=1.0, min_gain_dBFS=-15, max_gain_dBFS=15), noise=dict(prob=0.5),
                              reverb=dict(prob=0.5), spec_aug=dict(prob=0.5, freq_mask_ratio=0.1, n_freq_masks=1, time_mask_ratio=0.05,
                                                                   n_time_masks=1, max_time_warp=0)))
    ds = PPVectorDataset(str(lst), fz, max_duration=3, min_duration=0.3, mode='train', aug_conf=aug, num_speakers=3)
    assert len(ds) == 5 and [int(l) for l in ds.labels] == [0, 1, 2, 0, 1]
    assert ds.max_samples == 48000 and ds.max_feature_len == 298
    assert ds.spec_augment is not None and ds.volume_conf == dict(prob=1.0, min_gain_dBFS=-15.0, max_gain_dBFS=15.0)
    random.seed(3)
    it = ds[0]
    assert it['samples'].dtype == np.float32 and it['samples'].shape == (48000,) and it['start'] == 0 and it['label'] == 0
    assert -15.0 <= it['gain_dB'] <= 15.0 and abs(float(np.abs(it['samples']).max())) <= 1.0
    short = ds[1]                                                   # too short: falls through to the next entry (reader.py:89-91)
    assert short['samples'].shape == (70000,) and short['label'] == 2
    starts = {ds[2]['start'] for _ in range(20)}                    # 70000 samples > max: uniform random crop start
    assert len(starts) > 5 and all(0 <= s <= 70000 - 48000 for s in starts)
    # speed perturbation: rate drawn from {1.0, 0.9, 1.1}; 3-class mode offsets the label; the crop start lives on the new length
    aug3 = dict_to_object(dict(speed=dict(prob=1.0, speed_perturb_3_class=True), volume=None, noise=None, reverb=None, spec_aug=None))
    d3 = PPVectorDataset(str(lst), fz, max_duration=3, min_duration=0.3, mode='train', aug_conf=aug3, num_speakers=3)
    seen = set()
    for _ in range(40):
        it = d3[0]
        k = {1.0: 0, 0.9: 1, 1.1: 2}[it['speed']]
        seen.add(k)
        assert it['label'] == 0 + 3 * k and it['samples'].shape == (48000,)          # samples stay raw: the GPU resamples
        new_n = 48000 if k == 0 else int(48000 / it['speed'])
        assert 0 <= it['start'] <= max(0, new_n - 48000)
    assert seen == {0, 1, 2}
    # eval: sorted by duration, crop starts at 0, no augmentation objects
    ev = PPVectorDataset(str(lst), fz, max_duration=20, min_duration=0.3, mode='eval')
    assert [ev[i]['samples'].shape[0] for i in range(5)] == sorted(lens) and all(ev[i]['start'] == 0 and ev[i]['gain_dB'] == 0.0 for i in range(5))
    assert ev.spec_augment is None and [int(l) for l in ev.labels] == [1, 1, 0, 0, 2]
    # .npy features: cropped to the frame count of max_duration in train mode, from frame 0 in eval mode
    feats = np.arange(400 * 80, dtype=np.float32).reshape(400, 80)
    np.save(tmp_path / 'f0.npy', feats)
    np.save(tmp_path / 'f1.npy', feats[:100])
    fl = tmp_path / 'feat_list.txt'
    fl.write_text(f'{tmp_path}/f0.npy\t4\n{tmp_path}/f1.npy\t5\n')
    dn = PPVectorDataset(str(fl), fz, max_duration=3, mode='train')
    a = dn[0]
    assert a['feature'].shape == (298, 80) and a['label'] == 4 and a['feature'][0, 0] % 80 == 0 and dn[1]['feature'].shape == (100, 80)
    de = PPVectorDataset(str(fl), fz, max_duration=3, mode='eval')
    assert de[0]['feature'].shape == (100, 80) and np.array_equal(de[1]['feature'], feats[:298])
    with pytest.raises(AssertionError):
        PPVectorDataset(str(lst), fz, mode='test')
