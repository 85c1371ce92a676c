"""PPVectorPredictor on the MI355X engine -- the inference API of ppvector/predict.py:24-396.

Hot path (what runs in HIP): ``predict`` (:218-233), ``predict_batch`` (:235-269: zero-padded waveforms, ratio
mask, one featurizer call, backbone in chunks), ``contrast`` (:271-283) and the 1:N search of ``recognition``
(:173-187, :337-342) -- Fbank+CMN, backbone forward and cosine scoring all go through libvpmi.
Host bookkeeping kept in Python as in the reference: audio decoding / resampling / dB normalisation
(``_load_audio`` :189-216, yeaudio semantics restated for PCM WAV + ndarray input), the user index
(pickle ``audio_indexes.bin`` with the reference's key names, :86-109).  Speaker diarisation is out of scope.
"""
import io
import os
import pickle
import shutil
import wave
from io import BufferedReader

import numpy as np
import torch
import yaml
from torch import nn

from ppvector.data_utils.featurizer import AudioFeaturizer
from ppvector.metric.metrics import cosine_score_matrix
from ppvector.models import build_model
from ppvector.utils.utils import dict_to_object


class AudioSegment:
    """Minimal stand-in for yeaudio.audio.AudioSegment (mono float32 samples in [-1, 1])."""

    def __init__(self, samples, sample_rate):
        samples = np.asarray(samples)
        if samples.dtype.kind in 'iu':
            samples = samples.astype(np.float32) / float(2 ** (8 * samples.dtype.itemsize - 1))
        samples = samples.astype(np.float32)
        if samples.ndim == 2:                      # (frames, channels) -> mono
            samples = samples.mean(axis=1)
        self.samples, self.sample_rate = samples, int(sample_rate)

    @classmethod
    def from_file(cls, f):
        with wave.open(f, 'rb') as w:
            n, ch, sw, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
            raw = w.readframes(n)
        if sw not in (1, 2, 4):
            raise Exception(f'不支持该数据类型: {8 * sw}-bit PCM')
        dt = {1: np.uint8, 2: np.int16, 4: np.int32}[sw]
        x = np.frombuffer(raw, dtype=dt)
        if sw == 1:
            x = (x.astype(np.int16) - 128).astype(np.int8)
        return cls(x.reshape(-1, ch), sr)

    @classmethod
    def from_bytes(cls, b):
        return cls.from_file(io.BytesIO(b))

    @classmethod
    def from_ndarray(cls, x, sample_rate=16000):
        return cls(x, sample_rate)

    @property
    def duration(self):
        return self.samples.shape[0] / float(self.sample_rate)

    @property
    def rms_db(self):
        return 10.0 * np.log10(max(float(np.mean(self.samples.astype(np.float64) ** 2)), 1e-20))

    def resample(self, target_sample_rate):
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(target_sample_rate), self.sample_rate)
        self.samples = resample_poly(self.samples, int(target_sample_rate) // g, self.sample_rate // g).astype(np.float32)
        self.sample_rate = int(target_sample_rate)

    def normalize(self, target_db=-20, max_gain_db=300.0):
        gain = min(target_db - self.rms_db, max_gain_db)
        self.samples = (self.samples * (10.0 ** (gain / 20.0))).astype(np.float32)


class PPVectorPredictor:
    def __init__(self, configs, threshold=0.6, audio_db_path=None, model_path='models/CAMPPlus_Fbank/best_model/',
                 use_gpu=True):
        """声纹识别预测工具 (same arguments as the reference; ``use_gpu`` must be True: no CPU fallback)."""
        if not use_gpu:
            raise RuntimeError('the MI355X engine has no CPU path (use_gpu=False is not available)')
        assert torch.cuda.is_available(), 'GPU不可用'
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.threshold = threshold
        if isinstance(configs, str):
            with open(configs, 'r', encoding='utf-8') as f:
                configs = yaml.load(f.read(), Loader=yaml.FullLoader)
        self.configs = dict_to_object(configs)
        self._audio_featurizer = AudioFeaturizer(feature_method=self.configs.preprocess_conf.feature_method,
                                                 method_args=self.configs.preprocess_conf.get('method_args', {}))
        backbone = build_model(input_size=self._audio_featurizer.feature_dim, configs=self.configs)
        self.predictor = nn.Sequential(backbone)
        if isinstance(model_path, dict):                       # an in-memory state dict ("0.<...>" keys)
            state = model_path
        else:
            if not os.path.exists(model_path):
                raise Exception("模型文件不存在，请检查{}是否存在！".format(model_path))
            if os.path.isdir(model_path):                      # the reference's checkpoint directory (predict.py:61-62)
                model_path = os.path.join(model_path, 'model.pdparams')
            assert os.path.exists(model_path), f"{model_path} 模型不存在！"
            from ppvector.utils.checkpoint import load_pdparams
            state = load_pdparams(model_path)
        state = {k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v for k, v in state.items()}
        own = self.predictor.state_dict()
        # load_pretrained semantics (utils/checkpoint.py:11-42): keep what matches by name and shape
        self.predictor.load_state_dict({k: v for k, v in state.items() if k in own and tuple(v.shape) == tuple(own[k].shape)},
                                       strict=False)
        self.predictor.to(self.device).eval()
        self.audio_feature = None
        self.users_name, self.users_audio_path = [], []
        self.audio_db_path = audio_db_path
        if self.audio_db_path is not None:
            self.audio_indexes_path = os.path.join(audio_db_path, "audio_indexes.bin")
            self.__load_audio_db(self.audio_db_path)

    # ------------------------------------------------------------------ voice-print index (host bookkeeping)
    def __load_audio_indexes(self):
        if not os.path.exists(self.audio_indexes_path):
            return
        with open(self.audio_indexes_path, "rb") as f:
            indexes = pickle.load(f)
        for name, feature, path in zip(indexes["users_name"], indexes["faces_feature"], indexes["users_image_path"]):
            if not os.path.exists(path):
                continue
            self.users_name.append(name)
            self.users_audio_path.append(path)
            self.audio_feature = feature[None] if self.audio_feature is None else np.vstack((self.audio_feature, feature))

    def __write_index(self):
        with open(self.audio_indexes_path, "wb") as f:
            pickle.dump({"users_name": self.users_name, "faces_feature": self.audio_feature,
                         "users_image_path": self.users_audio_path}, f)

    def __load_audio_db(self, audio_db_path):
        self.__load_audio_indexes()
        os.makedirs(audio_db_path, exist_ok=True)
        todo = []
        for name in sorted(os.listdir(audio_db_path)):
            audio_dir = os.path.join(audio_db_path, name)
            if not os.path.isdir(audio_dir):
                continue
            for file in sorted(os.listdir(audio_dir)):
                p = os.path.join(audio_dir, file).replace('\\', '/')
                if p not in self.users_audio_path:
                    todo.append((name, p))
        for name, p in todo:
            feat = self.predict(p)
            self.users_name.append(name)
            self.users_audio_path.append(p)
            self.audio_feature = feat[None] if self.audio_feature is None else np.vstack((self.audio_feature, feat))
        if todo:
            self.__write_index()

    # ------------------------------------------------------------------ audio front end (host)
    def _load_audio(self, audio_data, sample_rate=16000):
        if isinstance(audio_data, str):
            seg = AudioSegment.from_file(audio_data)
        elif isinstance(audio_data, BufferedReader):
            seg = AudioSegment.from_file(audio_data)
        elif isinstance(audio_data, np.ndarray):
            seg = AudioSegment.from_ndarray(audio_data, sample_rate)
        elif isinstance(audio_data, bytes):
            seg = AudioSegment.from_bytes(audio_data)
        elif isinstance(audio_data, AudioSegment):
            seg = audio_data
        else:
            raise Exception(f'不支持该数据类型，当前数据类型为：{type(audio_data)}')
        ds = self.configs.dataset_conf.dataset
        assert seg.duration >= ds.min_duration, f'音频太短，最小应该为{ds.min_duration}s，当前音频为{seg.duration}s'
        if seg.sample_rate != ds.sample_rate:
            seg.resample(ds.sample_rate)
        if ds.use_dB_normalization:
            seg.normalize(target_db=ds.target_dB)
        return seg

    # ------------------------------------------------------------------ hot path
    def predict(self, audio_data, sample_rate=16000):
        """预测一个音频的特征 -> (embd_dim,) float32."""
        seg = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        x = torch.from_numpy(seg.samples).to(self.device).unsqueeze(0)
        feat = self._audio_featurizer(x, want_bf16=self._want_bf16())
        return self.predictor(feat).cpu().numpy()[0]

    def predict_batch(self, audios_data, sample_rate=16000, batch_size=32):
        """预测一批音频的特征: zero-pad the WAVEFORMS to the longest, ratio mask, as the reference does."""
        waves = [self._load_audio(audio_data=a, sample_rate=sample_rate).samples for a in audios_data]
        longest = max(w.shape[0] for w in waves)
        inputs = np.zeros((len(waves), longest), dtype=np.float32)
        ratios = []
        for i, w in enumerate(waves):
            inputs[i, :w.shape[0]] = w
            ratios.append(w.shape[0] / longest)
        x = torch.from_numpy(inputs).to(self.device)
        r = torch.tensor(ratios, dtype=torch.float32, device=self.device)
        feat = self._audio_featurizer(x, r)
        outs = [self.predictor(feat[i:i + batch_size].contiguous()).cpu().numpy() for i in range(0, len(waves), batch_size)]
        return np.concatenate(outs, axis=0)

    def contrast(self, audio_data1, audio_data2):
        """声纹对比 -> cosine similarity."""
        f = np.stack([self.predict(audio_data1), self.predict(audio_data2)])
        s = cosine_score_matrix(torch.from_numpy(f[:1]).to(self.device), torch.from_numpy(f[1:]).to(self.device))
        return float(s[0, 0])

    def _want_bf16(self):
        import ppvector
        return ppvector.get_compute_dtype() == 'bfloat16'

    # ------------------------------------------------------------------ 1:N
    def register(self, audio_data, user_name: str, sample_rate=16000):
        """声纹注册."""
        seg = self._load_audio(audio_data=audio_data, sample_rate=sample_rate)
        feat = self.predict(audio_data=seg)
        if self.audio_db_path is not None:
            d = os.path.join(self.audio_db_path, user_name)
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f'{len(os.listdir(d))}.wav').replace('\\', '/')
            pcm = np.clip(seg.samples * 32768.0, -32768, 32767).astype(np.int16)
            with wave.open(path, 'wb') as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(seg.sample_rate); w.writeframes(pcm.tobytes())
        else:
            path = f'<memory:{len(self.users_name)}>'
        self.users_name.append(user_name)
        self.users_audio_path.append(path)
        self.audio_feature = feat[None] if self.audio_feature is None else np.vstack((self.audio_feature, feat))
        if self.audio_db_path is not None:
            self.__write_index()
        return True, "注册成功"

    def recognition(self, audio_data, threshold=None, sample_rate=16000):
        """声纹识别 -> (name, score) of the best enrolled user, or (None, None) under the threshold."""
        if threshold:
            self.threshold = threshold
        if self.audio_feature is None:
            return None, None
        feat = self.predict(audio_data, sample_rate=sample_rate)
        users = sorted(set(self.users_name))
        means = np.stack([self.audio_feature[[i for i, n in enumerate(self.users_name) if n == u]].mean(axis=0) for u in users])
        s = cosine_score_matrix(torch.from_numpy(feat[None]).to(self.device),
                                torch.from_numpy(means.astype(np.float32)).to(self.device)).cpu().numpy()[0]
        i = int(np.argmax(s))
        if s[i] >= self.threshold:
            return users[i], round(float(s[i]), 5)
        return None, None

    def get_users(self):
        return self.users_name

    def remove_user(self, user_name):
        if user_name not in self.users_name:
            return False
        keep = [i for i, n in enumerate(self.users_name) if n != user_name]
        if self.audio_db_path is not None:
            shutil.rmtree(os.path.join(self.audio_db_path, user_name), ignore_errors=True)
        self.users_name = [self.users_name[i] for i in keep]
        self.users_audio_path = [self.users_audio_path[i] for i in keep]
        self.audio_feature = self.audio_feature[keep] if keep else None
        if self.audio_db_path is not None:
            self.__write_index()
        return True
