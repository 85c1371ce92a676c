"""AudioFeaturizer on the MI355X engine.

Mirrors ppvector/data_utils/featurizer.py:7-80 (constructor arguments, call signature,
``feature_dim``, the exceptions for unknown methods).  'Fbank' (KaldiFbank, featurizer.py:83-101)
runs in one fused HIP pipeline (csrc/fbank.hip): framing, DC removal, pre-emphasis, Povey window,
512-point FFT, Kaldi mel bank, log, time-mean subtraction and the length mask.
"""
import ctypes as C
import math

import torch
from torch import nn

from ppvector import _native as N

_KALDI_KW = {'sr': 'sample_rate', 'n_mels': 'n_mels', 'frame_length': 'frame_length_ms',
             'frame_shift': 'frame_shift_ms', 'preemphasis_coefficient': 'preemph',
             'remove_dc_offset': 'remove_dc', 'low_freq': 'low_freq', 'high_freq': 'high_freq'}
_KALDI_FIXED = {'dither': 0.0, 'window_type': 'povey', 'snip_edges': True, 'use_energy': False,
                'use_log_fbank': True, 'use_power': True, 'round_to_power_of_two': True,
                'subtract_mean': False, 'htk_compat': False, 'raw_energy': True, 'channel': -1,
                'vtln_warp': 1.0, 'energy_floor': 1.0, 'blackman_coeff': 0.42, 'vtln_low': 100.0,
                'vtln_high': -500.0}


class AudioFeaturizer(nn.Module):
    """音频特征器 (feature_method: 'Fbank' on the HIP engine).

    :param feature_method: 所使用的预处理方法
    :param method_args: 预处理方法的参数
    """

    def __init__(self, feature_method='MelSpectrogram', method_args={}):
        super().__init__()
        self._method_args = method_args
        self._feature_method = feature_method
        if feature_method == 'Fbank':
            self._opts = self._fbank_opts(method_args)
        elif feature_method == 'MelSpectrogram':
            self._opts = self._mel_opts(method_args)
        elif feature_method == 'LogMelSpectrogram':
            self._opts = self._mel_opts(method_args, log=True)
        elif feature_method == 'MFCC':
            args = dict(method_args)
            self._n_mfcc = int(args.pop('n_mfcc', 40))
            self._opts = self._mel_opts(args, log=True)
            if self._n_mfcc > self._opts.n_mels:
                raise AssertionError('n_mfcc cannot be larger than n_mels: %d vs %d' % (self._n_mfcc, self._opts.n_mels))
            self._dct = None
        elif feature_method == 'Spectrogram':
            self._opts = None       # known to the reference; not built on the HIP engine yet
        else:
            raise Exception(f'预处理方法 {self._feature_method} 不存在!')
        self._ws = N.Workspace()

    @staticmethod
    def _fbank_opts(method_args):
        o = N.FbankOpts()
        N.lib().vp_fbank_default_opts(C.byref(o))
        for k, v in dict(method_args).items():
            if k in _KALDI_KW:
                setattr(o, _KALDI_KW[k], type(getattr(o, _KALDI_KW[k]))(v))
            elif k in _KALDI_FIXED:
                if v != _KALDI_FIXED[k]:
                    raise NotImplementedError(f'Fbank option {k}={v} is not built on the HIP engine')
            else:
                raise TypeError(f"fbank() got an unexpected keyword argument '{k}'")
        return o

    @staticmethod
    def _mel_opts(method_args, log=False):
        """paddle.audio.features.MelSpectrogram keyword surface (defaults: sr 22050, n_fft 2048, hop_length 512,
        win_length None, window 'hann', power 2.0, center True, pad_mode 'reflect', n_mels 64, f_min 50.0,
        f_max None, htk False, norm 'slaney').  log=True: paddle.audio.features.LogMelSpectrogram (featurizer.py:20-21) =
        the same mel spectrogram through power_to_db(ref_value=1.0, amin=1e-10, top_db=None); its n_fft default is 512."""
        o = N.MelOpts()
        N.lib().vp_mel_default_opts(C.byref(o))
        if log:
            o.log_db, o.n_fft = 1, 512
        names = {'sr': 'sample_rate', 'n_fft': 'n_fft', 'hop_length': 'hop_length', 'win_length': 'win_length',
                 'n_mels': 'n_mels', 'f_min': 'f_min', 'f_max': 'f_max', 'power': 'power'}
        if log:
            names.update(ref_value='ref_value', amin='amin')
        fixed = {'window': 'hann', 'center': True, 'pad_mode': 'reflect', 'htk': False, 'norm': 'slaney', 'dtype': 'float32'}
        if log:
            fixed['top_db'] = None
        for k, v in dict(method_args).items():
            if k in names:
                if v is None:
                    v = 0
                setattr(o, names[k], type(getattr(o, names[k]))(v))
            elif k in fixed:
                if v != fixed[k]:
                    raise NotImplementedError(f'MelSpectrogram option {k}={v} is not built on the HIP engine')
            else:
                raise TypeError(f"__init__() got an unexpected keyword argument '{k}'")
        if dict(method_args).get('hop_length') is None:              # paddle: hop_length defaults to win_length // 4
            o.hop_length = (o.win_length or o.n_fft) // 4
        if o.f_max > 0.5 * o.sample_rate:
            raise ValueError(f'f_max {o.f_max} is above the Nyquist frequency of sr {o.sample_rate}')
        return o

    def num_frames(self, n_samples):
        if self._feature_method in ('MelSpectrogram', 'LogMelSpectrogram', 'MFCC'):
            return N.lib().vp_mel_num_frames(C.byref(self._opts), int(n_samples))
        return N.lib().vp_fbank_num_frames(C.byref(self._opts), int(n_samples))

    def forward(self, waveforms, input_lens_ratio=None, want_bf16=False):
        """waveforms (L,) or (B, L) float32 on the GPU -> (B, T, F) float32.

        With want_bf16 the bf16 copy the bf16 network consumes is produced by the same kernel and
        attached to the result as ``._vp_bf16``.
        """
        if self._opts is None:
            raise NotImplementedError(f'feature_method {self._feature_method} is not built on the HIP engine yet '
                                      '(Fbank is); there is no CPU fallback')
        if waveforms.dim() == 1:
            waveforms = waveforms.unsqueeze(0)
        if not waveforms.is_cuda:
            raise N.VpmiError('AudioFeaturizer needs GPU tensors: the engine has no CPU fallback')
        # 16-bit PCM as decoded (torch.int16): widened inside the Fbank frame kernel (x 1 / 32768, what the reference's readers do on
        # the host), so an upload in front of this call carries half the bytes; other methods widen first
        pcm16 = waveforms.dtype == torch.int16 and self._feature_method == 'Fbank' and waveforms.shape[-1] % 2 == 0
        if waveforms.dtype == torch.int16 and not pcm16:
            waveforms = waveforms.float() * (1.0 / 32768.0)
        wav = waveforms.contiguous() if pcm16 else waveforms.contiguous().float()
        B, L = wav.shape
        lib, ctx = N.lib(), N.ctx(wav.device)
        mel = self._feature_method in ('MelSpectrogram', 'LogMelSpectrogram', 'MFCC')
        T = (lib.vp_mel_num_frames if mel else lib.vp_fbank_num_frames)(C.byref(self._opts), L)
        if T <= 0:
            raise ValueError(f'{L} samples are shorter than one analysis window')
        F = self._opts.n_mels
        out = torch.empty((B, T, F), dtype=torch.float32, device=wav.device)
        out16 = torch.empty((B, T, F), dtype=torch.bfloat16, device=wav.device) if want_bf16 else None
        ratio = None
        if input_lens_ratio is not None:
            ratio = input_lens_ratio.to(device=wav.device, dtype=torch.float32).contiguous()
        nws = (lib.vp_mel_workspace_bytes if mel else lib.vp_fbank_workspace_bytes)(C.byref(self._opts), B, L)
        ws = self._ws.get(nws, wav.device)
        if pcm16:
            N.check(lib.vp_fbank_cmn_pcm16(ctx, N.ptr(wav), 1.0 / 32768.0, N.ptr(ratio), B, L, C.byref(self._opts), N.ptr(out), N.ptr(out16),
                                           N.ptr(ws), ws.numel(), N.stream_ptr()), ctx)
        else:
            fn = lib.vp_melspec_cmn_f32 if mel else lib.vp_fbank_cmn_f32
            N.check(fn(ctx, N.ptr(wav), N.ptr(ratio), B, L, C.byref(self._opts), N.ptr(out), N.ptr(out16), N.ptr(ws),
                       ws.numel(), N.stream_ptr()), ctx)
        if self._feature_method == 'MFCC':
            # paddle.audio.features.MFCC (featurizer.py:26-27): log-mel @ create_dct(n_mfcc, n_mels, norm='ortho').  The DCT is
            # linear, so it commutes with the mean subtraction and the zeroed rows already applied to the log-mel features.
            if self._dct is None or self._dct.device != wav.device:
                n = torch.arange(F, dtype=torch.float64)
                k = torch.arange(self._n_mfcc, dtype=torch.float64).unsqueeze(1)
                dct = torch.cos(math.pi / F * (n + 0.5) * k)
                dct[0] *= 1.0 / math.sqrt(2.0)
                self._dct = (dct * math.sqrt(2.0 / F)).t().contiguous().float().to(wav.device)        # (n_mels, n_mfcc)
            mf = torch.empty((B, T, self._n_mfcc), dtype=torch.float32, device=wav.device)
            N.check(lib.vp_dense_f32(ctx, N.ptr(out), F, N.ptr(self._dct), 1, None, B * T, self._n_mfcc, F, N.VP_ACT_NONE, N.ptr(mf),
                                     self._n_mfcc, N.stream_ptr()), ctx)
            out, out16 = mf, (torch.empty_like(mf, dtype=torch.bfloat16) if want_bf16 else None)
            if out16 is not None:
                N.check(lib.vp_cast_f32_bf16(ctx, N.ptr(out), N.ptr(out16), out.numel(), N.stream_ptr()), ctx)
        if out16 is not None:
            out._vp_bf16 = out16
        return out

    def forward_ragged(self, waveforms, n_samples, want_bf16=False):
        """The training loader's semantics for a ragged batch: utterance b occupies the first n_samples[b] samples of row b;
        it is featurised as if alone (time mean over ITS frames -- the reference's per-utterance call, reader.py:102-103) and
        the rows past its last frame are zero (collate_fn.py:5-23).  Returns (features (B, T, F), input_lens (B,) int64).
        One batched launch for 'Fbank'; other methods run utterance by utterance and are packed by vp_pad_batch."""
        if not waveforms.is_cuda:
            raise N.VpmiError('AudioFeaturizer needs GPU tensors: the engine has no CPU fallback')
        wav = waveforms.contiguous().float()
        B, L = wav.shape
        ns = torch.as_tensor(n_samples).to(device=wav.device, dtype=torch.int32).contiguous()
        if self._feature_method != 'Fbank':
            from ppvector.data_utils.collate_fn import collate_fn
            feats = [self.forward(wav[b, :int(n)])[0] for b, n in enumerate(ns.tolist())]
            out, _, lens = collate_fn([(f, 0) for f in feats])
            return out, lens
        lib, ctx = N.lib(), N.ctx(wav.device)
        T = lib.vp_fbank_num_frames(C.byref(self._opts), L)
        if T <= 0:
            raise ValueError(f'{L} samples are shorter than one analysis window')
        F = self._opts.n_mels
        out = torch.empty((B, T, F), dtype=torch.float32, device=wav.device)
        out16 = torch.empty((B, T, F), dtype=torch.bfloat16, device=wav.device) if want_bf16 else None
        nf = torch.empty((B,), dtype=torch.int32, device=wav.device)
        ws = self._ws.get(lib.vp_fbank_workspace_bytes(C.byref(self._opts), B, L), wav.device)
        N.check(lib.vp_fbank_cmn_ragged_f32(ctx, N.ptr(wav), N.ptr(ns), B, L, C.byref(self._opts), N.ptr(out), N.ptr(out16), N.ptr(nf),
                                            N.ptr(ws), ws.numel(), N.stream_ptr()), ctx)
        if out16 is not None:
            out._vp_bf16 = out16
        return out, nf.to(torch.int64)

    @property
    def feature_dim(self):
        """返回特征大小"""
        if self._feature_method == 'LogMelSpectrogram':
            return self._method_args.get('n_mels', 128)
        elif self._feature_method == 'MelSpectrogram':
            return self._method_args.get('n_mels', 64)
        elif self._feature_method == 'Spectrogram':
            return self._method_args.get('n_fft', 512) // 2 + 1
        elif self._feature_method == 'MFCC':
            return self._method_args.get('n_mfcc', 40)
        elif self._feature_method == 'Fbank':
            return self._method_args.get('n_mels', 23)
        else:
            raise Exception('没有{}预处理方法'.format(self._feature_method))
