"""GPU parity tests, kernel level: each C-ABI entry point of libvpmi against the CPU oracle /
a float64 restatement of the same reference op, on seeded inputs.  Run with -m gpu on an MI355X.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fbank as ofb
from oracle import models as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from ppvector import _native as N
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: these tests must run on an MI355X (no CPU fallback exists)')
    N.ctx(0)
    return N


def dev(x, dtype=None):
    t = torch.as_tensor(np.asarray(x)) if not isinstance(x, torch.Tensor) else x
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


# --------------------------------------------------------------------------------------- Fbank
@pytest.mark.parametrize('B,L', [(1, 400), (3, 16000), (2, 48000), (5, 7919)])
def test_fbank_matches_oracle(N, B, L):
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(B, L, seed=11 + L, lowpass=0.8 if L % 2 else 0.0)
    ref = ofb.featurize(w, method_args=dict(sr=16000, n_mels=80))
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    out = fz(dev(w), want_bf16=True)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    # tolerance: f32 FFT/log round-off on log-mel energies (values span ~[-12, 8])
    assert np.max(np.abs(got - ref)) < 2e-3, np.max(np.abs(got - ref))
    assert np.mean(np.abs(got - ref)) < 2e-5
    twin = out._vp_bf16.float().cpu().numpy()
    assert np.max(np.abs(twin - got)) <= np.max(np.abs(got)) * 2 ** -8 + 1e-6


def test_fbank_pcm16_equals_float_path(N):
    """16-bit PCM input (vp_fbank_cmn_pcm16: samples widened x 1/32768 inside the frame kernel, what the reference's readers do on the
    host) against the float32 entry over the widened samples: bit for bit, incl. the length mask and the bf16 twin; and against the
    oracle on the widened samples.  An odd number of samples per row takes the widen-first route."""
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(3, 16000, seed=77)
    pcm = np.clip(np.round(w / np.abs(w).max() * 20000.0), -32768, 32767).astype(np.int16)
    wf = pcm.astype(np.float32) * np.float32(1.0 / 32768.0)
    ratio = np.asarray([1.0, 0.6, 0.31], np.float32)
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    a = fz(dev(torch.from_numpy(pcm)), dev(ratio), want_bf16=True)
    b = fz(dev(wf), dev(ratio), want_bf16=True)
    torch.cuda.synchronize()
    assert a.dtype == torch.float32 and torch.equal(a, b)
    assert torch.equal(a._vp_bf16, b._vp_bf16)
    ref = ofb.featurize(wf, ratio, method_args=dict(sr=16000, n_mels=80))
    assert np.max(np.abs(a.cpu().numpy() - ref)) < 2e-3
    odd = fz(dev(torch.from_numpy(pcm[:, :15999])))
    assert torch.equal(odd, fz(dev(wf[:, :15999])))


def test_fbank_mask_and_1d(N):
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(4, 12000, seed=5)
    ratio = np.asarray([1.0, 0.75, 0.5, 0.13], np.float32)
    ref = ofb.featurize(w, ratio, method_args=dict(sr=16000, n_mels=80))
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    got = fz(dev(w), dev(ratio)).cpu().numpy()
    assert np.max(np.abs(got - ref)) < 2e-3
    T = ref.shape[1]
    lens = (ratio * np.float32(T)).astype(np.int32)
    for b in range(4):
        assert np.all(got[b, lens[b]:] == 0)
    one = fz(dev(w[0])).cpu().numpy()
    assert one.shape == (1, T, 80)
    with pytest.raises(ValueError):
        fz(dev(w[:, :300]))
    f23 = AudioFeaturizer('Fbank', dict(sr=16000))
    assert f23.feature_dim == 23
    r23 = ofb.featurize(w[:1], method_args=dict(sr=16000))
    assert np.max(np.abs(f23(dev(w[:1])).cpu().numpy() - r23)) < 2e-3


def test_fbank_ragged_is_per_utterance_featurise_then_collate(N):
    """The training loader's semantics (reader.py:102-103 + collate_fn.py:5-23) in one batched launch: every utterance
    featurised alone (time mean over its own frames), features zero-padded to the longest."""
    from oracle import augment as oa
    from ppvector.data_utils.featurizer import AudioFeaturizer
    lens = [48000, 30000, 16001, 400, 47999]
    w = ofb.synth_waves(len(lens), 48000, seed=21)
    garbage = w.copy()
    for b, n in enumerate(lens):
        garbage[b, n:] = 7.0                                                   # samples past the utterance must not matter
    per = [ofb.featurize(w[b:b + 1, :n], method_args=dict(sr=16000, n_mels=80))[0] for b, n in enumerate(lens)]
    ref, _, ref_lens = oa.collate([(f, 0) for f in per])
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    got, got_lens = fz.forward_ragged(dev(garbage), torch.tensor(lens), want_bf16=True)
    assert got_lens.dtype == torch.int64 and np.array_equal(got_lens.cpu().numpy(), ref_lens)
    g = got.cpu().numpy()
    assert g.shape == ref.shape and np.max(np.abs(g - ref)) < 2e-3 and np.mean(np.abs(g - ref)) < 2e-5
    for b, n in enumerate(ref_lens):
        assert np.all(g[b, n:] == 0) and abs(float(g[b, :n].mean(axis=0).max())) < 1e-4       # own-frame CMN
    assert np.max(np.abs(got._vp_bf16.float().cpu().numpy() - g)) <= np.max(np.abs(g)) * 2 ** -8 + 1e-6
    # other feature methods take the per-utterance route with the same contract
    mz = AudioFeaturizer('MelSpectrogram', dict(sr=16000, n_fft=1024, hop_length=320, win_length=1024, n_mels=64, f_min=50.0))
    mg, ml = mz.forward_ragged(dev(w[:2]), torch.tensor([48000, 20000]))
    alone = mz(dev(w[1:2, :20000]))
    assert mg.shape[0] == 2 and int(ml[1]) == alone.shape[1] and torch.equal(mg[1, :alone.shape[1]], alone[0]) and float(mg[1, alone.shape[1]:].abs().max()) == 0.0


def test_log_mel_spectrogram_matches_restatement(N):
    """feature_method 'LogMelSpectrogram' (featurizer.py:20-21): the mel kernel with the power_to_db epilogue."""
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(3, 24000, seed=31, lowpass=0.5)
    w[2, 12000:] = 0.0                                                     # digital silence: the amin floor (-100 dB)
    for args in (dict(sr=16000, n_fft=1024, hop_length=320, win_length=1024, n_mels=64, f_min=50.0),
                 dict(sr=16000, hop_length=160, n_mels=80, ref_value=0.5, amin=1e-8)):      # n_fft defaults to 512 here
        ref = ofb.featurize_mel(w, method_args=args, log=True)
        fz = AudioFeaturizer('LogMelSpectrogram', args)
        got = fz(dev(w)).cpu().numpy()
        assert fz.feature_dim == args['n_mels'] and got.shape == ref.shape
        d = np.abs(got - ref)
        assert d.max() < 5e-2 and d.mean() < 1e-3, (d.max(), d.mean())    # dB of f32-FFT mel energies; the floor rows are exact
    with pytest.raises(NotImplementedError):
        AudioFeaturizer('LogMelSpectrogram', dict(top_db=80.0))
    assert AudioFeaturizer('LogMelSpectrogram', {}).feature_dim == 128        # the reference's default (featurizer.py:69-70)


def test_mfcc_matches_restatement(N):
    """feature_method 'MFCC' (featurizer.py:26-27): log-mel -> orthonormal DCT-II, then the featurizer's CMN and mask."""
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(3, 16000, seed=41, lowpass=0.6)
    ratio = np.asarray([1.0, 0.6, 0.31], np.float32)
    for args in (dict(sr=16000, n_mfcc=20, n_fft=512, hop_length=160, n_mels=40, f_min=20.0), dict(sr=16000)):
        ref = ofb.featurize_mel(w, ratio, method_args=args, log='mfcc')
        fz = AudioFeaturizer('MFCC', args)
        got = fz(dev(w), dev(ratio), want_bf16=True)
        g = got.cpu().numpy()
        assert fz.feature_dim == args.get('n_mfcc', 40) and g.shape == ref.shape
        d = np.abs(g - ref)
        assert d.max() < 0.2 and d.mean() < 5e-3, (d.max(), d.mean())       # sums of 40-64 dB values from f32-FFT mel energies
        assert np.max(np.abs(got._vp_bf16.float().cpu().numpy() - g)) <= np.max(np.abs(g)) * 2 ** -8 + 1e-6
    with pytest.raises(AssertionError):
        AudioFeaturizer('MFCC', dict(n_mfcc=80, n_mels=64))


def test_fbank_real_speech_golden(N, golden_dir):
    from ppvector.data_utils.featurizer import AudioFeaturizer
    g = np.load(f'{golden_dir}/wavs_3s.npz')
    wav = g['pcm'].astype(np.float32) / 32768.0
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    got = fz(dev(wav)).cpu().numpy()
    assert got.shape == (4, 298, 80)
    d = np.abs(got - g['feats'])
    assert d.max() < 5e-3 and d.mean() < 5e-5, (d.max(), d.mean())


# --------------------------------------------------------------------------------------- conv1d
def conv_ref(x, w, bias, kw, dil, pad_mode, stride=1):
    """x (B,T,Cin) f64, w (Cout, Cin, kw) -> (B,T_out,Cout) f64, the reference's conv semantics."""
    xt = x.transpose(1, 2)
    pad = dil * (kw - 1) // 2
    if pad_mode == 'reflect' and pad > 0:
        xt = F.pad(xt, (pad, pad), mode='reflect')
    elif pad_mode == 'zero' and pad > 0:
        xt = F.pad(xt, (pad, pad))
    y = F.conv1d(xt, w, bias, stride=stride, dilation=dil)
    return y.transpose(1, 2)


def run_conv(N, x, w, bias, kw, dil, pad_mode, dtype, out_dtype=None, relu=False, bn=None, act2=0, rowbias=None,
             add_in=None, want_sums=False, ysplit=0, T_out=None, amp=False, res=None):
    lib, ctx = N.lib(), N.ctx(0)
    B, T, Cin = x.shape
    Cout = w.shape[0]
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    odt = tdt if out_dtype is None else (torch.bfloat16 if out_dtype == 'bf16' else torch.float32)
    pm = {'none': N.VP_PAD_NONE, 'reflect': N.VP_PAD_REFLECT, 'zero': N.VP_PAD_ZERO}[pad_mode]
    if T_out is None:
        T_out = T if pad_mode != 'none' else T - dil * (kw - 1)
    xd = dev(x, tdt)
    wd = dev(w.permute(0, 2, 1).reshape(Cout, kw * Cin), tdt)
    y = torch.zeros((B, T_out, Cout), dtype=odt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in, d.dtype_out = N.dtype_id(tdt), N.dtype_id(odt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T_out, Cin, Cout, kw, dil, 1
    d.pad_mode, d.pad_left = pm, (0 if pad_mode == 'none' else dil * (kw - 1) // 2)
    d.x, d.ldx, d.xoff = xd.data_ptr(), Cin, 0
    d.mfma_bf16 = int(amp)
    keep = [xd, wd, y]
    y2 = None
    if ysplit:
        y2 = torch.zeros((B, T_out, 2 * ysplit), dtype=odt, device='cuda')
        keep.append(y2)
        d.y2, d.ldy2, d.y2off, d.ysplit = y2.data_ptr(), 2 * ysplit, ysplit, ysplit
    d.w = wd.data_ptr()
    if bias is not None:
        bd = dev(bias, torch.float32); keep.append(bd); d.bias = bd.data_ptr()
    if rowbias is not None:
        rd = dev(rowbias, torch.float32); keep.append(rd); d.rowbias = rd.data_ptr()
    d.act = N.VP_ACT_RELU if relu else N.VP_ACT_NONE
    if bn is not None:
        sc, sh = dev(bn[0], torch.float32), dev(bn[1], torch.float32)
        keep += [sc, sh]
        d.bn_scale, d.bn_shift = sc.data_ptr(), sh.data_ptr()
    d.act2 = act2
    d.y, d.ldy, d.yoff = y.data_ptr(), Cout, 0
    aux = None
    if add_in is not None:
        ad = dev(add_in, odt)
        aux = torch.zeros_like(y)
        keep += [ad, aux]
        d.add_in, d.ld_add, d.add_off = ad.data_ptr(), Cout, 0
        d.aux, d.ld_aux, d.aux_off = aux.data_ptr(), Cout, 0
    if res is not None:
        rs = dev(res, odt); keep.append(rs)
        d.res, d.ld_res, d.res_off = rs.data_ptr(), Cout, 0
    ps = pq = None
    if want_sums:
        tiles, nseg = lib.vp_conv1d_tiles_m(B, T_out), lib.vp_conv1d_nseg(T_out)
        ps = torch.full((tiles, nseg, Cout), float('nan'), dtype=torch.float32, device='cuda')
        pq = torch.full((tiles, nseg, Cout), float('nan'), dtype=torch.float32, device='cuda')
        d.psum, d.psumsq = ps.data_ptr(), pq.data_ptr()
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    if ysplit:
        return y, y2
    return y, aux, ps, pq


def q(t, dtype):
    """Quantise a float64 tensor the way the kernel's inputs are quantised."""
    return t.to(torch.bfloat16).double() if dtype == 'bf16' else t.float().double()


CONV_CASES = [
    # B, T, Cin, Cout, kw, dil, pad
    (2, 64, 80, 512, 5, 1, 'reflect'),      # ECAPA block0 geometry (K = 400: ragged last K stage)
    (3, 50, 64, 64, 3, 2, 'reflect'),       # Res2 conv, BN=64 tile, crosses utterances inside a tile
    (2, 77, 64, 64, 3, 4, 'reflect'),
    (1, 298, 512, 512, 1, 1, 'reflect'),    # 1x1
    (2, 40, 128, 192, 1, 1, 'reflect'),     # Cout not a multiple of the tile
    (2, 64, 80, 128, 5, 1, 'none'),         # TDNN un-padded
    (2, 60, 128, 64, 3, 3, 'none'),
    (2, 33, 64, 128, 3, 2, 'zero'),
]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv1d_plain(N, case, dtype):
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(1000 + CONV_CASES.index(case))
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    ref = conv_ref(q(x, dtype), q(w, dtype), bias, kw, dil, pad)
    y, _, _, _ = run_conv(N, x, w, bias, kw, dil, pad, dtype, out_dtype='f32' if dtype == 'f32' else 'f32')
    err = (y.double().cpu() - ref).abs().max().item()
    # same quantised inputs on both sides: only accumulation order / f32 accumulate differs
    assert err < 2e-4, err
    assert y.shape == ref.shape


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv1d_mixed_precision(N, case):
    """vp_conv1d_desc.mfma_bf16: f32 tensors, operands rounded to bf16 while staging, bf16 matrix cores, f32 accumulate and f32
    out (the training engine under enable_amp) == float64 conv of the bf16-rounded operands; and it is NOT the exact-f32 result."""
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(7 * kw + dil)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    ref = conv_ref(q(x, 'bf16'), q(w, 'bf16'), bias, kw, dil, pad)
    y, _, ps, pq = run_conv(N, x, w, bias, kw, dil, pad, 'f32', amp=True, want_sums=True)
    assert y.dtype == torch.float32
    assert (y.double().cpu() - ref).abs().max().item() < 2e-4
    exact = conv_ref(q(x, 'f32'), q(w, 'f32'), bias, kw, dil, pad)
    assert (y.double().cpu() - exact).abs().max().item() > 1e-3          # bf16 rounding of the operands is really there
    assert not torch.isnan(ps).any()


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv1d_split_precision(N, case):
    """vp_conv1d_desc.mfma_bf16 = 2 (the 'float32x3' engine): f32 tensors, each operand split into bf16 hi + lo while staging, three bf16
    MFMAs per k-step (hi*hi + hi*lo + lo*hi), f32 accumulate.  Against the float64 conv of the UNROUNDED f32 operands: ~2^-17 per
    product, i.e. two orders below a single bf16 pass and within a small factor of the exact-f32 matrix cores -- the precision that
    carries the reference's 1e-4 score tolerance (tests/test_gpu_models.py::test_score_parity_at_trained_weights)."""
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(11 * kw + dil)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    exact = conv_ref(q(x, 'f32'), q(w, 'f32'), bias, kw, dil, pad)
    y, _, ps, pq = run_conv(N, x, w, bias, kw, dil, pad, 'f32', amp=2, want_sums=True)
    assert y.dtype == torch.float32 and y.shape == exact.shape
    e3 = (y.double().cpu() - exact).abs().max().item()
    y1, _, _, _ = run_conv(N, x, w, bias, kw, dil, pad, 'f32', amp=1)
    e1 = (y1.double().cpu() - exact).abs().max().item()
    print(f'[conv x3 {case}] max |err| vs float64: split precision {e3:.2e}, single bf16 pass {e1:.2e}')
    assert e3 < 5e-5, e3                     # outputs are O(1): ~2^-17 per product, random signs over K terms
    assert e1 > 40 * e3                      # and it really is two orders better than one bf16 pass
    assert not torch.isnan(ps).any() and not torch.isnan(pq).any()


HL_CASES = [
    # name, B, T, Cin, Cout, kw, in_fmt, out_fmt, extras
    ('block0: f32 features in, hl32 out (128-wide kernel, taps)', 3, 298, 80, 512, 5, 'f32', 'hl', ()),
    ('asp tdnn: hl32 in / out, tanh + rowbias (128-wide kernel)', 3, 298, 1536, 128, 1, 'hl', 'hl', ('tanh', 'rowbias')),
    ('tdnn1: ring, hl32 in / out + pass-through chunk to y2', 16, 298, 512, 512, 1, 'hl', 'hl', ('ysplit',)),
    ('tdnn2: ring, hl32 in / out + fused time sums', 16, 298, 512, 512, 1, 'hl', 'hl', ('sums',)),
    ('mfa: ring, K = 1536, both sums, ragged last tile', 15, 298, 1536, 1536, 1, 'hl', 'hl', ('sums',)),
    ('ring, hl32 in, f32 out', 16, 298, 512, 256, 1, 'hl', 'f32', ()),
    ('ring, slow rows: residual + aux', 16, 298, 512, 256, 1, 'hl', 'hl', ('res', 'aux')),
]


@pytest.mark.parametrize('case', HL_CASES, ids=[c[0].split(':')[0] + f'_{i}' for i, c in enumerate(HL_CASES)])
def test_conv1d_hl32(N, case):
    """Split bf16 planes (VP_HL32) as a STORAGE format of the split-precision engine: the conv GEMMs of the ECAPA fast path read / write
    them (128-wide kernel: f32 -> hl32 and hl32 -> hl32; 128 x 256 LDS-DMA ring: hl32 -> hl32 / f32).  Against the float64 conv of the
    values the inputs actually carry (hi + lo), the output is held to split precision's error plus its own hl32 rounding (2^-17)."""
    from ppvector.models.utils import pack_hl32, unpack_hl32
    name, B, T, Cin, Cout, kw, fin, fout, extras = case
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(HL_CASES.index(case) + 50)
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(Cout, Cin, kw, generator=g) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    wp = w.permute(0, 2, 1).reshape(Cout, kw * Cin).contiguous()
    if fin == 'hl':
        xd, wd = pack_hl32(x).cuda(), pack_hl32(wp).cuda()
        xv, wv = unpack_hl32(xd).double().cpu(), unpack_hl32(wd).double().cpu().reshape(Cout, kw, Cin).permute(0, 2, 1)
    else:
        xd, wd = x.cuda(), wp.cuda()
        xv, wv = x.double(), w.double()
    y = torch.zeros(B, T, Cout, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = N.VP_HL32 if fin == 'hl' else N.VP_F32
    d.dtype_out = N.VP_HL32 if fout == 'hl' else N.VP_F32
    d.mfma_bf16 = 2
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, Cin, Cout, kw, 1, 1
    d.pad_mode, d.pad_left = N.VP_PAD_REFLECT, (kw - 1) // 2
    d.x, d.ldx, d.w, d.y, d.ldy = xd.data_ptr(), Cin, wd.data_ptr(), y.data_ptr(), Cout
    bd, scd, shd = bias.cuda(), sc.cuda(), sh.cuda()
    d.bias, d.act, d.bn_scale, d.bn_shift = bd.data_ptr(), N.VP_ACT_RELU, scd.data_ptr(), shd.data_ptr()
    z = conv_ref(xv, wv, bias.double(), kw, 1, 'reflect')
    keep = []
    if 'rowbias' in extras:
        rb = torch.randn(B, Cout, generator=g); rbd = rb.cuda(); keep.append(rbd)
        d.rowbias = rbd.data_ptr()
        z = z + rb.double()[:, None, :]
    ref = torch.relu(z) * sc.double() + sh.double()
    if 'res' in extras:
        res = torch.randn(B, T, Cout, generator=g)
        rsd = pack_hl32(res).cuda(); keep.append(rsd)
        d.res, d.ld_res = rsd.data_ptr(), Cout
        ref = ref + unpack_hl32(rsd).double().cpu()
    if 'tanh' in extras:
        d.act2 = N.VP_ACT_TANH
        ref = torch.tanh(ref)
    y2 = aux = ps = pq = None
    if 'ysplit' in extras:
        y2 = torch.zeros(B, T, 128, device='cuda'); d.y2, d.ldy2, d.y2off, d.ysplit = y2.data_ptr(), 128, 64, 64
    if 'aux' in extras:
        add = torch.randn(B, T, Cout, generator=g)
        addd = pack_hl32(add).cuda(); aux = torch.zeros(B, T, Cout, device='cuda'); keep.append(addd)
        d.add_in, d.ld_add, d.aux, d.ld_aux = addd.data_ptr(), Cout, aux.data_ptr(), Cout
    if 'sums' in extras:
        tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
        ps = torch.full((tiles, nseg, Cout), float('nan'), device='cuda'); pq = torch.full((tiles, nseg, Cout), float('nan'), device='cuda')
        d.psum, d.psumsq = ps.data_ptr(), pq.data_ptr()
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    got = (unpack_hl32(y) if fout == 'hl' else y).double().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f'[conv hl32] {name}: max |err| {err:.2e} (outputs up to {scale:.1f})')
    assert err < 6e-5 * max(1.0, scale), err
    if y2 is not None:
        got2 = unpack_hl32(y2).double().cpu()
        assert (got2[..., 64:128] - ref[..., :64]).abs().max().item() < 6e-5 * max(1.0, scale)
        assert (y2[..., :64] == 0).all()
    if aux is not None:
        assert (unpack_hl32(aux).double().cpu() - (ref + unpack_hl32(addd).double().cpu())).abs().max().item() < 1.2e-4 * max(1.0, scale)
    if ps is not None:
        ps, pq = ps.double().cpu(), pq.double().cpu()
        for b in range(B):
            s1 = torch.zeros(Cout, dtype=torch.float64); s2 = torch.zeros(Cout, dtype=torch.float64)
            for tm in range((b * T) // 128, ((b + 1) * T - 1) // 128 + 1):
                sg = b - (tm * 128) // T
                s1 += ps[tm, sg]; s2 += pq[tm, sg]
            dref = ref[b] - sh.double()
            assert (s1 - dref.sum(0)).abs().max().item() < 1e-4 * max(1.0, dref.sum(0).abs().max().item())
            assert (s2 - (dref ** 2).sum(0)).abs().max().item() < 1e-4 * (dref ** 2).sum(0).abs().max().item()


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv1d_full_epilogue(N, dtype):
    B, T, Cin, Cout, kw, dil = 3, 70, 64, 192, 3, 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    rowbias = torch.randn(B, Cout, generator=g, dtype=torch.float64)
    sc = torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cout, generator=g, dtype=torch.float64)
    add = torch.randn(B, T, Cout, generator=g, dtype=torch.float64)
    z = conv_ref(q(x, dtype), q(w, dtype), bias, kw, dil, 'reflect') + rowbias[:, None, :]
    ref = torch.relu(z) * sc + sh
    y, aux, ps, pq = run_conv(N, x, w, bias, kw, dil, 'reflect', dtype, relu=True, bn=(sc, sh), rowbias=rowbias,
                              add_in=add, want_sums=True)
    tol = 2e-4
    if dtype == 'bf16':                              # + half an ulp of the bf16 output
        tol = tol + 2.0 ** -8 * ref.abs().max().item()
    assert (y.double().cpu() - ref).abs().max().item() < tol
    addq = q(add, dtype)
    assert (aux.double().cpu() - (ref + addq)).abs().max().item() < 2 * tol
    # time sums of (y - shift): rebuild per utterance from the per-tile partials
    lib = N.lib()
    nseg = lib.vp_conv1d_nseg(T)
    ps, pq = ps.double().cpu(), pq.double().cpu()
    assert not torch.isnan(ps).any() and not torch.isnan(pq).any()
    for b in range(B):
        s1 = torch.zeros(Cout, dtype=torch.float64)
        s2 = torch.zeros(Cout, dtype=torch.float64)
        for tm in range((b * T) // 128, ((b + 1) * T - 1) // 128 + 1):
            sg = b - (tm * 128) // T
            s1 += ps[tm, sg]
            s2 += pq[tm, sg]
        dref = ref[b] - sh
        # the sums are taken from the f32 accumulators (before any bf16 rounding of y)
        assert (s1 - dref.sum(0)).abs().max().item() < 1e-4 * max(1.0, dref.sum(0).abs().max().item())
        assert (s2 - (dref ** 2).sum(0)).abs().max().item() < 1e-4 * (dref ** 2).sum(0).abs().max().item()
    # tanh second activation
    y2, _, _, _ = run_conv(N, x, w, bias, kw, dil, 'reflect', dtype, relu=True, bn=(sc, sh), act2=N.VP_ACT_TANH)
    assert (y2.double().cpu() - torch.tanh(torch.relu(z - rowbias[:, None, :]) * sc + sh)).abs().max().item() < tol


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv1d_split_destination(N, dtype):
    """tdnn1 also drops its first Res2 chunk (y_0 = x_0) into the concat buffer: columns [0, ysplit)
    are stored twice, the second time at a column offset of another tensor."""
    B, T, Cin, Cout = 2, 45, 128, 256
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 1, generator=g, dtype=torch.float64) / Cin ** 0.5
    ref = conv_ref(q(x, dtype), q(w, dtype), None, 1, 1, 'reflect')
    y, y2 = run_conv(N, x, w, None, 1, 1, 'reflect', dtype, out_dtype='f32', ysplit=64)
    assert (y.double().cpu() - ref).abs().max().item() < 2e-4
    assert torch.equal(y2[:, :, 64:], y[:, :, :64])
    assert torch.all(y2[:, :, :64] == 0)


@pytest.mark.parametrize('case', [
    # B, T, Cin, Cout, kw, dil, pad  -- M = B*T >= 16384 rows, Cin % 64 == 0, Cout >= 256: the 256 x 256 LDS-DMA kernel
    (70, 241, 128, 320, 1, 1, 'reflect'),    # ragged M (not a multiple of 128), ragged N, utterances straddle tiles
    (57, 298, 512, 512, 1, 1, 'reflect'),    # the SE-Res2 1x1 layers at the bench utterance length (8 K-steps; M % 128 != 0)
    (60, 200, 192, 256, 1, 1, 'reflect'),    # odd K-step count (3): the ring's padded step must contribute zeros
    (66, 250, 64, 256, 3, 2, 'reflect'),     # taps: one tap per 64-wide K step, reflect at both utterance ends
    (64, 260, 128, 256, 3, 3, 'none'),       # un-padded (TDNN)
    (65, 255, 64, 384, 5, 1, 'zero'),        # zero padding = out-of-range DMA offsets
    (70, 241, 80, 512, 5, 1, 'reflect'),     # ECAPA block0 geometry: Cin % 64 != 0, K-steps straddle taps, ragged K tail
    (66, 250, 72, 256, 3, 2, 'zero'),
])
@pytest.mark.parametrize('sched', [3, 4, 5, 6])  # 3 = two-stage role-split schedule, 4 = half-tile ring, 5 = ring + resident workgroups, 6 = 128 x 256 ring, two workgroups per CU
def test_conv1d_wide_tiles_bf16(N, case, sched):
    prev = N.lib().vp_conv256_select(sched)
    try:
        _wide_tiles_case(N, case)
    finally:
        N.lib().vp_conv256_select(prev)


def _wide_tiles_case(N, case):
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(77 + kw)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    rowbias = torch.randn(B, Cout, generator=g, dtype=torch.float64)
    sc = torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cout, generator=g, dtype=torch.float64)
    z = conv_ref(q(x, 'bf16'), q(w, 'bf16'), bias, kw, dil, pad) + rowbias[:, None, :]
    ref = torch.relu(z) * sc + sh
    To = ref.shape[1]
    add = torch.randn(B, To, Cout, generator=g, dtype=torch.float64)
    y, aux, ps, pq = run_conv(N, x, w, bias, kw, dil, pad, 'bf16', relu=True, bn=(sc, sh), rowbias=rowbias, add_in=add,
                              want_sums=True)
    tol = 2e-4 + 2.0 ** -8 * ref.abs().max().item()
    assert (y.double().cpu() - ref).abs().max().item() < tol
    assert (aux.double().cpu() - (ref + q(add, 'bf16'))).abs().max().item() < 2 * tol
    ps, pq = ps.double().cpu(), pq.double().cpu()
    assert not torch.isnan(ps).any() and not torch.isnan(pq).any()
    dref = ref - sh
    for b in (0, 1, B // 2, B - 1):
        s1 = torch.zeros(Cout, dtype=torch.float64)
        s2 = torch.zeros(Cout, dtype=torch.float64)
        for tm in range((b * To) // 128, ((b + 1) * To - 1) // 128 + 1):
            s1 += ps[tm, b - (tm * 128) // To]
            s2 += pq[tm, b - (tm * 128) // To]
        assert (s1 - dref[b].sum(0)).abs().max().item() < 1e-4 * max(1.0, dref[b].sum(0).abs().max().item())
        assert (s2 - (dref[b] ** 2).sum(0)).abs().max().item() < 1e-4 * (dref[b] ** 2).sum(0).abs().max().item()
    ya, yb = run_conv(N, x, w, None, kw, dil, pad, 'bf16', ysplit=64)
    assert torch.equal(yb[:, :, 64:], ya[:, :, :64])
    assert (ya.double().cpu() - (z - rowbias[:, None, :] - bias)).abs().max().item() < tol
    # per-channel terms only (the kernel's fast row loop) with the fused time sums
    ref2 = torch.relu(z - rowbias[:, None, :]) * sc + sh
    y, _, ps, pq = run_conv(N, x, w, bias, kw, dil, pad, 'bf16', relu=True, bn=(sc, sh), want_sums=True)
    assert (y.double().cpu() - ref2).abs().max().item() < tol
    ps, pq = ps.double().cpu(), pq.double().cpu()
    assert not torch.isnan(ps).any() and not torch.isnan(pq).any()
    dref = ref2 - sh
    for b in (0, 1, B // 2, B - 1):
        s1 = sum(ps[tm, b - (tm * 128) // To] for tm in range((b * To) // 128, ((b + 1) * To - 1) // 128 + 1))
        s2 = sum(pq[tm, b - (tm * 128) // To] for tm in range((b * To) // 128, ((b + 1) * To - 1) // 128 + 1))
        assert (s1 - dref[b].sum(0)).abs().max().item() < 1e-4 * max(1.0, dref[b].sum(0).abs().max().item())
        assert (s2 - (dref[b] ** 2).sum(0)).abs().max().item() < 1e-4 * (dref[b] ** 2).sum(0).abs().max().item()


@pytest.mark.parametrize('case', [(40, 149, 512, 512), (33, 298, 1536, 512), (35, 130, 256, 320)])
def test_conv1d_wide_bf16_operands_f32_output(N, case):
    """bf16 operands -> f32 output on the 128 x 256 ring kernel (the training engine's data-gradient GEMMs over bf16 dz, with the
    other path's gradient added in the epilogue): float64 conv of the same bf16 operands + the f32 residual; with and without
    the residual, bias and ReLU; ragged M and N."""
    B, T, Cin, Cout = case
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 1, generator=g, dtype=torch.float64) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    res = torch.randn(B, T, Cout, generator=g, dtype=torch.float64)
    z = conv_ref(q(x, 'bf16'), q(w, 'bf16'), None, 1, 1, 'zero')
    y, _, _, _ = run_conv(N, x, w, None, 1, 1, 'zero', 'bf16', out_dtype='f32', res=res)
    assert y.dtype == torch.float32
    e1 = (y.double().cpu() - (z + q(res, 'f32'))).abs().max().item()
    y, _, _, _ = run_conv(N, x, w, None, 1, 1, 'zero', 'bf16', out_dtype='f32')
    e2 = (y.double().cpu() - z).abs().max().item()
    y, _, _, _ = run_conv(N, x, w, bias, 1, 1, 'zero', 'bf16', out_dtype='f32', relu=True)
    e3 = (y.double().cpu() - torch.relu(z + bias)).abs().max().item()
    print(f'[conv bf16 -> f32, {B}x{T} rows, {Cin} -> {Cout}] max abs err: with residual {e1:.2e}, plain {e2:.2e}, bias + relu {e3:.2e}')
    assert max(e1, e2, e3) < 2e-4


def test_conv1d_rejects_bad_shapes(N):
    x = torch.randn(1, 8, 20, dtype=torch.float64)
    w = torch.randn(16, 20, 3, dtype=torch.float64)
    with pytest.raises(N.VpmiError):
        run_conv(N, x, w, None, 3, 1, 'reflect', 'bf16')            # Cin % 8 != 0
    x = torch.randn(1, 2, 16, dtype=torch.float64)
    w = torch.randn(16, 16, 3, dtype=torch.float64)
    with pytest.raises(N.VpmiError):
        run_conv(N, x, w, None, 3, 4, 'reflect', 'f32')             # reflect pad >= T
    # maximum sizes: the kernels address x through 32-bit buffer offsets.  A batch whose activations pass 4 GiB runs as batch slices
    # (test_conv1d_activations_past_4gib_run_as_batch_slices); ONE utterance past 4 GiB is refused before any launch
    lib, ctx = N.lib(), N.ctx(0)
    small = torch.zeros(64, dtype=torch.bfloat16, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_BF16
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = 1, 4200000, 4200000, 512, 512, 1, 1, 1
    d.pad_mode = N.VP_PAD_REFLECT
    d.x, d.ldx, d.w, d.y, d.ldy = small.data_ptr(), 512, small.data_ptr(), small.data_ptr(), 512
    rc = lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr())
    assert rc == N.VP_EUNSUP and b'4 GiB' in lib.vp_last_error(ctx)
    d.B = 0                                                          # empty batch
    assert lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()) == N.VP_EINVAL


def test_conv1d_activations_past_4gib_run_as_batch_slices(N):
    """BASELINE configs[4] at its per-GPU batch (ERes2Net-large, 128 utterances) has activation tensors past 4 GiB; the conv kernels
    address x through a 32-bit buffer offset, so vp_conv1d_fwd runs such a launch as consecutive batch slices (csrc/conv_gemm.hip) --
    with the fused time sums cut on M-tile boundaries.  Here: a 64-channel slice of a 4.4 GB f32 buffer (leading dimension 8192),
    136 utterances, against the same conv over the two halves launched separately and against float64 on a few utterances."""
    lib, ctx = N.lib(), N.ctx(0)
    B, T, Cin, Cout, ld = 136, 1024, 64, 64, 8192
    g = torch.Generator(device='cuda').manual_seed(9)
    xbig = torch.empty((B, T, ld), device='cuda')
    assert xbig.numel() * 4 > 2 ** 32
    xbig[:, :, 256:256 + Cin] = torch.randn((B, T, Cin), device='cuda', generator=g)
    w = torch.randn((Cout, Cin), device='cuda', generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, device='cuda', generator=g)

    def run(b0, nb, y, ps):
        d = N.Conv1dDesc()
        d.dtype_in = d.dtype_out = N.VP_F32
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = nb, T, T, Cin, Cout, 1, 1, 1
        d.pad_mode = N.VP_PAD_REFLECT
        d.x, d.ldx, d.xoff, d.w, d.bias = xbig[b0:].data_ptr(), ld, 256, w.data_ptr(), bias.data_ptr()
        d.y, d.ldy = y.data_ptr(), Cout
        d.psum = ps.data_ptr()
        N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)

    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    y = torch.zeros((B, T, Cout), device='cuda')
    ps = torch.zeros((tiles, nseg, Cout), device='cuda')
    run(0, B, y, ps)                                                   # 4.4 GB: sliced inside the library
    y2 = torch.zeros_like(y)
    ps2 = torch.zeros_like(ps)
    h = B // 2
    run(0, h, y2[:h], ps2[:h * T // 128])                              # the same in two launches, each under 4 GiB
    run(h, B - h, y2[h:], ps2[h * T // 128:])
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(ps, ps2)
    for b in (0, 67, 68, 135):                                         # utterances on both sides of the library's cut
        ref = xbig[b, :, 256:256 + Cin].double() @ w.double().t() + bias.double()
        assert (y[b].double() - ref).abs().max().item() < 1e-4


def test_empty_and_degenerate_inputs_are_refused(N):
    """Empty batches, utterances shorter than one analysis window and single-frame pooling are errors with a message,
    as in the reference (featurizer.py asserts / paddle shape errors) -- never a silent zero-size launch."""
    from ppvector.data_utils.featurizer import AudioFeaturizer
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))
    with pytest.raises((ValueError, N.VpmiError)):
        fz(torch.zeros((0, 16000), device='cuda'))
    with pytest.raises((ValueError, N.VpmiError)):
        fz(torch.zeros((2, 399), device='cuda'))                    # < one 25 ms window
    assert fz(torch.zeros((1, 400), device='cuda')).shape == (1, 1, 80)
    m = EcapaTdnn(80).cuda().eval()
    with pytest.raises((ValueError, N.VpmiError)):
        m(torch.zeros((0, 50, 80), device='cuda'))
    one = m(torch.randn((1, 30, 80), device='cuda'))                # B = 1 works
    assert one.shape == (1, 192) and torch.isfinite(one).all()


# --------------------------------------------------------------------------------------- small ops
@pytest.mark.parametrize('M,Nn,K,kn', [(256, 192, 3072, 0), (7, 128, 512, 0), (33, 50, 37, 0), (256, 2796, 192, 1),
                                      (5, 17, 19, 1)])
def test_dense_f32(N, M, Nn, K, kn):
    g = torch.Generator().manual_seed(M + Nn + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, Nn, generator=g) if kn else torch.randn(Nn, K, generator=g)
    bias = torch.randn(Nn, generator=g)
    ref = a.double() @ (w.double() if kn else w.double().t()) + bias.double()
    out = torch.empty(M, Nn, device='cuda')
    lib, ctx = N.lib(), N.ctx(0)
    ad, wd, bd = dev(a), dev(w), dev(bias)
    for act, fn in ((N.VP_ACT_NONE, lambda t: t), (N.VP_ACT_RELU, torch.relu), (N.VP_ACT_SIGMOID, torch.sigmoid)):
        N.check(lib.vp_dense_f32(ctx, ad.data_ptr(), K, wd.data_ptr(), kn, bd.data_ptr(), M, Nn, K, act,
                                 out.data_ptr(), Nn, N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        err = (out.double().cpu() - fn(ref)).abs().max().item()
        assert err < 1e-5 * max(1.0, K ** 0.5), (act, err)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_se_scale_residual_and_cast(N, dtype):
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    B, T, Cc = 3, 37, 128
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, Cc, generator=g).to(tdt)
    r = torch.randn(B, T, 2 * Cc, generator=g).to(tdt)
    s = torch.rand(B, Cc, generator=g)
    out = torch.zeros(B, T, 3 * Cc, dtype=tdt, device='cuda')
    lib, ctx = N.lib(), N.ctx(0)
    xd, rd, sd = dev(x), dev(r), dev(s)
    N.check(lib.vp_se_scale_residual(ctx, N.dtype_id(tdt), xd.data_ptr(), Cc, 0, sd.data_ptr(), rd.data_ptr(), 2 * Cc,
                                     Cc, out.data_ptr(), 3 * Cc, 2 * Cc, B, T, Cc, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    ref = (x.float() * s[:, None, :] + r[:, :, Cc:].float()).to(tdt)
    got = out.cpu()
    assert torch.equal(got[:, :, 2 * Cc:].float(), ref.float()) or \
        (got[:, :, 2 * Cc:].float() - ref.float()).abs().max().item() <= 2 ** -7 * ref.float().abs().max().item()
    assert torch.all(got[:, :, :2 * Cc] == 0)
    v = torch.randn(1001, generator=g)
    y = torch.empty(1001, dtype=torch.bfloat16, device='cuda')
    vd = dev(v)
    N.check(lib.vp_cast_f32_bf16(ctx, vd.data_ptr(), y.data_ptr(), 1001, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), v.to(torch.bfloat16))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_asp_softmax_stats(N, dtype):
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    B, T, Cc = 3, 61, 192
    g = torch.Generator().manual_seed(4)
    e = torch.randn(B, T, Cc, generator=g) * 3
    x = (torch.randn(B, T, Cc, generator=g) + 2.0).to(tdt)
    lib, ctx = N.lib(), N.ctx(0)
    pooled = torch.empty(B, 2 * Cc, device='cuda')
    ed, xd = dev(e), dev(x)
    N.check(lib.vp_asp_softmax_stats(ctx, N.dtype_id(tdt), ed.data_ptr(), xd.data_ptr(), Cc, 0, B, T, Cc, 1e-12,
                                     pooled.data_ptr(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    a = torch.softmax(e.double(), dim=1)
    xd64 = x.double()
    mean = (a * xd64).sum(1)
    std = torch.sqrt(((a * (xd64 - mean[:, None, :]) ** 2).sum(1)).clamp(min=1e-12))
    got = pooled.double().cpu()
    assert (got[:, :Cc] - mean).abs().max().item() < 1e-5
    assert (got[:, Cc:] - std).abs().max().item() < 1e-4


# --------------------------------------------------------------------------------------- head
def test_cosine_head_and_aam_golden(N, golden_dir):
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.loss.aamloss import AAMLoss
    g = np.load(f'{golden_dir}/ecapa_ref_small.npz')
    W = om.head_params(192, 2796, seed=int(g['head_seed']))
    head = SpeakerIdentification(input_dim=192, num_speakers=2796, classifier_type='Cosine')
    head.load_state_dict({'weight': W})
    head.cuda().eval()                  # eval: plain logits (in train mode they carry the autograd tape, tests/test_gpu_train.py)
    out = head(dev(g['emb_eval']))
    assert out['features'].shape == (2, 192)
    assert np.max(np.abs(out['logits'].cpu().numpy() - g['logits'])) < 1e-6
    labels = dev(g['labels'])
    for (margin, ls, easy), ref in zip(g['loss_cfg'], g['losses']):
        crit = AAMLoss(margin=0.2, scale=32, easy_margin=bool(easy), label_smoothing=float(ls))
        crit.update(margin=float(margin))
        loss = crit(out, labels)
        assert abs(float(loss) - ref) < 2e-5 * max(1.0, abs(ref)), (margin, ls, easy, float(loss), ref)


def test_aam_loss_random_batch_vs_oracle(N):
    from ppvector.loss.aamloss import AAMLoss
    g = torch.Generator().manual_seed(9)
    B, D, Cn = 64, 192, 2796
    emb = torch.randn(B, D, generator=g)
    W = om.head_params(D, Cn, seed=3)
    labels = torch.randint(0, Cn, (B,), generator=g)
    logits = om.cosine_head(emb, W)
    lib, ctx = N.lib(), N.ctx(0)
    ws = torch.empty(lib.vp_cosine_aam_workspace_bytes(B, D, Cn), dtype=torch.uint8, device='cuda')
    loss = torch.empty(1, device='cuda'); row = torch.empty(B, device='cuda'); lg = torch.empty(B, Cn, device='cuda')
    ed, Wd, ld = dev(emb), dev(W), dev(labels)
    for margin, ls in ((0.0, 0.0), (0.2, 0.0), (0.3, 0.1)):
        N.check(lib.vp_cosine_aam_ce_fwd(ctx, ed.data_ptr(), Wd.data_ptr(), ld.data_ptr(), B, D, Cn, margin, 32.0, ls, 0,
                                         loss.data_ptr(), lg.data_ptr(), row.data_ptr(), ws.data_ptr(), ws.numel(),
                                         N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        ref = om.aam_loss(logits.double(), labels, margin, 32.0, False, ls)
        assert abs(float(loss) - float(ref)) < 2e-5 * float(ref)
        assert (lg.cpu() - logits).abs().max().item() < 1e-6


def test_cosine_scores(N):
    from ppvector.metric.metrics import cosine_score_matrix
    from oracle import scoring as osc
    g = np.random.RandomState(0)
    a, b = g.standard_normal((37, 192)).astype(np.float32), g.standard_normal((9, 192)).astype(np.float32)
    got = cosine_score_matrix(dev(a), dev(b)).cpu().numpy()
    assert np.max(np.abs(got - osc.cosine_matrix(a.astype(np.float64), b.astype(np.float64)))) < 1e-6


def test_evaluate_trials_matches_reference_loop(N):
    """evaluate_trials vs the reference's per-trial loop restated in oracle/scoring.py (trainer.py:416-431, metrics.py:4-37)."""
    from oracle import scoring as osc
    from ppvector.metric.metrics import evaluate_trials
    g = np.random.RandomState(3)
    spk = g.standard_normal((6, 192))
    el = np.repeat(np.arange(6), 3)
    tl = g.randint(0, 6, 40)
    enroll = (spk[el] + 0.8 * g.standard_normal((len(el), 192))).astype(np.float32)
    trials = (spk[tl] + 0.8 * g.standard_normal((len(tl), 192))).astype(np.float32)
    sc, lab = osc.trial_scores(trials, tl, enroll, el)
    fnr, fpr = osc.fnr_fpr(sc, lab)[:2]
    ref_eer = osc.eer(fnr, fpr, sc)
    ref_eer = ref_eer[0] if isinstance(ref_eer, tuple) else ref_eer
    ref_dcf = osc.min_dcf(fnr, fpr)
    eer, dcf, thr = evaluate_trials(dev(enroll), el, dev(trials), tl)
    assert abs(eer - float(ref_eer)) < 1e-4 and abs(dcf - float(ref_dcf)) < 1e-4 and -1.0 < thr < 1.0


# --------------------------------------------------------------------------------------- conv extensions (CAM++)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('stride_f,kw', [(1, 9), (2, 9), (2, 1)])
def test_conv2d_resblock_epilogue(N, dtype, stride_f, kw):
    """2-D conv over (time, freq) on (B,T,F,C) tensors with BN, residual add and ReLU
    (BasicResBlock, campplus.py:238-243; shortcut 1x1 stride (2,1))."""
    B, T, Fq, Cin, Cout = 2, 21, 12, 32, 32
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g = torch.Generator().manual_seed(5 + stride_f + kw)
    x = torch.randn(B, T, Fq, Cin, generator=g, dtype=torch.float64)
    k = 3 if kw == 9 else 1
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / (Cin * k * k) ** 0.5     # (Cout,Cin,kF,kT)
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    sc = torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cout, generator=g, dtype=torch.float64)
    Fo = (Fq - 1) // stride_f + 1 if (kw == 9 or stride_f == 2) else Fq
    res = torch.randn(B, T, Fo, Cout, generator=g, dtype=torch.float64)
    xq, wq, rq = q(x, dtype), q(w, dtype), q(res, dtype)
    ref = F.conv2d(xq.permute(0, 3, 2, 1), wq, bias, stride=(stride_f, 1), padding=1 if kw == 9 else 0)   # (B,C,F,T)
    ref = torch.relu(ref.permute(0, 3, 2, 1) * sc + sh + rq)
    lib, ctx = N.lib(), N.ctx(0)
    xd = dev(x, tdt)
    wd = dev(w.permute(0, 3, 2, 1).reshape(Cout, kw * Cin), tdt)
    bd, scd, shd, rd = dev(bias, torch.float32), dev(sc, torch.float32), dev(sh, torch.float32), dev(res, tdt)
    y = torch.zeros((B, T, Fo, Cout), dtype=tdt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.dtype_id(tdt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, Cin, Cout, kw, 1, 1
    d.KF, d.F_in, d.F_out, d.stride_f = (3 if kw == 9 else 1), Fq, Fo, stride_f
    d.pad_mode, d.pad_left, d.pad_f = N.VP_PAD_ZERO, (1 if kw == 9 else 0), (1 if kw == 9 else 0)
    d.x, d.ldx, d.w, d.bias = xd.data_ptr(), Cin, wd.data_ptr(), bd.data_ptr()
    d.bn_scale, d.bn_shift, d.act2 = scd.data_ptr(), shd.data_ptr(), N.VP_ACT_RELU
    d.res, d.ld_res = rd.data_ptr(), Cout
    d.y, d.ldy = y.data_ptr(), Cout
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    tol = 2e-4 + (2.0 ** -8 * ref.abs().max().item() if dtype == 'bf16' else 0.0)
    assert (y.double().cpu() - ref).abs().max().item() < tol


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv1d_prologue_gate_stride(N, dtype):
    """CAM++ 1-D pieces: input BN+ReLU prologue on a column slice of a wide buffer (campplus.py:137-143),
    k3 dilated conv with the per-(utterance, 100-frame segment) gate (campplus.py:88-94), and the
    k5 stride-2 zero-padded TDNN (campplus.py:299-305)."""
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(12)
    B, T, Cw, Cin, Cout = 3, 131, 320, 192, 128
    x = torch.randn(B, T, Cw, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, generator=g, dtype=torch.float64) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    ps = torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5
    ph = torch.randn(Cin, generator=g, dtype=torch.float64)
    sc = torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cout, generator=g, dtype=torch.float64)
    xin = q(torch.relu(q(x[:, :, :Cin], dtype) * ps.float().double() + ph.float().double()), dtype)   # prologue output is re-quantised
    ref = torch.relu((xin @ q(w, dtype).t() + bias) * sc + sh)
    xd, wd = dev(x, tdt), dev(w, tdt)
    keep = [dev(v, torch.float32) for v in (bias, ps, ph, sc, sh)]
    y = torch.zeros((B, T, Cout), dtype=tdt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.dtype_id(tdt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride, d.pad_mode = B, T, T, Cin, Cout, 1, 1, 1, N.VP_PAD_ZERO
    d.x, d.ldx, d.w, d.bias = xd.data_ptr(), Cw, wd.data_ptr(), keep[0].data_ptr()
    d.pro_scale, d.pro_shift, d.bn_scale, d.bn_shift, d.act2 = keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), N.VP_ACT_RELU
    d.y, d.ldy = y.data_ptr(), Cout
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    tol = 3e-4 + (2.0 ** -7 * ref.abs().max().item() if dtype == 'bf16' else 0.0)
    assert (y.double().cpu() - ref).abs().max().item() < tol
    # gated k3 dilated conv, 32 output channels written into a column slice
    Cb, Cg, dil, seg = 128, 32, 2, 100
    nseg = (T + seg - 1) // seg
    h = torch.randn(B, T, Cb, generator=g, dtype=torch.float64)
    wl = torch.randn(Cg, Cb, 3, generator=g, dtype=torch.float64) / (3 * Cb) ** 0.5
    bl = torch.randn(Cg, generator=g, dtype=torch.float64)
    gate = torch.rand(B, nseg, Cg, generator=g, dtype=torch.float64)
    refc = conv_ref(q(h, dtype), q(wl, dtype), bl, 3, dil, 'zero')
    segidx = torch.arange(T) // seg
    refg = refc * gate.float().double()[:, segidx, :]
    hd, wld = dev(h, tdt), dev(wl.permute(0, 2, 1).reshape(Cg, 3 * Cb), tdt)
    bld, gd = dev(bl, torch.float32), dev(gate, torch.float32)
    out = torch.zeros((B, T, 96), dtype=tdt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.dtype_id(tdt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, Cb, Cg, 3, dil, 1
    d.pad_mode, d.pad_left = N.VP_PAD_ZERO, dil
    d.x, d.ldx, d.w, d.bias = hd.data_ptr(), Cb, wld.data_ptr(), bld.data_ptr()
    d.gate, d.gate_len, d.gate_nseg = gd.data_ptr(), seg, nseg
    d.y, d.ldy, d.yoff = out.data_ptr(), 96, 64
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    tol = 2e-4 + (2.0 ** -8 * refg.abs().max().item() if dtype == 'bf16' else 0.0)
    assert (out[:, :, 64:].double().cpu() - refg).abs().max().item() < tol
    assert torch.all(out[:, :, :64] == 0)
    # k5 stride-2 zero-pad-2
    Ci, Co = 64, 128
    xs = torch.randn(B, T, Ci, generator=g, dtype=torch.float64)
    ws5 = torch.randn(Co, Ci, 5, generator=g, dtype=torch.float64) / (5 * Ci) ** 0.5
    refs = F.conv1d(q(xs, dtype).transpose(1, 2), q(ws5, dtype), None, stride=2, padding=2).transpose(1, 2)
    Tn = (T - 1) // 2 + 1
    assert refs.shape[1] == Tn
    xsd, wsd = dev(xs, tdt), dev(ws5.permute(0, 2, 1).reshape(Co, 5 * Ci), tdt)
    ys = torch.zeros((B, Tn, Co), dtype=torch.float32, device='cuda') if dtype == 'f32' else torch.zeros((B, Tn, Co), dtype=tdt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.dtype_id(tdt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, Tn, Ci, Co, 5, 1, 2
    d.pad_mode, d.pad_left = N.VP_PAD_ZERO, 2
    d.x, d.ldx, d.w, d.y, d.ldy = xsd.data_ptr(), Ci, wsd.data_ptr(), ys.data_ptr(), Co
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    tol = 2e-4 + (2.0 ** -8 * refs.abs().max().item() if dtype == 'bf16' else 0.0)
    assert (ys.double().cpu() - refs).abs().max().item() < tol


# --------------------------------------------------------------------------------------- MelSpectrogram
@pytest.mark.parametrize('args,L', [
    (dict(sr=16000, n_fft=1024, hop_length=320, win_length=1024, n_mels=64, f_min=50.0), 48000),   # README.md:288-296
    (dict(sr=16000, n_fft=512, hop_length=160, win_length=400, n_mels=80, f_min=20.0), 16000),
    (dict(sr=22050, n_mels=64), 30000),                                                          # class defaults: n_fft 2048, hop 512
])
def test_melspectrogram_matches_oracle(N, args, L):
    from ppvector.data_utils.featurizer import AudioFeaturizer
    w = ofb.synth_waves(3, L, seed=L % 97, lowpass=0.9)
    ref = ofb.featurize_mel(w, method_args=args, dtype=np.float64)
    fz = AudioFeaturizer('MelSpectrogram', args)
    assert fz.feature_dim == args['n_mels']
    out = fz(dev(w), want_bf16=True)
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    # linear power features: tolerance relative to the utterance's dynamic range (f32 FFT round-off)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(got - ref)) < 2e-5 * scale, np.max(np.abs(got - ref)) / scale
    ratio = np.asarray([1.0, 0.6, 0.31], np.float32)
    refm = ofb.featurize_mel(w, ratio, method_args=args, dtype=np.float64)
    gotm = fz(dev(w), dev(ratio)).cpu().numpy()
    assert np.max(np.abs(gotm - refm)) < 2e-5 * scale
    with pytest.raises(ValueError):
        AudioFeaturizer('MelSpectrogram', dict(sr=16000, n_fft=1024, f_max=14000.0))   # above Nyquist (README's f_max)


# --------------------------------------------------------------------------------------- batch assembly
def test_spec_augment_matches_restatement(N):
    """Same Python `random` draws on both sides -> same masks; the kernel fills with the running mean (f32 sum order
    differs from NumPy's: tolerance 1e-5 on values of order 10)."""
    import random
    from oracle import augment as oa
    from ppvector.data_utils.spec_aug import SpecAugmentor
    conf = dict(prob=0.7, freq_mask_ratio=0.2, n_freq_masks=2, time_mask_ratio=0.1, n_time_masks=2, max_time_warp=0)
    rng = np.random.RandomState(4)
    x = (rng.standard_normal((9, 130, 80)) * 4 + 1).astype(np.float32)
    random.seed(123)
    ref = np.stack([oa.spec_augment(x[b], **conf) for b in range(x.shape[0])])
    random.seed(123)
    aug = SpecAugmentor(**conf)
    got = aug.batch(dev(x)).cpu().numpy()
    assert (ref != x).any() and np.max(np.abs(got - ref)) < 1e-5
    random.seed(5)
    refz = np.stack([oa.spec_augment(x[b], replace_with_zero=True, **conf) for b in range(x.shape[0])])
    random.seed(5)
    gotz = SpecAugmentor(replace_with_zero=True, **conf).batch(dev(x)).cpu().numpy()
    assert np.array_equal(gotz, refz)
    random.seed(9)                      # the reference's per-sample call form
    one_ref = oa.spec_augment(x[0], **conf)
    random.seed(9)
    one = SpecAugmentor(**conf)(dev(x[0])).cpu().numpy()
    assert np.max(np.abs(one - one_ref)) < 1e-5
    random.seed(9)
    one_np = SpecAugmentor(**conf)(x[0])                      # NumPy in -> NumPy out (reader.py:106)
    assert isinstance(one_np, np.ndarray) and np.max(np.abs(one_np - one_ref)) < 1e-5
    with pytest.raises(NotImplementedError):
        SpecAugmentor(max_time_warp=5)


def test_assemble_waves_normalises_crops_and_pads(N):
    """reader.py:97-101 + predict.py:246-254 as one launch: dB normalisation over the whole utterance, crop, zero pad."""
    from oracle import augment as oa
    from ppvector.data_utils.wave_batch import assemble_waves
    rng = np.random.RandomState(5)
    lens = (48000, 16001, 7, 80000, 31999)
    waves = [(rng.standard_normal(n) * s).astype(np.float32) for n, s in zip(lens, (0.1, 0.01, 0.5, 0.3, 1e-4))]
    waves.append(np.zeros(1000, np.float32))                                   # silence: gain stays 1, no NaN
    # predict_batch geometry: no crop, pad to the longest
    ref, nv = oa.wave_batch(waves, normalize=True, target_db=-20.0)
    out, ratio = assemble_waves([dev(w) for w in waves], use_dB_normalization=True, target_dB=-20.0)
    assert out.shape == ref.shape and np.max(np.abs(out.cpu().numpy() - ref)) < 5e-6 * np.max(np.abs(ref))
    assert np.allclose(ratio.cpu().numpy(), nv / ref.shape[1], atol=1e-7)
    rms_db = 10 * np.log10(np.mean(out[0, :48000].double().cpu().numpy() ** 2))
    assert abs(rms_db + 20.0) < 1e-3
    assert not torch.isnan(out).any() and float(out[5].abs().max()) == 0.0
    # training geometry: random crop starts, fixed 3 s rows; and the un-normalised volume-perturbation path
    starts = [0, 16001 - 5, 3, 80000 - 48000, 100, 0]
    ref, nv = oa.wave_batch(waves, L=48000, starts=starts, normalize=True, target_db=-23.5)
    out, ratio = assemble_waves([dev(w) for w in waves], max_len=48000, starts=starts, target_dB=-23.5)
    assert np.max(np.abs(out.cpu().numpy() - ref)) < 5e-6 * np.max(np.abs(ref)) and np.array_equal((ratio.cpu().numpy() * 48000).round().astype(np.int32), nv)
    gains = [-15.0, 0.0, 6.0, 15.0, -3.0, 1.0]
    ref, _ = oa.wave_batch(waves, L=20000, normalize=False, gains_db=gains)
    out, _ = assemble_waves([dev(w) for w in waves], max_len=20000, use_dB_normalization=False, gains_dB=gains)
    assert np.max(np.abs(out.cpu().numpy() - ref)) < 5e-6 * np.max(np.abs(ref))
    with pytest.raises(N.VpmiError):
        assemble_waves([torch.zeros(10)])


def test_speed_perturb_matches_linear_interpolation_restatement(N):
    """SpeedPerturbAugmentor -> AudioSegment.change_speed (reader.py:155-156): np.interp onto int(len / rate) points."""
    from oracle import augment as oa
    from ppvector.data_utils.wave_batch import speed_perturb
    rng = np.random.RandomState(9)
    waves = [rng.standard_normal(n).astype(np.float32) * 0.1 for n in (48000, 16001, 33333, 9, 70000)]
    rates = [0.9, 1.0, 1.1, 0.9, 1.1]
    out = speed_perturb([dev(w) for w in waves], rates)
    for w, r, o in zip(waves, rates, out):
        ref = oa.change_speed(w, r)
        assert o.shape == ref.shape and o.dtype == torch.float32
        assert np.max(np.abs(o.cpu().numpy() - ref)) < 1e-6
    assert out[1].data_ptr() != 0 and torch.equal(out[1].cpu(), torch.from_numpy(waves[1]))        # rate 1.0: untouched
    with pytest.raises(N.VpmiError):
        speed_perturb([torch.zeros(10)], [0.9])


def test_collate_fn_pads_like_reference(N):
    from oracle import augment as oa
    from ppvector.data_utils.collate_fn import collate_fn
    rng = np.random.RandomState(2)
    batch = [(rng.standard_normal((t, 80)).astype(np.float32), l) for t, l in ((298, 3), (120, 0), (1, 7), (297, 2))]
    rf, rl, rn = oa.collate(batch)
    f, l, n = collate_fn([(dev(x), lab) for x, lab in batch])
    assert f.dtype == torch.float32 and l.dtype == torch.int64 and n.dtype == torch.int64
    assert np.array_equal(f.cpu().numpy(), rf) and np.array_equal(l.cpu().numpy(), rl) and np.array_equal(n.cpu().numpy(), rn)


# ----------------------------------------------------------------- the three fused bf16 kernels of the ECAPA forward, one by one
def _bf(t):
    """Round a float64 tensor to bf16 (round-to-nearest-even, as the kernels store) and come back to float64."""
    return t.float().to(torch.bfloat16).double()


def _tdnn_layers(N, ws, bs, scs, shs, dil):
    arr = (N.TdnnLayer * len(ws))()
    keep = []
    for j, (w, b, sc, sh) in enumerate(zip(ws, bs, scs, shs)):
        wd = dev(w.permute(0, 2, 1).reshape(w.shape[0], -1), torch.bfloat16)       # [Cout][tap*Cin + c]
        bd, sd, hd = dev(b, torch.float32), dev(sc, torch.float32), dev(sh, torch.float32)
        keep += [wd, bd, sd, hd]
        arr[j].w, arr[j].bias, arr[j].bn_scale, arr[j].bn_shift = wd.data_ptr(), bd.data_ptr(), sd.data_ptr(), hd.data_ptr()
        arr[j].cin, arr[j].cout, arr[j].kw, arr[j].dil = w.shape[1], w.shape[0], w.shape[2], dil
    return arr, keep


@pytest.mark.parametrize('B,T,dil', [(3, 28, 2), (2, 298, 3), (5, 298, 4), (2, 384, 4), (4, 17, 2), (3, 33, 4)])
def test_res2_chain_kernel_vs_float64(N, B, T, dil):
    """vp_res2_chain_fwd against a float64 restatement of Res2NetBlock.forward (ecapa_tdnn.py:36-47) with the kernel's
    rounding points (bf16 inputs / weights, y_j stored as bf16, next input = bf16(y_j + x_{j+1}) from the unrounded y_j):
    every frame of every slice, so a wrong tap at one reflected boundary frame (t < dil, t >= T - dil) cannot hide."""
    lib, ctx = N.lib(), N.ctx(0)
    Cc, wdt, nconv = 512, 64, 7
    g = torch.Generator().manual_seed(100 * T + dil)
    t1 = _bf(torch.randn(B, T, Cc, generator=g, dtype=torch.float64))
    ws = [_bf(torch.randn(wdt, wdt, 3, generator=g, dtype=torch.float64) / (3 * wdt) ** 0.5) for _ in range(nconv)]
    bs = [0.1 * torch.randn(wdt, generator=g, dtype=torch.float64).float().double() for _ in range(nconv)]
    scs = [(torch.rand(wdt, generator=g, dtype=torch.float64) + 0.5).float().double() for _ in range(nconv)]
    shs = [0.2 * torch.randn(wdt, generator=g, dtype=torch.float64).float().double() for _ in range(nconv)]
    ref = torch.zeros(B, T, Cc, dtype=torch.float64)
    cur = t1[:, :, wdt:2 * wdt]
    for j in range(nconv):
        xt = F.pad(cur.transpose(1, 2), (dil, dil), mode='reflect')
        v = (torch.relu(F.conv1d(xt, ws[j], bs[j], dilation=dil)) * scs[j][None, :, None] + shs[j][None, :, None]).transpose(1, 2)
        ref[:, :, (j + 1) * wdt:(j + 2) * wdt] = _bf(v)
        if j + 1 < nconv:
            cur = _bf(v + t1[:, :, (j + 2) * wdt:(j + 3) * wdt])
    layers, keep = _tdnn_layers(N, ws, bs, scs, shs, dil)
    t1d = dev(t1, torch.bfloat16)
    r2 = torch.full((B, T, Cc), 7.0, dtype=torch.bfloat16, device='cuda')
    N.check(lib.vp_res2_chain_fwd(ctx, layers, nconv, t1d.data_ptr(), r2.data_ptr(), B, T, Cc, wdt, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    out = r2.double().cpu()
    assert torch.all(out[:, :, :wdt] == 7.0)                                  # slice 0 belongs to the producing conv
    err = (out[:, :, wdt:] - ref[:, :, wdt:]).abs()
    scale = ref.abs().max().item()
    # f32 vs float64 accumulation can flip a bf16 rounding (1 ulp = 2^-8 relative) and the flip feeds the next conv
    print(f'[res2_chain B={B} T={T} d={dil}] max err {err.max().item():.3e} (max |ref| {scale:.2f}), mean {err.mean().item():.2e}')
    assert err.max().item() < 2.0 ** -7 * scale, err.max().item()
    assert err.mean().item() < 2e-4 * scale
    edge = torch.cat([err[:, :2 * dil], err[:, -2 * dil:]], dim=1)            # the reflected frames, separately
    assert edge.max().item() < 2.0 ** -7 * scale


@pytest.mark.parametrize('B,T,dil', [(3, 28, 2), (2, 298, 2), (2, 298, 3), (5, 298, 4), (2, 400, 4), (4, 17, 2), (3, 33, 4), (2, 600, 3), (1, 2000, 4)])
def test_res2_chain_x3_kernel_vs_float64(N, B, T, dil):
    """vp_res2_chain_x3_fwd (split precision, tensors as split bf16 planes, the utterance cut into time segments with recomputed halos)
    against the float64 Res2NetBlock.forward (ecapa_tdnn.py:36-47) of the values the planes carry -- NO intermediate rounding in the
    reference: every frame of every slice incl. the reflected boundary frames and the frames on both sides of a segment cut (T = 298:
    two segments cut at 160; T = 400 / 600: three / four), held to split precision's own error."""
    from ppvector.models.utils import pack_hl32, unpack_hl32
    lib, ctx = N.lib(), N.ctx(0)
    Cc, wdt, nconv = 512, 64, 7
    g = torch.Generator().manual_seed(100 * T + dil)
    t1d = pack_hl32(torch.randn(B, T, Cc, generator=g)).cuda()
    t1 = unpack_hl32(t1d).double().cpu()
    ws = [torch.randn(wdt, wdt, 3, generator=g) / (3 * wdt) ** 0.5 for _ in range(nconv)]
    bs = [0.1 * torch.randn(wdt, generator=g, dtype=torch.float64).float().double() for _ in range(nconv)]
    scs = [(torch.rand(wdt, generator=g, dtype=torch.float64) + 0.5).float().double() for _ in range(nconv)]
    shs = [0.2 * torch.randn(wdt, generator=g, dtype=torch.float64).float().double() for _ in range(nconv)]
    layers, keep = _tdnn_layers(N, [w.double() for w in ws], bs, scs, shs, dil)
    wv = []
    for j, w in enumerate(ws):                                         # k = tap * 64 + channel, as split planes
        whl = pack_hl32(w.permute(0, 2, 1).reshape(wdt, 3 * wdt)).cuda()
        keep.append(whl)
        layers[j].w_hl = whl.data_ptr()
        wv.append(unpack_hl32(whl).double().cpu().reshape(wdt, 3, wdt).permute(0, 2, 1))
    ref = torch.zeros(B, T, Cc, dtype=torch.float64)
    cur = t1[:, :, wdt:2 * wdt]
    for j in range(nconv):
        xt = F.pad(cur.transpose(1, 2), (dil, dil), mode='reflect')
        v = (torch.relu(F.conv1d(xt, wv[j], bs[j], dilation=dil)) * scs[j][None, :, None] + shs[j][None, :, None]).transpose(1, 2)
        ref[:, :, (j + 1) * wdt:(j + 2) * wdt] = v
        if j + 1 < nconv:
            cur = v + t1[:, :, (j + 2) * wdt:(j + 3) * wdt]
    r2 = pack_hl32(torch.full((B, T, Cc), 7.0)).cuda()
    N.check(lib.vp_res2_chain_x3_fwd(ctx, layers, nconv, t1d.data_ptr(), r2.data_ptr(), B, T, Cc, wdt, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    out = unpack_hl32(r2).double().cpu()
    assert torch.all(out[:, :, :wdt] == 7.0)                                  # slice 0 belongs to the producing conv
    err = (out[:, :, wdt:] - ref[:, :, wdt:]).abs()
    scale = ref.abs().max().item()
    print(f'[res2_chain_x3 B={B} T={T} d={dil}] max err {err.max().item():.3e} (max |ref| {scale:.2f}), mean {err.mean().item():.2e}')
    assert err.max().item() < 1e-4 * max(1.0, scale), err.max().item()         # seven chained convs of ~2^-17 products + hl32 stores
    assert err.mean().item() < 1e-5 * max(1.0, scale)


def test_res2_chain_x3_refuses_shapes_it_does_not_cover(N):
    from ppvector.models.utils import pack_hl32
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(64, 64, 3, generator=g, dtype=torch.float64) for _ in range(7)]
    z = [torch.zeros(64, dtype=torch.float64) for _ in range(7)]
    layers, keep = _tdnn_layers(N, ws, z, z, z, 2)
    t1 = torch.zeros((1, 100, 512), device='cuda')
    r2 = torch.zeros_like(t1)
    assert lib.vp_res2_chain_x3_fwd(ctx, layers, 7, t1.data_ptr(), r2.data_ptr(), 1, 100, 512, 64, N.stream_ptr()) == N.VP_EUNSUP   # no split weights
    for j in range(7):
        whl = pack_hl32(ws[j].float().permute(0, 2, 1).reshape(64, 192)).cuda(); keep.append(whl)
        layers[j].w_hl = whl.data_ptr()
    assert lib.vp_res2_chain_x3_fwd(ctx, layers, 7, t1.data_ptr(), r2.data_ptr(), 1, 100, 512, 32, N.stream_ptr()) == N.VP_EUNSUP   # width != 64
    assert lib.vp_res2_chain_x3_fwd(ctx, layers, 7, t1.data_ptr(), r2.data_ptr(), 1, 100, 512, 64, N.stream_ptr()) == N.VP_OK


@pytest.mark.parametrize('B,T,Cc', [(3, 28, 1536), (2, 298, 1536), (3, 512, 512), (2, 37, 224)])
def test_asp_fused_x3_kernel_vs_float64(N, B, T, Cc):
    """vp_asp_fused_x3_fwd (split precision: h and x as split bf16 planes, f32 weights split in registers) against float64: logits GEMM
    (128 -> C) + softmax over time + weighted mean / std (pooling.py:105-123); T not a multiple of the 16-frame tile, C not a multiple of
    128.  Two orders tighter than the bf16 kernel's bound."""
    from ppvector.models.utils import pack_hl32, unpack_hl32
    lib, ctx = N.lib(), N.ctx(0)
    att = 128
    g = torch.Generator().manual_seed(T + Cc)
    hd = pack_hl32(torch.tanh(torch.randn(B, T, att, generator=g))).cuda()
    h = unpack_hl32(hd).double().cpu()
    w32 = torch.randn(Cc, att, generator=g) * (3.0 / att ** 0.5)                                # logits spread over several units
    bias = torch.randn(Cc, generator=g, dtype=torch.float64).float().double()
    xd = pack_hl32(torch.randn(B, T, Cc, generator=g) * 2 + 0.5 * torch.randn(1, 1, Cc, generator=g)).cuda()
    x = unpack_hl32(xd).double().cpu()
    e = h @ w32.double().t() + bias
    al = torch.softmax(e, dim=1)
    mu = (al * x).sum(1)
    sd = torch.sqrt(((al * (x - mu[:, None]) ** 2).sum(1)).clamp(min=1e-12))
    ref = torch.cat([mu, sd], 1)
    center = torch.cat([x.mean(1), x.std(1, unbiased=False)], 1)                                # the global-context stats buffer
    wd, bd, cd = w32.cuda(), dev(bias, torch.float32), dev(center, torch.float32)
    pooled = torch.full((B, 2 * Cc), float('nan'), dtype=torch.float32, device='cuda')
    N.check(lib.vp_asp_fused_x3_fwd(ctx, hd.data_ptr(), wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), Cc, cd.data_ptr(), 2 * Cc, B, T, Cc, att,
                                    1e-12, pooled.data_ptr(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    err = (pooled.double().cpu() - ref).abs()
    print(f'[asp_fused_x3 B={B} T={T} C={Cc}] max err {err.max().item():.3e} (max |ref| {ref.abs().max().item():.2f})')
    assert not torch.isnan(pooled).any()
    assert err.max().item() < 2e-5 * max(1.0, ref.abs().max().item()), err.max().item()


def test_res2_chain_refuses_shapes_it_does_not_cover(N):
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(64, 64, 3, generator=g, dtype=torch.float64) for _ in range(7)]
    z = [torch.zeros(64, dtype=torch.float64) for _ in range(7)]
    layers, keep = _tdnn_layers(N, ws, z, z, z, 2)
    t1 = torch.zeros((1, 512, 512), dtype=torch.bfloat16, device='cuda')
    r2 = torch.zeros_like(t1)
    assert lib.vp_res2_chain_fwd(ctx, layers, 7, t1.data_ptr(), r2.data_ptr(), 1, 512, 512, 64, N.stream_ptr()) == N.VP_EUNSUP   # T > 384
    assert lib.vp_res2_chain_fwd(ctx, layers, 7, t1.data_ptr(), r2.data_ptr(), 1, 100, 512, 32, N.stream_ptr()) == N.VP_EUNSUP   # width != 64


@pytest.mark.parametrize('B,T,Cc', [(3, 28, 1536), (2, 298, 1536), (3, 512, 512), (2, 37, 200)])
def test_asp_fused_kernel_vs_float64(N, B, T, Cc):
    """vp_asp_fused_fwd against float64: logits GEMM (128 -> C) + softmax over time + weighted mean / std (pooling.py:105-123),
    bf16 operands, every utterance and channel; T not a multiple of the kernel's 16-frame tile, C not a multiple of 128."""
    lib, ctx = N.lib(), N.ctx(0)
    att = 128
    g = torch.Generator().manual_seed(T + Cc)
    h = _bf(torch.tanh(torch.randn(B, T, att, generator=g, dtype=torch.float64)))
    w = _bf(torch.randn(Cc, att, generator=g, dtype=torch.float64) * (3.0 / att ** 0.5))       # logits spread over several units
    bias = torch.randn(Cc, generator=g, dtype=torch.float64).float().double()
    x = _bf(torch.randn(B, T, Cc, generator=g, dtype=torch.float64) * 2 + 0.5 * torch.randn(1, 1, Cc, generator=g, dtype=torch.float64))
    e = h @ w.t() + bias
    al = torch.softmax(e, dim=1)
    mu = (al * x).sum(1)
    sd = torch.sqrt(((al * (x - mu[:, None]) ** 2).sum(1)).clamp(min=1e-12))
    ref = torch.cat([mu, sd], 1)
    center = torch.cat([x.mean(1), x.std(1, unbiased=False)], 1)                                # the global-context stats buffer
    hd, wd, xd = dev(h, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
    bd, cd = dev(bias, torch.float32), dev(center, torch.float32)
    pooled = torch.full((B, 2 * Cc), float('nan'), dtype=torch.float32, device='cuda')
    N.check(lib.vp_asp_fused_fwd(ctx, hd.data_ptr(), wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), Cc, cd.data_ptr(), 2 * Cc, B, T, Cc, att,
                                 1e-12, pooled.data_ptr(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    err = (pooled.double().cpu() - ref).abs()
    print(f'[asp_fused B={B} T={T} C={Cc}] max err {err.max().item():.3e} (max |ref| {ref.abs().max().item():.2f})')
    assert not torch.isnan(pooled).any()
    assert err.max().item() < 2e-4 * max(1.0, ref.abs().max().item()), err.max().item()


@pytest.mark.parametrize('B,T,Cc', [(3, 298, 1536), (2, 64, 1536), (5, 241, 512), (2, 304, 192), (4, 17, 64)])
def test_asp_utt_kernel_vs_float64(N, B, T, Cc):
    """vp_asp_utt_fwd -- the whole AttentiveStatisticsPooling.forward (pooling.py:105-123) of the bf16 engine in one kernel per
    utterance -- against float64 over the same bf16 operands: attention TDNN (1x1 + per-utterance bias + ReLU + BN + tanh), logits,
    softmax over time, weighted mean / std.  Lengths that leave whole frame quarters empty (T = 64: three of four; T = 17), that end
    inside a quarter (241), the longest covered (304); channel counts of ECAPA (1536), TDNN-style (512) and narrow (192, 64).
    Also: twenty launches must be bit-identical (a first version with inline-asm d16 loads was not)."""
    lib, ctx = N.lib(), N.ctx(0)
    att = 128
    g = torch.Generator().manual_seed(T + Cc)
    x = _bf(torch.randn(B, T, Cc, generator=g, dtype=torch.float64) * 2 + 0.5 * torch.randn(1, 1, Cc, generator=g, dtype=torch.float64))
    wt = _bf(torch.randn(att, Cc, generator=g, dtype=torch.float64) / Cc ** 0.5)
    wc = _bf(torch.randn(Cc, att, generator=g, dtype=torch.float64) * (3.0 / att ** 0.5))
    bias = torch.randn(att, generator=g, dtype=torch.float64).float().double()
    sc = (torch.rand(att, generator=g, dtype=torch.float64) + 0.5).float().double()
    sh = (torch.randn(att, generator=g, dtype=torch.float64) * 0.1).float().double()
    rb = (torch.randn(B, att, generator=g, dtype=torch.float64) * 0.3).float().double()
    cb = torch.randn(Cc, generator=g, dtype=torch.float64).float().double()           # constant over time: must not matter
    h = _bf(torch.tanh(torch.relu(x @ wt.t() + bias + rb[:, None, :]) * sc + sh))      # the kernel keeps h as bf16
    e = h @ wc.t() + cb
    al = torch.softmax(e, dim=1)
    mu = (al * x).sum(1)
    sd = torch.sqrt(((al * x * x).sum(1) - mu * mu).clamp(min=1e-12))
    ref = torch.cat([mu, sd], 1)
    xd, wtd, wcd = dev(x.reshape(B * T, Cc), torch.bfloat16), dev(wt, torch.bfloat16), dev(wc, torch.bfloat16)
    bd, scd, shd, rbd, cbd = (dev(t, torch.float32) for t in (bias, sc, sh, rb, cb))
    L = N.TdnnLayer()
    L.w, L.bias, L.bn_scale, L.bn_shift, L.cin, L.cout, L.kw, L.dil = wtd.data_ptr(), bd.data_ptr(), scd.data_ptr(), shd.data_ptr(), Cc, att, 1, 1
    outs = []
    for _ in range(20):
        pooled = torch.full((B, 2 * Cc), float('nan'), dtype=torch.float32, device='cuda')
        N.check(lib.vp_asp_utt_fwd(ctx, xd.data_ptr(), Cc, C.byref(L), rbd.data_ptr(), wcd.data_ptr(), cbd.data_ptr(), B, T, Cc, att, 1e-12,
                                   pooled.data_ptr(), N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        outs.append(pooled)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    assert not torch.isnan(outs[0]).any()
    err = (outs[0].double().cpu() - ref).abs()
    # h is rounded to bf16 between the two GEMMs (as the two-launch path stores it): a logit moves by ~|w_c| 2^-9 sqrt(att)
    print(f'[asp_utt B={B} T={T} C={Cc}] max err mean {err[:, :Cc].max().item():.3e} std {err[:, Cc:].max().item():.3e} (max |ref| {ref.abs().max().item():.2f})')
    # measured (round 4, MI355X): <= 1.3e-4 over the five shapes, for |ref| ~ 2.5-3.5; the bound is ~2x that
    assert err.max().item() < 3e-4, err.max().item()


def test_asp_fused_softmax_survives_a_spike(N):
    """One frame whose logit dominates (the online softmax's new-maximum path): weights collapse on it, mean = its x, std = floor."""
    lib, ctx = N.lib(), N.ctx(0)
    B, T, Cc, att = 1, 100, 128, 128
    h = torch.zeros(B, T, att, dtype=torch.float64)
    h[0, 61] = 1.0                                                  # frame 61: logit 128 * 0.5 = 64 above the rest
    w = torch.full((Cc, att), 0.5, dtype=torch.float64)
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(B, T, Cc, generator=g, dtype=torch.float64))
    center = torch.cat([x.mean(1), x.std(1, unbiased=False)], 1)
    hd, wd, xd = dev(h, torch.bfloat16), dev(w, torch.bfloat16), dev(x, torch.bfloat16)
    bd, cd = torch.zeros(Cc, device='cuda'), dev(center, torch.float32)
    pooled = torch.empty((B, 2 * Cc), dtype=torch.float32, device='cuda')
    N.check(lib.vp_asp_fused_fwd(ctx, hd.data_ptr(), wd.data_ptr(), bd.data_ptr(), xd.data_ptr(), Cc, cd.data_ptr(), 2 * Cc, B, T, Cc, att,
                                 1e-12, pooled.data_ptr(), N.stream_ptr()), ctx)
    p = pooled.double().cpu()
    assert (p[0, :Cc] - x[0, 61]).abs().max().item() < 1e-5
    assert p[0, Cc:].max().item() < 1e-3


@pytest.mark.parametrize('B,T', [(5, 28), (3, 298), (2, 512), (7, 130)])
def test_se_gate_kernel_vs_float64(N, B, T):
    """vp_se_gate_fwd from the producing conv's partial time sums (128-row tiles straddling utterances) against float64."""
    lib, ctx = N.lib(), N.ctx(0)
    Cc, H = 512, 128
    g = torch.Generator().manual_seed(B * 1000 + T)
    y = torch.randn(B, T, Cc, generator=g, dtype=torch.float64)
    shift = 0.3 * torch.randn(Cc, generator=g, dtype=torch.float64).float().double()
    w1 = (torch.randn(Cc, H, generator=g, dtype=torch.float64) / Cc ** 0.5).float().double()
    b1 = 0.1 * torch.randn(H, generator=g, dtype=torch.float64).float().double()
    w2 = (torch.randn(H, Cc, generator=g, dtype=torch.float64) / H ** 0.5).float().double()
    b2 = 0.1 * torch.randn(Cc, generator=g, dtype=torch.float64).float().double()
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    ps = torch.full((tiles, nseg, Cc), float('nan'), dtype=torch.float64)
    rows = (y - shift).reshape(B * T, Cc)
    for tm in range(tiles):
        b0 = (tm * 128) // T
        for r0 in range(tm * 128, min((tm + 1) * 128, B * T)):
            seg = r0 // T - b0
            ps[tm, seg] = torch.where(torch.isnan(ps[tm, seg]), torch.zeros_like(ps[tm, seg]), ps[tm, seg]) + rows[r0]
    mean = y.mean(1)
    ref = torch.sigmoid(torch.relu(mean @ w1 + b1) @ w2 + b2)
    out = torch.empty((B, Cc), dtype=torch.float32, device='cuda')
    psd = dev(ps, torch.float32)                       # slots no tile row touched stay NaN: the kernel must not read them
    keep = [dev(t, torch.float32) for t in (shift, w1, b1, w2, b2)]
    N.check(lib.vp_se_gate_fwd(ctx, psd.data_ptr(), keep[0].data_ptr(), B, T, Cc, H, keep[1].data_ptr(), keep[2].data_ptr(),
                               keep[3].data_ptr(), keep[4].data_ptr(), out.data_ptr(), N.stream_ptr()), ctx)
    err = (out.double().cpu() - ref).abs().max().item()
    print(f'[se_gate B={B} T={T}] max err {err:.3e}')
    assert err < 2e-6, err


# --------------------------------------------------------------------------------------- CAM++ fused kernels (round 2)
def _layer(N, w, b, sc, sh, dil=1, keep=None):
    """vp_tdnn_layer over a (Cout, Cin, K) float64 weight: [Cout][tap * Cin + c] bf16, f32 epilogue terms (None = absent)."""
    L = N.TdnnLayer()
    wd = dev(w.permute(0, 2, 1).reshape(w.shape[0], -1), torch.bfloat16)
    keep.append(wd)
    L.w = wd.data_ptr()
    for name, v in (('bias', b), ('bn_scale', sc), ('bn_shift', sh)):
        if v is not None:
            d_ = dev(v, torch.float32)
            keep.append(d_)
            setattr(L, name, d_.data_ptr())
    L.cin, L.cout, L.kw, L.dil = w.shape[1], w.shape[0], w.shape[2], dil
    return L


@pytest.mark.parametrize('B,Tn,ch0,nl,seg', [(3, 149, 128, 3, 100), (2, 37, 256, 2, 100), (2, 160, 192, 2, 50), (1, 16, 128, 1, 100)])
def test_cam_block_kernel_vs_float64(N, B, Tn, ch0, nl, seg):
    """vp_cam_block_fwd against a float64 restatement of CAMDenseTDNNBlock.forward (campplus.py:145-173; CAMDenseTDNNLayer :109-142,
    CAMLayer :67-106) with the kernel's rounding points: bf16 concat buffer and weights, the BN1 + ReLU input staged as bf16, h stored
    as bf16 (the context is taken over that bf16 h), outputs stored as bf16.  Every frame and channel of every layer: two context
    segments (T' = 149, seg 100), four (T' = 160, seg 50), an utterance shorter than a segment, a single 16-frame tile."""
    lib, ctx = N.lib(), N.ctx(0)
    bnc, gr, H = 128, 32, 64
    g = torch.Generator().manual_seed(Tn * 7 + ch0)
    ld = ch0 + nl * gr + 32                                              # a concat buffer wider than the block needs
    cat = torch.zeros(B, Tn, ld, dtype=torch.float64)
    cat[:, :, :ch0] = _bf(torch.randn(B, Tn, ch0, generator=g, dtype=torch.float64))
    ref = cat.clone()
    layers = (N.CamLayer * nl)()
    keep = []
    f32 = lambda t: t.float().double()                                  # noqa: E731 -- parameters the kernel holds in f32
    ch = ch0
    for l in range(nl):
        dil = 1 + l                                                      # the reference's blocks use dilation 1 / 2: cover both and 3
        s1, h1 = f32(torch.rand(ch, generator=g, dtype=torch.float64) + 0.5), f32(0.3 * torch.randn(ch, generator=g, dtype=torch.float64))
        w1 = _bf(torch.randn(bnc, ch, 1, generator=g, dtype=torch.float64) / ch ** 0.5)
        b1 = f32(0.1 * torch.randn(bnc, generator=g, dtype=torch.float64))
        s2, h2 = f32(torch.rand(bnc, generator=g, dtype=torch.float64) + 0.5), f32(0.2 * torch.randn(bnc, generator=g, dtype=torch.float64))
        wl = _bf(torch.randn(gr, bnc, 3, generator=g, dtype=torch.float64) / (3 * bnc) ** 0.5)
        bl = f32(0.1 * torch.randn(gr, generator=g, dtype=torch.float64))
        cw1, cb1 = f32(torch.randn(bnc, H, generator=g, dtype=torch.float64) / bnc ** 0.5), f32(0.1 * torch.randn(H, generator=g, dtype=torch.float64))
        cw2, cb2 = f32(torch.randn(H, gr, generator=g, dtype=torch.float64) / H ** 0.5), f32(0.1 * torch.randn(gr, generator=g, dtype=torch.float64))
        # float64 reference with the kernel's rounding points
        xin = _bf(torch.relu(ref[:, :, :ch] * s1 + h1))
        hmid = _bf(torch.relu((xin @ w1[:, :, 0].t() + b1) * s2 + h2))                          # (B, Tn, 128)
        nseg = (Tn + seg - 1) // seg
        segmean = torch.stack([hmid[:, s * seg:min((s + 1) * seg, Tn)].mean(1) for s in range(nseg)], 1)      # (B, nseg, 128)
        ctxv = hmid.mean(1, keepdim=True) + segmean
        gate = torch.sigmoid(torch.relu(ctxv @ cw1 + cb1) @ cw2 + cb2)                           # (B, nseg, 32)
        loc = F.conv1d(hmid.transpose(1, 2), wl, bl, padding=dil, dilation=dil).transpose(1, 2)  # zero 'same' padding
        segidx = torch.arange(Tn) // seg
        ref[:, :, ch:ch + gr] = _bf(loc * gate[:, segidx])
        # device-side layer
        L = layers[l]
        for name, v in (('bn1_scale', s1), ('bn1_shift', h1), ('ctx_w1', cw1), ('ctx_b1', cb1), ('ctx_w2', cw2), ('ctx_b2', cb2)):
            d_ = dev(v, torch.float32)
            keep.append(d_)
            setattr(L, name, d_.data_ptr())
        L.linear1 = _layer(N, w1, b1, s2, h2, 1, keep)
        L.local = _layer(N, wl, bl, None, None, dil, keep)
        ch += gr
    catd = dev(cat, torch.bfloat16)
    N.check(lib.vp_cam_block_fwd(ctx, layers, nl, catd.data_ptr(), ld, ch0, B, Tn, seg, bnc, gr, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    out = catd.double().cpu()
    assert torch.equal(out[:, :, :ch0], cat[:, :, :ch0])                 # the block input is read-only
    assert torch.all(out[:, :, ch:] == 0)                                # columns past the block are untouched
    err = (out[:, :, ch0:ch] - ref[:, :, ch0:ch]).abs()
    scale = ref[:, :, ch0:ch].abs().max().item()
    print(f'[cam_block B={B} Tn={Tn} ch0={ch0} layers={nl} seg={seg}] max err {err.max().item():.3e} (max |ref| {scale:.2f}), mean {err.mean().item():.2e}')
    # f32 vs float64 accumulation can flip a bf16 rounding of h (1 ulp = 2^-8 relative), which the next layers read
    assert err.max().item() < 2.0 ** -6 * scale, err.max().item()
    assert err.mean().item() < 4e-4 * scale


def test_cam_block_refuses_shapes_it_does_not_cover(N):
    lib, ctx = N.lib(), N.ctx(0)
    keep = []
    layers = (N.CamLayer * 1)()
    z = torch.zeros(128, dtype=torch.float64)
    layers[0].linear1 = _layer(N, torch.zeros(128, 128, 1, dtype=torch.float64), z, z, z, 1, keep)
    layers[0].local = _layer(N, torch.zeros(32, 128, 3, dtype=torch.float64), z[:32], None, None, 1, keep)
    cat = torch.zeros((1, 200, 256), dtype=torch.bfloat16, device='cuda')
    assert lib.vp_cam_block_fwd(ctx, layers, 1, cat.data_ptr(), 256, 128, 1, 200, 100, 128, 32, N.stream_ptr()) == N.VP_EUNSUP   # T' > 160
    assert lib.vp_cam_block_fwd(ctx, layers, 1, cat.data_ptr(), 256, 128, 1, 100, 100, 64, 32, N.stream_ptr()) == N.VP_EUNSUP    # bottleneck != 128


@pytest.mark.parametrize('B,T,F_in,stride,mode', [(2, 37, 40, 1, 'res'), (3, 20, 80, 2, 'shortcut'), (2, 9, 20, 2, 'plain'),
                                                  (1, 298, 10, 1, 'res'), (2, 23, 80, 2, 'fused_c1'), (1, 5, 11, 2, 'shortcut')])
def test_conv3x3_c32_kernel_vs_float64(N, B, T, F_in, stride, mode):
    """vp_conv3x3_c32_fwd against float64 conv2d over the same bf16 operands (BasicResBlock / FCM, campplus.py:211-281): stride 1 with
    the residual, stride 2 with the fused 1x1 shortcut, the plain stride-2 FCM.conv2, and the variant whose input map is FCM.conv1
    (1 -> 32, BN, ReLU, rounded to bf16 as the unfused path stores it) produced inside the kernel.  Every position, incl. the map's
    border rows / columns (zero padding), an odd F and more time rows than one workgroup's slab."""
    lib, ctx = N.lib(), N.ctx(0)
    Cc = 32
    g = torch.Generator().manual_seed(T * 100 + F_in + stride)
    keep = []
    f32 = lambda t: t.float().double()                                  # noqa: E731
    w = _bf(torch.randn(Cc, Cc, 3, 3, generator=g, dtype=torch.float64) / (9 * Cc) ** 0.5)     # (out, in, kt, kf)
    bias = f32(0.1 * torch.randn(Cc, generator=g, dtype=torch.float64))
    sc, sh = f32(torch.rand(Cc, generator=g, dtype=torch.float64) + 0.5), f32(0.2 * torch.randn(Cc, generator=g, dtype=torch.float64))
    F_out = (F_in - 1) // stride + 1
    feats = c1 = None
    if mode == 'fused_c1':
        feats = _bf(torch.randn(B, T, F_in, generator=g, dtype=torch.float64) * 3)
        c1w = f32(torch.randn(Cc, 1, 3, 3, generator=g, dtype=torch.float64) / 3)
        c1b, c1s, c1h = (f32(0.1 * torch.randn(Cc, generator=g, dtype=torch.float64)), f32(torch.rand(Cc, generator=g, dtype=torch.float64) + 0.5),
                         f32(0.2 * torch.randn(Cc, generator=g, dtype=torch.float64)))
        y1 = F.conv2d(feats[:, None], c1w, c1b, padding=1)               # (B, 32, T, F)
        x = _bf(torch.relu(y1 * c1s[None, :, None, None] + c1h[None, :, None, None])).permute(0, 2, 3, 1)     # (B, T, F, 32)
        c1 = [dev(c1w.reshape(Cc, 9), torch.float32), dev(c1b, torch.float32), dev(c1s, torch.float32), dev(c1h, torch.float32)]
    else:
        x = _bf(torch.randn(B, T, F_in, Cc, generator=g, dtype=torch.float64))
    conv = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=(1, stride), padding=1)              # (B, 32, T, F_out)
    y = conv * sc[None, :, None, None] + sh[None, :, None, None]
    res = None
    if mode == 'res':
        res = _bf(torch.randn(B, T, F_out, Cc, generator=g, dtype=torch.float64))
        y = y + res.permute(0, 3, 1, 2)
    ref = _bf(torch.relu(y)).permute(0, 2, 3, 1)
    # time-major taps: k = (kt * 3 + kf) * 32 + c
    L = N.TdnnLayer()
    wd = dev(w.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc), torch.bfloat16)
    bd, sd, hd = dev(bias, torch.float32), dev(sc, torch.float32), dev(sh, torch.float32)
    keep += [wd, bd, sd, hd]
    L.w, L.bias, L.bn_scale, L.bn_shift, L.cin, L.cout, L.kw, L.dil = wd.data_ptr(), bd.data_ptr(), sd.data_ptr(), hd.data_ptr(), Cc, Cc, 9, 1
    S = None
    ref2 = None
    if mode in ('shortcut', 'fused_c1'):
        w2 = _bf(torch.randn(Cc, Cc, 1, 1, generator=g, dtype=torch.float64) / Cc ** 0.5)
        b2 = f32(0.1 * torch.randn(Cc, generator=g, dtype=torch.float64))
        s2, h2 = f32(torch.rand(Cc, generator=g, dtype=torch.float64) + 0.5), f32(0.2 * torch.randn(Cc, generator=g, dtype=torch.float64))
        y2 = F.conv2d(x.permute(0, 3, 1, 2), w2, b2, stride=(1, stride))
        ref2 = _bf(y2 * s2[None, :, None, None] + h2[None, :, None, None]).permute(0, 2, 3, 1)
        S = N.TdnnLayer()
        w2d, b2d, s2d, h2d = dev(w2.reshape(Cc, Cc), torch.bfloat16), dev(b2, torch.float32), dev(s2, torch.float32), dev(h2, torch.float32)
        keep += [w2d, b2d, s2d, h2d]
        S.w, S.bias, S.bn_scale, S.bn_shift, S.cin, S.cout, S.kw, S.dil = w2d.data_ptr(), b2d.data_ptr(), s2d.data_ptr(), h2d.data_ptr(), Cc, Cc, 1, 1
    xd = dev(x, torch.bfloat16)
    yd = torch.full((B, T, F_out, Cc), 7.0, dtype=torch.bfloat16, device='cuda')
    y2d = torch.full((B, T, F_out, Cc), 7.0, dtype=torch.bfloat16, device='cuda')
    resd = dev(res, torch.bfloat16) if res is not None else None
    fd = dev(feats, torch.bfloat16) if feats is not None else None
    N.check(lib.vp_conv3x3_c32_fwd(ctx, None if fd is not None else xd.data_ptr(), yd.data_ptr(), C.byref(L), resd.data_ptr() if resd is not None else None, 1,
                                   C.byref(S) if S is not None else None, y2d.data_ptr() if S is not None else None, B, T, F_in, stride,
                                   fd.data_ptr() if fd is not None else None, *( [t.data_ptr() for t in c1] if c1 else [None] * 4), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    out = yd.double().cpu()
    scale = ref.abs().max().item()
    err = (out - ref).abs()
    print(f'[conv3x3_c32 B={B} T={T} F={F_in} s={stride} {mode}] max err {err.max().item():.3e} (max |ref| {scale:.2f})')
    # f32 accumulation of bf16 products vs float64: below half a bf16 ulp of the output plus accumulation noise
    assert err.max().item() < 2e-4 * scale + 2.0 ** -8 * scale, err.max().item()
    assert err.mean().item() < 2e-4 * scale
    if ref2 is not None:
        e2 = (y2d.double().cpu() - ref2).abs()
        assert e2.max().item() < 2e-4 * ref2.abs().max().item() + 2.0 ** -8 * ref2.abs().max().item(), e2.max().item()
    else:
        assert torch.all(y2d == 7.0)


@pytest.mark.parametrize('K,Nn,T,relu2,psum', [(32, 128, 23840, False, True), (128, 32, 5000, True, False), (64, 64, 300, True, True),
                                               (32, 32, 40000, False, True), (64, 128, 777, False, True), (32, 64, 5960, 'res_hardtanh', False),
                                               (64, 256, 5960, False, True), (256, 64, 5960, True, False), (128, 512, 1490, False, True)])
def test_pointwise_kernel_vs_float64(N, K, Nn, T, relu2, psum):
    """vp_pointwise_fwd (the streaming 1x1 conv of the 2-D backbones' full-resolution stages, resnet_se.py:8-45) against float64 over the
    same bf16 operands: every output, and the fused per-utterance column sums in the conv GEMM's psum layout ((tile, segment, channel)
    of 128-row tiles, values = y - bn_shift) from which the SE gate takes its mean -- utterance lengths that are not multiples of 128
    (every tile position of the boundary), a ragged last tile, a column-offset input view."""
    lib, ctx = N.lib(), N.ctx(0)
    B = max(2, -(-40000 // T))
    M = B * T
    g = torch.Generator().manual_seed(K * 1000 + Nn + T)
    ldx, xoff = K + 32, 16
    xfull = _bf(torch.randn(M, ldx, generator=g, dtype=torch.float64))
    x = xfull[:, xoff:xoff + K]
    w = _bf(torch.randn(Nn, K, generator=g, dtype=torch.float64) / K ** 0.5)
    f32 = lambda t: t.float().double()                                  # noqa: E731
    bias = f32(0.1 * torch.randn(Nn, generator=g, dtype=torch.float64))
    sc, sh = f32(torch.rand(Nn, generator=g, dtype=torch.float64) + 0.5), f32(0.3 * torch.randn(Nn, generator=g, dtype=torch.float64))
    pre = (x @ w.t() + bias) * sc
    y = pre + sh
    res = None
    if relu2 == 'res_hardtanh':                                         # ERes2Net's conv3: hardtanh(bn(conv) + residual, 0, 20)
        res = _bf(torch.randn(M, Nn, generator=g, dtype=torch.float64) * 8)
        y = torch.clamp(y + res, 0.0, 20.0)
    elif relu2:
        y = torch.relu(y)
    ref = _bf(y)
    xd, wd = dev(xfull, torch.bfloat16), dev(w, torch.bfloat16)
    bd, sd, hd = dev(bias, torch.float32), dev(sc, torch.float32), dev(sh, torch.float32)
    yd = torch.full((M, Nn), 7.0, dtype=torch.bfloat16, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_BF16
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, K, Nn, 1, 1, 1
    d.pad_mode = N.VP_PAD_ZERO
    d.x, d.ldx, d.xoff, d.w, d.bias = xd.data_ptr(), ldx, xoff, wd.data_ptr(), bd.data_ptr()
    d.bn_scale, d.bn_shift = sd.data_ptr(), hd.data_ptr()
    d.act2 = N.VP_ACT_HARDTANH20 if relu2 == 'res_hardtanh' else (N.VP_ACT_RELU if relu2 else N.VP_ACT_NONE)
    resd = dev(res, torch.bfloat16) if res is not None else None
    if resd is not None:
        d.res, d.ld_res = resd.data_ptr(), Nn
    d.y, d.ldy = yd.data_ptr(), Nn
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    ps = torch.full((tiles, nseg, Nn), float('nan'), dtype=torch.float32, device='cuda')
    if psum:
        d.psum = ps.data_ptr()
    N.check(lib.vp_pointwise_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    err = (yd.double().cpu() - ref).abs()
    scale = ref.abs().max().item()
    print(f'[pointwise K={K} N={Nn} T={T}] max err {err.max().item():.3e} (max |ref| {scale:.2f})')
    assert err.max().item() < 2e-4 * scale + 2.0 ** -8 * scale
    assert err.mean().item() < 1e-4 * scale
    if psum:
        # per-utterance sums of (y - shift) recovered from the tiles the way se_gate_kernel walks them
        p = ps.double().cpu()
        dvals = (y - sh)
        for b in range(B):
            t0, t1 = (b * T) // 128, ((b + 1) * T - 1) // 128
            tot = torch.zeros(Nn, dtype=torch.float64)
            for tm in range(t0, t1 + 1):
                seg = b - (tm * 128) // T
                tot += p[tm, seg]
            want = dvals[b * T:(b + 1) * T].sum(0)
            assert not torch.isnan(tot).any()
            assert (tot - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()) * T ** 0.5, (b, (tot - want).abs().max().item())


@pytest.mark.parametrize('B,T,F_in', [(2, 37, 40), (3, 50, 20), (1, 298, 10), (2, 5, 11)])
def test_resblock_c32_kernel_vs_float64(N, B, T, F_in):
    """vp_resblock_c32_fwd (a stride-1 BasicResBlock of the FCM head in one launch, campplus.py:211-243) against float64 over the same
    bf16 operands, h rounded to bf16 as the two-launch path stores it: every position incl. the map's border rows / columns, more
    time rows than one workgroup's slab (the two-row halo across slabs), an odd F."""
    lib, ctx = N.lib(), N.ctx(0)
    Cc = 32
    g = torch.Generator().manual_seed(T * 10 + F_in)
    f32 = lambda t: t.float().double()                                  # noqa: E731
    x = _bf(torch.randn(B, T, F_in, Cc, generator=g, dtype=torch.float64))
    keep, Ls, ws = [], [], []
    for _ in range(2):
        w = _bf(torch.randn(Cc, Cc, 3, 3, generator=g, dtype=torch.float64) / (9 * Cc) ** 0.5)
        bias = f32(0.1 * torch.randn(Cc, generator=g, dtype=torch.float64))
        sc, sh = f32(torch.rand(Cc, generator=g, dtype=torch.float64) + 0.5), f32(0.2 * torch.randn(Cc, generator=g, dtype=torch.float64))
        ws.append((w, bias, sc, sh))
        L = N.TdnnLayer()
        wd = dev(w.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc), torch.bfloat16)
        bd, sd, hd = dev(bias, torch.float32), dev(sc, torch.float32), dev(sh, torch.float32)
        keep += [wd, bd, sd, hd]
        L.w, L.bias, L.bn_scale, L.bn_shift, L.cin, L.cout, L.kw, L.dil = wd.data_ptr(), bd.data_ptr(), sd.data_ptr(), hd.data_ptr(), Cc, Cc, 9, 1
        Ls.append(L)
    xc = x.permute(0, 3, 1, 2)
    (w1, b1, s1, h1), (w2, b2, s2, h2) = ws
    h = _bf(torch.relu(F.conv2d(xc, w1, b1, padding=1) * s1[None, :, None, None] + h1[None, :, None, None]))
    ref = _bf(torch.relu(F.conv2d(h, w2, b2, padding=1) * s2[None, :, None, None] + h2[None, :, None, None] + xc)).permute(0, 2, 3, 1)
    xd = dev(x, torch.bfloat16)
    yd = torch.full((B, T, F_in, Cc), 7.0, dtype=torch.bfloat16, device='cuda')
    N.check(lib.vp_resblock_c32_fwd(ctx, xd.data_ptr(), yd.data_ptr(), C.byref(Ls[0]), C.byref(Ls[1]), B, T, F_in, N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    err = (yd.double().cpu() - ref).abs()
    scale = ref.abs().max().item()
    print(f'[resblock_c32 B={B} T={T} F={F_in}] max err {err.max().item():.3e} (max |ref| {scale:.2f}), mean {err.mean().item():.2e}')
    # a bf16 rounding of h can flip (f32 vs float64 accumulation) and feeds nine taps of conv2
    assert err.max().item() < 2.0 ** -6 * scale, err.max().item()
    assert err.mean().item() < 3e-4 * scale
