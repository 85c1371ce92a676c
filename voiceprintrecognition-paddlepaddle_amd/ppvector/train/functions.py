"""torch.autograd.Function wrappers: forward AND backward of each op run in libvpmi (f32 engine).

Activations are position-major 2-D tensors (B*T, C) as everywhere in the engine.  What paddle's autograd provides the
reference for free (ppvector/trainer.py:213-219: loss.backward()) is spelled out here per op:
  ConvBlock   conv1d (+ per-utterance bias) (+ ReLU) (+ BatchNorm with batch statistics) (+ tanh)
              -- Conv1D / TDNNBlock of models/utils.py:22-148 and the Conv1D + ReLU + BatchNorm1D chain of models/tdnn.py:46-60
  TimeStats   [mean | sqrt(clip(var))] over time       (models/pooling.py:97-104, mask of ones)
  AttnStats   softmax over time + weighted mean / std   (models/pooling.py:114-123)
  BNRows      BatchNorm1D on (B, C)                     (models/utils.py:96-119)
  HeadLoss    cosine classifier + AAM-softmax loss      (models/fc.py:41-53, loss/aamloss.py:28-47)
"""
import ctypes as C
import os
import weakref

import torch

import ppvector

from ppvector import _native as N

_PAD = {'none': N.VP_PAD_NONE, 'zero': N.VP_PAD_ZERO, 'reflect': N.VP_PAD_REFLECT}


_CONSTS = {}


def _const(value, n, device):
    """A read-only vector of n zeros / ones on `device`, made once (two fill launches per BatchNorm unit and step otherwise).  Never
    created under a stream capture: GraphedTrainStep runs eager warm-up steps first, and a capture that finds the cache empty falls back
    to a fresh tensor."""
    key = (float(value), int(n), str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = torch.full((n,), float(value), dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _CONSTS[key] = t
    return t


def _chk(rc, ctx):
    N.check(rc, ctx)


def _bytes(n, dev):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)


def _f32c(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise N.VpmiError('training functions take f32 GPU tensors (no CPU fallback)')
    if getattr(t, '_vp_bf16_only', False) and getattr(t, '_vp_bf16', None) is not None:
        return t._vp_bf16.float()         # a memory-less placeholder (_placeholder): a consumer without a bf16 path gets the values converted
    return t.contiguous()


def _f32_rows(t):
    """_f32c, except that a column slice of a wider f32 tensor (unit channel stride, 16-byte aligned rows) is taken as it is: the conv
    kernels read their input with a row pitch (Conv1dDesc.ldx), so a chunk of a torch.split needs no copy."""
    if (t.dtype == torch.float32 and t.is_cuda and t.dim() == 2 and not getattr(t, '_vp_bf16_only', False) and t.stride(1) == 1
            and t.stride(0) >= t.shape[1] and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        return t
    return _f32c(t)


def _conv_desc(x, B, T_in, T_out, Cin, Cout, KW, dil, pad_mode, pad_left, w, bias=None, rowbias=None, relu=False):
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_F32
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T_in, T_out, Cin, Cout, KW, dil, 1
    d.pad_mode, d.pad_left = pad_mode, pad_left
    d.x, d.ldx, d.xoff = x.data_ptr(), (x.stride(0) if x.dim() == 2 and x.shape[0] > 1 else Cin), 0     # (a column slice of a wider buffer keeps its pitch)
    d.w = w.data_ptr()
    if bias is not None:
        d.bias = bias.data_ptr()
    if rowbias is not None:
        d.rowbias = rowbias.data_ptr()
    d.act = N.VP_ACT_RELU if relu else N.VP_ACT_NONE
    d.ldy = Cout
    # enable_amp: bf16 matrix cores over f32 tensors (forward, dgrad and wgrad); set_train_x3: the same three GEMMs in split precision
    d.mfma_bf16 = 1 if ppvector.get_train_amp() else (2 if ppvector.get_train_x3() else 0)
    return d


def col_sums(a, b=None, bmean=None, bscale=None):
    """-> (2, C): sum_m a and (when b) sum_m a * (b - bmean) * bscale."""
    lib, ctx = N.lib(), N.ctx(a.device)
    M, Cc = a.shape
    out = torch.empty((2, Cc), dtype=torch.float32, device=a.device)
    ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cc), a.device)
    _chk(lib.vp_col_sums_f32(ctx, a.data_ptr(), Cc, b.data_ptr() if b is not None else None, Cc,
                             bmean.data_ptr() if b is not None else None, bscale.data_ptr() if b is not None else None,
                             M, Cc, out.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
    return out


def _wide_bf16(M, Cin, Cout, KW, bias, rowbias, gamma, cfg):
    """Mixed precision, wide 1x1 TDNN blocks (tdnn1 / tdnn2 / MFA of ECAPA): the GEMM operands x and dz are kept as bf16 tensors --
    what the matrix cores would round them to anyway -- so the forward, data-gradient and weight-gradient kernels read half the
    bytes and skip the conversion (vp_conv1d_fwd bf16 -> f32, vp_conv1d_wgrad_bf16_oik).  VPMI_TRAIN_BF16_OPS=0 keeps f32 operands."""
    if cfg.get('wide_taps') and KW > 1:
        # a tapped layer the caller names (ECAPA's block 0: 80 -> 512, k = 5): the same bf16 operand / bf16 pre-BatchNorm form on the
        # 256-wide tapped LDS-DMA kernel the inference engine runs it on (f32 operands: 163 us forward, 228 us weight gradient)
        if not (ppvector.get_train_amp() and gamma is not None and bias is not None and rowbias is None and Cin % 8 == 0 and Cout % 64 == 0
                and Cout >= 256 and M >= 4096 and not cfg.get('tanh', False) and cfg.get('T', 0) >= 128):
            return 0
        return int(os.environ.get('VPMI_TRAIN_BF16_OPS', '2'))
    if not (ppvector.get_train_amp() and KW == 1 and gamma is not None and bias is not None and rowbias is None and Cin % 64 == 0
            and Cin >= 256 and Cout >= 256 and Cout % 4 == 0 and M >= 4096 and not cfg.get('tanh', False)):
        return 0
    return int(os.environ.get('VPMI_TRAIN_BF16_OPS', '2'))


# _wide_bf16 levels: 1 = bf16 GEMM operands (x, dz), same roundings as the f32-operand kernels;  2 (default) = also the pre-BatchNorm
# activation z stored as bf16 -- one more rounding (2^-9 relative per element, what Paddle's O1 does to every conv output) that lets
# the forward run the 256-wide LDS-DMA kernel (bf16 -> bf16, statistics from the f32 accumulators) and halves z's three later reads.


_ZERO = {}


def _only16(t):
    """The bf16 tensor that carries the values of `t` when `t` is a memory-less f32 placeholder on the tape (see ConvBlock cfg['y_bf16'],
    SEBlockFn), else None."""
    return getattr(t, '_vp_bf16', None) if getattr(t, '_vp_bf16_only', False) else None


def _placeholder(shape, device):
    """f32 tensor of `shape` that owns one element: what autograd sees of an activation that exists as bf16 only (autograd hands a
    tensor's gradient over in the tensor's dtype, so a bf16 tensor on the tape would get its f32 gradient cast down).  The element is
    NaN: a consumer that reads the placeholder's own memory instead of its bf16 twin (a missed _f32c / _only16) poisons the loss and
    every gradient behind it at the first step, instead of training on zeros (tests/test_gpu_train.py:
    test_placeholder_read_as_data_is_loud)."""
    key = (device.type, device.index)
    z = _ZERO.get(key)
    if z is None:                  # one element per device, made once (a fill per placeholder is a 5 us launch, 16 of them per step)
        z = _ZERO[key] = torch.full((1,), float('nan'), dtype=torch.float32, device=device)
    return z.expand(shape)


def _narrow_to_wide(M, Cin, Cout, KW):
    """A plain 1x1 GEMM with few input and many output channels under enable_amp: worth a bf16 copy of its small operand."""
    return (ppvector.get_train_amp() and KW == 1 and Cin % 64 == 0 and Cout >= 256 and 4 * Cin <= Cout and M >= 4096
            and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') != '0')


# ---------------------------------------------------------------------------------------------------- bf16 weight panels
# The wide mixed-precision layers read their weights as bf16 twice per step: W in the forward GEMM, W^T in the data-gradient GEMM.  Each
# used to be a conversion launch of its own (13 + 9 per ECAPA step).  A backbone's training forward now hands ALL of them to ONE launch
# up front (vp_prep_weights_bf16); the panels live in this registry -- keyed by the parameter and the column slice, stamped with the
# weights epoch (an optimiser step / a graph replay moves it) AND the parameter's torch version counter (load_state_dict, a checkpoint
# resume, a manual in-place edit move that one: ADVICE r05): stale panels are never read -- and ConvBlock looks them up.
_W16 = {}


def drop_weight_panels():
    """Forget every bf16 panel (GraphedTrainStep drops its captures / a fault was handled: the panels of a captured step live in that
    graph's private pool)."""
    _W16.clear()


def prep_weights_bf16(items):
    """items: [(weight (Cout, Cin_total, 1) f32 parameter, col0, ncols)] -- the 1x1 conv weights (or column slices of one: the x part of
    ASP's attention TDNN) whose bf16 panel W [Cout][ncols] and transpose W^T [ncols][Cout] the step will read.  One launch."""
    items = [(w, c0, nc) for w, c0, nc in items if w is not None and w.is_cuda and w.dtype == torch.float32 and w.dim() == 3
             and w.shape[2] == 1 and w.is_contiguous()]
    if not items:
        return
    lib, hctx = N.lib(), N.ctx(items[0][0].device)
    n = len(items)
    dev = items[0][0].device
    total = sum(w.shape[0] * nc for w, _, nc in items)
    pool = torch.empty(2 * total, dtype=torch.bfloat16, device=dev)
    ws, w16s, wts, rows, cols, lds, at = [], [], [], [], [], [], 0
    epoch = N.weights_epoch()
    for w, c0, nc in items:
        co, ct = w.shape[0], w.shape[1]
        a = pool[at:at + co * nc].view(co, nc)
        b = pool[at + co * nc:at + 2 * co * nc].view(nc, co)
        at += 2 * co * nc
        ws.append(w.data_ptr() + 4 * c0)
        w16s.append(a.data_ptr())
        wts.append(b.data_ptr())
        rows.append(co)
        cols.append(nc)
        lds.append(ct)
        _W16[(id(w), c0, nc)] = (weakref.ref(w), (epoch, w._version), a, b)
    arr = lambda t, v: (t * n)(*v)
    _chk(lib.vp_prep_weights_bf16(hctx, arr(C.c_void_p, ws), arr(C.c_void_p, w16s), arr(C.c_void_p, wts), arr(C.c_int, rows),
                                  arr(C.c_int, cols), arr(C.c_int, lds), n, N.stream_ptr()), hctx)


def _panels16(w, c0=0, nc=None):
    """(W bf16 [Cout][ncols], W^T bf16 [ncols][Cout]) of this step's prep_weights_bf16 launch, or None."""
    if w is None or w.dim() != 3:
        return None
    e = _W16.get((id(w), c0, w.shape[1] if nc is None else nc))
    if e is None or e[0]() is not w or e[1] != (N.weights_epoch(), w._version) or os.environ.get('VPMI_NO_WPREP'):
        return None
    return e[2], e[3]


class ConvBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, rowbias, gamma, beta, run_mean, run_var, cfg):
        lib, hctx = N.lib(), N.ctx(x.device)
        p16 = cfg.get('_w16') or _panels16(weight)              # this step's bf16 panels of the weight (prep_weights_bf16), or None
        weight = _f32c(weight)
        B, T_in, dil = cfg['B'], cfg['T'], cfg.get('dilation', 1)
        Cout, Cin, KW = weight.shape
        if KW != 1 or p16 is None or tuple(p16[0].shape) != (Cout, Cin):
            p16 = None
        wide = _wide_bf16(B * T_in, Cin, Cout, KW, bias, rowbias, gamma, cfg)
        if x.dtype == torch.bfloat16:
            if not wide and KW == 1 and ppvector.get_train_amp() and Cin % 8 == 0 and Cout % 4 == 0 and bias is not None:
                wide = 1                                      # a producer handed over bf16 (ASP's attention TDNN on the bf16 MFA output): bf16 operands, f32 z
            if not wide:
                raise N.VpmiError('ConvBlock: a bf16 input outside the wide mixed-precision layers')
            if x.stride(1) != 1 or x.stride(0) % 4:
                x = x.contiguous()
        elif _only16(x) is not None:                    # the producer wrote x as bf16 ONLY: x is a placeholder, the twin is the operand
            if not wide:
                raise N.VpmiError('ConvBlock: an input that exists as bf16 only outside the wide mixed-precision layers')
            x = _only16(x)
        else:
            shadow = getattr(x, '_vp_bf16', None)         # the producer already wrote x as bf16 (SEBlockFn: a slice of the MFA operand)
            x = _f32c(x)
            if wide:
                x = shadow if shadow is not None and shadow.shape == x.shape else x.to(torch.bfloat16)
        pad = cfg.get('pad', 'none')
        pad_left = 0 if pad == 'none' else dil * (KW - 1) // 2
        T_out = T_in - dil * (KW - 1) if pad == 'none' else T_in
        relu, bn = cfg.get('relu', False), gamma is not None
        tanh = N.VP_ACT_TANH if cfg.get('tanh', False) else (N.VP_ACT_SIGMOID if cfg.get('sigmoid', False) else 0)
        if x.shape != (B * T_in, Cin) or T_out < 1:
            raise ValueError(f'ConvBlock: x {tuple(x.shape)} does not match B {B}, T {T_in}, Cin {Cin}')
        w2 = None
        if KW > 1:          # the forward panel (Cout, tap, Cin) and the data-gradient one (Cin, reversed tap, Cout) in one launch
            wp = torch.empty((Cout, KW * Cin), dtype=torch.float32, device=x.device)
            if ctx.needs_input_grad[0]:
                w2 = torch.empty((Cin, KW * Cout), dtype=torch.float32, device=x.device)
            _chk(lib.vp_conv_weight_layouts_f32(hctx, weight.data_ptr(), Cout, Cin, KW, wp.data_ptr(),
                                                w2.data_ptr() if w2 is not None else None, N.stream_ptr()), hctx)
        else:
            wp = weight.view(Cout, Cin)
        narrow = bool(_narrow_to_wide(B * T_in, Cin, Cout, KW) and not wide and not bn and rowbias is None and cfg.get('pad', 'none') == 'none')
        out16 = narrow and bool(cfg.get('out_bf16')) and not cfg.get('tanh', False) and not cfg.get('sigmoid', False)
        z = torch.empty((B * T_out, Cout), dtype=torch.bfloat16 if (wide >= 2 or out16) else torch.float32, device=x.device)
        if wide:
            wp = p16[0] if p16 is not None else wp.to(torch.bfloat16)
        xin = x
        if narrow:
            # enable_amp, few inputs -> many outputs (ASP's logits conv, 128 -> 1536): the kernel rounds x to bf16 anyway; rounding it up
            # front (39 MB) lets the launch take the LDS-DMA ring kernel with f32 output instead of the 128-wide register-staged one,
            # which is bound by its 469 MB of stores (206 -> ~115 us).  Saved for backward: the f32 x, as before.
            xin, wp = x.to(torch.bfloat16), (p16[0] if p16 is not None else wp.to(torch.bfloat16))
        d = _conv_desc(xin, B, T_in, T_out, Cin, Cout, KW, dil, _PAD[pad], pad_left, wp, bias, rowbias, relu)
        if wide or xin is not x:
            d.dtype_in = N.VP_BF16
            if wide >= 2 or out16:
                d.dtype_out = N.VP_BF16
        d.y = z.data_ptr()
        ps = pq = None
        if bn and lib.vp_conv1d_nseg(T_out) <= 8:              # the conv's fused column sums (utterances >= ~19 frames)
            tiles, nseg = lib.vp_conv1d_tiles_m(B, T_out), lib.vp_conv1d_nseg(T_out)
            ps = torch.empty((tiles * nseg, Cout), dtype=torch.float32, device=x.device)
            pq = torch.empty_like(ps)
            d.psum, d.psumsq = ps.data_ptr(), pq.data_ptr()
        if T_in == 1 and KW == 1 and not bn and rowbias is None and not wide and not narrow and B <= 4096 and Cin >= 1024:
            # a per-utterance dense layer (ASP's context term, the embedding layer): B rows are one or two tiles of the conv GEMM, which
            # then walks K = 3072 serially (70 us at B = 256); the small-M dense kernel takes 11 us (exact f32).  Long K only: the short
            # per-utterance layers (the SE block's unfused form) keep the rounding their fused twin is tested against
            _chk(lib.vp_dense_f32(hctx, x.data_ptr(), d.ldx, wp.data_ptr(), 0, bias.data_ptr() if bias is not None else None, B, Cout, Cin,
                                  N.VP_ACT_RELU if relu else N.VP_ACT_NONE, z.data_ptr(), Cout, N.stream_ptr()), hctx)
        else:
            _chk(lib.vp_conv1d_fwd(hctx, C.byref(d), N.stream_ptr()), hctx)
        if bn and ps is None and wide >= 2:
            raise N.VpmiError('ConvBlock: bf16 pre-BN activation needs the conv\'s fused column sums (utterances of >= ~19 frames)')
        if bn and ps is None:                                   # very short utterances: a separate column-sum pass
            zeros, ones = _const(0.0, Cout, x.device), _const(1.0, Cout, x.device)
            sums = col_sums(z, z, zeros, ones)
            ps, pq = sums[0:1], sums[1:2]
        mean = invstd = None
        y = z
        if bn:
            mean, invstd, scale, shift = (torch.empty(Cout, dtype=torch.float32, device=x.device) for _ in range(4))
            _chk(lib.vp_bn_train_finalize(hctx, ps.data_ptr(), pq.data_ptr(), ps.shape[0], B * T_out, Cout, gamma.data_ptr(),
                                          beta.data_ptr(), run_mean.data_ptr() if run_mean is not None else None,
                                          run_var.data_ptr() if run_var is not None else None, cfg.get('momentum', 0.9),
                                          cfg.get('eps', 1e-5), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                          shift.data_ptr(), N.stream_ptr()), hctx)
            into, add = cfg.get('y_into'), cfg.get('aux_add')
            y16 = wide >= 2 and cfg.get('y_bf16') and not tanh
            if y16 and cfg.get('_no_apply'):
                # the consumer folds the BatchNorm apply into its own read of z (ConvSEFn: the SE gate + residual pass): y is never stored
                y = _placeholder(z.shape, x.device)
                cfg['_zaff'] = (z, scale, shift)
            elif y16:
                # the output's only consumer reads it as bf16 and takes its time statistics from the fused sums (ECAPA's MFA -> ASP):
                # 234 MB written instead of 469, and every later pass over it reads half
                y16t = torch.empty(z.shape, dtype=torch.bfloat16, device=x.device)
                _chk(lib.vp_affine_rows_b16_b16(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), B * T_out, Cout,
                                                y16t.data_ptr(), Cout, 0, N.stream_ptr()), hctx)
                # autograd hands a tensor's gradient over in the tensor's dtype: a bf16 y would get its 469 MB f32 gradient cast down.
                # The tape therefore sees an f32 PLACEHOLDER of the right shape that owns no memory (one zero, expanded); the values
                # travel as its bf16 twin, the way producers' twins do everywhere else (`_vp_bf16`), marked as the only copy.
                y = _placeholder(z.shape, x.device)
                cfg['_y16'] = y16t
            elif wide >= 2:
                y = torch.empty(z.shape, dtype=torch.float32, device=x.device)
                _chk(lib.vp_affine_rows_b16_f32(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), B * T_out, Cout,
                                                y.data_ptr(), Cout, 0, N.stream_ptr()), hctx)
            elif into is not None or add is not None:
                # y straight into a channel slice of a wider tensor, and aux = y + add (the next Res2Net chunk's input) in the same pass
                y = into if into is not None else torch.empty_like(z)
                aux = torch.empty_like(z) if add is not None else None
                _chk(lib.vp_affine_rows_aux_f32(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), B * T_out, Cout,
                                                y.data_ptr(), y.stride(0), add.data_ptr() if add is not None else None,
                                                add.stride(0) if add is not None else 0, aux.data_ptr() if aux is not None else None,
                                                Cout, N.stream_ptr()), hctx)
                ctx.aux_out = aux
            elif cfg.get('y_bf16') and not wide and not tanh and ppvector.get_train_amp():
                # a layer outside the wide set whose consumers read bf16 only (ECAPA's block 0): the apply pass writes the bf16 form directly
                y16t = torch.empty(z.shape, dtype=torch.bfloat16, device=x.device)
                _chk(lib.vp_affine_rows_f32_b16(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), B * T_out, Cout,
                                                y16t.data_ptr(), Cout, 0, N.stream_ptr()), hctx)
                y = _placeholder(z.shape, x.device)
                cfg['_y16'] = y16t
            else:
                y = torch.empty_like(z)
                _chk(lib.vp_affine_rows_f32(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), B * T_out, Cout,
                                            y.data_ptr(), Cout, 0, N.stream_ptr()), hctx)
        if tanh:
            yt = torch.empty_like(y)
            _chk(lib.vp_act_f32(hctx, tanh, y.data_ptr(), y.numel(), yt.data_ptr(), N.stream_ptr()), hctx)
            y = yt
        if bn and not tanh and cfg.get('want_tsums') and ps is not None and ps.shape[0] > 1:
            # the conv's fused per-(tile, utterance) sums of z + this layer's affine: a consumer's time statistics of y need no pass over y
            cfg['_tsums'] = (ps, pq, scale, shift, B, T_out)
        ctx.save_for_backward(x, weight, z, mean, invstd, gamma, y if tanh else None, w2)
        ctx.geom = (B, T_in, T_out, Cin, Cout, KW, dil, pad, pad_left, relu, bn, tanh, bias is not None, rowbias is not None)
        ctx.wide = wide
        ctx.wt16 = p16[1] if p16 is not None else None           # W^T as bf16: the data-gradient GEMM's panel, from the same launch
        ctx.zero_dbias = bool(cfg.get('zero_bias_grad', False))
        return y

    @staticmethod
    def backward(ctx, dy):
        return _conv_block_bwd(ctx, dy)


def _conv_block_bwd(ctx, dy, skip=None, fold=None, utt=None):
    """ConvBlock's backward.  skip: a gradient that reached x along another path (the block residual, another consumer of the
    same tensor), added in the data-gradient conv's epilogue instead of by a separate pass.  fold: {'dx': slice view of a wider
    gradient tensor, 'add': slice view or None} -- d x is written into that slice and the returned tensor is d x + add (or None):
    the Res2Net hand-off (Res2Fn).  utt: ('se', s (B, Cout), dm (B, Cout), T) -- the output gradient is dy * s[b] + dm[b] / T per
    utterance (ConvSEFn: the SE block behind this conv never stores its input gradient) -- or ('ctx', alpha, beta, T, bn_scale, bn_shift)
    -- dy + alpha[b] + beta[b] * y (MfaAspFn: the pooling layer's context-statistics gradient), formed on the fly by the two
    BatchNorm-backward passes."""
    x, weight, z, mean, invstd, gamma, yt, w2 = ctx.saved_tensors
    B, T_in, T_out, Cin, Cout, KW, dil, pad, pad_left, relu, bn, tanh, has_bias, has_rb = ctx.geom
    lib, hctx = N.lib(), N.ctx(x.device)
    dev = x.device
    M = B * T_out
    wide = getattr(ctx, 'wide', False)
    if dy.dtype == torch.bfloat16:
        # a plain conv (no BatchNorm / activation behind it) handed its output gradient as bf16 -- the ASP logits conv, whose d e the
        # statistics' backward writes in that form: dz = dy is the bf16 GEMM operand, x (small here) is converted to match
        if bn or relu or tanh or has_rb or KW != 1 or not ppvector.get_train_amp():
            raise N.VpmiError('ConvBlock backward: a bf16 output gradient is only taken by a plain 1x1 conv under enable_amp')
        dy, wide = dy.contiguous(), wide or 1
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
    else:
        dy = _f32c(dy)
    if tanh:
        t = torch.empty_like(dy)
        _chk(lib.vp_act_bwd_f32(hctx, tanh, dy.data_ptr(), yt.data_ptr(), dy.numel(), t.data_ptr(), N.stream_ptr()), hctx)
        dy = t
    dgamma = dbeta = None
    if bn or relu:
        if utt is not None and not (bn and z.dtype == torch.bfloat16 and has_bias and Cout % 4 == 0 and not tanh):
            raise N.VpmiError('ConvBlock backward: a per-utterance term on the output gradient needs the bf16 pre-BatchNorm form')
        if bn:
            if z.dtype == torch.bfloat16 and utt is not None and utt[0] == 'ctx':
                # d y = dy + alpha[b] + beta[b] * y: a consumer's context-statistics gradient (MfaAspFn), y re-formed from z
                _, al, be, Tu, bsc, bsh = utt
                sums = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cout), dev)
                _chk(lib.vp_col_sums_f32_b16_ctx(hctx, dy.data_ptr(), Cout, al.data_ptr(), be.data_ptr(), int(Tu), bsc.data_ptr(), bsh.data_ptr(),
                                                 z.data_ptr(), Cout, mean.data_ptr(), invstd.data_ptr(), M, Cout, sums.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), N.stream_ptr()), hctx)
            elif z.dtype == torch.bfloat16 and utt is not None:
                # d y = dy * s[b] + dm[b] / T: the SE block behind this conv (ConvSEFn)
                sums = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cout), dev)
                _chk(lib.vp_col_sums_f32_b16_utt(hctx, dy.data_ptr(), Cout, utt[1].data_ptr(), utt[2].data_ptr(), int(utt[3]), z.data_ptr(), Cout,
                                                 mean.data_ptr(), invstd.data_ptr(), M, Cout, sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 N.stream_ptr()), hctx)
            elif z.dtype == torch.bfloat16:
                sums = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cout), dev)
                _chk(lib.vp_col_sums_f32_b16(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mean.data_ptr(), invstd.data_ptr(), M, Cout,
                                             sums.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            else:
                sums = col_sums(dy, z, mean, invstd)
            dgamma, dbeta = sums[1], sums[0]             # views of a buffer this call owns: no copies
            mu, istd, g = mean, invstd, gamma
        else:                                   # ReLU alone: the BN backward formula with identity statistics
            sums = torch.zeros((2, Cout), dtype=torch.float32, device=dev)
            mu, istd = _const(0.0, Cout, dev), _const(1.0, Cout, dev)
            g = None
        dz = torch.empty_like(dy, dtype=torch.bfloat16 if wide else torch.float32)
        if has_bias and Cout % 4 == 0:          # the bias gradient (column sums of dz) from the pass that writes dz
            dbias = torch.empty(Cout, dtype=torch.float32, device=dev)
            ws = _bytes(lib.vp_bn_relu_bwd_dbias_workspace_bytes(M, Cout), dev)
            fn = (lib.vp_bn_relu_bwd_dbias_b16 if z.dtype == torch.bfloat16 else
                  lib.vp_bn_relu_bwd_dbias_bf16out if wide else lib.vp_bn_relu_bwd_dbias_f32)
            if utt is not None and utt[0] == 'ctx':
                _, al, be, Tu, bsc, bsh = utt
                _chk(lib.vp_bn_relu_bwd_dbias_b16_ctx(hctx, dy.data_ptr(), Cout, al.data_ptr(), be.data_ptr(), int(Tu), bsc.data_ptr(), bsh.data_ptr(),
                                                      z.data_ptr(), Cout, mu.data_ptr(), istd.data_ptr(), g.data_ptr() if g is not None else None,
                                                      sums.data_ptr(), M, Cout, int(relu), dz.data_ptr(), Cout, dbias.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            elif utt is not None:
                _chk(lib.vp_bn_relu_bwd_dbias_b16_utt(hctx, dy.data_ptr(), Cout, utt[1].data_ptr(), utt[2].data_ptr(), int(utt[3]), z.data_ptr(),
                                                      Cout, mu.data_ptr(), istd.data_ptr(), g.data_ptr() if g is not None else None,
                                                      sums.data_ptr(), M, Cout, int(relu), dz.data_ptr(), Cout, dbias.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            else:
                _chk(fn(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mu.data_ptr(), istd.data_ptr(),
                        g.data_ptr() if g is not None else None, sums.data_ptr(), M, Cout, int(relu),
                        dz.data_ptr(), Cout, dbias.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
        else:
            _chk(lib.vp_bn_relu_bwd_f32(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mu.data_ptr(), istd.data_ptr(),
                                        g.data_ptr() if g is not None else None, sums.data_ptr(), M, Cout, int(relu),
                                        dz.data_ptr(), Cout, N.stream_ptr()), hctx)
            dbias = col_sums(dz)[0] if has_bias else None
    else:
        dz = dy
        if has_bias and getattr(ctx, 'given_dbias', None) is not None:      # the caller's own pass over dz already summed its columns
            dbias = ctx.given_dbias
        elif has_bias and getattr(ctx, 'zero_dbias', False):      # the caller knows sum_rows dz == 0 (a bias in front of a softmax over time)
            dbias = torch.zeros(Cout, dtype=torch.float32, device=dev)
        else:
            dbias = col_sums(dz if dz.dtype == torch.float32 else dz.float())[0] if has_bias else None
    drb = None
    if has_rb:
        drb = torch.empty((B, Cout), dtype=torch.float32, device=dev)
        fn = lib.vp_utt_sums_b16 if dz.dtype == torch.bfloat16 else lib.vp_utt_sums_f32
        _chk(fn(hctx, dz.data_ptr(), Cout, B, T_out, Cout, drb.data_ptr(), N.stream_ptr()), hctx)
    # weight gradient
    d = _conv_desc(x, B, T_in, T_out, Cin, Cout, KW, dil, _PAD[pad], pad_left, weight)
    dW = torch.empty((Cout, Cin, KW), dtype=torch.float32, device=dev)      # reduced straight into the model's layout
    ws = _bytes(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(d)), dev)
    if wide:                                            # x (saved as bf16) and dz both bf16 in memory
        d.dtype_in = N.VP_BF16
        _chk(lib.vp_conv1d_wgrad_bf16_oik(hctx, C.byref(d), dz.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(),
                                          N.stream_ptr()), hctx)
    else:
        _chk(lib.vp_conv1d_wgrad_oik_f32(hctx, C.byref(d), dz.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(),
                                         N.stream_ptr()), hctx)
    # data gradient: the forward kernel over dz with reversed taps and swapped channel roles
    dx = None
    if ctx.needs_input_grad[0]:
        wt16 = getattr(ctx, 'wt16', None) if KW == 1 else None
        # (a many -> few layer's data gradient is a few -> many GEMM over dz and runs on bf16 operands too: see the last branch below)
        narrow_d = bool(not wide and KW == 1 and dz.dtype == torch.float32 and pad == 'none' and _narrow_to_wide(B * T_out, Cout, Cin, KW))
        if w2 is None and (wide or narrow_d) and wt16 is not None:     # KW = 1: W^T, already there as bf16 (this step's weight-prep launch)
            w2 = wt16
        elif w2 is None:
            w2 = weight.view(Cout, Cin).t().to(torch.bfloat16 if wide else torch.float32, memory_format=torch.contiguous_format)
            if not wide:
                w2 = w2.contiguous()
        splits = getattr(ctx, 'out_splits', None) if (KW == 1 and skip is None and fold is None) else None
        dx = torch.empty((B * T_in, Cin), dtype=torch.float32, device=dev) if not splits else None
        if pad == 'reflect' and pad_left > 0:
            # gradient w.r.t. the reflect-PADDED input (a "full" zero-padded conv), then fold the mirrored frames back
            Tp = T_in + 2 * pad_left
            dxp = torch.empty((B * Tp, Cin), dtype=torch.float32, device=dev)
            d2 = _conv_desc(dz, B, T_out, Tp, Cout, Cin, KW, dil, N.VP_PAD_ZERO, dil * (KW - 1), w2)
            d2.y = dxp.data_ptr()
            _chk(lib.vp_conv1d_fwd(hctx, C.byref(d2), N.stream_ptr()), hctx)
            if fold is not None:
                into, add = fold['dx'], fold.get('add')
                dx = torch.empty((B * T_in, Cin), dtype=torch.float32, device=dev) if add is not None else None
                _chk(lib.vp_reflect_fold_into_f32(hctx, dxp.data_ptr(), B, T_in, pad_left, Cin, into.data_ptr(), into.stride(0),
                                                  add.data_ptr() if add is not None else None, add.stride(0) if add is not None else 0,
                                                  dx.data_ptr() if dx is not None else None, N.stream_ptr()), hctx)
                fold = None
            else:
                _chk(lib.vp_reflect_fold_f32(hctx, dxp.data_ptr(), B, T_in, pad_left, Cin, dx.data_ptr(), N.stream_ptr()), hctx)
            if skip is not None:
                dx += skip
        elif splits:
            # a concatenated input (CatConvBlock): one launch per input over its rows of W^T, each gradient a contiguous tensor of its
            # own -- a column slice of one wide d x is copied once more when autograd stores it as a leaf's .grad (3 x 55 us on the MFA layer)
            dx, at = [], 0
            for wd in splits:
                part = torch.empty((B * T_in, wd), dtype=torch.float32, device=dev)
                d2 = _conv_desc(dz, B, T_out, T_in, Cout, wd, 1, 1, N.VP_PAD_ZERO, 0, w2[at:at + wd])
                # the descriptor's operand type follows the operands actually handed over (never from the gating flags alone)
                if w2.dtype == torch.bfloat16:
                    if dz.dtype != torch.bfloat16:
                        raise N.VpmiError('conv backward: bf16 W^T rows with an f32 dz in the split data-gradient branch')
                    d2.dtype_in = N.VP_BF16
                d2.y = part.data_ptr()
                _chk(lib.vp_conv1d_fwd(hctx, C.byref(d2), N.stream_ptr()), hctx)
                dx.append(part)
                at += wd
            dx = tuple(dx)
        else:
            dzin = dz
            if narrow_d:
                # the data gradient of a many-inputs -> few-outputs layer (ASP's attention TDNN, 1536 -> 128) is a few -> many GEMM over
                # dz: the same up-front rounding of the small operand, the same kernel (341 -> ~215 us with the other path's gradient
                # added in its epilogue)
                dzin, w2 = dz.to(torch.bfloat16), (w2 if w2.dtype == torch.bfloat16 else w2.to(torch.bfloat16))
            d2 = _conv_desc(dzin, B, T_out, T_in, Cout, Cin, KW, dil, N.VP_PAD_ZERO, dil * (KW - 1) - pad_left, w2)
            if (w2.dtype == torch.bfloat16) != (dzin.dtype == torch.bfloat16):
                raise N.VpmiError(f'conv backward: operand types disagree (W^T {w2.dtype}, dz {dzin.dtype})')
            if w2.dtype == torch.bfloat16:
                d2.dtype_in = N.VP_BF16
            d2.y = dx.data_ptr()
            if skip is not None:
                skip = _f32c(skip)
                d2.res, d2.ld_res, d2.res_off = skip.data_ptr(), Cin, 0
            _chk(lib.vp_conv1d_fwd(hctx, C.byref(d2), N.stream_ptr()), hctx)
    elif skip is not None:
        dx = skip
    if fold is not None and dx is not None:          # (a conv without reflect padding: the same hand-off with tensor ops)
        fold['dx'].copy_(dx)
        dx = dx + fold['add'] if fold.get('add') is not None else None
    return dx, dW, dbias, drb, dgamma, dbeta, None, None, None


class CatConvBlock(torch.autograd.Function):
    """TDNNBlock over the channel concatenation of several (B*T, C_i) tensors -- the MFA layer (ecapa_tdnn.py:262-263) -- as one
    tape entry.  In the wide mixed-precision mode the concatenation is built directly as the bf16 operand (one converting copy per
    input into its slice; no f32 concatenation exists)."""

    @staticmethod
    def forward(ctx, cfg, weight, bias, gamma, beta, run_mean, run_var, *xs):
        M, widths = xs[0].shape[0], [x.shape[1] for x in xs]
        Cout, Cin, KW = weight.shape
        if _wide_bf16(M, Cin, Cout, KW, bias, None, gamma, cfg):
            xcat = cfg.get('xcat')                         # filled slice by slice by the producers of xs (SEBlockFn's bf16 shadow), or:
            if xcat is None or tuple(xcat.shape) != (M, Cin) or xcat.dtype != torch.bfloat16:
                if any(_only16(x) is not None for x in xs):
                    raise N.VpmiError('CatConvBlock: inputs that exist as bf16 only need the shared bf16 concatenation (cfg[\'xcat\'])')
                xs = [_f32c(x) for x in xs]
                xcat = torch.empty((M, Cin), dtype=torch.bfloat16, device=xs[0].device)
                at = 0
                for x, wd in zip(xs, widths):
                    xcat[:, at:at + wd].copy_(x)
                    at += wd
        else:
            xcat = torch.cat([_f32c(x) for x in xs], dim=1)
        tp = _Tape((True,) * 9)
        y = ConvBlock.forward(tp, xcat, weight, bias, None, gamma, beta, run_mean, run_var, cfg)
        ctx.save_for_backward(*tp.saved_tensors)
        ctx.inner = (tp.geom, tp.wide, widths, tp.wt16)
        return y

    @staticmethod
    def backward(ctx, dy):
        tp = _Tape((True,) * 9)
        tp.saved_tensors = ctx.saved_tensors
        tp.geom, tp.wide, widths, tp.wt16 = ctx.inner
        tp.out_splits = widths
        r = _conv_block_bwd(tp, dy)
        dxs = r[0] if isinstance(r[0], tuple) else r[0].split(widths, dim=1)
        return (None, r[1], r[2], r[4], r[5], None, None, *dxs)


class ConvBlockSkip(torch.autograd.Function):
    """ConvBlock that also hands its input on: (y, x).  A consumer of the second output (the SE-Res2 block's residual,
    ecapa_tdnn.py:139-141) sends its gradient back here, where the data-gradient conv adds it in its epilogue -- instead of
    autograd summing two (B*T, C) tensors afterwards."""

    @staticmethod
    def forward(ctx, x, weight, bias, rowbias, gamma, beta, run_mean, run_var, cfg):
        y = ConvBlock.forward(ctx, x, weight, bias, rowbias, gamma, beta, run_mean, run_var, cfg)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        return _conv_block_bwd(ctx, dy, dskip)


class SEScale(torch.autograd.Function):
    """out = x * s[b] + res  (SEBlock gate, ecapa_tdnn.py:82, and the block residual, :139-141)."""

    @staticmethod
    def forward(ctx, x, s, res, B, T):
        lib, hctx = N.lib(), N.ctx(x.device)
        x, s, res = _f32c(x), _f32c(s), _f32c(res)
        Cc = x.shape[1]
        out = torch.empty_like(x)
        _chk(lib.vp_se_scale_residual(hctx, N.VP_F32, x.data_ptr(), Cc, 0, s.data_ptr(), res.data_ptr(), Cc, 0, out.data_ptr(), Cc, 0,
                                      B, T, Cc, N.stream_ptr()), hctx)
        ctx.save_for_backward(x, s)
        ctx.geom = (B, T)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        B, T = ctx.geom
        lib, hctx = N.lib(), N.ctx(x.device)
        dy = _f32c(dy)
        Cc = x.shape[1]
        dx, ds = torch.empty_like(x), torch.empty_like(s)
        rc = N.VP_EUNSUP
        if T >= 1024 and Cc % 4 == 0:                # many positions per utterance (2-D feature maps): spread over the chip
            ws = _bytes(lib.vp_scale_rows_bwd_workspace_bytes(B, T, Cc), x.device)
            rc = lib.vp_scale_rows_bwd_ws_f32(hctx, dy.data_ptr(), x.data_ptr(), s.data_ptr(), B, T, Cc, dx.data_ptr(), ds.data_ptr(),
                                              ws.data_ptr(), ws.numel(), N.stream_ptr())
            if rc not in (0, N.VP_EUNSUP):
                _chk(rc, hctx)
        if rc == N.VP_EUNSUP:
            _chk(lib.vp_scale_rows_bwd_f32(hctx, dy.data_ptr(), x.data_ptr(), s.data_ptr(), B, T, Cc, dx.data_ptr(), ds.data_ptr(),
                                           N.stream_ptr()), hctx)
        return dx, ds, dy, None, None


class Res2Fn(torch.autograd.Function):
    """Res2NetBlock (ecapa_tdnn.py:11-47) as one tape entry: y_0 = x_0, y_1 = f_1(x_1), y_i = f_i(x_i + y_{i-1}), concat -- f_i a
    TDNNBlock (conv k3 'same' reflect -> ReLU -> BatchNorm).  Each chunk's BatchNorm pass writes y_i into its slice of the output and
    the next chunk's input y_i + x_{i+1} beside it; in backward each chunk's reflect-fold writes d x_i into its slice of d x and hands
    d y_{i-1} = d(input_i) + d out[:, i-1] on.  (As separate entries: 6 strided adds and a concatenation each way per block.)
    params: for chunks 1..scale-1: conv weight, conv bias, BN weight, BN bias, BN running mean, BN running variance."""

    @staticmethod
    def _fused_desc(x, out, cfg, params, S):
        d = N.Res2TrainDesc()
        d.B, d.T, d.C, d.scale, d.width, d.dil = cfg['B'], cfg['T'], x.shape[1], S, x.shape[1] // S, cfg['dilation']
        d.momentum, d.eps = cfg['momentum'], cfg['eps']
        d.x, d.out = x.data_ptr(), (out.data_ptr() if out is not None else None)
        d.x_is_bf16 = int(x.dtype == torch.bfloat16)
        for i in range(S - 1):
            wt, bs, g, b, rm, rv = params[6 * i:6 * i + 6]
            d.w[i], d.bias[i], d.gamma[i], d.beta[i] = wt.data_ptr(), bs.data_ptr(), g.data_ptr(), b.data_ptr()
            d.run_mean[i] = rm.data_ptr() if rm is not None else None
            d.run_var[i] = rv.data_ptr() if rv is not None else None
        return d

    @staticmethod
    def _fused_ok(x, cfg, params, S):
        """The one-launch-per-direction chain (csrc/res2_train.hip): enable_amp steps, 64-channel chunks, contiguous f32 parameters
        in the model's layout.  VPMI_RES2_TRAIN_UNFUSED=1 keeps the per-chunk launches (A/B, parity tests)."""
        if (not ppvector.get_train_amp() or os.environ.get('VPMI_RES2_TRAIN_UNFUSED') or not ppvector.get_fused_grid_kernels()
                or x.shape[1] != 64 * S or not 2 <= S <= 8):
            return False
        for i in range(S - 1):
            wt, bs, g, b = params[6 * i:6 * i + 4]
            if wt is None or bs is None or g is None or b is None or tuple(wt.shape) != (64, 64, 3):
                return False
            if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (wt, bs, g, b)):
                return False
        return True

    @staticmethod
    def forward(ctx, x, cfg, *params):
        x16 = _only16(x)                                      # the producer (tdnn1) wrote the block input as bf16 only: the fused kernel reads that
        B, T, S = cfg['B'], cfg['T'], cfg['scale']
        fused_ok = Res2Fn._fused_ok(x, cfg, params, S)
        if x16 is not None and not (fused_ok and x16.is_contiguous()):
            x, x16 = x16.float(), None                        # (the per-chunk path wants f32)
        elif x16 is None:
            x = _f32c(x)
        xk = x16 if x16 is not None else x                    # what the kernel reads
        w = x.shape[1] // S
        # out16_only (with bf16_twin): the caller's only consumer reads the bf16 copy -- the fused kernel then skips the f32 store and the tape
        # gets a memory-less placeholder (_placeholder); the per-chunk fallback below always writes f32
        only16 = bool(cfg.get('bf16_twin') and cfg.get('out16_only'))
        out = None if (only16 and fused_ok) else torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ctx.fused = False
        if fused_ok:
            lib, hctx = N.lib(), N.ctx(x.device)
            M = x.shape[0]
            z = torch.empty((S - 1, M, 64), dtype=torch.float32, device=x.device)
            inb = torch.empty((S - 1, M, 64), dtype=torch.bfloat16, device=x.device)
            stats = torch.empty((S - 1, 2, 64), dtype=torch.float32, device=x.device)
            d = Res2Fn._fused_desc(xk, out, cfg, params, S)
            outb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if cfg.get('bf16_twin') else None
            d.z, d.inb, d.stats = z.data_ptr(), inb.data_ptr(), stats.data_ptr()
            d.out_bf16 = outb.data_ptr() if outb is not None else None
            ws = _bytes(lib.vp_res2_train_workspace_bytes(B, S), x.device)
            rc = lib.vp_res2_train_fwd(hctx, C.byref(d), ws.data_ptr(), ws.numel(), N.stream_ptr())
            if rc == 0:
                ctx.save_for_backward(z, inb, stats, *[params[6 * i + k] for i in range(S - 1) for k in (0, 2)])
                ctx.fused, ctx.split, ctx.cfg = True, (S, w), dict(cfg)
                cfg['_twin'] = outb                                      # (handed to the caller, who hangs it on the returned tensor)
                if out is None:
                    cfg['_twin_only'] = True
                    return _placeholder(x.shape, x.device)
                return out
            if rc != N.VP_EUNSUP:
                _chk(rc, hctx)
            if out is None:
                out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            if x16 is not None:
                x = x16.float()
        out[:, :w].copy_(x[:, :w])
        inp = x[:, w:2 * w].contiguous()
        saved, meta = [], []
        for i in range(1, S):
            wt, bs, g, b, rm, rv = params[6 * (i - 1):6 * i]
            tp = _Tape((True,) * 9)
            ConvBlock.forward(tp, inp, wt, bs, None, g, b, rm, rv,
                              dict(B=B, T=T, dilation=cfg['dilation'], pad='reflect', relu=True, momentum=cfg['momentum'], eps=cfg['eps'],
                                   y_into=out[:, i * w:(i + 1) * w], aux_add=x[:, (i + 1) * w:(i + 2) * w] if i + 1 < S else None))
            inp = tp.aux_out
            saved.extend(tp.saved_tensors)
            meta.append((len(tp.saved_tensors), tp.geom))
        ctx.save_for_backward(*saved)
        ctx.meta, ctx.split = meta, (S, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        S, w = ctx.split
        if ctx.fused:
            lib, hctx = N.lib(), N.ctx(dout.device)
            cfg = ctx.cfg
            B, T = cfg['B'], cfg['T']
            z, inb, stats = ctx.saved_tensors[:3]
            wg = ctx.saved_tensors[3:]                                   # conv weight, BN weight per chunk
            M = dout.shape[0]
            dx = torch.empty_like(dout)
            dzb = torch.empty((S - 1, M, 64), dtype=torch.bfloat16, device=dout.device)
            dvec = torch.empty((S - 1, 3, 64), dtype=torch.float32, device=dout.device)
            d = N.Res2TrainDesc()
            d.B, d.T, d.C, d.scale, d.width, d.dil = B, T, dout.shape[1], S, w, cfg['dilation']
            d.momentum, d.eps = cfg['momentum'], cfg['eps']
            d.x, d.out = dout.data_ptr(), dx.data_ptr()
            for i in range(S - 1):
                d.w[i], d.gamma[i] = wg[2 * i].data_ptr(), wg[2 * i + 1].data_ptr()
            d.z, d.stats, d.dzb, d.dvec = z.data_ptr(), stats.data_ptr(), dzb.data_ptr(), dvec.data_ptr()
            ws = _bytes(lib.vp_res2_train_workspace_bytes(B, S), dout.device)
            _chk(lib.vp_res2_train_bwd(hctx, C.byref(d), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            grads = [None] * (6 * (S - 1))
            dw_all = torch.empty((S - 1, 64, 64, 3), dtype=torch.float32, device=dout.device)
            wd = N.Conv1dDesc()
            wd.dtype_in = wd.dtype_out = N.VP_BF16
            wd.B, wd.T_in, wd.T_out, wd.Cin, wd.Cout, wd.KW, wd.dilation, wd.stride = B, T, T, 64, 64, 3, cfg['dilation'], 1
            wd.pad_mode, wd.pad_left, wd.ldx, wd.xoff, wd.ldy, wd.mfma_bf16 = N.VP_PAD_REFLECT, cfg['dilation'], 64, 0, 64, 1
            wws = _bytes(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(wd)) * (S - 1), dout.device)
            wd.x, wd.w = inb.data_ptr(), wg[0].data_ptr()                # weight gradients dz_i^T in_i of all chunks in one launch
            _chk(lib.vp_conv1d_wgrad_bf16_oik_batched(hctx, C.byref(wd), dzb.data_ptr(), 64, dw_all.data_ptr(), S - 1, M * 64, M * 64,
                                                      wws.data_ptr(), wws.numel(), N.stream_ptr()), hctx)
            for i in range(S - 1):
                grads[6 * i:6 * i + 4] = [dw_all[i], dvec[i, 0], dvec[i, 1], dvec[i, 2]]
            return (dx, None, *grads)
        saved, at, tapes = ctx.saved_tensors, 0, []
        for n, geom in ctx.meta:
            tp = _Tape((True,) * 9)
            tp.saved_tensors, tp.geom = saved[at:at + n], geom
            tapes.append(tp)
            at += n
        dx = torch.empty_like(dout)
        dx[:, :w].copy_(dout[:, :w])
        dy = dout[:, (S - 1) * w:].contiguous()
        grads = [None] * (6 * (S - 1))
        for i in range(S - 1, 0, -1):
            r = _conv_block_bwd(tapes[i - 1], dy, None,
                                dict(dx=dx[:, i * w:(i + 1) * w], add=dout[:, (i - 1) * w:i * w] if i > 1 else None))
            dy = r[0]
            grads[6 * (i - 1):6 * (i - 1) + 4] = [r[1], r[2], r[4], r[5]]
        return (dx, None, *grads)


class _Tape:
    """What ConvBlock.forward / .backward need of an autograd context, for the functions that run a ConvBlock inside their own
    forward and backward."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _time_stats_launch(x, B, T, eps, unbiased, stats):
    """vp_time_stats_f32, or its many-frames flavour (frames spread over the chip) for the 2-D feature maps."""
    lib, hctx = N.lib(), N.ctx(x.device)
    Cc = x.shape[1]
    if T >= 1024 and Cc % 4 == 0:
        ws = _bytes(lib.vp_time_stats_workspace_bytes(B, T, Cc), x.device)
        rc = lib.vp_time_stats_ws_f32(hctx, x.data_ptr(), Cc, B, T, Cc, eps, int(unbiased), stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                      N.stream_ptr())
        if rc == 0:
            return
        if rc != N.VP_EUNSUP:
            _chk(rc, hctx)
    _chk(lib.vp_time_stats_f32(hctx, x.data_ptr(), Cc, B, T, Cc, eps, int(unbiased), stats.data_ptr(), N.stream_ptr()), hctx)


def _time_stats(x, B, T, eps, want_std):
    """[mean | std] (or the mean alone) over time of x (B*T, C): from the producing conv's fused sums when it left them on the tensor
    (`_vp_tsums`: ConvBlock with cfg['want_tsums']), else by a pass over x."""
    lib, hctx = N.lib(), N.ctx(x.device)
    Cc = x.shape[1]
    ts = getattr(x, '_vp_tsums', None)
    # (mixed precision only: the f32 engine is the parity instrument and takes its statistics from the tensor itself -- the
    # E[z^2] - E[z]^2 form of the fused sums costs it 3e-5 on the embeddings)
    if ts is not None and ppvector.get_train_amp() and ts[4] == B and ts[5] == T and ts[0].shape[1] == Cc and not os.environ.get('VPMI_NO_TSUMS'):
        ps, pq, scale, shift = ts[:4]
        stats = torch.empty((B, 2 * Cc if want_std else Cc), dtype=torch.float32, device=x.device)
        _chk(lib.vp_moments_finalize_affine(hctx, ps.data_ptr(), pq.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, T, Cc, eps,
                                            int(want_std), stats.data_ptr(), N.stream_ptr()), hctx)
        return stats
    stats = torch.empty((B, 2 * Cc), dtype=torch.float32, device=x.device)
    _time_stats_launch(x, B, T, eps, 0, stats)
    return stats if want_std else stats[:, :Cc].contiguous()


class SEBlockFn(torch.autograd.Function):
    """SEBlock (ecapa_tdnn.py:50-82, lengths=None) and the block residual (:139-141) as one tape entry:
    out = h * sigmoid(W2 relu(W1 mean_t(h) + b1) + b2) + res.  The two dense layers are ConvBlocks at T = 1 (same kernels, same
    mixed-precision flavour as everywhere else).  Backward in two passes over the big tensors: ds = sum_t dout * h, the dense
    layers' backward down to d mean, then dh = dout * s + d mean / T written once -- instead of dh = dout * s, a separate
    mean-backward tensor and autograd's sum of the two (8 tensor passes -> 4)."""

    @staticmethod
    def forward(ctx, h, res, w1, b1, w2, b2, B, T, shadow=None):
        lib, hctx = N.lib(), N.ctx(h.device)
        Cc = h.shape[1]
        h16 = _only16(h)
        res16 = getattr(res, '_vp_bf16', None)
        # all-bf16 form (enable_amp at scale): h exists as bf16 only (tdnn2's cfg['y_bf16']), the residual has a bf16 twin, and the block
        # output is written ONCE, as bf16, into its slice of the MFA operand -- 312 MB per block instead of 546
        all16 = (h16 is not None and res16 is not None and shadow is not None and res16.shape == h.shape and res16.stride(1) == 1
                 and shadow.stride(1) == 1 and Cc % 8 == 0 and res16.stride(0) % 8 == 0 and shadow.stride(0) % 8 == 0)
        if h16 is not None and not all16:
            raise N.VpmiError('SEBlockFn: an input that exists as bf16 only needs a bf16 residual twin and a bf16 output slice')
        if all16:
            if getattr(h, '_vp_tsums', None) is None or not ppvector.get_train_amp() or os.environ.get('VPMI_NO_TSUMS'):
                raise N.VpmiError('SEBlockFn: a bf16-only input needs its producer\'s fused time sums')
        else:
            res = _f32c(res)
            h = _f32c(h)                                      # (a contiguous tensor comes back as itself, with what its producer hung on it)
        mean = _time_stats(h, B, T, 1e-12, False)
        needs = (True,) * 9
        t1, t2 = _Tape(needs), _Tape(needs)
        H = w1.shape[0]
        fused = (not os.environ.get('VPMI_SE_DENSE_UNFUSED') and tuple(w1.shape) == (H, Cc, 1) and tuple(w2.shape) == (Cc, H, 1)
                 and Cc <= 1024 and H <= 1024 and Cc % 4 == 0 and H % 4 == 0 and b1 is not None and b2 is not None
                 and all(t.dtype == torch.float32 and t.is_contiguous() for t in (w1, b1, w2, b2)))
        if fused:                                    # the two dense layers: one launch (csrc/se_train.hip) instead of two GEMM launches
            a = torch.empty((B, H), dtype=torch.float32, device=h.device)
            s = torch.empty((B, Cc), dtype=torch.float32, device=h.device)
            _chk(lib.vp_se_dense_train_fwd(hctx, mean.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), B, Cc, H,
                                           int(ppvector.get_train_amp()), a.data_ptr(), s.data_ptr(), N.stream_ptr()), hctx)
        else:
            a = ConvBlock.forward(t1, mean, w1, b1, None, None, None, None, None, dict(B=B, T=1, relu=True))
            s = ConvBlock.forward(t2, a, w2, b2, None, None, None, None, None, dict(B=B, T=1, sigmoid=True))
        if all16:
            _chk(lib.vp_se_scale_residual(hctx, N.VP_BF16, h16.data_ptr(), h16.stride(0), 0, s.data_ptr(), res16.data_ptr(), res16.stride(0), 0,
                                          shadow.data_ptr(), shadow.stride(0), 0, B, T, Cc, N.stream_ptr()), hctx)
            out = _placeholder(h.shape, h.device)             # the caller hangs `shadow` on it as its only copy
            h = h16
        else:
            out = torch.empty_like(h)
        if all16:
            pass
        elif shadow is not None:                     # (B*T, Cc) bf16 view, unit column stride: the block output as the next GEMMs read it
            _chk(lib.vp_se_scale_residual_shadow(hctx, h.data_ptr(), Cc, 0, s.data_ptr(), res.data_ptr(), Cc, 0, out.data_ptr(), Cc, 0,
                                                 shadow.data_ptr(), shadow.stride(0), 0, B, T, Cc, N.stream_ptr()), hctx)
        else:
            _chk(lib.vp_se_scale_residual(hctx, N.VP_F32, h.data_ptr(), Cc, 0, s.data_ptr(), res.data_ptr(), Cc, 0, out.data_ptr(), Cc, 0,
                                          B, T, Cc, N.stream_ptr()), hctx)
        if fused:
            ctx.save_for_backward(h, s, mean, a, w1, w2)
            ctx.n1, ctx.geoms = -1, (B, T, int(ppvector.get_train_amp()), None)
        else:
            ctx.save_for_backward(h, s, *t1.saved_tensors, *t2.saved_tensors)
            ctx.n1 = len(t1.saved_tensors)
            ctx.geoms = (B, T, t1.geom, t2.geom)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        h, s = saved[0], saved[1]
        B, T, g1, g2 = ctx.geoms
        lib, hctx = N.lib(), N.ctx(h.device)
        dout = _f32c(dout)
        Cc = h.shape[1]
        ds = torch.empty_like(s)
        if h.dtype == torch.bfloat16:
            if not h.is_contiguous():
                h = h.contiguous()
            _chk(lib.vp_utt_dot_x16(hctx, dout.data_ptr(), h.data_ptr(), B, T, Cc, ds.data_ptr(), N.stream_ptr()), hctx)
        else:
            _chk(lib.vp_utt_dot_f32(hctx, dout.data_ptr(), h.data_ptr(), B, T, Cc, ds.data_ptr(), N.stream_ptr()), hctx)
        if ctx.n1 < 0:                               # the fused dense layers: d mean and the four parameter gradients in two launches
            mean, a, w1, w2 = saved[2:6]
            H = w1.shape[0]
            dm = torch.empty_like(mean)
            dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
            db1 = torch.empty(H, dtype=torch.float32, device=h.device)
            db2 = torch.empty(Cc, dtype=torch.float32, device=h.device)
            ws = _bytes(lib.vp_se_dense_train_bwd_workspace_bytes(B, Cc, H), h.device)
            _chk(lib.vp_se_dense_train_bwd(hctx, ds.data_ptr(), mean.data_ptr(), a.data_ptr(), s.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                           B, Cc, H, g1, dm.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                                           ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            dh = torch.empty(h.shape, dtype=torch.float32, device=h.device)
            _chk(lib.vp_scale_shift_rows_f32(hctx, dout.data_ptr(), s.data_ptr(), dm.data_ptr(), B, T, Cc, dh.data_ptr(), N.stream_ptr()), hctx)
            return dh, dout, dw1, db1, dw2, db2, None, None, None
        needs = (True,) * 9
        t1, t2 = _Tape(needs), _Tape(needs)
        t1.saved_tensors, t1.geom = saved[2:2 + ctx.n1], g1
        t2.saved_tensors, t2.geom = saved[2 + ctx.n1:], g2
        da, dw2, db2 = ConvBlock.backward(t2, ds)[:3]
        dm, dw1, db1 = ConvBlock.backward(t1, da)[:3]
        dh = torch.empty(h.shape, dtype=torch.float32, device=h.device)
        _chk(lib.vp_scale_shift_rows_f32(hctx, dout.data_ptr(), s.data_ptr(), dm.data_ptr(), B, T, Cc, dh.data_ptr(), N.stream_ptr()), hctx)
        return dh, dout, dw1, db1, dw2, db2, None, None, None


class ConvSEFn(torch.autograd.Function):
    """tdnn2 -> SEBlock -> + residual of an SE-Res2 block (ecapa_tdnn.py:125-142) as ONE tape entry, for the all-bf16 form of the mixed-
    precision step: conv (bf16 pre-BatchNorm output z, fused time sums) -> batch statistics -> [the BatchNorm apply is folded into the
    gate pass: h = BN(z) is never stored] -> squeeze mean from the fused sums -> the two dense layers -> out = h * s + res written once, as
    bf16, into its slice of the MFA operand.  Backward: ds = sum_t dout * h (h re-formed from z), the dense layers' backward down to
    dmean, then the conv's BatchNorm + ReLU backward reads dout and forms the SE block's input gradient dh = dout * s + dmean / T on the
    fly [dh is never stored].  As two entries (ConvBlock + SEBlockFn) the step wrote and re-read h (2 x 78 MB per block at 256 x 298
    frames) and dh (156 MB f32 written, read twice); the arithmetic is the same to the last bit
    (tests/test_gpu_train.py::test_conv_se_tail_as_one_tape_entry_changes_nothing).  VPMI_SE_TAIL_UNFUSED=1 keeps the two entries."""

    @staticmethod
    def usable(x, res, conv_w, w1, w2, b1, b2, shadow, B, T):
        Cc, H = conv_w.shape[0], w1.shape[0]
        Cin = conv_w.shape[1]
        res16 = getattr(res, '_vp_bf16', None)
        if not (B * T >= 4096 and Cin % 64 == 0 and Cin >= 256 and Cc >= 256):          # (the conv must take _wide_bf16's level 2)
            return False
        return bool(ppvector.get_train_amp() and not os.environ.get('VPMI_SE_TAIL_UNFUSED') and not os.environ.get('VPMI_SE_DENSE_UNFUSED')
                    and not os.environ.get('VPMI_NO_TSUMS') and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') == '2'
                    and shadow is not None and res16 is not None and tuple(res16.shape) == (B * T, Cc) and res16.stride(1) == 1
                    and shadow.stride(1) == 1 and Cc % 8 == 0 and res16.stride(0) % 8 == 0 and shadow.stride(0) % 8 == 0
                    and tuple(w1.shape) == (H, Cc, 1) and tuple(w2.shape) == (Cc, H, 1) and Cc <= 1024 and H <= 1024 and H % 4 == 0
                    and b1 is not None and b2 is not None and all(t.dtype == torch.float32 and t.is_contiguous() for t in (w1, b1, w2, b2))
                    and conv_w.shape[2] == 1 and lib_nseg_ok(T))

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, run_mean, run_var, res, w1, b1, w2, b2, cfg, shadow):
        lib, hctx = N.lib(), N.ctx(weight.device)
        B, T = cfg['B'], cfg['T']
        tp = _Tape((True,) * 9)
        c2 = dict(cfg, want_tsums=True, y_bf16=True, _no_apply=True)
        ConvBlock.forward(tp, x, weight, bias, None, gamma, beta, run_mean, run_var, c2)
        if c2.get('_zaff') is None or c2.get('_tsums') is None:
            raise N.VpmiError('ConvSEFn: the conv did not take the bf16 pre-BatchNorm form with fused time sums (check ConvSEFn.usable)')
        z, scale, shift = c2['_zaff']
        ps, pq = c2['_tsums'][:2]
        Cc, H = z.shape[1], w1.shape[0]
        dev = z.device
        mean = torch.empty((B, Cc), dtype=torch.float32, device=dev)          # squeeze: time mean of h from the conv's fused sums of z
        _chk(lib.vp_moments_finalize_affine(hctx, ps.data_ptr(), pq.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, T, Cc, 1e-12, 0,
                                            mean.data_ptr(), N.stream_ptr()), hctx)
        a = torch.empty((B, H), dtype=torch.float32, device=dev)
        s = torch.empty((B, Cc), dtype=torch.float32, device=dev)
        _chk(lib.vp_se_dense_train_fwd(hctx, mean.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), B, Cc, H,
                                       int(ppvector.get_train_amp()), a.data_ptr(), s.data_ptr(), N.stream_ptr()), hctx)
        res16 = res._vp_bf16
        _chk(lib.vp_se_scale_residual_z16(hctx, z.data_ptr(), Cc, scale.data_ptr(), shift.data_ptr(), s.data_ptr(), res16.data_ptr(),
                                          res16.stride(0), 0, shadow.data_ptr(), shadow.stride(0), 0, B, T, Cc, N.stream_ptr()), hctx)
        ctx.save_for_backward(*tp.saved_tensors, s, mean, a, w1, w2, scale, shift)
        ctx.n_conv = len(tp.saved_tensors)
        ctx.inner = (tp.geom, tp.wide, getattr(tp, 'wt16', None), tp.zero_dbias, B, T, int(ppvector.get_train_amp()))
        return _placeholder(z.shape, dev)                      # the caller hangs `shadow` on it as its only copy

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        tp = _Tape((True,) * 9)
        tp.saved_tensors = saved[:ctx.n_conv]
        tp.geom, tp.wide, tp.wt16, tp.zero_dbias, B, T, amp = ctx.inner
        s, mean, a, w1, w2, scale, shift = saved[ctx.n_conv:]
        z = tp.saved_tensors[2]
        lib, hctx = N.lib(), N.ctx(z.device)
        dout = _f32c(dout)
        Cc, H = z.shape[1], w1.shape[0]
        ds = torch.empty_like(s)
        _chk(lib.vp_utt_dot_z16(hctx, dout.data_ptr(), z.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, T, Cc, ds.data_ptr(),
                                N.stream_ptr()), hctx)
        dm = torch.empty_like(mean)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        db1 = torch.empty(H, dtype=torch.float32, device=z.device)
        db2 = torch.empty(Cc, dtype=torch.float32, device=z.device)
        ws = _bytes(lib.vp_se_dense_train_bwd_workspace_bytes(B, Cc, H), z.device)
        _chk(lib.vp_se_dense_train_bwd(hctx, ds.data_ptr(), mean.data_ptr(), a.data_ptr(), s.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                       B, Cc, H, amp, dm.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                                       ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
        r = _conv_block_bwd(tp, dout, utt=('se', s, dm, T))
        return r[0], r[1], r[2], r[4], r[5], None, None, dout, dw1, db1, dw2, db2, None, None


def lib_nseg_ok(T):
    """The conv GEMM's fused per-utterance sums exist for utterances of >= ~19 frames (at most 8 segments per 128-row tile)."""
    return N.lib().vp_conv1d_nseg(int(T)) <= 8


class TimeStats(torch.autograd.Function):
    """[mean | std] over time.  tstp=False: sqrt(clip(var_biased, 1e-12)) (ASP context, pooling.py:97-104);
    tstp=True: sqrt(var_unbiased + 1e-8) (TemporalStatsPool, pooling.py:128-146)."""

    @staticmethod
    def forward(ctx, x, B, T, tstp=False, eps=None):
        lib, hctx = N.lib(), N.ctx(x.device)
        x = _f32c(x)
        Cc = x.shape[1]
        eps = eps if eps is not None else (1e-8 if tstp else 1e-12)
        stats = torch.empty((B, 2 * Cc), dtype=torch.float32, device=x.device)
        _time_stats_launch(x, B, T, eps, int(tstp), stats)
        ctx.save_for_backward(x, stats)
        ctx.geom = (B, T, eps, int(tstp))
        return stats

    @staticmethod
    def backward(ctx, ds):
        x, stats = ctx.saved_tensors
        B, T, eps, tstp = ctx.geom
        lib, hctx = N.lib(), N.ctx(x.device)
        Cc = x.shape[1]
        dx = torch.empty_like(x)
        _chk(lib.vp_time_stats_bwd_f32(hctx, x.data_ptr(), Cc, stats.data_ptr(), _f32c(ds).data_ptr(), B, T, Cc, eps, tstp,
                                       dx.data_ptr(), Cc, N.stream_ptr()), hctx)
        return dx, None, None, None, None


class AttnStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, x, B, T):
        lib, hctx = N.lib(), N.ctx(x.device)
        e, x = _f32c(e), _f32c(x)
        Cc = x.shape[1]
        pooled = torch.empty((B, 2 * Cc), dtype=torch.float32, device=x.device)
        _chk(lib.vp_asp_softmax_stats(hctx, N.VP_F32, e.data_ptr(), x.data_ptr(), Cc, 0, B, T, Cc, 1e-12, pooled.data_ptr(),
                                      N.stream_ptr()), hctx)
        ctx.save_for_backward(e, x, pooled)
        ctx.geom = (B, T)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        e, x, pooled = ctx.saved_tensors
        B, T = ctx.geom
        lib, hctx = N.lib(), N.ctx(x.device)
        Cc = x.shape[1]
        de, dx = torch.empty_like(e), torch.empty_like(x)
        _chk(lib.vp_attn_stats_bwd_f32(hctx, e.data_ptr(), x.data_ptr(), Cc, pooled.data_ptr(), _f32c(dp).data_ptr(), B, T, Cc,
                                       1e-12, de.data_ptr(), dx.data_ptr(), Cc, N.stream_ptr()), hctx)
        return de, dx, None, None


def asp16_min_rows():
    """Rows (B * T) from which the MFA output, ASP's logits and their gradients take the all-bf16 form (VPMI_ASP16_MIN_ROWS for A/B)."""
    return int(os.environ.get('VPMI_ASP16_MIN_ROWS', '4096'))


def _asp_de16(B, T, Cc):
    """enable_amp at scale: the statistics' backward writes d e as bf16 (the operand the logits conv's two backward GEMMs round it to)."""
    return bool(ppvector.get_train_amp() and B * T >= asp16_min_rows() and Cc % 4 == 0 and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') != '0')


class AspFn(torch.autograd.Function):
    """AttentiveStatisticsPooling.forward with lengths=None (pooling.py:86-125) as one tape entry: x (B*T, C) -> (B, 2C).
    x has three consumers (the context statistics, the attention TDNN, the weighted statistics); as separate entries their
    three (B*T, C) gradients are written and then summed pairwise by autograd (8 tensor passes over 469 MB at C = 1536).  Here
    the TDNN's data-gradient conv adds the weighted statistics' gradient in its epilogue and the statistics' backward adds that
    sum in its own pass.  w: (att, 3C | C, 1) as stored; the 2C context columns act on a per-utterance constant, i.e. a
    per-utterance bias (rowbias)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, run_mean, run_var, w2, b2, cfg):
        lib, hctx = N.lib(), N.ctx(x.device)
        twin = getattr(x, '_vp_bf16', None) if getattr(x, '_vp_bf16_only', False) else None
        x16 = twin is not None                                # the producer wrote x as bf16 ONLY (ConvBlock cfg['y_bf16']): x itself is a placeholder
        if x16:
            ts = getattr(x, '_vp_tsums', None)
            x = twin
            x._vp_tsums = ts
        else:
            x = _f32c(x)                                      # (may carry the producing conv's fused time sums)
        B, T, gc = cfg['B'], cfg['T'], cfg['global_context']
        Cc = x.shape[1]
        needs = (True,) * 9
        t0, t1, t2 = _Tape(needs), _Tape(needs), _Tape(needs)
        stats = rowbias = None
        if x16 and not (gc and getattr(x, '_vp_tsums', None) is not None and _asp_de16(B, T, Cc) and T <= 320):
            raise N.VpmiError('AspFn: a bf16 input needs the producer\'s fused time sums, global context and the bf16 statistics kernels (T <= 320)')
        if gc:
            stats = _time_stats(x, B, T, 1e-12, True)
            rowbias = ConvBlock.forward(t0, stats, w[:, Cc:].contiguous(), None, None, None, None, None, None, dict(B=B, T=1))
        # (the x columns of the attention TDNN's weight: their bf16 panels come from the step's weight-prep launch when the backbone asked)
        h = ConvBlock.forward(t1, x, w[:, :Cc].contiguous() if gc else w, bias, rowbias, gamma, beta, run_mean, run_var,
                              dict(B=B, T=T, relu=True, tanh=True, momentum=cfg['momentum'], eps=cfg['eps'],
                                   _w16=_panels16(w, 0, Cc) if gc else None))
        # the logits' bias shifts every frame of an utterance alike and the softmax over time removes it: its gradient,
        # sum_t alpha_t (dalpha_t - S) = S - S, is exactly zero -- no 469 MB column-sum pass over d e for it
        # enable_amp: the logits leave their conv as bf16 (what Paddle's O1 conv hands the f32 softmax) -- 234 instead of 469 MB written
        # once and read twice (here and in backward); only where backward also writes d e as bf16 (_asp_de16)
        e = ConvBlock.forward(t2, h, w2, b2, None, None, None, None, None,
                              dict(B=B, T=T, zero_bias_grad=True, out_bf16=_asp_de16(B, T, Cc) and T <= 320))
        pooled = torch.empty((B, 2 * Cc), dtype=torch.float32, device=x.device)
        if e.dtype == torch.bfloat16:
            _chk(lib.vp_asp_softmax_stats_l16(hctx, e.data_ptr(), x.data_ptr(), N.VP_BF16 if x16 else N.VP_F32, Cc, 0, B, T, Cc, 1e-12,
                                              pooled.data_ptr(), N.stream_ptr()), hctx)
        elif x16:
            raise N.VpmiError('AspFn: bf16 x with f32 logits is not built')
        else:
            _chk(lib.vp_asp_softmax_stats(hctx, N.VP_F32, e.data_ptr(), x.data_ptr(), Cc, 0, B, T, Cc, 1e-12, pooled.data_ptr(),
                                          N.stream_ptr()), hctx)
        tapes = (t0, t1, t2) if gc else (t1, t2)
        ctx.save_for_backward(x, stats, e, pooled, *(t for tp in tapes for t in tp.saved_tensors))
        ctx.tape_meta = [(len(tp.saved_tensors), tp.geom, tp.zero_dbias, getattr(tp, 'wide', 0), getattr(tp, 'wt16', None)) for tp in tapes]
        ctx.geom = (B, T, gc)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        return _asp_backward(ctx, dp)[0]


def _asp_backward(ctx, dp, defer_ctx=False):
    """AspFn's backward.  defer_ctx: the context statistics' gradient is NOT added to d x here; it comes back as (stats, d stats) for the
    caller to fold into the producing layer's BatchNorm backward (MfaAspFn).  -> (AspFn's gradient tuple, (stats, dstats) or None)"""
    saved = ctx.saved_tensors
    x, stats, e, pooled = saved[:4]
    B, T, gc = ctx.geom
    lib, hctx = N.lib(), N.ctx(x.device)
    Cc = x.shape[1]
    tapes, at = [], 4
    for n, geom, zero_dbias, wide, wt16 in ctx.tape_meta:
        tp = _Tape((True,) * 9)
        tp.saved_tensors, tp.geom, tp.zero_dbias, tp.wide, tp.wt16 = saved[at:at + n], geom, zero_dbias, wide, wt16
        tapes.append(tp)
        at += n
    t2, t1 = tapes[-1], tapes[-2]
    de16 = _asp_de16(B, T, Cc)
    de = torch.empty_like(e, dtype=torch.bfloat16 if de16 else torch.float32)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    x16 = x.dtype == torch.bfloat16
    if e.dtype == torch.bfloat16 and not de16:
        raise N.VpmiError('AspFn: bf16 logits without a bf16 logit gradient (enable_amp changed between forward and backward?)')
    if e.dtype == torch.bfloat16:
        _chk(lib.vp_attn_stats_bwd_e16(hctx, e.data_ptr(), x.data_ptr(), N.VP_BF16 if x16 else N.VP_F32, Cc, pooled.data_ptr(),
                                       _f32c(dp).data_ptr(), B, T, Cc, 1e-12, de.data_ptr(), dx.data_ptr(), Cc, N.stream_ptr()), hctx)
    else:
        fn = lib.vp_attn_stats_bwd_de16 if de16 else lib.vp_attn_stats_bwd_f32
        _chk(fn(hctx, e.data_ptr(), x.data_ptr(), Cc, pooled.data_ptr(), _f32c(dp).data_ptr(), B, T, Cc, 1e-12, de.data_ptr(),
                dx.data_ptr(), Cc, N.stream_ptr()), hctx)
    dh, dw2, db2 = _conv_block_bwd(t2, de)[:3]
    dx, dwx, dbias, drb, dgamma, dbeta = _conv_block_bwd(t1, dh, dx)[:6]        # dx: TDNN's + the weighted statistics'
    dw = dwx
    deferred = None
    if gc:
        dstats, dwc = _conv_block_bwd(tapes[0], drb)[:2]
        if defer_ctx:
            deferred = (stats, dstats)
        else:
            fn = lib.vp_time_stats_bwd_add_x16 if x16 else lib.vp_time_stats_bwd_add_f32
            _chk(fn(hctx, x.data_ptr(), Cc, stats.data_ptr(), dstats.data_ptr(), B, T, Cc, 1e-12, 0,
                    dx.data_ptr(), Cc, dx.data_ptr(), Cc, N.stream_ptr()), hctx)
        dw = torch.cat([dwx, dwc], dim=1)
    return (dx, dw, dbias, dgamma, dbeta, None, None, dw2, db2, None), deferred


class MfaAspFn(torch.autograd.Function):
    """ECAPA's MFA TDNNBlock over the concatenated block outputs and the AttentiveStatisticsPooling behind it (ecapa_tdnn.py:262-267,
    pooling.py:86-125) as ONE tape entry, for the all-bf16 form of the mixed-precision step.  As two entries (CatConvBlock + AspFn) the
    pooling layer's three gradient contributions to the MFA output y -- weighted statistics, attention TDNN, context statistics -- were
    summed into one (B*T, 1536) f32 tensor by three passes; the third (read y, read + write the gradient: 1.17 GB at 256 x 298 frames)
    exists only to add alpha[b, c] + beta[b, c] * y.  Here that term is handed to the MFA layer's two BatchNorm-backward passes, which
    re-form y from the bf16 z they read anyway (_conv_block_bwd utt=('ctx', ...)).  VPMI_MFA_ASP_UNFUSED=1 keeps the two entries."""

    @staticmethod
    def forward(ctx, cfg, acfg, weight, bias, gamma, beta, run_mean, run_var, aw, abias, agamma, abeta, arm, arv, aw2, ab2, *xs):
        widths = [t.shape[1] for t in xs]
        xcat = cfg['xcat']
        tpM, tpA = _Tape((True,) * 9), _Tape((True,) * 10)
        y = ConvBlock.forward(tpM, xcat, weight, bias, None, gamma, beta, run_mean, run_var, cfg)
        if cfg.get('_y16') is None or cfg.get('_tsums') is None:
            raise N.VpmiError('MfaAspFn: the MFA conv did not take the all-bf16 form (bf16 output, fused time sums)')
        y._vp_bf16, y._vp_bf16_only, y._vp_tsums = cfg['_y16'], True, cfg['_tsums']
        scale, shift = cfg['_tsums'][2], cfg['_tsums'][3]
        pooled = AspFn.forward(tpA, y, aw, abias, agamma, abeta, arm, arv, aw2, ab2, acfg)
        nM, nA = len(tpM.saved_tensors), len(tpA.saved_tensors)
        ctx.save_for_backward(*tpM.saved_tensors, *tpA.saved_tensors, scale, shift)
        ctx.counts = (nM, nA)
        ctx.metaM = (tpM.geom, tpM.wide, widths, getattr(tpM, 'wt16', None))
        ctx.metaA = (tpA.tape_meta, tpA.geom)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        saved = ctx.saved_tensors
        nM, nA = ctx.counts
        tpM, tpA = _Tape((True,) * 9), _Tape((True,) * 10)
        tpM.saved_tensors, tpA.saved_tensors = saved[:nM], saved[nM:nM + nA]
        scale, shift = saved[nM + nA:]
        tpM.geom, tpM.wide, widths, tpM.wt16 = ctx.metaM
        tpM.out_splits = widths
        tpA.tape_meta, tpA.geom = ctx.metaA
        (dx, dw, dbias, dgamma, dbeta, _, _, dw2, db2, _), deferred = _asp_backward(tpA, dp, defer_ctx=True)
        utt = None
        if deferred is not None:
            stats, dstats = deferred
            B, T = tpA.geom[0], tpA.geom[1]
            Cc = stats.shape[1] // 2
            lib, hctx = N.lib(), N.ctx(stats.device)
            ab = torch.empty((2, B, Cc), dtype=torch.float32, device=stats.device)
            _chk(lib.vp_time_stats_bwd_coeffs(hctx, stats.data_ptr(), dstats.data_ptr(), B, T, Cc, 1e-12, ab.data_ptr(), N.stream_ptr()), hctx)
            utt = ('ctx', ab[0], ab[1], T, scale, shift)
        r = _conv_block_bwd(tpM, dx, utt=utt)
        dxs = r[0] if isinstance(r[0], tuple) else r[0].split(widths, dim=1)
        return (None, None, r[1], r[2], r[4], r[5], None, None, dw, dbias, dgamma, dbeta, None, None, dw2, db2, *dxs)


class BNRows(torch.autograd.Function):
    """BatchNorm1D with batch statistics on a (M, C) tensor [-> ReLU].  With the ReLU, its backward is folded into the two
    BatchNorm-backward passes (they re-evaluate x * scale + shift > 0: vp_col_sums_masked_f32 / vp_bn_relu_bwd_masked_f32), so neither
    the output nor a d(activation) tensor is kept (VPMI_BN_RELU_UNFOLDED=1: the three-pass form)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, relu=False):
        lib, hctx = N.lib(), N.ctx(x.device)
        x = _f32c(x)
        M, Cc = x.shape
        zeros, ones = _const(0.0, Cc, x.device), _const(1.0, Cc, x.device)
        sums = col_sums(x, x, zeros, ones)                     # [sum x | sum x^2]
        mean, invstd, scale, shift = (torch.empty(Cc, dtype=torch.float32, device=x.device) for _ in range(4))
        _chk(lib.vp_bn_train_finalize(hctx, sums[0].data_ptr(), sums[1].data_ptr(), 1, M, Cc, gamma.data_ptr(), beta.data_ptr(),
                                      run_mean.data_ptr() if run_mean is not None else None,
                                      run_var.data_ptr() if run_var is not None else None, momentum, eps, mean.data_ptr(),
                                      invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), N.stream_ptr()), hctx)
        y = torch.empty_like(x)
        _chk(lib.vp_affine_rows_f32(hctx, x.data_ptr(), Cc, scale.data_ptr(), shift.data_ptr(), M, Cc, y.data_ptr(), Cc, int(relu),
                                    N.stream_ptr()), hctx)
        fold = bool(relu) and Cc % 4 == 0 and not os.environ.get('VPMI_BN_RELU_UNFOLDED')
        ctx.fold = (scale, shift) if fold else None
        ctx.save_for_backward(x, mean, invstd, gamma, y if relu and not fold else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, invstd, gamma, yr = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(x.device)
        dy = _f32c(dy)
        M, Cc = x.shape
        fold = ctx.fold
        if fold is not None:
            ms, mh = fold
            sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
            ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cc), x.device)
            rc = lib.vp_col_sums_masked_f32(hctx, dy.data_ptr(), Cc, x.data_ptr(), Cc, mean.data_ptr(), invstd.data_ptr(), ms.data_ptr(),
                                            mh.data_ptr(), 0.0, M, Cc, sums.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr())
            if rc == N.VP_EUNSUP:                     # (misaligned views: materialise the mask the plain way)
                yr = torch.empty_like(x)
                _chk(lib.vp_affine_rows_f32(hctx, x.data_ptr(), Cc, ms.data_ptr(), mh.data_ptr(), M, Cc, yr.data_ptr(), Cc, 1, N.stream_ptr()), hctx)
                fold = None
            else:
                _chk(rc, hctx)
                dx = torch.empty_like(x)
                _chk(lib.vp_bn_relu_bwd_masked_f32(hctx, dy.data_ptr(), Cc, x.data_ptr(), Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                                   sums.data_ptr(), ms.data_ptr(), mh.data_ptr(), 0.0, M, Cc, dx.data_ptr(), Cc, 0, N.stream_ptr()), hctx)
                return dx, sums[1], sums[0], None, None, None, None, None
        if yr is not None:
            t = torch.empty_like(dy)
            _chk(lib.vp_act_bwd_f32(hctx, N.VP_ACT_RELU, dy.data_ptr(), yr.data_ptr(), dy.numel(), t.data_ptr(), N.stream_ptr()), hctx)
            dy = t
        sums = col_sums(dy, x, mean, invstd)
        dx = torch.empty_like(x)
        _chk(lib.vp_bn_relu_bwd_f32(hctx, dy.data_ptr(), Cc, x.data_ptr(), Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                    sums.data_ptr(), M, Cc, 0, dx.data_ptr(), Cc, N.stream_ptr()), hctx)
        return dx, sums[1], sums[0], None, None, None, None, None


class HeadLoss(torch.autograd.Function):
    """loss = AAMLoss(cosine_classifier(emb, W), labels); the kernels produce d emb and d W with the forward value.  Returns
    (loss, pred): pred (B int32, non-differentiable) = argmax of the un-margined cosines = what trainer.py:233-236 reads off
    outputs["logits"], or an empty tensor when the logits-tensor path ran (B > 128 or D != 192)."""

    @staticmethod
    def forward(ctx, emb, W, labels, margin, scale, label_smoothing, easy_margin):
        lib, hctx = N.lib(), N.ctx(emb.device)
        emb, W = _f32c(emb), _f32c(W)
        B, D = emb.shape
        Cc = W.shape[1]
        demb, dW = torch.empty_like(emb), torch.empty_like(W)
        loss = torch.empty(1, dtype=torch.float32, device=emb.device)
        if not emb.is_cuda:
            raise N.VpmiError('HeadLoss needs GPU tensors: the engine has no CPU fallback')
        lab = labels.to(device=emb.device, dtype=torch.int64).reshape(-1).contiguous()       # (labels may arrive on the host)
        rc = N.VP_EUNSUP
        pred = torch.empty((0,), dtype=torch.int32, device=emb.device)
        if B <= 128 and D == 192 and os.environ.get('VPMI_HEAD_UNTILED') is None:
            # class-tiled head (csrc/head_tiled.hip): no (B, C) cosine / gradient tensors -- 2 x 102 MB at 200 000 classes x 128 utterances
            pred = torch.empty((B,), dtype=torch.int32, device=emb.device)
            ws = _bytes(lib.vp_cosine_aam_tiled_bwd_workspace_bytes(B, D, Cc), emb.device)
            rc = lib.vp_cosine_aam_tiled_bwd(hctx, emb.data_ptr(), W.data_ptr(), lab.data_ptr(), B, D, Cc, float(margin), float(scale),
                                             float(label_smoothing), int(easy_margin), 1.0, demb.data_ptr(), dW.data_ptr(), loss.data_ptr(),
                                             pred.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr())
            if rc not in (N.VP_OK, N.VP_EUNSUP):
                _chk(rc, hctx)
            if rc != N.VP_OK:
                pred = torch.empty((0,), dtype=torch.int32, device=emb.device)
        if rc != N.VP_OK:
            ws = _bytes(lib.vp_cosine_aam_ce_bwd_workspace_bytes(B, D, Cc), emb.device)
            _chk(lib.vp_cosine_aam_ce_bwd(hctx, emb.data_ptr(), W.data_ptr(), lab.data_ptr(), B, D, Cc, float(margin), float(scale),
                                          float(label_smoothing), int(easy_margin), 1.0, demb.data_ptr(), dW.data_ptr(),
                                          loss.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
        ctx.save_for_backward(demb, dW)
        ctx.mark_non_differentiable(pred)
        return loss[0], pred

    @staticmethod
    def backward(ctx, g, _gp):
        demb, dW = ctx.saved_tensors
        return demb * g, dW * g, None, None, None, None, None


class CosineLogits(torch.autograd.Function):
    """cos = normalize(emb, axis=1) @ normalize(W, axis=0)  (SpeakerIdentification 'Cosine', models/fc.py:41-53)."""

    @staticmethod
    def forward(ctx, emb, W):
        lib, hctx = N.lib(), N.ctx(emb.device)
        emb, W = _f32c(emb), _f32c(W)
        B, D = emb.shape
        Cc = W.shape[1]
        logits = torch.empty((B, Cc), dtype=torch.float32, device=emb.device)
        ws = _bytes(lib.vp_cosine_logits_workspace_bytes(B, D, Cc), emb.device)
        _chk(lib.vp_cosine_logits_f32(hctx, emb.data_ptr(), W.data_ptr(), B, D, Cc, logits.data_ptr(), ws.data_ptr(), ws.numel(),
                                      N.stream_ptr()), hctx)
        ctx.save_for_backward(emb, W)
        return logits

    @staticmethod
    def backward(ctx, dcos):
        emb, W = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(emb.device)
        B, D = emb.shape
        Cc = W.shape[1]
        demb, dW = torch.empty_like(emb), torch.empty_like(W)
        ws = _bytes(lib.vp_cosine_logits_bwd_workspace_bytes(B, D, Cc), emb.device)
        _chk(lib.vp_cosine_logits_bwd(hctx, emb.data_ptr(), W.data_ptr(), _f32c(dcos).data_ptr(), B, D, Cc, demb.data_ptr(),
                                      dW.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
        return demb, dW


class AamCe(torch.autograd.Function):
    """loss = CrossEntropy(label_smoothing)(scale * margin(cos), labels)  (AAMLoss.forward, loss/aamloss.py:28-47)."""

    @staticmethod
    def forward(ctx, logits, labels, margin, scale, label_smoothing, easy_margin):
        lib, hctx = N.lib(), N.ctx(logits.device)
        logits = _f32c(logits)
        B, Cc = logits.shape
        lab = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        dl = torch.empty_like(logits)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        row = torch.empty(B, dtype=torch.float32, device=logits.device)
        _chk(lib.vp_aam_ce_bwd(hctx, logits.data_ptr(), lab.data_ptr(), B, Cc, float(margin), float(scale), float(label_smoothing),
                               int(bool(easy_margin)), 1.0, dl.data_ptr(), loss.data_ptr(), row.data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


class Dense(torch.autograd.Function):
    """y (M, N) = x (M, K) @ W (K, N) + bias in exact f32 (vp_dense_f32) with its three backward products -- the Linear
    output and the DenseLayer stages of the classifier head (models/fc.py:27-29, :36-37, :56-71); any N, K."""

    @staticmethod
    def forward(ctx, x, w_kn, bias):
        lib, hctx = N.lib(), N.ctx(x.device)
        x, w_kn = _f32c(x), _f32c(w_kn)
        M, K = x.shape
        Nn = w_kn.shape[1]
        y = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
        _chk(lib.vp_dense_f32(hctx, x.data_ptr(), K, w_kn.data_ptr(), 1, bias.data_ptr() if bias is not None else None, M, Nn, K,
                              N.VP_ACT_NONE, y.data_ptr(), Nn, N.stream_ptr()), hctx)
        ctx.save_for_backward(x, w_kn)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_kn = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(x.device)
        dy = _f32c(dy)
        M, K = x.shape
        Nn = w_kn.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)                                   # dy @ W^T: W read as [N' = K][K' = N]
            _chk(lib.vp_dense_f32(hctx, dy.data_ptr(), Nn, w_kn.data_ptr(), 0, None, M, K, Nn, N.VP_ACT_NONE, dx.data_ptr(), K,
                                  N.stream_ptr()), hctx)
        if ctx.needs_input_grad[1]:
            xt = x.t().contiguous()                                    # x^T @ dy
            dw = torch.empty_like(w_kn)
            _chk(lib.vp_dense_f32(hctx, xt.data_ptr(), M, dy.data_ptr(), 1, None, K, Nn, M, N.VP_ACT_NONE, dw.data_ptr(), Nn,
                                  N.stream_ptr()), hctx)
        db = col_sums(dy)[0] if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class MarginCe(torch.autograd.Function):
    """The AM / ARM / CE / SubCenter (/ AAM) losses over the head's logits (loss/amloss.py:14-25, armloss.py:14-31,
    celoss.py:11-19, subcenterloss.py:32-54): value and d loss / d logits from one launch (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, logits, labels, kind, K, margin, scale, label_smoothing, easy_margin):
        lib, hctx = N.lib(), N.ctx(logits.device)
        logits = _f32c(logits)
        B, CK = logits.shape
        lab = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        dl = torch.empty_like(logits)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        row = torch.empty(B, dtype=torch.float32, device=logits.device)
        _chk(lib.vp_margin_ce_bwd(hctx, logits.data_ptr(), lab.data_ptr(), B, CK // K, K, kind, float(margin), float(scale),
                                  float(label_smoothing), int(bool(easy_margin)), 1.0, dl.data_ptr(), loss.data_ptr(), row.data_ptr(),
                                  N.stream_ptr()), hctx)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None, None, None


class SphereFace2Fn(torch.autograd.Function):
    """SphereFace2.forward (loss/sphereface2.py:47-69) with its gradients w.r.t. the logits and the bias parameter."""

    @staticmethod
    def forward(ctx, logits, labels, bias, margin, scale, lanbuda, t, type_a):
        lib, hctx = N.lib(), N.ctx(logits.device)
        logits, bias = _f32c(logits), _f32c(bias)
        B, Cc = logits.shape
        lab = labels.to(device=logits.device, dtype=torch.int64).reshape(-1).contiguous()
        dl = torch.empty_like(logits)
        out = torch.empty(2 + 2 * B, dtype=torch.float32, device=logits.device)       # loss, dbias, row_loss, row_dbias
        _chk(lib.vp_sphereface2(hctx, logits.data_ptr(), lab.data_ptr(), bias.data_ptr(), B, Cc, float(margin), float(scale),
                                float(lanbuda), int(t), int(bool(type_a)), 1.0, out.data_ptr(), out[2:].data_ptr(), dl.data_ptr(),
                                out[1:].data_ptr(), out[2 + B:].data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(dl, out)
        ctx.bias_shape = bias.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        dl, out = ctx.saved_tensors
        return dl * g, None, (out[1] * g).reshape(ctx.bias_shape), None, None, None, None, None


class Conv2dBlock(torch.autograd.Function):
    """2-D conv over (B, T, F, C) positions (zero padding (k-1)/2, stride s on both axes) [-> BatchNorm (batch statistics)]
    [-> ReLU]: the Conv2D -> BatchNorm2D -> ReLU units of models/resnet_se.py:8-45,72-74 (BN BEFORE the ReLU, unlike TDNNBlock).
    x (B*T*F, Cin) f32, weight (Cout, Cin, kF, kT) as stored by the reference."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, run_mean, run_var, cfg):
        lib, hctx = N.lib(), N.ctx(x.device)
        x, weight = _f32_rows(x), _f32c(weight)
        B, T, Fq = cfg['B'], cfg['T'], cfg['F']
        st, sf = cfg.get('stride_t', cfg.get('stride', 1)), cfg.get('stride_f', cfg.get('stride', 1))
        dil = cfg.get('dilation', 1)                       # along time
        Cout, Cin, KF, KT = weight.shape
        pad, padf = dil * (KT - 1) // 2, (KF - 1) // 2
        To, Fo = (T + 2 * pad - dil * (KT - 1) - 1) // st + 1, (Fq + 2 * padf - (KF - 1) - 1) // sf + 1
        s = (st, sf, dil, padf)
        act = {None: 0, 'relu': N.VP_ACT_RELU, 'hardtanh': N.VP_ACT_HARDTANH20, 'silu': N.VP_ACT_SILU, 'tanh': N.VP_ACT_TANH}[
            cfg.get('act', 'relu' if cfg.get('relu', False) else None)]
        relu, bn = act != 0, gamma is not None
        w2 = None
        if KT * KF > 1:     # the forward panel (Cout, kt, kf, Cin) and the data-gradient one (Cin, reversed kt, reversed kf, Cout) in one launch
            wp = torch.empty((Cout, KT * KF * Cin), dtype=torch.float32, device=x.device)
            if ctx.needs_input_grad[0]:
                w2 = torch.empty((Cin, KT * KF * Cout), dtype=torch.float32, device=x.device)
            _chk(lib.vp_conv2d_weight_layouts_f32(hctx, weight.data_ptr(), Cout, Cin, KF, KT, wp.data_ptr(),
                                                  w2.data_ptr() if w2 is not None else None, N.stream_ptr()), hctx)
        else:
            wp = weight.view(Cout, Cin)
        z = torch.empty((B * To * Fo, Cout), dtype=torch.float32, device=x.device)
        d = _conv_desc(x, B, T, To, Cin, Cout, KT * KF, dil, N.VP_PAD_ZERO, pad, wp, bias)
        d.F_in, d.F_out, d.KF, d.stride, d.stride_f, d.pad_f = Fq, Fo, KF, st, sf, padf
        d.y = z.data_ptr()
        _chk(lib.vp_conv1d_fwd(hctx, C.byref(d), N.stream_ptr()), hctx)
        mean = invstd = None
        y = z
        clamp = 0
        if bn:
            zeros, ones = _const(0.0, Cout, x.device), _const(1.0, Cout, x.device)
            sums = col_sums(z, z, zeros, ones)
            mean, invstd, scale, shift = (torch.empty(Cout, dtype=torch.float32, device=x.device) for _ in range(4))
            _chk(lib.vp_bn_train_finalize(hctx, sums[0].data_ptr(), sums[1].data_ptr(), 1, z.shape[0], Cout, gamma.data_ptr(),
                                          beta.data_ptr(), run_mean.data_ptr() if run_mean is not None else None,
                                          run_var.data_ptr() if run_var is not None else None, cfg.get('momentum', 0.9),
                                          cfg.get('eps', 1e-5), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                          shift.data_ptr(), N.stream_ptr()), hctx)
            # ReLU and Hardtanh(0, 20) (ERes2Net's "ReLU", eres2net.py:14-22) leave with the apply pass; the other activations get their own
            clamp = {N.VP_ACT_RELU: 1, N.VP_ACT_HARDTANH20: 2}.get(act, 0)
            if clamp == 2 and os.environ.get('VPMI_BN_RELU_UNFOLDED'):
                clamp = 0
            y = torch.empty_like(z)
            _chk(lib.vp_affine_rows_f32(hctx, z.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), z.shape[0], Cout, y.data_ptr(),
                                        Cout, clamp, N.stream_ptr()), hctx)
        pre = y                                       # the activation's input (SiLU's backward needs it)
        if act and not (bn and clamp):
            y = torch.empty_like(pre)
            _chk(lib.vp_act_f32(hctx, act, pre.data_ptr(), pre.numel(), y.data_ptr(), N.stream_ptr()), hctx)
        # BatchNorm -> ReLU / Hardtanh: the backward folds the activation's mask into the BatchNorm-backward passes (they re-evaluate
        # 0 < z * scale + shift [< 20]); then the output is not kept for backward
        fold = bn and clamp and Cout % 4 == 0 and not os.environ.get('VPMI_BN_RELU_UNFOLDED')
        ctx.fold = (scale, shift, 20.0 if clamp == 2 else 0.0) if fold else None
        ctx.save_for_backward(x, weight, z, mean, invstd, gamma, None if fold else ((pre if act == N.VP_ACT_SILU else y) if act else None), w2)
        ctx.geom = (B, T, Fq, To, Fo, Cin, Cout, KT, KF, s, pad, act, bn, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, mean, invstd, gamma, yr, w2 = ctx.saved_tensors
        B, T, Fq, To, Fo, Cin, Cout, KT, KF, s, pad, relu, bn, has_bias = ctx.geom
        lib, hctx = N.lib(), N.ctx(x.device)
        dev = x.device
        dy = _f32c(dy)
        M = B * To * Fo
        fold = getattr(ctx, 'fold', None)
        if fold is not None:                          # BatchNorm -> ReLU with the mask folded into the two passes below
            ms, mh, hi = fold
            sums = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cout), dev)
            rc = lib.vp_col_sums_masked_f32(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mean.data_ptr(), invstd.data_ptr(), ms.data_ptr(),
                                            mh.data_ptr(), hi, M, Cout, sums.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr())
            if rc == N.VP_EUNSUP:                     # (misaligned views: materialise the mask the plain way)
                yr = torch.empty_like(z)
                _chk(lib.vp_affine_rows_f32(hctx, z.data_ptr(), Cout, ms.data_ptr(), mh.data_ptr(), M, Cout, yr.data_ptr(), Cout,
                                            2 if hi else 1, N.stream_ptr()), hctx)
                fold = None
            else:
                _chk(rc, hctx)
        if fold is not None:
            dgamma, dbeta = sums[1], sums[0]
            dz = torch.empty_like(dy)
            _chk(lib.vp_bn_relu_bwd_masked_f32(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mean.data_ptr(), invstd.data_ptr(),
                                               gamma.data_ptr(), sums.data_ptr(), ms.data_ptr(), mh.data_ptr(), hi, M, Cout, dz.data_ptr(), Cout,
                                               0, N.stream_ptr()), hctx)
            bn = False                                # (done)
        elif relu:                                    # `relu` holds the activation code here
            t = torch.empty_like(dy)
            _chk(lib.vp_act_bwd_f32(hctx, relu, dy.data_ptr(), yr.data_ptr(), dy.numel(), t.data_ptr(), N.stream_ptr()), hctx)
            dy = t
        if fold is None:
            dgamma = dbeta = None
            dz = dy
        if bn:
            sums = col_sums(dy, z, mean, invstd)
            dgamma, dbeta = sums[1], sums[0]
            dz = torch.empty_like(dy)
            _chk(lib.vp_bn_relu_bwd_f32(hctx, dy.data_ptr(), Cout, z.data_ptr(), Cout, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                        sums.data_ptr(), M, Cout, 0, dz.data_ptr(), Cout, N.stream_ptr()), hctx)
        # a conv bias directly in front of a train-mode BatchNorm: sum_rows dz is IDENTICALLY zero (dz = gamma invstd (dy - mean(dy) - zhat
        # mean(dy zhat)), sum zhat = 0) -- what a column-sum pass returns there is rounding noise (1e-9 of the gradient scale, in the
        # reference's autograd as here); the pass is not launched.  VPMI_BN_BIAS_GRAD_SUMS=1 runs it.
        if has_bias and ctx.geom[12] and not os.environ.get('VPMI_BN_BIAS_GRAD_SUMS'):
            dbias = torch.zeros(Cout, dtype=torch.float32, device=dev)
        else:
            dbias = col_sums(dz)[0] if has_bias else None
        st, sf, dil, padf = s
        d = _conv_desc(x, B, T, To, Cin, Cout, KT * KF, dil, N.VP_PAD_ZERO, pad, weight)
        d.F_in, d.F_out, d.KF, d.stride, d.stride_f, d.pad_f = Fq, Fo, KF, st, sf, padf
        dW = torch.empty((Cout, Cin, KF, KT), dtype=torch.float32, device=dev)       # reduced straight into the model's layout
        ws = _bytes(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(d)), dev)
        _chk(lib.vp_conv1d_wgrad_oik_f32(hctx, C.byref(d), dz.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
        dx = None
        if ctx.needs_input_grad[0]:
            src = dz
            strided = st > 1 or sf > 1
            if strided:                                 # zero-insertion: the strided data gradient as a stride-1 conv
                src = torch.empty((B * T * Fq, Cout), dtype=torch.float32, device=dev)
                _chk(lib.vp_zero_insert_2d_f32(hctx, dz.data_ptr(), B, To, Fo, Cout, T, Fq, st, sf, src.data_ptr(), N.stream_ptr()), hctx)
            if w2 is None:                              # 1x1: W^T
                w2 = weight.view(Cout, Cin).t().contiguous()
            dx = torch.empty((B * T * Fq, Cin), dtype=torch.float32, device=dev)
            Ts, Fs = (T, Fq) if strided else (To, Fo)
            d2 = _conv_desc(src, B, Ts, T, Cout, Cin, KT * KF, dil, N.VP_PAD_ZERO, dil * (KT - 1) - pad, w2)
            d2.F_in, d2.F_out, d2.KF, d2.stride, d2.stride_f, d2.pad_f = Fs, Fq, KF, 1, 1, KF - 1 - padf
            d2.y = dx.data_ptr()
            _chk(lib.vp_conv1d_fwd(hctx, C.byref(d2), N.stream_ptr()), hctx)
        return dx, dW, dbias, dgamma, dbeta, None, None, None


class AffCombine(torch.autograd.Function):
    """o = x (1 + t) + y (1 - t)  (AFF, models/eres2net.py:48-51; t = tanh of the local attention)."""

    @staticmethod
    def forward(ctx, t, x, y):
        lib, hctx = N.lib(), N.ctx(x.device)
        t, x, y = _f32c(t), _f32c(x), _f32c(y)
        out = torch.empty_like(x)
        _chk(lib.vp_aff_combine_f32(hctx, t.data_ptr(), x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(t, x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        t, x, y = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(x.device)
        g = _f32c(g)
        dx, dy, dt = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        _chk(lib.vp_aff_combine_bwd_f32(hctx, g.data_ptr(), t.data_ptr(), x.data_ptr(), y.data_ptr(), x.numel(), dx.data_ptr(), dy.data_ptr(),
                                        dt.data_ptr(), N.stream_ptr()), hctx)
        return dt, dx, dy


class Act(torch.autograd.Function):
    """Elementwise activation with its backward in libvpmi ('relu' | 'hardtanh' = clamp(0, 20))."""

    @staticmethod
    def forward(ctx, x, kind):
        lib, hctx = N.lib(), N.ctx(x.device)
        x = _f32c(x)
        code = {'relu': N.VP_ACT_RELU, 'hardtanh': N.VP_ACT_HARDTANH20}[kind]
        y = torch.empty_like(x)
        _chk(lib.vp_act_f32(hctx, code, x.data_ptr(), x.numel(), y.data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(y)
        ctx.code = code
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(y.device)
        g = _f32c(g)
        dz = torch.empty_like(g)
        _chk(lib.vp_act_bwd_f32(hctx, ctx.code, g.data_ptr(), y.data_ptr(), g.numel(), dz.data_ptr(), N.stream_ptr()), hctx)
        return dz, None


class SegCtx(torch.autograd.Function):
    """ctx[b, s] = mean_t x[b] + mean over 100-frame segment s of x[b]  (CAMLayer context, models/campplus.py:88-106)."""

    @staticmethod
    def forward(ctx, x, B, T, seg_len):
        lib, hctx = N.lib(), N.ctx(x.device)
        x = _f32c(x)
        Cc = x.shape[1]
        nseg = (T + seg_len - 1) // seg_len
        out = torch.empty((B * nseg, Cc), dtype=torch.float32, device=x.device)
        _chk(lib.vp_seg_ctx_f32(hctx, x.data_ptr(), B, T, Cc, seg_len, out.data_ptr(), N.stream_ptr()), hctx)
        ctx.geom = (B, T, Cc, seg_len)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, Cc, seg_len = ctx.geom
        lib, hctx = N.lib(), N.ctx(g.device)
        g = _f32c(g)
        dx = torch.empty((B * T, Cc), dtype=torch.float32, device=g.device)
        _chk(lib.vp_seg_ctx_bwd_f32(hctx, g.data_ptr(), B, T, Cc, seg_len, dx.data_ptr(), N.stream_ptr()), hctx)
        return dx, None, None, None


class SegScale(torch.autograd.Function):
    """out[b, t] = y[b, t] * m[b, seg(t)]  (CAMLayer gate, models/campplus.py:94)."""

    @staticmethod
    def forward(ctx, y, m, B, T, seg_len):
        lib, hctx = N.lib(), N.ctx(y.device)
        y, m = _f32c(y), _f32c(m)
        Cc = y.shape[1]
        out = torch.empty_like(y)
        _chk(lib.vp_seg_scale_f32(hctx, y.data_ptr(), m.data_ptr(), B, T, Cc, seg_len, out.data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(y, m)
        ctx.geom = (B, T, seg_len)
        return out

    @staticmethod
    def backward(ctx, g):
        y, m = ctx.saved_tensors
        B, T, seg_len = ctx.geom
        lib, hctx = N.lib(), N.ctx(y.device)
        g = _f32c(g)
        dy, dm = torch.empty_like(y), torch.empty_like(m)
        _chk(lib.vp_seg_scale_bwd_f32(hctx, g.data_ptr(), y.data_ptr(), m.data_ptr(), B, T, y.shape[1], seg_len, dy.data_ptr(), dm.data_ptr(),
                                      N.stream_ptr()), hctx)
        return dy, dm, None, None, None


class CamLayerFn(torch.autograd.Function):
    """CAMLayer (models/campplus.py:88-95) as one tape entry: out = linear_local(h) * sigmoid(linear2(relu(linear1(mean_t h + segment means of h)))).
    The local conv is a ConvBlock (same kernels as everywhere); everything behind it is vp_cam_gate_fwd_f32 forward (one launch for the
    segment means, both dense layers and the gate) and vp_cam_gate_bwd_f32 + vp_cam_gate_wgrad_f32 backward; the gradient that reaches h
    through the context is added in the local conv's data-gradient epilogue, the conv's bias gradient comes out of the gate's backward pass.
    h (B*T, C) f32; weights as the reference stores them: wl (O, C, k), w1 (H, C, 1), w2 (O, H, 1)."""

    @staticmethod
    def usable(h, wl, w1, w2, T, seg_len):
        Cc, O, H = wl.shape[1], wl.shape[0], w1.shape[0]
        nseg = (T + seg_len - 1) // seg_len
        # (the kernels' layout: float4 rows, C <= 1024, O <= 128, everything of one utterance's context in 128 KB of LDS)
        return (h.dtype == torch.float32 and Cc % 4 == 0 and Cc <= 1024 and O % 4 == 0 and O <= 128
                and (nseg * (Cc + H + 2 * O) + Cc + 16 * O + 8192 + 16) * 4 <= 128 * 1024 and not os.environ.get('VPMI_CAM_LAYER_UNFUSED'))

    @staticmethod
    def forward(ctx, h, wl, bl, w1, b1, w2, b2, cfg):
        lib, hctx = N.lib(), N.ctx(h.device)
        B, T, seg_len = cfg['B'], cfg['T'], cfg['seg_len']
        h = _f32c(h)
        dev = h.device
        O, Cc, _ = wl.shape
        H = w1.shape[0]
        nseg = (T + seg_len - 1) // seg_len
        tape = _Tape((ctx.needs_input_grad[0], True, bl is not None, False, False, False, False, False, False))
        y = ConvBlock.forward(tape, h, wl, bl, None, None, None, None, None, dict(B=B, T=T, dilation=cfg.get('dilation', 1), pad='zero'))
        w1c, w2c = _f32c(w1).view(H, Cc), _f32c(w2).view(O, H)
        cx = torch.empty((B * nseg, Cc), dtype=torch.float32, device=dev)
        hid = torch.empty((B * nseg, H), dtype=torch.float32, device=dev)
        m = torch.empty((B * nseg, O), dtype=torch.float32, device=dev)
        out = cfg.get('out_into')                    # a column slice of a DenseNet block's buffer (CamDenseBlockFn), or a tensor of its own
        if out is None:
            out = torch.empty((B * T, O), dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (B * T, O) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError('CamLayerFn: out_into does not match the layer output')
        _chk(lib.vp_cam_gate_fwd_f32(hctx, h.data_ptr(), Cc, y.data_ptr(), O, w1c.data_ptr(), b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(),
                                     B, T, Cc, H, O, seg_len, cx.data_ptr(), hid.data_ptr(), m.data_ptr(), out.data_ptr(), out.stride(0),
                                     N.stream_ptr()), hctx)
        ctx.tape = tape
        ctx.save_for_backward(y, cx, hid, m, w1c, w2c)
        ctx.geom = (B, T, Cc, H, O, seg_len, nseg, bl is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        y, cx, hid, m, w1c, w2c = ctx.saved_tensors
        B, T, Cc, H, O, seg_len, nseg, has_bl = ctx.geom
        lib, hctx = N.lib(), N.ctx(y.device)
        dev = y.device
        if g.dtype != torch.float32 or g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
            g = _f32c(g)
        ldg = g.stride(0)                        # (a column slice of the DenseNet concatenation's gradient keeps its pitch: no copy)
        dy = torch.empty_like(y)
        dp1 = torch.empty((B * nseg, H), dtype=torch.float32, device=dev)
        dp2 = torch.empty((B * nseg, O), dtype=torch.float32, device=dev)
        dh_ctx = torch.empty((B * T, Cc), dtype=torch.float32, device=dev)
        dyb = torch.empty((B, O), dtype=torch.float32, device=dev)
        _chk(lib.vp_cam_gate_bwd_f32(hctx, g.data_ptr(), ldg, y.data_ptr(), O, hid.data_ptr(), m.data_ptr(), w1c.data_ptr(), w2c.data_ptr(),
                                     B, T, Cc, H, O, seg_len, dy.data_ptr(), O, dp1.data_ptr(), dp2.data_ptr(), dh_ctx.data_ptr(), Cc,
                                     dyb.data_ptr(), N.stream_ptr()), hctx)
        dw1 = torch.empty((H, Cc, 1), dtype=torch.float32, device=dev)
        dw2 = torch.empty((O, H, 1), dtype=torch.float32, device=dev)
        db1 = torch.empty(H, dtype=torch.float32, device=dev)
        db2 = torch.empty(O, dtype=torch.float32, device=dev)
        dbl = torch.empty(O, dtype=torch.float32, device=dev) if has_bl else None
        _chk(lib.vp_cam_gate_wgrad_f32(hctx, dp1.data_ptr(), dp2.data_ptr(), cx.data_ptr(), hid.data_ptr(), dyb.data_ptr(), B, nseg, Cc, H, O,
                                       dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), dbl.data_ptr() if has_bl else None,
                                       N.stream_ptr()), hctx)
        tape = ctx.tape
        tape.given_dbias = dbl
        r = _conv_block_bwd(tape, dy, skip=dh_ctx)
        tape.given_dbias = None              # (a second reference would make autograd COPY the gradient into .grad instead of adopting it)
        del dbl
        return r[0], r[1], r[2], dw1, db1, dw2, db2, None


class CamDenseBlockFn(torch.autograd.Function):
    """CAMDenseTDNNBlock (models/campplus.py:137-171: x = concat([x, layer(x)]) layer after layer) on ONE preallocated (B*T, C_final) buffer:
    a layer reads its input as the first C_l columns of the buffer IN PLACE (row pitch C_final) and its CAMLayer writes its 32 output
    channels into the next column slice -- no torch.cat; backward walks the layers in reverse over ONE gradient buffer: a layer's output
    gradient is a column slice of it, and the layer's BatchNorm + ReLU backward ADDS its input gradient into the first C_l columns
    (vp_bn_relu_bwd_masked_f32, accumulate) -- no narrow copies, none of autograd's strided adds of the concatenation's gradient.
    Per layer (nonlinear1 -> linear1 -> nonlinear2 -> CAMLayer, campplus.py:108-134) the kernels are the ones the per-layer tape runs
    (BNRows' passes, ConvBlock, CamLayerFn).  params: 12 per layer -- bn1 weight, bias; linear1 weight, bias; bn2 weight, bias;
    cam_layer.linear_local weight, bias; .linear1 weight, bias; .linear2 weight, bias.  cfg['bufs'][l] = (bn1 running mean, variance,
    momentum, eps, bn2 running mean, variance, momentum, eps, dilation)."""

    @staticmethod
    def usable(x, layers, T, seg_len):
        if os.environ.get('VPMI_CAM_BLOCK_UNFUSED') or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] % 4:
            return False
        for lay in layers:
            cl = lay.cam_layer
            wl = cl.linear_local.weight
            if wl.shape[0] % 4 or lay.linear1.bias is None or cl.linear_local.bias is None:
                return False
            if not CamLayerFn.usable(x, wl, cl.linear1.weight, cl.linear2.weight, T, seg_len):
                return False
        return True

    @staticmethod
    def forward(ctx, x0, cfg, *params):
        lib, hctx = N.lib(), N.ctx(x0.device)
        B, T, seg_len, bufs = cfg['B'], cfg['T'], cfg['seg_len'], cfg['bufs']
        x0 = _f32c(x0)
        dev = x0.device
        M, C0 = x0.shape
        L = len(params) // 12
        G = params[6].shape[0]
        Cf = C0 + L * G
        X = torch.empty((M, Cf), dtype=torch.float32, device=dev)
        X[:, :C0].copy_(x0)
        tapes = []
        for l in range(L):
            g1, be1, wl1, bl1, g2, be2, wloc, bloc, w1, b1, w2, b2 = params[12 * l:12 * l + 12]
            rm1, rv1, mom1, eps1, rm2, rv2, mom2, eps2, dil = bufs[l]
            Cl = C0 + l * G
            # nonlinear1: BatchNorm (batch statistics) -> ReLU over the first Cl columns, read where they are
            zeros, ones = _const(0.0, Cl, dev), _const(1.0, Cl, dev)
            sums = torch.empty((2, Cl), dtype=torch.float32, device=dev)
            ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cl), dev)
            _chk(lib.vp_col_sums_f32(hctx, X.data_ptr(), Cf, X.data_ptr(), Cf, zeros.data_ptr(), ones.data_ptr(), M, Cl, sums.data_ptr(),
                                     ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            mean, invstd, scale, shift = (torch.empty(Cl, dtype=torch.float32, device=dev) for _ in range(4))
            _chk(lib.vp_bn_train_finalize(hctx, sums[0].data_ptr(), sums[1].data_ptr(), 1, M, Cl, g1.data_ptr(), be1.data_ptr(),
                                          rm1.data_ptr() if rm1 is not None else None, rv1.data_ptr() if rv1 is not None else None, mom1, eps1,
                                          mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), N.stream_ptr()), hctx)
            h0 = torch.empty((M, Cl), dtype=torch.float32, device=dev)
            _chk(lib.vp_affine_rows_f32(hctx, X.data_ptr(), Cf, scale.data_ptr(), shift.data_ptr(), M, Cl, h0.data_ptr(), Cl, 1,
                                        N.stream_ptr()), hctx)
            t2, t3, t4 = _Tape((True,) * 9), _Tape((True,) * 8), _Tape((True,) * 8)
            # (linear1's bias sits directly in front of nonlinear2's BatchNorm: its gradient is identically zero, see Conv2dBlock.backward)
            z1 = ConvBlock.forward(t2, h0, wl1, bl1, None, None, None, None, None,
                                   dict(B=B, T=T, zero_bias_grad=not os.environ.get('VPMI_BN_BIAS_GRAD_SUMS')))
            h = BNRows.forward(t3, z1, g2, be2, rm2, rv2, mom2, eps2, True)
            CamLayerFn.forward(t4, h, wloc, bloc, w1, b1, w2, b2, dict(B=B, T=T, seg_len=seg_len, dilation=dil, out_into=X[:, Cl:Cl + G]))
            tapes.append((Cl, mean, invstd, scale, shift, g1, t2, t3, t4))
        ctx.tapes = tapes
        ctx.geom = (M, C0, G, L, Cf)
        ctx.save_for_backward(X)
        return X

    @staticmethod
    def backward(ctx, g):
        (X,) = ctx.saved_tensors
        M, C0, G, L, Cf = ctx.geom
        lib, hctx = N.lib(), N.ctx(X.device)
        dev = X.device
        # ONE gradient buffer for the whole block, updated in place.  A copy of the incoming gradient (one pass per block): autograd's
        # tensors are not ours to write into (a retain_grad() or a hook on the block output would see the sums)
        Gb = g.float().contiguous()
        if Gb is g or Gb.data_ptr() == g.data_ptr():
            Gb = g.clone()
        grads = [None] * (12 * L)
        for l in reversed(range(L)):
            Cl, mean, invstd, scale, shift, g1, t2, t3, t4 = ctx.tapes[l]
            r4 = CamLayerFn.backward(t4, Gb[:, Cl:Cl + G])
            r3 = BNRows.backward(t3, r4[0])
            r2 = _conv_block_bwd(t2, r3[0])
            dh0 = r2[0]
            sums = torch.empty((2, Cl), dtype=torch.float32, device=dev)
            ws = _bytes(lib.vp_col_sums_workspace_bytes(M, Cl), dev)
            _chk(lib.vp_col_sums_masked_f32(hctx, dh0.data_ptr(), Cl, X.data_ptr(), Cf, mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                            shift.data_ptr(), 0.0, M, Cl, sums.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), hctx)
            _chk(lib.vp_bn_relu_bwd_masked_f32(hctx, dh0.data_ptr(), Cl, X.data_ptr(), Cf, mean.data_ptr(), invstd.data_ptr(), g1.data_ptr(),
                                               sums.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0.0, M, Cl, Gb.data_ptr(), Cf, 1,
                                               N.stream_ptr()), hctx)
            grads[12 * l:12 * l + 12] = [sums[1], sums[0], r2[1], r2[2], r3[1], r3[2], r4[1], r4[2], r4[3], r4[4], r4[5], r4[6]]
        return (Gb[:, :C0].contiguous(), None, *grads)


class SEDenseFn(torch.autograd.Function):
    """s = sigmoid(W2 relu(W1 m + b1) + b2) over m (B, C): the two dense layers of a squeeze-excitation gate (resnet_se.py:48-63;
    ecapa_tdnn.py:50-82 runs the same kernels inside SEBlockFn) as ONE launch forward and TWO backward (csrc/se_train.hip:
    vp_se_dense_train_fwd / _bwd) instead of two conv-GEMM launches over B rows forward and ~14 launches backward (activation backward,
    bias sums, weight and data gradient of each layer with their partial-sum stages).  w1 (H, C, 1), w2 (C, H, 1) contiguous f32."""

    @staticmethod
    def usable(m, w1, b1, w2, b2):
        Cc, H = m.shape[1], w1.shape[0]
        return (not os.environ.get('VPMI_SE_DENSE_UNFUSED') and m.dtype == torch.float32 and tuple(w1.shape) == (H, Cc, 1)
                and tuple(w2.shape) == (Cc, H, 1) and Cc <= 1024 and H <= 1024 and Cc % 4 == 0 and H % 4 == 0 and b1 is not None and b2 is not None)

    @staticmethod
    def forward(ctx, m, w1, b1, w2, b2):
        lib, hctx = N.lib(), N.ctx(m.device)
        m, w1, b1, w2, b2 = (_f32c(t) for t in (m, w1, b1, w2, b2))
        B, Cc = m.shape
        H = w1.shape[0]
        a = torch.empty((B, H), dtype=torch.float32, device=m.device)
        s = torch.empty((B, Cc), dtype=torch.float32, device=m.device)
        amp = int(ppvector.get_train_amp())
        _chk(lib.vp_se_dense_train_fwd(hctx, m.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), B, Cc, H, amp,
                                       a.data_ptr(), s.data_ptr(), N.stream_ptr()), hctx)
        ctx.save_for_backward(m, a, s, w1, w2)
        ctx.amp = amp
        return s

    @staticmethod
    def backward(ctx, ds):
        m, a, s, w1, w2 = ctx.saved_tensors
        lib, hctx = N.lib(), N.ctx(m.device)
        ds = _f32c(ds)
        B, Cc = m.shape
        H = w1.shape[0]
        dm = torch.empty_like(m)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        db1 = torch.empty(H, dtype=torch.float32, device=m.device)
        db2 = torch.empty(Cc, dtype=torch.float32, device=m.device)
        ws = _bytes(lib.vp_se_dense_train_bwd_workspace_bytes(B, Cc, H), m.device)
        _chk(lib.vp_se_dense_train_bwd(hctx, ds.data_ptr(), m.data_ptr(), a.data_ptr(), s.data_ptr(), w1.data_ptr(), w2.data_ptr(), B, Cc, H,
                                       ctx.amp, dm.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(),
                                       ws.numel(), N.stream_ptr()), hctx)
        return dm, dw1, db1, dw2, db2
