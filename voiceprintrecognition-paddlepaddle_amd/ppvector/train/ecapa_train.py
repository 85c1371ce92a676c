"""Training-mode forward of ECAPA-TDNN (ppvector/models/ecapa_tdnn.py:245-276) through the autograd functions of
functions.py.  Every conv / BatchNorm / SE gate / ASP / projection runs in libvpmi (forward and backward); the Res2Net
chunking, the hand-off adds (y_{i-1} + x_i) and the two concatenations are tensor slicing / torch.cat / `+` on 2-D
(B*T, C) tensors -- data movement and three elementwise adds per block that PyTorch's tape needs to see.
Input (B, T, F) f32 on the GPU -> embeddings (B, embd_dim)."""
import os

import torch

import ppvector
from ppvector.train.functions import asp16_min_rows, BNRows, CatConvBlock, ConvBlock, ConvBlockSkip, ConvSEFn, MfaAspFn, Res2Fn, SEBlockFn, prep_weights_bf16
from ppvector.train.segments import cut
from ppvector.train.tdnn_train import asp_forward


def tdnn_block(blk, x, B, T, want_tsums=False, y_bf16=False, wide_taps=False):
    """TDNNBlock (models/utils.py:122-148): BN(ReLU(Conv1d 'same' reflect)).  want_tsums: the consumer takes time statistics of the
    output (SE squeeze): the conv's fused per-utterance sums travel with the tensor instead of a pass over it."""
    conv, norm = blk.conv.conv, blk.norm.norm
    cfg = dict(B=B, T=T, dilation=blk.conv.dilation, pad='reflect', relu=True, momentum=norm.momentum, eps=norm.eps, want_tsums=want_tsums,
               y_bf16=y_bf16, wide_taps=wide_taps)
    y = ConvBlock.apply(x, conv.weight, conv.bias, None, norm.weight, norm.bias, norm._mean, norm._variance, cfg)
    if cfg.get('_tsums') is not None:
        y._vp_tsums = cfg.pop('_tsums')
    if cfg.get('_y16') is not None:                              # y is a memory-less f32 placeholder for the tape; the values are this twin
        y._vp_bf16, y._vp_bf16_only = cfg.pop('_y16'), True
    return y


def res2net_block(r2, x, B, T):
    """Res2NetBlock (ecapa_tdnn.py:11-47) as one tape entry (functions.Res2Fn)."""
    blocks = list(r2.blocks)
    params = []
    for blk in blocks:
        conv, norm = blk.conv.conv, blk.norm.norm
        params += [conv.weight, conv.bias, norm.weight, norm.bias, norm._mean, norm._variance]
    n0 = blocks[0].norm.norm
    if any(b.conv.dilation != blocks[0].conv.dilation or b.norm.norm.momentum != n0.momentum or b.norm.norm.eps != n0.eps
           for b in blocks):
        raise NotImplementedError('Res2NetBlock chunks with different dilation / BatchNorm settings')
    # bf16_twin: under enable_amp the fused chain kernel writes its output once more as bf16 -- the operand tdnn2 reads
    cfg = dict(B=B, T=T, scale=r2.scale, dilation=blocks[0].conv.dilation, momentum=n0.momentum, eps=n0.eps,
               bf16_twin=bool(ppvector.get_train_amp()) and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') != '0'
               and not os.environ.get('VPMI_NO_SHADOW') and B * T >= 4096 and x.shape[1] >= 256)
    cfg['out16_only'] = cfg['bf16_twin'] and not os.environ.get('VPMI_RES2_F32_OUT')      # tdnn2 reads the bf16 copy and nothing else reads the f32 one
    out = Res2Fn.apply(x, cfg, *params)
    if cfg.get('_twin') is not None:
        out._vp_bf16 = cfg['_twin']
        if cfg.get('_twin_only'):
            out._vp_bf16_only = True                             # `out` is a memory-less placeholder for the tape
    return out


def se_res2net_block(blk, x, B, T, shadow=None):
    if blk.shortcut is not None:
        raise NotImplementedError('SERes2NetBlock with a shortcut conv is not built')
    conv, norm = blk.tdnn1.conv.conv, blk.tdnn1.norm.norm      # tdnn1 also hands x on as the residual (its gradient comes back here)
    # (enable_amp at scale: tdnn1's output leaves its BatchNorm pass as bf16 only -- the fused Res2 chain reads it in that form)
    cfg1 = dict(B=B, T=T, dilation=blk.tdnn1.conv.dilation, pad='reflect', relu=True, momentum=norm.momentum, eps=norm.eps,
                y_bf16=shadow is not None and not os.environ.get('VPMI_TDNN1_F32_OUT'))
    h, residual = ConvBlockSkip.apply(x, conv.weight, conv.bias, None, norm.weight, norm.bias, norm._mean, norm._variance, cfg1)
    if cfg1.get('_y16') is not None:
        h._vp_bf16, h._vp_bf16_only = cfg1.pop('_y16'), True
    h = res2net_block(blk.res2net_block, h, B, T)
    # all-bf16 SE stage (enable_amp at scale, the MFA operand buffer given): tdnn2's output, the residual and the block output exist as
    # bf16 only; the tape sees f32 placeholders (functions._placeholder)
    res16 = getattr(x, '_vp_bf16', None)
    all16 = shadow is not None and res16 is not None and not os.environ.get('VPMI_SE_F32') and not os.environ.get('VPMI_NO_TSUMS')
    if res16 is not None:
        residual._vp_bf16 = res16                                # (a view of x: attributes do not travel with it)
        if getattr(x, '_vp_bf16_only', False):
            residual._vp_bf16_only = True
    se = blk.se_block                                           # squeeze, two dense layers, gate, + residual: one tape entry
    c2, n2 = blk.tdnn2.conv.conv, blk.tdnn2.norm.norm
    if (all16 and getattr(h, '_vp_bf16_only', False) and c2.bias is not None
            and ConvSEFn.usable(h, residual, c2.weight, se.conv1.conv.weight, se.conv2.conv.weight, se.conv1.conv.bias, se.conv2.conv.bias,
                                shadow, B, T)):
        # tdnn2 + SE gate + residual as ONE tape entry: neither tdnn2's BatchNorm output nor the SE block's input gradient is ever stored
        cfg2 = dict(B=B, T=T, dilation=blk.tdnn2.conv.dilation, pad='reflect', relu=True, momentum=n2.momentum, eps=n2.eps)
        out = ConvSEFn.apply(h, c2.weight, c2.bias, n2.weight, n2.bias, n2._mean, n2._variance, residual, se.conv1.conv.weight,
                             se.conv1.conv.bias, se.conv2.conv.weight, se.conv2.conv.bias, cfg2, shadow)
        out._vp_bf16, out._vp_bf16_only = shadow, True
        return out
    h = tdnn_block(blk.tdnn2, h, B, T, want_tsums=True, y_bf16=all16)
    out = SEBlockFn.apply(h, residual, se.conv1.conv.weight, se.conv1.conv.bias, se.conv2.conv.weight, se.conv2.conv.bias, B, T, shadow)
    if shadow is not None:
        out._vp_bf16 = shadow                                    # ConvBlock / CatConvBlock take the operand from here instead of converting
        if getattr(h, '_vp_bf16_only', False):
            out._vp_bf16_only = True                             # ... and it is the ONLY copy: `out` itself is a placeholder
    return out


def ecapa_forward_train(m, feats):
    B, T, F = feats.shape
    x = feats.reshape(B * T, F)
    blocks = list(m.blocks)[1:]
    Cb0 = m.blocks[0].conv.conv.weight.shape[0]
    use_xcat = (ppvector.get_train_amp() and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') != '0' and not os.environ.get('VPMI_NO_SHADOW')
                and B * T >= 4096 and Cb0 % 64 == 0 and Cb0 >= 256 and m.mfa.conv.conv.weight.shape[1] == Cb0 * len(blocks))
    # (with the bf16 operand path on, block 0's output is read as bf16 only -- tdnn1's operand and the first block's residual)
    if ppvector.get_train_amp() and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') != '0' and B * T >= 4096:
        # every bf16 weight panel the step will read (W for the forward GEMMs, W^T for the data-gradient GEMMs of the wide 1x1 layers
        # and of ASP's two convs) from ONE launch -- they were 22 conversion launches spread over the step
        items = []
        for blk in blocks:
            for t in (blk.tdnn1, blk.tdnn2):
                wt = t.conv.conv.weight
                items.append((wt, 0, wt.shape[1]))
        wt = m.mfa.conv.conv.weight
        items.append((wt, 0, wt.shape[1]))
        wa, wc = m.asp.tdnn.conv.conv.weight, m.asp.conv.conv.weight
        items.append((wa, 0, wc.shape[0] if m.asp.global_context else wa.shape[1]))       # the x columns of the attention TDNN
        items.append((wc, 0, wc.shape[1]))
        prep_weights_bf16([it for it in items if it[0].shape[2] == 1])
    x = tdnn_block(m.blocks[0], x, B, T, y_bf16=use_xcat and not os.environ.get('VPMI_BLOCK0_F32_OUT'),
                   wide_taps=use_xcat and not os.environ.get('VPMI_BLOCK0_F32_OPS'))
    outs = []
    # enable_amp: the block outputs are GEMM operands twice (next block's tdnn1, the MFA concatenation) -- the kernel that produces
    # them also writes them as bf16, straight into their column slice of the MFA operand (VPMI_TRAIN_BF16_OPS=0: f32 operands)
    Cb = x.shape[1]
    xcat = None
    if use_xcat:
        xcat = torch.empty((B * T, Cb * len(blocks)), dtype=torch.bfloat16, device=x.device)
    if xcat is not None and getattr(x, '_vp_bf16', None) is None:
        x._vp_bf16 = x.to(torch.bfloat16)                        # block 0's output as the first block reads it (GEMM operand and residual)
    for i, blk in enumerate(blocks):
        x = se_res2net_block(blk, x, B, T, xcat[:, i * Cb:(i + 1) * Cb] if xcat is not None else None)
        outs.append(x)
        # backward stage boundary (train/segments.py): every block output is live across it -- the next block reads the last
        # one, the MFA concatenation all of them.  Stages from the end: head + ASP + MFA | block 3 | block 2 | blocks 0-1
        outs = list(cut(*outs))
        x = outs[-1]
    conv, norm = m.mfa.conv.conv, m.mfa.norm.norm
    # ASP's context statistics come from the MFA conv's fused sums; under enable_amp (bf16 operand path) the MFA output itself leaves
    # its BatchNorm pass as bf16 -- ASP is its only consumer and reads it as a GEMM operand and in three statistics passes
    cfg = dict(B=B, T=T, dilation=m.mfa.conv.dilation, pad='reflect', relu=True, momentum=norm.momentum, eps=norm.eps, xcat=xcat,
               want_tsums=True, y_bf16=xcat is not None and bool(m.asp.global_context) and T <= 320 and B * T >= asp16_min_rows()
               and not os.environ.get('VPMI_MFA_F32_OUT'))
    if cfg['y_bf16'] and os.environ.get('VPMI_TRAIN_BF16_OPS', '2') == '2' and not os.environ.get('VPMI_MFA_ASP_UNFUSED'):
        # MFA + ASP as ONE tape entry: the pooling layer's context-statistics gradient is folded into the MFA layer's BatchNorm backward
        # instead of a pass of its own over the (B*T, 1536) tensors (functions.MfaAspFn)
        a = m.asp
        ac, an, a2 = a.tdnn.conv.conv, a.tdnn.norm.norm, a.conv.conv
        acfg = dict(B=B, T=T, global_context=bool(a.global_context), momentum=an.momentum, eps=an.eps)
        p = MfaAspFn.apply(cfg, acfg, conv.weight, conv.bias, norm.weight, norm.bias, norm._mean, norm._variance, ac.weight, ac.bias,
                           an.weight, an.bias, an._mean, an._variance, a2.weight, a2.bias, *outs)
        n = m.asp_bn.norm
        p = BNRows.apply(p, n.weight, n.bias, n._mean, n._variance, n.momentum, n.eps)
        fc = m.fc.conv
        return ConvBlock.apply(p, fc.weight, fc.bias, None, None, None, None, None, dict(B=B, T=1))
    x = CatConvBlock.apply(cfg, conv.weight, conv.bias, norm.weight, norm.bias, norm._mean, norm._variance, *outs)
    if cfg.get('_tsums') is not None:
        x._vp_tsums = cfg.pop('_tsums')
    if cfg.get('_y16') is not None:                              # x is a memory-less f32 placeholder for the tape; the values are this twin
        x._vp_bf16, x._vp_bf16_only = cfg.pop('_y16'), True
    p = asp_forward(m.asp, x, B, T)
    n = m.asp_bn.norm
    p = BNRows.apply(p, n.weight, n.bias, n._mean, n._variance, n.momentum, n.eps)
    fc = m.fc.conv
    return ConvBlock.apply(p, fc.weight, fc.bias, None, None, None, None, None, dict(B=B, T=1))
