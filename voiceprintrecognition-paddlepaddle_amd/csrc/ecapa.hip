// Whole-backbone forward (eval mode) for ECAPA-TDNN and TDNN: the launch graph over the kernels in
// conv_gemm.hip / small_ops.hip.  Host-only code; everything it enqueues is asynchronous on the
// caller's stream and touches only the caller's workspace, so the sequence is hipGraph-capturable.
//
// Reference graphs: EcapaTdnn.forward (ppvector/models/ecapa_tdnn.py:245-276) with SERes2NetBlock
// (:129-142), Res2NetBlock (:36-47), SEBlock (:69-82), AttentiveStatisticsPooling (pooling.py:86-125),
// asp_bn + fc (:271-274); TDNN.forward (ppvector/models/tdnn.py:46-68).
// What never materialises compared with the reference: the (B,F,T) transpose, reflect-padded
// copies, torch.chunk / concat copies (slices of one (B*T, n*C) buffer are addressed in place),
// the x_i + y_{i-1} temporaries' extra pass (written by the producing conv), the (B, 3C, T)
// global-context concat of ASP (its mean/std columns collapse to a per-utterance bias).
// Three arithmetic modes per backbone (`dtype` of the weights struct): VP_F32 exact f32 matrix cores; VP_BF16 bf16 tensors with the fused
// bf16 kernels; VP_F32X3 split precision -- f32 tensors on the generic kernels, or, for ECAPA with split weights present, tensors as
// split bf16 planes on the LDS-DMA ring / fused hl32 kernels (ecapa_fwd_hl below).
#include "common.h"

#include <stdlib.h>

namespace {

struct Carver {
    char* base; size_t off; size_t cap;
    Carver(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
    void* take(size_t bytes) {
        size_t o = off;
        off += vp_align_up(bytes ? bytes : 1, 256);
        return base ? (void*)(base + o) : nullptr;
    }
};

void tdnn_desc(vp_conv1d_desc& d, const vp_tdnn_layer& L, int dtype, int B, int T_in, int T_out, int pad_mode) {
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dtype);
    d.B = B; d.T_in = T_in; d.T_out = T_out;
    d.Cin = L.cin; d.Cout = L.cout; d.KW = L.kw; d.dilation = L.dil; d.stride = 1;
    d.pad_mode = pad_mode;
    d.pad_left = pad_mode == VP_PAD_NONE ? 0 : L.dil * (L.kw - 1) / 2;
    vp_desc_weights(d, L);
    d.bias = L.bias; d.act = VP_ACT_RELU; d.bn_scale = L.bn_scale; d.bn_shift = L.bn_shift;
}


}  // namespace

// Attentive statistics pooling over x (B*T, C).  When w.psum != NULL the global-context mean/std come
// from the producing conv's partial sums (shift = its bn_shift or NULL); otherwise w.stats must
// already hold [mean | std] per utterance.  Leaves pooled (B, 2C) = [mean | std].
int vp_run_asp(vp_ctx* ctx, const vp_asp_weights& A, int dtc, const void* x, int ldx, const float* shift,
               int B, int T, const VpAspBufs& w, hipStream_t st) {
    const int C = A.C;
    const int dtype = vp_storage_dtype(dtc);
    int rc = VP_OK;
    if (w.psum) rc = vp_moments_finalize(ctx, w.psum, w.psumsq, shift, B, T, C, 1e-12f, 1, w.stats, st);
    if (rc) return rc;
    if (A.w_ctx) {
        rc = vp_dense_f32_ex(ctx, w.stats, 2 * C, A.w_ctx, 0, nullptr, nullptr, nullptr, B, A.att, 2 * C, VP_ACT_NONE,
                             w.rowbias, A.att, st);
        if (rc) return rc;
    }
    if (dtc == VP_HL32) {
        // split-precision fast path: x is stored as split bf16 planes.  Attention TDNN (hl32 in / out, tanh) on the 128-wide kernel, then
        // logits + softmax + weighted statistics in one kernel (asp_x3.hip): no (B*T, C) logits tensor
        if (!A.tdnn.w_hl) VP_FAIL(ctx, VP_EINVAL, "asp: hl32 input needs the attention TDNN's split weights (w_hl)");
        vp_conv1d_desc d;
        tdnn_desc(d, A.tdnn, VP_F32X3, B, T, T, VP_PAD_REFLECT);
        d.dtype_in = d.dtype_out = VP_HL32; d.w = A.tdnn.w_hl;
        d.x = x; d.ldx = ldx; d.xoff = 0; d.rowbias = A.w_ctx ? w.rowbias : nullptr; d.act2 = VP_ACT_TANH;
        d.y = w.h; d.ldy = A.att; d.yoff = 0;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        rc = vp_asp_fused_x3(ctx, w.h, (const float*)A.conv_w, A.conv_b, x, ldx, w.stats, 2 * C, B, T, C, A.att, 1e-12f, w.pooled, st);
        if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "asp: shape not covered by the split-precision kernel (attention width 128, C %% 32 == 0)");
        return rc;
    }
    if (dtype == VP_BF16) {                       // one kernel per utterance: attention TDNN + logits + softmax + weighted statistics
        rc = vp_asp_utt_bf16(ctx, x, ldx, &A.tdnn, A.w_ctx ? w.rowbias : nullptr, A.conv_w, A.conv_b, B, T, C, A.att, 1e-12f, w.pooled, st);
        if (rc != VP_EUNSUP) return rc;
    }
    vp_conv1d_desc d;
    tdnn_desc(d, A.tdnn, dtc, B, T, T, VP_PAD_REFLECT);
    d.x = x; d.ldx = ldx; d.xoff = 0; d.rowbias = A.w_ctx ? w.rowbias : nullptr; d.act2 = VP_ACT_TANH;
    d.y = w.h; d.ldy = A.att; d.yoff = 0;
    rc = vp_conv1d_fwd(ctx, &d, st);
    if (rc) return rc;
    static const bool asp_unfused = getenv("VPMI_ASP_UNFUSED") != nullptr;           // A/B and race screens
    if (dtype == VP_BF16 && !asp_unfused) {       // logits GEMM + softmax + weighted stats in one kernel; no (B*T, C) f32 logits
        rc = vp_asp_fused_bf16(ctx, w.h, A.conv_w, A.conv_b, x, ldx, w.stats, 2 * C, B, T, C, A.att, 1e-12f, w.pooled, st);
        if (rc != VP_EUNSUP) return rc;
    }
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dtc);
    d.dtype_out = VP_F32; d.B = B; d.T_in = T; d.T_out = T; d.Cin = A.att; d.Cout = C;
    d.KW = 1; d.dilation = 1; d.stride = 1; d.pad_mode = VP_PAD_REFLECT; d.pad_left = 0;
    d.x = w.h; d.ldx = A.att; d.w = A.conv_w; d.bias = A.conv_b; d.y = w.e; d.ldy = C;
    rc = vp_conv1d_fwd(ctx, &d, st);
    if (rc) return rc;
    return vp_asp_softmax_stats_ex(ctx, dtype, w.e, x, ldx, 0, w.stats, 2 * C, B, T, C, 1e-12f, w.pooled, st);
}

namespace {

struct EcapaPlan {
    void *cat0, *cat, *t1, *r2, *t2, *tmpA, *tmpB, *mfa, *h, *im;
    float *e, *psum, *psumsq, *stats, *se_h, *se_s, *rowbias, *pooled;
    size_t total;
};

int plan_ecapa(const vp_ecapa_weights* w, int B, int T, void* ws, size_t cap, EcapaPlan& p) {
    const size_t es = vp_dtype_size(w->dtype);
    const size_t M = (size_t)B * T;
    const int C = w->block0.cout, Cm = w->mfa.cout, nb = w->n_blocks;
    const int width = C / w->res2_scale;
    const int cmax = C > Cm ? C : Cm;
    const size_t nps = (size_t)vp_conv1d_tiles_m(B, T) * vp_conv1d_nseg(T) * cmax * sizeof(float);
    Carver c(ws, cap);
    p.cat0 = c.take(M * C * es);
    p.cat = c.take(M * (size_t)nb * C * es);
    p.t1 = c.take(M * C * es);
    p.r2 = c.take(M * C * es);
    p.t2 = c.take(M * C * es);
    p.tmpA = c.take(M * width * es);
    p.tmpB = c.take(M * width * es);
    p.mfa = c.take(M * Cm * es);
    p.h = c.take(M * w->asp.att * es);
    p.e = (float*)c.take(M * Cm * sizeof(float));
    p.psum = (float*)c.take(nps);
    p.psumsq = (float*)c.take(nps);
    p.stats = (float*)c.take((size_t)B * 2 * cmax * sizeof(float));
    p.se_h = (float*)c.take((size_t)B * w->se_ch * sizeof(float));
    p.se_s = (float*)c.take((size_t)B * C * sizeof(float));
    p.rowbias = (float*)c.take((size_t)B * w->asp.att * sizeof(float));
    p.pooled = (float*)c.take((size_t)B * 2 * Cm * sizeof(float));
    // split-precision fast path: blocks[0]'s tapped operand rows as an hl32 matrix (B*T, roundup(kw * cin, 32))
    p.im = (w->dtype == VP_F32X3 && w->block0.w_hl) ? c.take(M * (size_t)vp_align_up((size_t)w->block0.kw * w->block0.cin, 32) * 4) : nullptr;
    p.total = c.off;
    return VP_OK;
}

int check_ecapa(vp_ctx* ctx, const vp_ecapa_weights* w) {
    if (!vp_backbone_dtype_ok(w->dtype)) VP_FAIL(ctx, VP_EINVAL, "ecapa: bad dtype");
    if (w->n_blocks < 1 || w->n_blocks > VP_MAX_SE_BLOCKS) VP_FAIL(ctx, VP_EINVAL, "ecapa: n_blocks %d", w->n_blocks);
    if (w->res2_scale < 2 || w->res2_scale - 1 > VP_MAX_RES2) VP_FAIL(ctx, VP_EINVAL, "ecapa: res2_scale %d", w->res2_scale);
    const int C = w->block0.cout;
    const int epc = w->dtype == VP_BF16 ? 8 : 4;
    if (C % w->res2_scale || (C / w->res2_scale) % epc) VP_FAIL(ctx, VP_EINVAL, "ecapa: channels %d not divisible", C);
    for (int i = 0; i < w->n_blocks; ++i) {
        const vp_se_res2_block& b = w->blk[i];
        if (b.tdnn1.cin != C || b.tdnn1.cout != C || b.tdnn2.cin != C || b.tdnn2.cout != C)
            VP_FAIL(ctx, VP_EUNSUP, "ecapa: SE-Res2 blocks must keep %d channels (shortcut conv not built)", C);
    }
    if (w->mfa.cin != w->n_blocks * C) VP_FAIL(ctx, VP_EINVAL, "ecapa: mfa.cin %d != %d", w->mfa.cin, w->n_blocks * C);
    if (w->asp.C != w->mfa.cout) VP_FAIL(ctx, VP_EINVAL, "ecapa: asp.C mismatch");
    return VP_OK;
}

// ---- split-precision fast path (dtype VP_F32X3): every big activation stored as split bf16 planes (vpmi.h: VP_HL32) -------------
// The wide 1x1 layers stream both operands by LDS-DMA on the 128 x 256 ring (conv_gemm256.hip, three MFMAs per fragment pair), the
// Res2 chain and the pooling run fused (res2_x3.hip, asp_x3.hip); producers split once per output element.  Same graph, same buffers
// (an hl32 tensor has f32's footprint) and the same arithmetic contract as the generic VP_F32X3 path (f32 tensors, operands split
// while staging) -- taken when every layer it touches has its split weights and the shapes fit; VPMI_X3_GENERIC=1 pins the generic path.
bool ecapa_hl_ok(const vp_ecapa_weights* w, int B, int T) {
    static const bool off = getenv("VPMI_X3_GENERIC") != nullptr;
    const int C = w->block0.cout, Cm = w->mfa.cout, sc = w->res2_scale;
    if (off || C % 32 || Cm % 32 || C % sc || w->asp.att != 128 || !w->mfa.w_hl || !w->asp.tdnn.w_hl || !w->asp.w_ctx || T < 128 ||
        (long long)B * T < 128 * 32 || w->block0.cout < 256 || Cm < 256)
        return false;
    for (int i = 0; i < w->n_blocks; ++i) {
        const vp_se_res2_block& b = w->blk[i];
        if (!b.tdnn1.w_hl || !b.tdnn2.w_hl || !vp_res2_chain_x3_ok(b.res2, sc - 1, B, T, C, C / sc)) return false;
    }
    return true;
}

int ecapa_fwd_hl(vp_ctx* ctx, const vp_ecapa_weights* w, const void* feats, int B, int T, float* emb, const EcapaPlan& p, hipStream_t st) {
    const int C = w->block0.cout, Cm = w->mfa.cout, nb = w->n_blocks, sc = w->res2_scale;
    const int width = C / sc, ldcat = nb * C;
    int rc;
    vp_conv1d_desc d;
    auto hl_layer = [&](const vp_tdnn_layer& L) {
        tdnn_desc(d, L, VP_F32X3, B, T, T, VP_PAD_REFLECT);
        d.dtype_in = d.dtype_out = VP_HL32; d.w = L.w_hl;
    };
    // blocks[0]: with split weights present (w_hl = [C][roundup(kw * F, 32)], zero-padded) the taps are laid out as operand rows (im2col
    // into split planes, 127 MB at 256 x 3 s) and the layer runs as a 1x1 GEMM on the LDS-DMA ring; else f32 features in (split while
    // staging) on the 128-wide tapped kernel.  hl32 out either way
    const int Kp0 = (int)vp_align_up((size_t)w->block0.kw * w->block0.cin, 32);
    if (w->block0.w_hl && p.im && w->feat_dim == w->block0.cin && w->block0.cin % 8 == 0) {
        const int pad = w->block0.dil * (w->block0.kw - 1) / 2;
        if ((rc = vp_im2col_hl32(ctx, (const float*)feats, B, T, w->block0.cin, w->block0.kw, w->block0.dil, pad, p.im, Kp0, st))) return rc;
        tdnn_desc(d, w->block0, VP_F32X3, B, T, T, VP_PAD_REFLECT);
        d.dtype_in = d.dtype_out = VP_HL32; d.w = w->block0.w_hl;
        d.Cin = Kp0; d.KW = 1; d.dilation = 1; d.pad_left = 0;
        d.x = p.im; d.ldx = Kp0; d.y = p.cat0; d.ldy = C;
    } else {
        tdnn_desc(d, w->block0, VP_F32X3, B, T, T, VP_PAD_REFLECT);
        d.dtype_out = VP_HL32; d.w = w->block0.w; d.mfma_bf16 = 2;          // (f32 weights, split while staging: the hl32-output kernel's form)
        d.x = feats; d.ldx = w->feat_dim; d.y = p.cat0; d.ldy = C;
    }
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
    const void* xin = p.cat0;
    int ld_in = C, off_in = 0;
    for (int i = 0; i < nb; ++i) {
        const vp_se_res2_block& blk = w->blk[i];
        hl_layer(blk.tdnn1);
        d.x = xin; d.ldx = ld_in; d.xoff = off_in; d.y = p.t1; d.ldy = C;
        d.y2 = p.r2; d.ldy2 = C; d.y2off = 0; d.ysplit = width;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        rc = vp_res2_chain_x3(ctx, blk.res2, sc - 1, p.t1, p.r2, B, T, C, width, st);
        if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "ecapa: Res2 chain left the split-precision kernel's range after the shape check");
        if (rc) return rc;
        hl_layer(blk.tdnn2);
        d.x = p.r2; d.ldx = C; d.xoff = 0; d.y = p.t2; d.ldy = C; d.psum = p.psum;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        if ((rc = vp_se_gate(ctx, p.psum, blk.tdnn2.bn_shift, B, T, C, w->se_ch, blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2, p.se_s, st)))
            return rc;
        if ((rc = vp_se_scale_residual_ex(ctx, VP_HL32, p.t2, C, 0, p.se_s, xin, ld_in, off_in, p.cat, ldcat, i * C, B, T, C, 0, st))) return rc;
        xin = p.cat; ld_in = ldcat; off_in = i * C;
    }
    hl_layer(w->mfa);
    d.x = p.cat; d.ldx = ldcat; d.y = p.mfa; d.ldy = Cm; d.psum = p.psum; d.psumsq = p.psumsq;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
    VpAspBufs ab{p.h, p.e, p.psum, p.psumsq, p.stats, p.rowbias, p.pooled};
    if ((rc = vp_run_asp(ctx, w->asp, VP_HL32, p.mfa, Cm, w->mfa.bn_shift, B, T, ab, st))) return rc;
    return vp_dense_f32_ex(ctx, p.pooled, 2 * Cm, w->fc_w, 0, w->fc_b, nullptr, nullptr, B, w->embd_dim, 2 * Cm,
                           VP_ACT_NONE, emb, w->embd_dim, st);
}

}  // namespace

extern "C" {

// C-ABI doors of the three fused bf16 kernels of the ECAPA forward (tests call them one by one; vp_ecapa_fwd calls the
// launchers directly)
int vp_res2_chain_fwd(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C,
                      int width, vp_stream stream) {
    if (!ctx || !layers || !t1 || !r2 || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "res2_chain: bad arguments");
    const int rc = vp_res2_chain_bf16(ctx, layers, nconv, t1, r2, B, T, C, width, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "res2_chain: shape not covered by the fused kernel (width 64, equal dilations, 2 <= T <= 384)");
    return rc;
}

int vp_asp_utt_fwd(vp_ctx* ctx, const void* x, int ldx, const vp_tdnn_layer* tdnn, const float* rowbias, const void* conv_w,
                   const float* conv_b, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream) {
    if (!ctx || !x || !tdnn || !conv_w || !pooled || B <= 0) VP_FAIL(ctx, VP_EINVAL, "asp_utt: bad arguments");
    const int rc = vp_asp_utt_bf16(ctx, x, ldx, tdnn, rowbias, conv_w, conv_b, B, T, C, att, eps, pooled, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "asp_utt: shape not covered (attention width 128, C %% 64 == 0, T <= 304, 1x1 TDNN with BatchNorm)");
    return rc;
}

int vp_asp_fused_fwd(vp_ctx* ctx, const void* h, const void* w, const float* bias, const void* x, int ldx, const float* center,
                     int ldc, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream) {
    if (!ctx || !h || !w || !x || !center || !pooled || B <= 0) VP_FAIL(ctx, VP_EINVAL, "asp_fused: bad arguments");
    const int rc = vp_asp_fused_bf16(ctx, h, w, bias, x, ldx, center, ldc, B, T, C, att, eps, pooled, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "asp_fused: shape not covered (attention width 128, C and ldx multiples of 8, 16-byte aligned)");
    return rc;
}

int vp_res2_chain_x3_fwd(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C,
                         int width, vp_stream stream) {
    if (!ctx || !layers || !t1 || !r2 || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "res2_chain_x3: bad arguments");
    const int rc = vp_res2_chain_x3(ctx, layers, nconv, t1, r2, B, T, C, width, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "res2_chain_x3: shape not covered (width 64, equal dilations, split weights present, segments fit the LDS)");
    return rc;
}

int vp_asp_fused_x3_fwd(vp_ctx* ctx, const void* h, const float* w, const float* bias, const void* x, int ldx, const float* center,
                        int ldc, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream) {
    if (!ctx || !h || !w || !bias || !x || !pooled || B <= 0) VP_FAIL(ctx, VP_EINVAL, "asp_fused_x3: bad arguments");
    const int rc = vp_asp_fused_x3(ctx, h, w, bias, x, ldx, center, ldc, B, T, C, att, eps, pooled, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "asp_fused_x3: shape not covered (attention width 128, C and ldx multiples of 32, 16-byte aligned)");
    return rc;
}

int vp_se_gate_fwd(vp_ctx* ctx, const float* psum, const float* shift, int B, int T, int C, int H, const float* w1, const float* b1,
                   const float* w2, const float* b2, float* out, vp_stream stream) {
    return vp_se_gate(ctx, psum, shift, B, T, C, H, w1, b1, w2, b2, out, (hipStream_t)stream);
}

size_t vp_ecapa_workspace_bytes(const vp_ecapa_weights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0) return 0;
    EcapaPlan p;
    plan_ecapa(w, B, T, nullptr, 0, p);
    return p.total;
}

int vp_ecapa_fwd(vp_ctx* ctx, const vp_ecapa_weights* w, const void* feats, int B, int T, float* emb,
                 void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !w || !feats || !emb || B <= 0 || T <= 0) VP_FAIL(ctx, VP_EINVAL, "ecapa: bad arguments");
    int rc = check_ecapa(ctx, w);
    if (rc) return rc;
    EcapaPlan p;
    plan_ecapa(w, B, T, ws, ws_bytes, p);
    if (!ws || ws_bytes < p.total) VP_FAIL(ctx, VP_EWORKSPACE, "ecapa: workspace %zu < %zu", ws_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    const int dtc = w->dtype, dt = vp_storage_dtype(dtc);
    const int C = w->block0.cout, Cm = w->mfa.cout, nb = w->n_blocks, sc = w->res2_scale;
    const int width = C / sc, ldcat = nb * C;
    vp_conv1d_desc d;
    if (dtc == VP_F32X3 && ecapa_hl_ok(w, B, T)) return ecapa_fwd_hl(ctx, w, feats, B, T, emb, p, st);

    // blocks[0]: TDNNBlock(F -> C, k5) on the (B, T, F) features
    tdnn_desc(d, w->block0, dtc, B, T, T, VP_PAD_REFLECT);
    d.x = feats; d.ldx = w->feat_dim; d.y = p.cat0; d.ldy = C;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;

    const void* xin = p.cat0;
    int ld_in = C, off_in = 0;
    for (int i = 0; i < nb; ++i) {
        const vp_se_res2_block& blk = w->blk[i];
        // tdnn1 (1x1)
        tdnn_desc(d, blk.tdnn1, dtc, B, T, T, VP_PAD_REFLECT);
        d.x = xin; d.ldx = ld_in; d.xoff = off_in; d.y = p.t1; d.ldy = C;
        d.y2 = p.r2; d.ldy2 = C; d.y2off = 0; d.ysplit = width;       // y_0 = x_0 goes straight into the concat
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        // Res2Net chain: fused per-utterance kernel when it fits LDS, else one launch per conv
        int fused = VP_EUNSUP;
        static const bool res2_unfused = getenv("VPMI_RES2_UNFUSED") != nullptr;      // A/B and race screens: one launch per conv
        if (dt == VP_BF16 && !res2_unfused) fused = vp_res2_chain_bf16(ctx, blk.res2, sc - 1, p.t1, p.r2, B, T, C, width, st);
        if (fused != VP_OK && fused != VP_EUNSUP) return fused;
        void* tin = nullptr;
        void* tout = p.tmpA;
        for (int j = 1; j < sc && fused != VP_OK; ++j) {
            tdnn_desc(d, blk.res2[j - 1], dtc, B, T, T, VP_PAD_REFLECT);
            if (j == 1) { d.x = p.t1; d.ldx = C; d.xoff = width; }
            else { d.x = tin; d.ldx = width; d.xoff = 0; }
            d.y = p.r2; d.ldy = C; d.yoff = j * width;
            if (j + 1 < sc) {
                d.add_in = p.t1; d.ld_add = C; d.add_off = (j + 1) * width;
                d.aux = tout; d.ld_aux = width; d.aux_off = 0;
            }
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
            tin = tout;
            tout = (tout == p.tmpA) ? p.tmpB : p.tmpA;
        }
        // tdnn2 (1x1) over concat(y_0 .. y_{s-1}) = r2, with the SE time sums fused
        tdnn_desc(d, blk.tdnn2, dtc, B, T, T, VP_PAD_REFLECT);
        d.x = p.r2; d.ldx = C; d.xoff = 0;
        d.y = p.t2; d.ldy = C; d.psum = p.psum;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        // SE: mean over time -> 1x1 -> ReLU -> 1x1 -> sigmoid
        if ((rc = vp_se_gate(ctx, p.psum, blk.tdnn2.bn_shift, B, T, C, w->se_ch, blk.se_w1, blk.se_b1, blk.se_w2, blk.se_b2,
                             p.se_s, st))) return rc;
        // gate + residual, written straight into slice i of the MFA input
        if ((rc = vp_se_scale_residual(ctx, dt, p.t2, C, 0, p.se_s, xin, ld_in, off_in, p.cat, ldcat, i * C, B, T, C, st)))
            return rc;
        xin = p.cat; ld_in = ldcat; off_in = i * C;
    }
    // MFA (1x1 over the concat), with the global-context time moments fused
    tdnn_desc(d, w->mfa, dtc, B, T, T, VP_PAD_REFLECT);
    d.x = p.cat; d.ldx = ldcat; d.y = p.mfa; d.ldy = Cm; d.psum = p.psum; d.psumsq = p.psumsq;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
    VpAspBufs ab{p.h, p.e, p.psum, p.psumsq, p.stats, p.rowbias, p.pooled};
    if ((rc = vp_run_asp(ctx, w->asp, dtc, p.mfa, Cm, w->mfa.bn_shift, B, T, ab, st))) return rc;
    // asp_bn (folded) + fc
    return vp_dense_f32_ex(ctx, p.pooled, 2 * Cm, w->fc_w, 0, w->fc_b, nullptr, nullptr, B, w->embd_dim, 2 * Cm,
                           VP_ACT_NONE, emb, w->embd_dim, st);
}

// ------------------------------------------------------------------------------------- TDNN
static void tdnn_T(const vp_tdnn_weights* w, int T, int Ts[6]) {
    Ts[0] = T;
    for (int i = 0; i < 5; ++i) Ts[i + 1] = Ts[i] - w->td[i].dil * (w->td[i].kw - 1);
}

struct TdnnPlan { void* a; void* b; void* h; float *e, *psum, *psumsq, *stats, *rowbias, *pooled; size_t total; };

static void plan_tdnn(const vp_tdnn_weights* w, int B, int T, void* ws, size_t cap, TdnnPlan& p) {
    const size_t es = vp_dtype_size(w->dtype);
    const int C = w->channels;
    int Ts[6];
    tdnn_T(w, T, Ts);
    const int T1 = Ts[1] > 0 ? Ts[1] : 1, T5 = Ts[5] > 0 ? Ts[5] : 1;
    Carver c(ws, cap);
    p.a = c.take((size_t)B * T1 * C * es);
    p.b = c.take((size_t)B * T1 * C * es);
    p.h = c.take((size_t)B * T5 * w->asp.att * es);
    p.e = (float*)c.take((size_t)B * T5 * C * sizeof(float));
    const size_t nps = (size_t)vp_conv1d_tiles_m(B, T5) * vp_conv1d_nseg(T5) * C * sizeof(float);
    p.psum = (float*)c.take(nps);
    p.psumsq = (float*)c.take(nps);
    p.stats = (float*)c.take((size_t)B * 2 * C * sizeof(float));
    p.rowbias = (float*)c.take((size_t)B * w->asp.att * sizeof(float));
    p.pooled = (float*)c.take((size_t)B * 2 * C * sizeof(float));
    p.total = c.off;
}

size_t vp_tdnn_workspace_bytes(const vp_tdnn_weights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0) return 0;
    TdnnPlan p;
    plan_tdnn(w, B, T, nullptr, 0, p);
    return p.total;
}

int vp_tdnn_fwd(vp_ctx* ctx, const vp_tdnn_weights* w, const void* feats, int B, int T, float* emb,
                void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !w || !feats || !emb || B <= 0 || T <= 0) VP_FAIL(ctx, VP_EINVAL, "tdnn: bad arguments");
    if (!vp_backbone_dtype_ok(w->dtype)) VP_FAIL(ctx, VP_EINVAL, "tdnn: bad dtype");
    int Ts[6];
    tdnn_T(w, T, Ts);
    if (Ts[5] < 1) VP_FAIL(ctx, VP_EINVAL, "tdnn: %d frames are fewer than the receptive field", T);
    TdnnPlan p;
    plan_tdnn(w, B, T, ws, ws_bytes, p);
    if (!ws || ws_bytes < p.total) VP_FAIL(ctx, VP_EWORKSPACE, "tdnn: workspace %zu < %zu", ws_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    const int C = w->channels;
    int rc;
    vp_conv1d_desc d;
    const void* x = feats;
    int ldx = w->feat_dim;
    void* outs[5] = {p.a, p.b, p.a, p.b, p.a};
    for (int i = 0; i < 5; ++i) {
        tdnn_desc(d, w->td[i], w->dtype, B, Ts[i], Ts[i + 1], VP_PAD_NONE);
        d.x = x; d.ldx = ldx; d.y = outs[i]; d.ldy = C;
        if (i == 4) { d.psum = p.psum; d.psumsq = p.psumsq; }
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        x = outs[i]; ldx = C;
    }
    VpAspBufs ab{p.h, p.e, p.psum, p.psumsq, p.stats, p.rowbias, p.pooled};
    if ((rc = vp_run_asp(ctx, w->asp, w->dtype, x, C, w->td[4].bn_shift, B, Ts[5], ab, st))) return rc;
    return vp_dense_f32_ex(ctx, p.pooled, 2 * C, w->lin_w, 0, w->lin_b, nullptr, nullptr, B, w->embd_dim, 2 * C,
                           VP_ACT_NONE, emb, w->embd_dim, st);
}

}  // extern "C"
