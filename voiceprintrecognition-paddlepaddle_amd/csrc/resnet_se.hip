// ResNetSE backbone forward (eval mode): launch graph.
//
// Reference: ResNetSE.forward (ppvector/models/resnet_se.py:121-139) = conv3x3(1->32)+BN+ReLU (:72-74) ->
// 16 SEBottleneck blocks (:8-45; SELayer :48-63), stage strides 1,2,2,2 on BOTH axes -> reshape
// (B, C*F/8, T/8) (:132) -> AttentiveStatisticsPooling -> BN -> Linear -> BN (:134-138).
// Layout: (B, T, F, C) position-major.  The 1x1 convs are plain GEMMs over positions (the SE global
// average falls out of conv3's fused column sums), the 3x3 / strided 1x1 convs use the conv GEMM's 2-D
// loader, SE gate + residual + ReLU is one elementwise pass, and the final reshape is a permutation of
// the ASP / Linear weights done at pack time (channel index f*C + c instead of c*F + f).
#include "common.h"

namespace {

struct Carver {
    char* base; size_t off;
    explicit Carver(void* p) : base((char*)p), off(0) {}
    void* take(size_t bytes) {
        size_t o = off;
        off += vp_align_up(bytes ? bytes : 1, 256);
        return base ? (void*)(base + o) : nullptr;
    }
};

struct RsePlan {
    void *xa, *xb, *o1, *o2, *o3, *res, *h;
    float *e, *psum, *stats, *se_h, *se_s, *rowbias, *pooled;
    size_t total;
    int T4, F4, C4;
};

int down(int v, int s) { return s == 2 ? (v - 1) / 2 + 1 : v; }

void plan_rse(const vp_resnetse_weights* w, int B, int T, void* ws, RsePlan& p) {
    const size_t es = vp_dtype_size(w->dtype);
    int t = T, f = w->feat_dim;
    size_t big = (size_t)B * t * f * w->c1_channels, small = 0, ps = 0;
    int cmax = w->c1_channels;
    for (int i = 0; i < w->n_blocks; ++i) {
        const vp_rse_block& b = w->blk[i];
        const size_t pin = (size_t)B * t * f;
        const int to = down(t, b.stride), fo = down(f, b.stride);
        const size_t pout = (size_t)B * to * fo;
        if (pin * b.conv1.cout > small) small = pin * b.conv1.cout;
        if (pout * b.conv3.cout > big) big = pout * b.conv3.cout;
        if (pin * b.conv1.cin > big) big = pin * b.conv1.cin;
        const size_t need = (size_t)vp_conv1d_tiles_m(B, to * fo) * vp_conv1d_nseg(to * fo) * b.conv3.cout * 4;
        if (need > ps) ps = need;
        if (b.conv3.cout > cmax) cmax = b.conv3.cout;
        t = to; f = fo;
    }
    p.T4 = t; p.F4 = f; p.C4 = w->blk[w->n_blocks - 1].conv3.cout;
    const int Casp = p.F4 * p.C4;
    Carver c(ws);
    p.xa = c.take(big * es); p.xb = c.take(big * es); p.o3 = c.take(big * es); p.res = c.take(big * es);
    p.o1 = c.take(small * es); p.o2 = c.take(small * es);
    p.h = c.take((size_t)B * t * w->asp.att * es);
    p.e = (float*)c.take((size_t)B * t * Casp * 4);
    p.psum = (float*)c.take(ps);
    p.stats = (float*)c.take((size_t)B * 2 * (Casp > cmax ? Casp : cmax) * 4);
    p.se_h = (float*)c.take((size_t)B * cmax * 4);
    p.se_s = (float*)c.take((size_t)B * cmax * 4);
    p.rowbias = (float*)c.take((size_t)B * w->asp.att * 4);
    p.pooled = (float*)c.take((size_t)B * 2 * Casp * 4);
    p.total = c.off;
}

void base_desc(vp_conv1d_desc& d, const vp_tdnn_layer& L, int dt) {
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dt); d.Cin = L.cin; d.Cout = L.cout; d.KW = L.kw; d.dilation = 1; d.stride = 1;
    d.pad_mode = VP_PAD_ZERO; d.ldx = L.cin; d.ldy = L.cout;
    vp_desc_weights(d, L);
    d.bias = L.bias; d.bn_scale = L.bn_scale; d.bn_shift = L.bn_shift;
}

// 1x1 convs over positions: the streaming kernel for the few-channel full-resolution stages (pointwise.hip), else the conv GEMM
int conv1x1(vp_ctx* ctx, const vp_conv1d_desc& d, hipStream_t st) {
    const int rc = vp_pointwise_bf16(ctx, &d, 1, st);
    return rc == VP_EUNSUP ? vp_conv1d_fwd(ctx, &d, st) : rc;
}

}  // namespace

extern "C" {

// C-ABI door of the streaming 1x1 kernel (tests; vp_resnetse_fwd calls the launcher)
int vp_pointwise_fwd(vp_ctx* ctx, const vp_conv1d_desc* d, vp_stream stream) {
    if (!ctx || !d || !d->x || !d->w || !d->y) VP_FAIL(ctx, VP_EINVAL, "pointwise: null argument");
    const int rc = vp_pointwise_bf16(ctx, d, 0, (hipStream_t)stream);
    if (rc == VP_EUNSUP)
        VP_FAIL(ctx, VP_EUNSUP, "pointwise: shape not covered (bf16 1x1 stride 1, Cin / Cout in {32, 64, 128} with Cin * Cout <= 8192, >= 32768 positions)");
    return rc;
}

size_t vp_resnetse_workspace_bytes(const vp_resnetse_weights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0 || w->n_blocks < 1 || w->n_blocks > VP_MAX_RSE_BLOCKS) return 0;
    RsePlan p;
    plan_rse(w, B, T, nullptr, p);
    return p.total;
}

int vp_resnetse_fwd(vp_ctx* ctx, const vp_resnetse_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !w || !feats || !emb || B <= 0 || T <= 0) VP_FAIL(ctx, VP_EINVAL, "resnetse: bad arguments");
    if (!vp_backbone_dtype_ok(w->dtype)) VP_FAIL(ctx, VP_EINVAL, "resnetse: bad dtype");
    if (w->c1_channels != 32 || w->n_blocks < 1 || w->n_blocks > VP_MAX_RSE_BLOCKS)
        VP_FAIL(ctx, VP_EUNSUP, "resnetse: geometry not built (stem of 32 channels, <= %d blocks)", VP_MAX_RSE_BLOCKS);
    RsePlan p;
    plan_rse(w, B, T, ws, p);
    if (!ws || ws_bytes < p.total) VP_FAIL(ctx, VP_EWORKSPACE, "resnetse: workspace %zu < %zu", ws_bytes, p.total);
    if (p.T4 < 2) VP_FAIL(ctx, VP_EINVAL, "resnetse: %d frames are too few", T);
    hipStream_t st = (hipStream_t)stream;
    const int dtc = w->dtype, dt = vp_storage_dtype(dtc);
    int rc;
    vp_conv1d_desc d;
    if ((rc = vp_conv3x3_c1(ctx, dt, feats, p.xa, w->c1_w, w->c1_b, w->c1_scale, w->c1_shift, B, T, w->feat_dim, 32, st))) return rc;
    void* x = p.xa;
    void* xn = p.xb;
    int t = T, f = w->feat_dim;
    for (int i = 0; i < w->n_blocks; ++i) {
        const vp_rse_block& b = w->blk[i];
        const int to = down(t, b.stride), fo = down(f, b.stride);
        const int C = b.conv3.cout;
        // o1 = relu(bn1(conv1x1(x))): a GEMM over the B*t*f positions
        base_desc(d, b.conv1, dtc);
        d.B = B; d.T_in = t * f; d.T_out = t * f; d.x = x; d.y = p.o1; d.act2 = VP_ACT_RELU;
        if ((rc = conv1x1(ctx, d, st))) return rc;
        // o2 = relu(bn2(conv3x3 stride s (o1))): the stage-1 blocks (32 -> 32 channels, stride 1, full-resolution maps) run on the
        // slab kernel of the CAM++ FCM head (fcm_conv.hip), the rest on the conv GEMM's 2-D loader
        int fast = VP_EUNSUP;
        if (dt == VP_BF16 && b.stride == 1 && b.conv2.cin == 32 && b.conv2.cout == 32)
            fast = vp_conv3x3_c32_bf16(ctx, p.o1, p.o2, &b.conv2, nullptr, 1, nullptr, nullptr, B, t, f, 1, nullptr, nullptr, nullptr, nullptr, nullptr, st);
        if (fast != VP_OK && fast != VP_EUNSUP) return fast;
        if (fast != VP_OK) {
            base_desc(d, b.conv2, dtc);
            d.B = B; d.T_in = t; d.T_out = to; d.F_in = f; d.F_out = fo; d.KF = 3; d.stride = b.stride; d.stride_f = b.stride;
            d.pad_left = 1; d.pad_f = 1; d.x = p.o1; d.y = p.o2; d.act2 = VP_ACT_RELU;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        }
        // o3 = bn3(conv1x1(o2)) with the per-utterance sums of the SE squeeze fused in
        base_desc(d, b.conv3, dtc);
        d.B = B; d.T_in = to * fo; d.T_out = to * fo; d.x = p.o2; d.y = p.o3; d.psum = p.psum;
        if ((rc = conv1x1(ctx, d, st))) return rc;
        if ((rc = vp_se_gate(ctx, p.psum, b.conv3.bn_shift, B, to * fo, C, C / 8, b.se_w1, b.se_b1, b.se_w2, b.se_b2, p.se_s, st)))
            return rc;
        const void* res = x;
        if (b.has_down) {          // bn(conv1x1 stride (s, s)(x))
            base_desc(d, b.down, dtc);
            d.B = B; d.T_in = t; d.T_out = to; d.F_in = f; d.F_out = fo; d.KF = 1; d.stride = b.stride; d.stride_f = b.stride;
            d.x = x; d.y = p.res;
            if (b.stride == 1) {           // a plain pointwise conv over the B * t * f positions
                d.T_in = t * f; d.T_out = t * f; d.F_in = 0; d.F_out = 0; d.KF = 0; d.stride = 1; d.stride_f = 0;
                if ((rc = conv1x1(ctx, d, st))) return rc;
            } else if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
            res = p.res;
        }
        // x <- relu(o3 * s + residual)
        if ((rc = vp_se_scale_residual_ex(ctx, dt, p.o3, C, 0, p.se_s, res, C, 0, xn, C, 0, B, to * fo, C, 1, st))) return rc;
        void* tmp = x; x = xn; xn = tmp;
        t = to; f = fo;
    }
    // ASP over time on (B, T4, F4*C4), then bn2 -> linear -> bn3 (folded + permuted at pack time)
    const int Casp = p.F4 * p.C4;
    if (w->asp.C != Casp) VP_FAIL(ctx, VP_EINVAL, "resnetse: asp.C %d != %d", w->asp.C, Casp);
    if ((rc = vp_time_moments(ctx, dt, x, Casp, B, t, Casp, 1e-12f, 0, p.stats, st))) return rc;
    VpAspBufs ab{p.h, p.e, nullptr, nullptr, p.stats, p.rowbias, p.pooled};
    if ((rc = vp_run_asp(ctx, w->asp, dtc, x, Casp, nullptr, B, t, ab, st))) return rc;
    return vp_dense_f32_ex(ctx, p.pooled, 2 * Casp, w->lin_w, 0, w->lin_b, nullptr, nullptr, B, w->embd_dim, 2 * Casp,
                           VP_ACT_NONE, emb, w->embd_dim, st);
}

}  // extern "C"
