// ERes2Net backbone forward (eval mode): launch graph.
//
// Reference: ERes2Net.forward (ppvector/models/eres2net.py:239-263): conv3x3(1 -> m) + BN + ReLU; four stages of
// BasicBlockERes2Net (:56-108) / BasicBlockERes2Net_diff_AFF (:111-169), strides 1,2,2,2 on both axes; stage outputs
// fused bottom-up by 3x3 stride-2 convs + AFF (:33-53); TemporalStatsPool (pooling.py:128-146); Linear.
// Layout (B, T, F, C) position-major.  Per block: 1x1 (2-D strided when stride 2) -> Hardtanh -> per width-chunk 3x3
// conv (input = a column slice of the previous tensor, output = a column slice of the concat buffer; the Res2 hand-off
// sp + spx[i] comes out of the previous conv's epilogue as `aux`) -> 1x1 with the residual and the Hardtanh fused.
// AFF = column copies into a (P, 2C) buffer, two 1x1 GEMMs (SiLU, tanh fused) and one combine pass.
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

struct ErePlan {
    void *xa, *xb, *o1, *o2, *spin, *cat, *a1, *att, *res;     // block scratch (sized for the largest block)
    void *stage[4];                                            // stage outputs (stage[3] aliases the running buffer copy)
    void *ds, *fcat, *fa1, *fatt, *fuse_a, *fuse_b;            // top-level fusion
    float *stats;
    size_t total;
    int t[5], f[5];                                            // (T, F) at the stem and after each stage
};

int down2(int v) { return (v - 1) / 2 + 1; }
size_t vp_max(size_t a, size_t b) { return a > b ? a : b; }

void plan_ere(const vp_eres2net_weights* w, int B, int T, void* ws, ErePlan& p) {
    const size_t es = vp_dtype_size(w->dtype);
    p.t[0] = T; p.f[0] = w->feat_dim;
    for (int s = 1; s <= 4; ++s) {
        p.t[s] = s == 1 ? p.t[0] : down2(p.t[s - 1]);
        p.f[s] = s == 1 ? p.f[0] : down2(p.f[s - 1]);
    }
    size_t io = (size_t)B * T * w->feat_dim * w->m_channels, mid = 0, sp = 0, cat = 0, a1 = 0;
    int bi = 0;
    for (int s = 0; s < 4; ++s) {
        const size_t pin_first = (size_t)B * p.t[s == 0 ? 1 : s] * p.f[s == 0 ? 1 : s];
        const size_t pout = (size_t)B * p.t[s + 1] * p.f[s + 1];
        for (int j = 0; j < w->stage_blocks[s]; ++j, ++bi) {
            const vp_ere_block& b = w->blk[bi];
            const size_t pin = j == 0 ? pin_first : pout;
            io = vp_max(io, vp_max(pin * b.conv1.cin, pout * b.conv3.cout));
            mid = vp_max(mid, pout * (size_t)(b.width * b.scale));
            sp = vp_max(sp, pout * (size_t)b.width);
            if (b.use_aff) {
                cat = vp_max(cat, pout * (size_t)(2 * b.width));
                a1 = vp_max(a1, pout * (size_t)b.fuse[0].c1.cout);
            }
        }
    }
    Carver c(ws);
    p.xa = c.take(io * es); p.xb = c.take(io * es); p.res = c.take(io * es);
    p.o1 = c.take(mid * es); p.o2 = c.take(mid * es);
    p.spin = c.take(sp * es); p.att = c.take(sp * es);
    p.cat = c.take(cat * es); p.a1 = c.take(a1 * es);
    bi = 0;
    for (int s = 0; s < 4; ++s) {
        bi += w->stage_blocks[s];
        const int C = w->blk[bi - 1].conv3.cout;
        p.stage[s] = c.take((size_t)B * p.t[s + 1] * p.f[s + 1] * C * es);
    }
    size_t fmax = 0, fa = 0;
    for (int k = 0; k < 3; ++k) {
        const size_t P = (size_t)B * p.t[k + 2] * p.f[k + 2];
        fmax = vp_max(fmax, P * (size_t)w->down[k].cout);
        fa = vp_max(fa, P * (size_t)w->fuse[k].c1.cout);
    }
    p.ds = c.take(fmax * es); p.fcat = c.take(2 * fmax * es); p.fa1 = c.take(fa * es); p.fatt = c.take(fmax * es);
    p.fuse_a = c.take(fmax * es); p.fuse_b = c.take(fmax * es);
    const int Cst = p.f[4] * w->blk[w->n_blocks - 1].conv3.cout;
    p.stats = (float*)c.take((size_t)B * 2 * Cst * 4);
    p.total = c.off;
}

void conv_desc(vp_conv1d_desc& d, const vp_tdnn_layer& L, int dt) {
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dt); d.Cin = L.cin; d.Cout = L.cout; d.KW = L.kw; d.dilation = 1; d.stride = 1;
    d.pad_mode = VP_PAD_ZERO; d.ldx = L.cin; d.ldy = L.cout;
    vp_desc_weights(d, L);
    d.bias = L.bias; d.bn_scale = L.bn_scale; d.bn_shift = L.bn_shift;
}

// x (B, t, f, .) -> 2-D conv geometry on d (3x3 pad 1 or 1x1, stride s on both axes)
void geom2d(vp_conv1d_desc& d, int B, int t, int f, int s, bool k3) {
    d.B = B; d.T_in = t; d.F_in = f; d.T_out = s == 2 ? down2(t) : t; d.F_out = s == 2 ? down2(f) : f;
    d.KF = k3 ? 3 : 1; d.stride = s; d.stride_f = s; d.pad_left = k3 ? 1 : 0; d.pad_f = k3 ? 1 : 0;
}

// 1x1 convs over positions: the streaming kernel for the few-channel full-resolution stages (pointwise.hip), else the conv GEMM
int conv1x1(vp_ctx* ctx, const vp_conv1d_desc& d, hipStream_t st) {
    const int rc = vp_pointwise_bf16(ctx, &d, 1, st);
    return rc == VP_EUNSUP ? vp_conv1d_fwd(ctx, &d, st) : rc;
}

// AFF (eres2net.py:33-53): out = x (1 + tanh(att)) + y (1 - tanh(att)), att = BN(conv(SiLU(BN(conv(cat(x, y))))))
int run_aff(vp_ctx* ctx, const vp_aff_weights& A, int dtc, const void* x, int ldx, int xoff, const void* y, int ldy, int yoff,
            void* out, int ldo, int ooff, long long P, int C, void* cat, void* a1, void* att, hipStream_t st) {
    const int dt = vp_storage_dtype(dtc);
    int rc;
    if ((rc = vp_copy_cols(ctx, dt, x, ldx, xoff, cat, 2 * C, 0, P, C, st))) return rc;
    if ((rc = vp_copy_cols(ctx, dt, y, ldy, yoff, cat, 2 * C, C, P, C, st))) return rc;
    if (P > 0x7fffffff / 4) VP_FAIL(ctx, VP_EINVAL, "aff: too many positions");
    vp_conv1d_desc d;
    conv_desc(d, A.c1, dtc);
    d.B = 1; d.T_in = (int)P; d.T_out = (int)P; d.x = cat; d.y = a1; d.act2 = VP_ACT_SILU;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
    conv_desc(d, A.c2, dtc);
    d.B = 1; d.T_in = (int)P; d.T_out = (int)P; d.x = a1; d.y = att; d.act2 = VP_ACT_TANH;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
    return vp_aff_combine(ctx, dt, att, C, x, ldx, xoff, y, ldy, yoff, out, ldo, ooff, P, C, st);
}

}  // namespace

extern "C" {

size_t vp_eres2net_workspace_bytes(const vp_eres2net_weights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0 || w->n_blocks < 4 || w->n_blocks > VP_MAX_ERE_BLOCKS) return 0;
    ErePlan p;
    plan_ere(w, B, T, nullptr, p);
    return p.total;
}

int vp_eres2net_fwd(vp_ctx* ctx, const vp_eres2net_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !w || !feats || !emb || B <= 0 || T <= 0) VP_FAIL(ctx, VP_EINVAL, "eres2net: bad arguments");
    if (!vp_backbone_dtype_ok(w->dtype)) VP_FAIL(ctx, VP_EINVAL, "eres2net: bad dtype");
    int nb = 0;
    for (int s = 0; s < 4; ++s) nb += w->stage_blocks[s];
    if (nb != w->n_blocks || w->n_blocks > VP_MAX_ERE_BLOCKS) VP_FAIL(ctx, VP_EINVAL, "eres2net: block counts do not add up");
    ErePlan p;
    plan_ere(w, B, T, ws, p);
    if (!ws || ws_bytes < p.total) VP_FAIL(ctx, VP_EWORKSPACE, "eres2net: workspace %zu < %zu", ws_bytes, p.total);
    if (p.t[4] < 2) VP_FAIL(ctx, VP_EINVAL, "eres2net: %d frames are too few (unbiased variance over T/8 frames)", T);
    hipStream_t st = (hipStream_t)stream;
    const int dtc = w->dtype, dt = vp_storage_dtype(dtc);
    int rc;
    vp_conv1d_desc d;
    if ((rc = vp_conv3x3_c1(ctx, dt, feats, p.xa, w->c1_w, w->c1_b, w->c1_scale, w->c1_shift, B, T, w->feat_dim, w->m_channels, st)))
        return rc;
    const void* x = p.xa;
    int bi = 0;
    for (int s = 0; s < 4; ++s) {
        for (int j = 0; j < w->stage_blocks[s]; ++j, ++bi) {
            const vp_ere_block& b = w->blk[bi];
            const int tin = j == 0 ? p.t[s == 0 ? 1 : s] : p.t[s + 1], fin = j == 0 ? p.f[s == 0 ? 1 : s] : p.f[s + 1];
            const int to = p.t[s + 1], fo = p.f[s + 1];
            const long long P = (long long)B * to * fo;
            const int wd = b.width, W2 = b.width * b.scale, Co = b.conv3.cout;
            if (b.scale < 1 || b.scale > VP_MAX_ERE_SCALE) VP_FAIL(ctx, VP_EINVAL, "eres2net: scale %d", b.scale);
            // o1 = hardtanh(bn1(conv1x1 stride s (x)))
            if (P > 0x7fffffff / 4) VP_FAIL(ctx, VP_EINVAL, "eres2net: too many positions");
            conv_desc(d, b.conv1, dtc);
            if (b.stride == 1) { d.B = 1; d.T_in = (int)P; d.T_out = (int)P; }
            else geom2d(d, B, tin, fin, b.stride, false);
            d.x = x; d.y = p.o1; d.act2 = VP_ACT_HARDTANH20;
            if ((rc = b.stride == 1 ? conv1x1(ctx, d, st) : vp_conv1d_fwd(ctx, &d, st))) return rc;
            // chunk chain: sp_i = hardtanh(bn_i(conv3x3(in_i))) into o2[:, i*wd : (i+1)*wd]
            for (int i = 0; i < b.scale; ++i) {
                conv_desc(d, b.convs[i], dtc);
                geom2d(d, B, to, fo, 1, true);
                if (i == 0) { d.x = p.o1; d.ldx = W2; d.xoff = 0; }
                else { d.x = p.spin; d.ldx = wd; d.xoff = 0; }
                d.y = p.o2; d.ldy = W2; d.yoff = i * wd; d.act2 = VP_ACT_HARDTANH20;
                const bool more = i + 1 < b.scale;
                if (more && !b.use_aff) {          // next input = sp_i + spx[i+1] straight from this conv's epilogue
                    d.add_in = p.o1; d.ld_add = W2; d.add_off = (i + 1) * wd;
                    d.aux = p.spin; d.ld_aux = wd; d.aux_off = 0;
                }
                if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
                if (more && b.use_aff) {
                    if ((rc = run_aff(ctx, b.fuse[i], dtc, p.o2, W2, i * wd, p.o1, W2, (i + 1) * wd, p.spin, wd, 0, P, wd, p.cat,
                                      p.a1, p.att, st))) return rc;
                }
            }
            // residual branch
            const void* res = x;
            int ld_res = b.conv1.cin;
            if (b.has_shortcut) {
                conv_desc(d, b.shortcut, dtc);
                if (b.stride == 1) { d.B = 1; d.T_in = (int)P; d.T_out = (int)P; }
                else geom2d(d, B, tin, fin, b.stride, false);
                d.x = x; d.y = p.res;
                if ((rc = b.stride == 1 ? conv1x1(ctx, d, st) : vp_conv1d_fwd(ctx, &d, st))) return rc;
                res = p.res; ld_res = Co;
            }
            // out = hardtanh(bn3(conv1x1(o2)) + residual); the last block of a stage lands in the stage buffer
            void* dst = j + 1 == w->stage_blocks[s] ? p.stage[s] : (x == p.xa ? p.xb : p.xa);
            conv_desc(d, b.conv3, dtc);
            d.B = 1; d.T_in = (int)P; d.T_out = (int)P; d.x = p.o2; d.y = dst;
            d.res = res; d.ld_res = ld_res; d.act2 = VP_ACT_HARDTANH20;
            if ((rc = conv1x1(ctx, d, st))) return rc;
            x = dst;
        }
    }
    // bottom-up fusion: f12 = AFF(o2, down1(o1)); f123 = AFF(o3, down2(f12)); f1234 = AFF(o4, down3(f123));
    // ERes2NetV2 (first_fuse = 2): only AFF(o4, layer3_ds(o3))
    if (w->first_fuse < 0 || w->first_fuse > 2) VP_FAIL(ctx, VP_EINVAL, "eres2net: first_fuse %d", w->first_fuse);
    const void* low = p.stage[w->first_fuse];
    void* fout = nullptr;
    for (int k = w->first_fuse; k < 3; ++k) {
        const int C = w->down[k].cout;
        const long long P = (long long)B * p.t[k + 2] * p.f[k + 2];
        conv_desc(d, w->down[k], dtc);
        geom2d(d, B, p.t[k + 1], p.f[k + 1], 2, true);
        d.x = low; d.y = p.ds;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        fout = (k & 1) ? p.fuse_b : p.fuse_a;
        if ((rc = run_aff(ctx, w->fuse[k], dtc, p.stage[k + 1], C, 0, p.ds, C, 0, fout, C, 0, P, C, p.fcat, p.fa1, p.fatt, st)))
            return rc;
        low = fout;
    }
    // TSTP over time per (freq, channel); the Linear's columns were permuted to this order at pack time
    const int Cst = p.f[4] * w->down[2].cout;
    if ((rc = vp_time_moments(ctx, dt, fout, Cst, B, p.t[4], Cst, 1e-8f, 1, p.stats, st))) return rc;
    return vp_dense_f32_ex(ctx, p.stats, 2 * Cst, w->seg_w, 0, w->seg_b, nullptr, nullptr, B, w->embd_dim, 2 * Cst, VP_ACT_NONE,
                           emb, w->embd_dim, st);
}

}  // extern "C"
