// CAM++ backbone forward (eval mode): the kernels that are specific to it and the launch graph.
//
// Reference: CAMPPlus.forward (ppvector/models/campplus.py:331-335) = FCM head (:246-281, BasicResBlock
// :211-243) -> TDNNLayer k5 s2 (:38-64) -> 3 x [CAMDenseTDNNBlock (:145-173) of CAMDenseTDNNLayer
// (:109-142, CAMLayer :67-106) -> TransitLayer (:176-189)] -> BN-ReLU -> statistics_pooling (:24-30)
// -> DenseLayer + BN (:192-208).
// Layout: FCM activations are (B, T, F', 32) -- position-major with channels fastest -- so the 2-D
// convs run on the conv GEMM (taps = row shifts in (t, f)), and the FCM output IS the (B, T, 320)
// frame-major input of the TDNN (the reference's reshape (B, C*F', T) becomes a weight permutation
// at pack time).  The DenseNet-style concat is ONE (B*T', Cmax) buffer per block: layer i reads
// columns [0, C_i) through the conv GEMM's input prologue (its own BN + ReLU applied while staging)
// and writes its 32 new channels in place -- none of the reference's 52 concat copies exist.
#include "common.h"

namespace {

// ------------------------------------------------------------------ FCM conv1: 1 -> 32 channels, 3x3, BN, ReLU
// HBM-bound: reads (B,T,F) once, writes (B,T,F,32).  4 threads per position, 8 channels each.
template <typename TI, typename TO>
struct Fcm1Args {
    const TI* x; TO* y; const float* w; const float* bias; const float* scale; const float* shift;   // w [32][9], tap = kt*3 + kf
    int B, T, F, C; long long total;
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void fcm_conv1_kernel(Fcm1Args<TI, TO> a) {
    __shared__ float sw[64 * 9], sb[64], ss[64], sh[64];
    const int groups = a.C >> 3;
    for (int i = threadIdx.x; i < a.C * 9; i += 256) sw[i] = a.w[i];
    if (threadIdx.x < a.C) { sb[threadIdx.x] = a.bias[threadIdx.x]; ss[threadIdx.x] = a.scale[threadIdx.x]; sh[threadIdx.x] = a.shift[threadIdx.x]; }
    __syncthreads();
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < a.total; idx += (long long)gridDim.x * 256) {
        const long long pos = idx / groups;
        const int cg = (int)(idx - pos * groups) * 8;
        const int f = (int)(pos % a.F);
        const long long bt = pos / a.F;
        const int t = (int)(bt % a.T);
        const long long b = bt / a.T;
        float in[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int ts = t + kt - 1, fs = f + kf - 1;
                const bool ok = ts >= 0 && ts < a.T && fs >= 0 && fs < a.F;
                in[kt * 3 + kf] = ok ? vp_to_f32(a.x[(b * a.T + (ok ? ts : 0)) * a.F + (ok ? fs : 0)]) : 0.f;
            }
        TO out[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float s = sb[cg + c];
#pragma unroll
            for (int k = 0; k < 9; ++k) s += sw[(cg + c) * 9 + k] * in[k];
            out[c] = vp_from_f32<TO>(fmaxf(s * ss[cg + c] + sh[cg + c], 0.f));
        }
        TO* dst = a.y + pos * a.C + cg;
        if constexpr (sizeof(TO) == 2) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
        } else {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
            *reinterpret_cast<uint4*>(dst + 4) = *reinterpret_cast<const uint4*>(out + 4);
        }
    }
}

// ------------------------------------------------------------------ CAM context: mean over time + segment means
// campplus.py:92: context = x.mean(-1) + seg_pooling(x) (100-frame ceil-mode, exclusive average).
// ctx[b*nseg + s][c] = mean_t x[b,t,c] + mean_{t in seg s} x[b,t,c].   One workgroup per utterance.
template <typename T>
struct CtxArgs { const T* x; float* ctx; int ldx, Tn, C, seg_len, nseg; };

template <typename T>
__global__ __launch_bounds__(256) void cam_ctx_kernel(CtxArgs<T> a) {
    extern __shared__ float sm[];                 // [groups][nseg][C]
    const int b = blockIdx.x;
    const int groups = 256 / a.C;                 // C <= 256 (host-checked), C divides 256
    const int c = threadIdx.x % a.C, gq = threadIdx.x / a.C;
    const T* xb = a.x + (size_t)b * a.Tn * a.ldx;
    if (gq < groups) {
        for (int s = 0; s < a.nseg; ++s) {
            const int t0 = s * a.seg_len, t1 = min(t0 + a.seg_len, a.Tn);
            float acc = 0.f;
            for (int t = t0 + gq; t < t1; t += groups) acc += vp_to_f32(xb[(size_t)t * a.ldx + c]);
            sm[(gq * a.nseg + s) * a.C + c] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x < a.C) {
        float tot = 0.f;
        for (int s = 0; s < a.nseg; ++s) {
            float v = 0.f;
            for (int q = 0; q < groups; ++q) v += sm[(q * a.nseg + s) * a.C + c];
            sm[s * a.C + c] = v;                  // group 0's slots now hold the segment sums
            tot += v;
        }
        const float mean = tot / (float)a.Tn;
        for (int s = 0; s < a.nseg; ++s) {
            const int len = min(a.seg_len, a.Tn - s * a.seg_len);
            a.ctx[((size_t)b * a.nseg + s) * a.C + c] = mean + sm[s * a.C + c] / (float)len;
        }
    }
}

// ------------------------------------------------------------------ CAM context gate in one launch
// gate[b, s, :] = sigmoid(W2 relu(W1 (mean_t x[b] + mean_{t in seg s} x[b]) + b1) + b2)   (campplus.py:88-94).
// One workgroup per utterance.  Phase 1 reads the utterance's (T, C) block with 16-byte loads, 16 lanes per frame and 16
// frame groups in flight (the first version walked ~75 frames per thread with dependent 2-byte loads: 20 us, plus two
// dense launches of ~5 us each per layer, 52 layers).  Phases 2 / 3: a thread per output, weights input-major [in][out].
template <typename T>
struct GateArgs { const T* x; const float* w1; const float* b1; const float* w2; const float* b2; float* gate; int ldx, Tn, C, H, G, seg_len, nseg; };

template <typename T>
__global__ __launch_bounds__(256) void cam_gate_kernel(GateArgs<T> a) {
    extern __shared__ float sm[];                 // part[16][nseg][C] | ctx[nseg][C] | h[nseg][H] | w1[C][H] | w2[H][G]
    constexpr int V = 16 / (int)sizeof(T);        // channels per 16-byte load
    float* part = sm;
    float* cx = part + 16 * a.nseg * a.C;
    float* hh = cx + a.nseg * a.C;
    float* w1s = hh + a.nseg * a.H;
    float* w2s = w1s + a.C * a.H;
    const int b = blockIdx.x, tid = threadIdx.x;
    // both weight matrices into LDS with one round of independent 16-byte loads (read per output from global, each of the
    // C + H loop trips paid a memory latency: 72 us per layer)
    for (int i = tid; i < a.C * a.H / 4; i += 256) reinterpret_cast<float4*>(w1s)[i] = reinterpret_cast<const float4*>(a.w1)[i];
    for (int i = tid; i < a.H * a.G / 4; i += 256) reinterpret_cast<float4*>(w2s)[i] = reinterpret_cast<const float4*>(a.w2)[i];
    const int lpr = a.C / V;                      // lanes per frame (host: lpr divides 256, lpr <= 16 ... 32)
    const int rgs = min(256 / lpr, 16);           // frame groups in flight (threads past 16 groups idle in phase 1)
    const int cq = (tid % lpr) * V, rg = tid / lpr;
    const T* xb = a.x + (size_t)b * a.Tn * a.ldx;
    for (int s = 0; s < a.nseg; ++s) {
        const int t0 = s * a.seg_len, t1 = min(t0 + a.seg_len, a.Tn);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll 4
        for (int t = t0 + rg; t < t1 && rg < rgs; t += rgs) {
            const uint4 raw = *reinterpret_cast<const uint4*>(xb + (size_t)t * a.ldx + cq);
            const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += vp_to_f32(v[e]);
        }
        if (rg < rgs) {
#pragma unroll
            for (int e = 0; e < V; ++e) part[(rg * a.nseg + s) * a.C + cq + e] = acc[e];
        }
    }
    __syncthreads();
    const int ng = rgs;
    for (int c = tid; c < a.C; c += 256) {
        float tot = 0.f;
        for (int s = 0; s < a.nseg; ++s) {
            float v = 0.f;
            for (int q = 0; q < ng; ++q) v += part[(q * a.nseg + s) * a.C + c];
            cx[s * a.C + c] = v;
            tot += v;
        }
        const float mean = tot / (float)a.Tn;
        for (int s = 0; s < a.nseg; ++s) {
            const int len = min(a.seg_len, a.Tn - s * a.seg_len);
            cx[s * a.C + c] = mean + cx[s * a.C + c] / (float)len;
        }
    }
    __syncthreads();
    for (int o = tid; o < a.nseg * a.H; o += 256) {
        const int s = o / a.H, j = o - s * a.H;
        float acc = a.b1[j];
#pragma unroll 16
        for (int k = 0; k < a.C; ++k) acc += cx[s * a.C + k] * w1s[k * a.H + j];
        hh[o] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int o = tid; o < a.nseg * a.G; o += 256) {
        const int s = o / a.G, j = o - s * a.G;
        float acc = a.b2[j];
#pragma unroll 16
        for (int k = 0; k < a.H; ++k) acc += hh[s * a.H + k] * w2s[k * a.G + j];
        a.gate[((size_t)b * a.nseg + s) * a.G + j] = 1.f / (1.f + __expf(-acc));
    }
}

// ------------------------------------------------------------------ out_nonlinear (BN + ReLU) + statistics pooling
// campplus.py:323-326: relu(bn(x)) -> mean and UNBIASED std over time -> (B, 2C)
template <typename T>
struct StatArgs { const T* x; const float* scale; const float* shift; float* out; int ldx, Tn, C; };

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_stats_kernel(StatArgs<T> a) {
    __shared__ float sm[2][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const float sc = a.scale[cc], sh = a.shift[cc];
    const T* xb = a.x + (size_t)b * a.Tn * a.ldx;
    // centre on the first frame's value for a well-conditioned variance
    const float c0 = fmaxf(vp_to_f32(xb[cc]) * sc + sh, 0.f);
    float s1 = 0.f, s2 = 0.f;
    for (int t = wv; t < a.Tn; t += 4) {
        const float v = fmaxf(vp_to_f32(xb[(size_t)t * a.ldx + cc]) * sc + sh, 0.f) - c0;
        s1 += v; s2 += v * v;
    }
    sm[0][wv][lane] = s1; sm[1][wv][lane] = s2;
    __syncthreads();
    if (wv == 0 && ok) {
        const float t1 = sm[0][0][lane] + sm[0][1][lane] + sm[0][2][lane] + sm[0][3][lane];
        const float t2 = sm[1][0][lane] + sm[1][1][lane] + sm[1][2][lane] + sm[1][3][lane];
        const float n = (float)a.Tn;
        const float md = t1 / n;
        const float var = (t2 - t1 * md) / (n - 1.f);                    // unbiased (paddle std default)
        a.out[(size_t)b * 2 * a.C + c] = c0 + md;
        a.out[(size_t)b * 2 * a.C + a.C + c] = sqrtf(fmaxf(var, 0.f));
    }
}

struct Carver {
    char* base; size_t off;
    explicit Carver(void* p) : base((char*)p), off(0) {}
    void* take(size_t bytes) {
        size_t o = off;
        off += vp_align_up(bytes ? bytes : 1, 256);
        return base ? (void*)(base + o) : nullptr;
    }
};

struct CamPlan {
    void *fa, *fb, *fc;                 // FCM ping-pong (B,T,F,32)
    void *cat[2];                       // D-TDNN concat buffers (B*T', Cmax)
    void* h2;                           // (B*T', bn_ch)
    float *ctx, *c1, *gate, *stats;
    size_t total;
    int Tn, Cmax, nseg;
};

int cam_Tn(int T) { return (T - 1) / 2 + 1; }          // k5 s2 pad 2

void plan_cam(const vp_campplus_weights* w, int B, int T, void* ws, CamPlan& p) {
    const size_t es = vp_dtype_size(w->dtype);
    p.Tn = cam_Tn(T);
    int ch = w->init_channels, cmax = ch;
    for (int b = 0; b < w->n_blocks; ++b) {
        ch += w->block_layers[b] * w->growth;
        if (ch > cmax) cmax = ch;
        ch /= 2;
    }
    p.Cmax = cmax;
    p.nseg = (p.Tn + w->seg_len - 1) / w->seg_len;
    const size_t pos0 = (size_t)B * T * w->feat_dim;
    Carver c(ws);
    p.fa = c.take(pos0 * w->m_channels * es);
    p.fb = c.take(pos0 / 2 * w->m_channels * es + 4096);
    p.fc = c.take(pos0 / 2 * w->m_channels * es + 4096);
    const size_t Mn = (size_t)B * p.Tn;
    p.cat[0] = c.take(Mn * cmax * es);
    p.cat[1] = c.take(Mn * cmax * es);
    p.h2 = c.take(Mn * w->bn_channels * es);
    p.ctx = (float*)c.take((size_t)B * p.nseg * w->bn_channels * 4);
    p.c1 = (float*)c.take((size_t)B * p.nseg * (w->bn_channels / 2) * 4);
    p.gate = (float*)c.take((size_t)B * p.nseg * w->growth * 4);
    p.stats = (float*)c.take((size_t)B * 2 * cmax * 4);
    p.total = c.off;
}

void conv2d_desc(vp_conv1d_desc& d, const vp_tdnn_layer& L, int dt, int B, int T, int F_in, int F_out, int stride_f) {
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dt); d.B = B; d.T_in = T; d.T_out = T;
    d.Cin = L.cin; d.Cout = L.cout; d.KW = L.kw; d.dilation = 1; d.stride = 1;
    d.KF = L.kw == 9 ? 3 : 1;
    d.pad_left = L.kw == 9 ? 1 : 0; d.pad_f = L.kw == 9 ? 1 : 0; d.pad_mode = VP_PAD_ZERO;
    d.F_in = F_in; d.F_out = F_out; d.stride_f = stride_f;
    d.ldx = L.cin; d.ldy = L.cout;
    vp_desc_weights(d, L);
    d.bias = L.bias; d.bn_scale = L.bn_scale; d.bn_shift = L.bn_shift;
}

}  // namespace

// 3x3 conv of a single-channel (B, T, F) map to 32 channels + BN + ReLU -> (B, T, F, 32); shared by the
// CAM++ FCM head (campplus.py:254-255,274) and ResNetSE's stem (resnet_se.py:72-74,124-126).
int vp_conv3x3_c1(vp_ctx* ctx, int dtype, const void* feats, void* out, const float* w, const float* bias,
                  const float* scale, const float* shift, int B, int T, int F, int C, hipStream_t st) {
    if (C < 8 || C > 64 || C % 8) VP_FAIL(ctx, VP_EUNSUP, "conv3x3_c1: %d output channels (8..64, multiple of 8)", C);
    const long long total = (long long)B * T * F * (C / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == VP_BF16) {
        Fcm1Args<bf16_t, bf16_t> a{(const bf16_t*)feats, (bf16_t*)out, w, bias, scale, shift, B, T, F, C, total};
        hipLaunchKernelGGL((fcm_conv1_kernel<bf16_t, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        Fcm1Args<float, float> a{(const float*)feats, (float*)out, w, bias, scale, shift, B, T, F, C, total};
        hipLaunchKernelGGL((fcm_conv1_kernel<float, float>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    VP_LAUNCH_CHECK(ctx, "conv3x3_c1");
    return VP_OK;
}

extern "C" {

// C-ABI doors of the two fused bf16 kernels of the CAM++ forward (tests call them one by one; vp_campplus_fwd calls the launchers)
int vp_cam_block_fwd(vp_ctx* ctx, const vp_cam_layer* layers, int n_layers, void* cat, int ld, int ch0, int B, int Tn, int seg_len,
                     int bn_channels, int growth, vp_stream stream) {
    if (!ctx || !layers || !cat || B <= 0 || Tn <= 0 || seg_len <= 0) VP_FAIL(ctx, VP_EINVAL, "cam_block: bad arguments");
    const int rc = vp_cam_block_bf16(ctx, layers, n_layers, cat, ld, ch0, B, Tn, seg_len, bn_channels, growth, (hipStream_t)stream);
    if (rc == VP_EUNSUP)
        VP_FAIL(ctx, VP_EUNSUP, "cam_block: shape not covered by the fused kernel (bottleneck 128, growth 32, k3 local convs, T' <= 160, "
                                "<= 4 context segments, <= 24 layers, <= 1024 channels)");
    return rc;
}

int vp_resblock_c32_fwd(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv1, const vp_tdnn_layer* conv2, int B, int T, int F,
                        vp_stream stream) {
    if (!ctx || !x || !y || !conv1 || !conv2 || B <= 0 || T <= 0 || F <= 0) VP_FAIL(ctx, VP_EINVAL, "resblock_c32: bad arguments");
    const int rc = vp_resblock_c32_bf16(ctx, x, y, conv1, conv2, B, T, F, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "resblock_c32: shape not covered (32 channels, 3x3, stride 1, distinct 16-byte aligned buffers)");
    return rc;
}

int vp_conv3x3_c32_fwd(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv, const void* res, int relu,
                       const vp_tdnn_layer* shortcut, void* y2, int B, int T, int F_in, int stride_f, const void* c1_feats,
                       const float* c1_w, const float* c1_b, const float* c1_scale, const float* c1_shift, vp_stream stream) {
    if (!ctx || !y || !conv || B <= 0 || T <= 0 || F_in <= 0) VP_FAIL(ctx, VP_EINVAL, "conv3x3_c32: bad arguments");
    const int rc = vp_conv3x3_c32_bf16(ctx, x, y, conv, res, relu, shortcut, y2, B, T, F_in, stride_f, c1_feats, c1_w, c1_b, c1_scale,
                                       c1_shift, (hipStream_t)stream);
    if (rc == VP_EUNSUP) VP_FAIL(ctx, VP_EUNSUP, "conv3x3_c32: shape not covered (32 -> 32 channels, 3x3, frequency stride 1 or 2, 16-byte aligned)");
    return rc;
}

size_t vp_campplus_workspace_bytes(const vp_campplus_weights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0) return 0;
    CamPlan p;
    plan_cam(w, B, T, nullptr, p);
    return p.total;
}

int vp_campplus_fwd(vp_ctx* ctx, const vp_campplus_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !w || !feats || !emb || B <= 0 || T <= 0) VP_FAIL(ctx, VP_EINVAL, "campplus: bad arguments");
    if (!vp_backbone_dtype_ok(w->dtype)) VP_FAIL(ctx, VP_EINVAL, "campplus: bad dtype");
    if (w->m_channels != 32 || w->feat_dim % 8 || w->n_blocks < 1 || w->n_blocks > 4 || w->bn_channels > 256 ||
        256 % w->bn_channels || w->seg_len < 1)
        VP_FAIL(ctx, VP_EUNSUP, "campplus: geometry not built (m_channels 32, feat_dim %% 8 == 0, bn_channels | 256)");
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "campplus: batch too large");
    CamPlan p;
    plan_cam(w, B, T, ws, p);
    if (!ws || ws_bytes < p.total) VP_FAIL(ctx, VP_EWORKSPACE, "campplus: workspace %zu < %zu", ws_bytes, p.total);
    if (p.Tn < 2) VP_FAIL(ctx, VP_EINVAL, "campplus: %d frames are too few", T);
    hipStream_t st = (hipStream_t)stream;
    const int dtc = w->dtype, dt = vp_storage_dtype(dtc);
    int rc;
    vp_conv1d_desc d;

    // ---- FCM head: conv1 (1 -> 32), 2 x [ResBlock s2, ResBlock s1], conv2 s2
    // conv1 (1 -> 32) is produced inside the first ResBlock's kernels' input slab when that path is taken (bf16 engine); its
    // output tensor (B, T, F, 32) then never exists
    bool c1_fused = false;
    if (dt == VP_BF16 && w->res[0].stride == 2 && w->res[0].has_shortcut) {
        const int fr = vp_conv3x3_c32_bf16(ctx, nullptr, p.fb, &w->res[0].conv1, nullptr, 1, &w->res[0].shortcut, p.fc, B, T, w->feat_dim, 2,
                                           feats, w->fcm1_w, w->fcm1_b, w->fcm1_scale, w->fcm1_shift, st);
        if (fr != VP_OK && fr != VP_EUNSUP) return fr;
        c1_fused = fr == VP_OK;
    }
    if (!c1_fused && (rc = vp_conv3x3_c1(ctx, dt, feats, p.fa, w->fcm1_w, w->fcm1_b, w->fcm1_scale, w->fcm1_shift, B, T, w->feat_dim, 32, st)))
        return rc;
    int F = w->feat_dim;
    void* cur = p.fa;
    void* t1 = p.fb;
    void* t2 = p.fc;
    for (int i = 0; i < 4; ++i) {
        const vp_resblock& R = w->res[i];
        const int Fo = R.stride == 2 ? (F - 1) / 2 + 1 : F;
        // a stride-1 block with the identity shortcut: both convs and the residual in one launch (h stays in LDS)
        if (dt == VP_BF16 && !R.has_shortcut && R.stride == 1) {
            const int fr = vp_resblock_c32_bf16(ctx, cur, t2, &R.conv1, &R.conv2, B, T, F, st);
            if (fr != VP_OK && fr != VP_EUNSUP) return fr;
            if (fr == VP_OK) {
                void* tmp = cur; cur = t2; t2 = tmp;
                continue;
            }
        }
        // h = relu(bn1(conv1(x))) [+ the stride-2 block's shortcut bn(conv1x1(x)) from the same input slab]
        const void* sc = cur;
        int fast = VP_EUNSUP;
        if (i == 0 && c1_fused) fast = VP_OK;            // launched above: t1 = h, t2 = shortcut
        else if (dt == VP_BF16)
            fast = vp_conv3x3_c32_bf16(ctx, cur, t1, &R.conv1, nullptr, 1, R.has_shortcut ? &R.shortcut : nullptr, t2, B, T, F, R.stride,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, st);
        if (fast != VP_OK && fast != VP_EUNSUP) return fast;
        if (fast == VP_OK) {
            if (R.has_shortcut) sc = t2;
        } else {
            conv2d_desc(d, R.conv1, dtc, B, T, F, Fo, R.stride);
            d.x = cur; d.y = t1; d.act2 = VP_ACT_RELU;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
            if (R.has_shortcut) {      // bn(conv1x1 stride (s,1))
                conv2d_desc(d, R.shortcut, dtc, B, T, F, Fo, R.stride);
                d.x = cur; d.y = t2;
                if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
                sc = t2;
            }
        }
        // out = relu(bn2(conv2(h)) + shortcut)
        void* outb = R.has_shortcut ? cur : t2;          // never the buffer that holds the shortcut
        fast = VP_EUNSUP;
        if (dt == VP_BF16) fast = vp_conv3x3_c32_bf16(ctx, t1, outb, &R.conv2, sc, 1, nullptr, nullptr, B, T, Fo, 1, nullptr, nullptr, nullptr, nullptr, nullptr, st);
        if (fast != VP_OK && fast != VP_EUNSUP) return fast;
        if (fast != VP_OK) {
            conv2d_desc(d, R.conv2, dtc, B, T, Fo, Fo, 1);
            d.x = t1; d.y = outb; d.res = sc; d.ld_res = R.conv2.cout; d.act2 = VP_ACT_RELU;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        }
        if (!R.has_shortcut) { void* tmp = cur; cur = t2; t2 = tmp; }
        F = Fo;
    }
    {
        const int Fo = (F - 1) / 2 + 1;
        int fast = VP_EUNSUP;
        if (dt == VP_BF16) fast = vp_conv3x3_c32_bf16(ctx, cur, t1, &w->fcm_conv2, nullptr, 1, nullptr, nullptr, B, T, F, 2, nullptr, nullptr, nullptr, nullptr, nullptr, st);
        if (fast != VP_OK && fast != VP_EUNSUP) return fast;
        if (fast != VP_OK) {
            conv2d_desc(d, w->fcm_conv2, dtc, B, T, F, Fo, 2);
            d.x = cur; d.y = t1; d.act2 = VP_ACT_RELU;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        }
        F = Fo;
    }
    const int Cf = F * w->m_channels;                    // = tdnn.cin: (B, T, F', 32) read as (B, T, F'*32)
    if (w->tdnn.cin != Cf) VP_FAIL(ctx, VP_EINVAL, "campplus: tdnn.cin %d != %d", w->tdnn.cin, Cf);

    // ---- TDNN: conv k5 stride 2 zero-pad 2 -> BN -> ReLU, into columns [0, init) of cat[0]
    const int Tn = p.Tn, ld = p.Cmax;
    memset(&d, 0, sizeof(d));
    vp_desc_dtype(d, dtc); d.B = B; d.T_in = T; d.T_out = Tn; d.Cin = Cf; d.Cout = w->tdnn.cout;
    d.KW = w->tdnn.kw; d.dilation = 1; d.stride = 2; d.pad_left = (w->tdnn.kw - 1) / 2; d.pad_mode = VP_PAD_ZERO;
    d.x = t1; d.ldx = Cf; vp_desc_weights(d, w->tdnn); d.bias = w->tdnn.bias; d.bn_scale = w->tdnn.bn_scale; d.bn_shift = w->tdnn.bn_shift;
    d.act2 = VP_ACT_RELU; d.y = p.cat[0]; d.ldy = ld;
    if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;

    // ---- D-TDNN blocks
    int ch = w->init_channels, li = 0, cb = 0;
    const int bnc = w->bn_channels, gr = w->growth, nseg = p.nseg;
    for (int b = 0; b < w->n_blocks; ++b) {
        void* cat = p.cat[cb];
        // the whole block as one kernel, a resident workgroup per utterance (cam_block.hip), when the shape is covered
        int fused = VP_EUNSUP;
        if (dt == VP_BF16)
            fused = vp_cam_block_bf16(ctx, &w->layers[li], w->block_layers[b], cat, ld, ch, B, Tn, w->seg_len, bnc, gr, st);
        if (fused != VP_OK && fused != VP_EUNSUP) return fused;
        if (fused == VP_OK) { li += w->block_layers[b]; ch += w->block_layers[b] * gr; }
        for (int l = 0; l < w->block_layers[b] && fused != VP_OK; ++l, ++li) {
            const vp_cam_layer& L = w->layers[li];
            // h2 = relu(bn2(linear1(relu(bn1(x[:, :ch])))))  -- bn1+relu is the conv's input prologue
            memset(&d, 0, sizeof(d));
            vp_desc_dtype(d, dtc); d.B = B; d.T_in = Tn; d.T_out = Tn; d.Cin = ch; d.Cout = bnc; d.KW = 1;
            d.dilation = 1; d.stride = 1; d.pad_mode = VP_PAD_ZERO;
            d.x = cat; d.ldx = ld; vp_desc_weights(d, L.linear1); d.bias = L.linear1.bias; d.pro_scale = L.bn1_scale; d.pro_shift = L.bn1_shift;
            d.bn_scale = L.linear1.bn_scale; d.bn_shift = L.linear1.bn_shift; d.act2 = VP_ACT_RELU; d.y = p.h2; d.ldy = bnc;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
            // context gate m = sigmoid(W2 relu(W1 (mean + segmean) + b1) + b2), one row per (utterance, segment)
            {
                const int V = dt == VP_BF16 ? 8 : 4, lpr = bnc / V;
                const size_t smem = (size_t)(16 * nseg * bnc + nseg * bnc + nseg * (bnc / 2) + bnc * (bnc / 2) + (bnc / 2) * gr) * sizeof(float);
                static bool gate_attr = false;
                if (!gate_attr) {
                    VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cam_gate_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cam_gate_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    gate_attr = true;
                }
                if (smem > 128 * 1024 || bnc % V || lpr > 256 || 256 % lpr || (bnc * (bnc / 2)) % 4 || ((bnc / 2) * gr) % 4)
                    VP_FAIL(ctx, VP_EUNSUP, "campplus: utterance too long / bottleneck width not supported by the context-gate kernel");
                if (dt == VP_BF16) {
                    GateArgs<bf16_t> ga{(const bf16_t*)p.h2, L.ctx_w1, L.ctx_b1, L.ctx_w2, L.ctx_b2, p.gate, bnc, Tn, bnc, bnc / 2, gr, w->seg_len, nseg};
                    hipLaunchKernelGGL(cam_gate_kernel<bf16_t>, dim3(B), dim3(256), smem, st, ga);
                } else {
                    GateArgs<float> ga{(const float*)p.h2, L.ctx_w1, L.ctx_b1, L.ctx_w2, L.ctx_b2, p.gate, bnc, Tn, bnc, bnc / 2, gr, w->seg_len, nseg};
                    hipLaunchKernelGGL(cam_gate_kernel<float>, dim3(B), dim3(256), smem, st, ga);
                }
                VP_LAUNCH_CHECK(ctx, "cam_gate");
            }
            // y = linear_local(h2) * m, written in place as the layer's new channels
            memset(&d, 0, sizeof(d));
            vp_desc_dtype(d, dtc); d.B = B; d.T_in = Tn; d.T_out = Tn; d.Cin = bnc; d.Cout = gr;
            d.KW = L.local.kw; d.dilation = L.local.dil; d.stride = 1; d.pad_left = L.local.dil * (L.local.kw - 1) / 2;
            d.pad_mode = VP_PAD_ZERO; d.x = p.h2; d.ldx = bnc; vp_desc_weights(d, L.local); d.bias = L.local.bias;
            d.gate = p.gate; d.gate_len = w->seg_len; d.gate_nseg = nseg; d.y = cat; d.ldy = ld; d.yoff = ch;
            if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
            ch += gr;
        }
        // transit: linear(relu(bn(x))) -> ch/2, into the other concat buffer
        const vp_transit& Tr = w->transit[b];
        memset(&d, 0, sizeof(d));
        vp_desc_dtype(d, dtc); d.B = B; d.T_in = Tn; d.T_out = Tn; d.Cin = ch; d.Cout = ch / 2; d.KW = 1;
        d.dilation = 1; d.stride = 1; d.pad_mode = VP_PAD_ZERO; d.x = cat; d.ldx = ld; vp_desc_weights(d, Tr.linear); d.bias = Tr.linear.bias;
        d.pro_scale = Tr.bn_scale; d.pro_shift = Tr.bn_shift; d.y = p.cat[cb ^ 1]; d.ldy = ld;
        if ((rc = vp_conv1d_fwd(ctx, &d, st))) return rc;
        cb ^= 1;
        ch /= 2;
    }
    // ---- out_nonlinear + stats pooling + dense (+BN folded)
    if (dt == VP_BF16) {
        StatArgs<bf16_t> sa{(const bf16_t*)p.cat[cb], w->out_bn_scale, w->out_bn_shift, p.stats, ld, Tn, ch};
        hipLaunchKernelGGL(bn_relu_stats_kernel<bf16_t>, dim3((ch + 63) / 64, B), dim3(256), 0, st, sa);
    } else {
        StatArgs<float> sa{(const float*)p.cat[cb], w->out_bn_scale, w->out_bn_shift, p.stats, ld, Tn, ch};
        hipLaunchKernelGGL(bn_relu_stats_kernel<float>, dim3((ch + 63) / 64, B), dim3(256), 0, st, sa);
    }
    VP_LAUNCH_CHECK(ctx, "bn_relu_stats");
    return vp_dense_f32_ex(ctx, p.stats, 2 * ch, w->dense_w, 0, w->dense_b, nullptr, nullptr, B, w->embd_dim, 2 * ch,
                           VP_ACT_NONE, emb, w->embd_dim, st);
}

}  // extern "C"
