// CAM++ context gate of a CAMDenseTDNNLayer in TRAINING mode (CAMLayer.forward, campplus.py:88-106), behind the local conv:
//     ctx[b, s]  = mean_t h[b] + mean over the 100-frame segment s of h[b]            (seg_pooling, avg, ceil_mode)
//     hid[b, s]  = relu(W1 ctx + b1)          m[b, s] = sigmoid(W2 hid + b2)
//     out[b, t]  = y[b, t] * m[b, seg(t)]                                             (y = linear_local(h), computed by the conv GEMM)
// The unfused tape runs this as a segment-mean pass, two 1x1 convs over B * nseg rows (128 rows at B = 64: one GEMM tile each) and a
// scale pass forward, and ~16 launches backward (activation backward, bias sums, weight and data gradient of each tiny conv, the two
// segment passes, autograd's add of the two gradients of h).  Here: ONE launch forward, TWO backward, one workgroup of 1 024 threads per
// utterance -- the two dense layers are a few thousand multiply-adds per utterance and live in LDS (weights included, when they fit)
// between the segment means and the gate.  The passes are latency-bound (150 frames x 128 channels per utterance at the bench shape), so
// the rows are spread over all 16 waves (float4 per thread) and reduced through LDS in a fixed order: 15 / 15 / 4 us per layer at B = 64
// (a first version with 256 threads and scalar loads: 51 / 38 / 27 us).
//   forward : cam_gate_fwd_kernel      -> out, and ctx / hid / m kept for backward
//   backward: cam_gate_bwd_kernel      -> d y (= g * m), the gradient that reaches h THROUGH THE CONTEXT (the caller hands it to the local
//                                         conv's data-gradient GEMM as its epilogue addend), d pre-activations of both dense layers,
//                                         per-utterance column sums of d y (the local conv's bias gradient)
//             cam_gate_wgrad_kernel    -> d W1, d b1, d W2, d b2, d bias of the local conv: sums over the B * nseg rows / B utterances, each
//                                         output by 8 lanes (row r to lane r % 8) folded by three shuffles -- a fixed order, no atomics
// f32 throughout (these layers hold < 0.1 % of a CAM++ step's flops; under enable_amp the unfused 1x1 convs round their operands to bf16,
// this path does not).
#include "common.h"

namespace {

constexpr int CAM_NT = 1024;                 // threads per utterance: the passes are latency-bound, 16 waves keep ~4 rows per thread
constexpr int CAM_RED = 4096;                // floats of one cross-row-group reduction slab: (CAM_NT / quads per row) x channels <= 4096

__host__ __device__ inline int cam_al4(int n) { return (n + 3) & ~3; }

struct CamFwdArgs {
    const float* h; int ldh;
    const float* y; int ldy;
    const float *w1, *b1, *w2, *b2;
    int T, C, H, O, seg_len, nseg;
    float *ctx, *hid, *m;
    float* out; int ldo;
    int w_lds;                               // W1 and W2 staged in LDS (they fit beside the rest)
};

__global__ __launch_bounds__(CAM_NT) void cam_gate_fwd_kernel(CamFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_ctx = sm;                                       // [nseg][C]
    float* s_hid = s_ctx + cam_al4(a.nseg * a.C);            // [nseg][H]
    float* s_m = s_hid + cam_al4(a.nseg * a.H);              // [nseg][O]
    float* red = s_m + cam_al4(a.nseg * a.O);                // [row groups][C]
    float* s_w1 = red + CAM_RED;                             // [H][C]   (w_lds)
    float* s_w2 = s_w1 + a.H * a.C;                          // [O][H]
    const int tid = threadIdx.x, b = blockIdx.x;
    // the two weight matrices on their way into LDS while the segment sums run (H * C and O * H are multiples of 4: C, O are)
    const float *w1 = a.w1, *w2 = a.w2;
    if (a.w_lds) {
        for (int i = tid * 4; i < a.H * a.C; i += CAM_NT * 4) *reinterpret_cast<float4*>(s_w1 + i) = *reinterpret_cast<const float4*>(a.w1 + i);
        for (int i = tid * 4; i < a.O * a.H; i += CAM_NT * 4) *reinterpret_cast<float4*>(s_w2 + i) = *reinterpret_cast<const float4*>(a.w2 + i);
        w1 = s_w1; w2 = s_w2;
    }
    const int QL = a.C >> 2, RGn = CAM_NT / QL;              // float4 per row, row groups in flight
    const int rgi = tid / QL, q = tid - rgi * QL;
    const bool act = rgi < RGn;
    const float* hb = a.h + (size_t)b * a.T * a.ldh;
    // 1. segment sums of h: float4 per thread, RGn rows in flight, partials through LDS, summed in row-group order
    for (int s = 0; s < a.nseg; ++s) {
        const int t0 = s * a.seg_len, t1 = min(a.T, t0 + a.seg_len);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            for (int t = t0 + rgi; t < t1; t += RGn) {
                float v[4];
                vp_load4(hb + (size_t)t * a.ldh + q * 4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
            vp_store4(red + rgi * a.C + q * 4, acc);
        }
        __syncthreads();
        if (tid < a.C) {
            float ssum = 0.f;
            for (int r = 0; r < RGn; ++r) ssum += red[r * a.C + tid];
            s_ctx[s * a.C + tid] = ssum;
        }
        __syncthreads();
    }
    // ctx[s] = segment mean + utterance mean (seg_pooling avg with ceil_mode: the last segment over its own frames)
    if (tid < a.C) {
        float tot = 0.f;
        for (int s = 0; s < a.nseg; ++s) tot += s_ctx[s * a.C + tid];
        const float um = tot / (float)a.T;
        for (int s = 0; s < a.nseg; ++s) {
            const int len = min(a.T, (s + 1) * a.seg_len) - s * a.seg_len;
            const float v = s_ctx[s * a.C + tid] / (float)len + um;
            s_ctx[s * a.C + tid] = v;
            a.ctx[((size_t)b * a.nseg + s) * a.C + tid] = v;
        }
    }
    __syncthreads();
    const int wv = tid >> 6, lc = tid & 63;
    // 2. hid = relu(W1 ctx + b1): one wave per output, lanes over the input channels
    for (int o = wv; o < a.nseg * a.H; o += CAM_NT / 64) {
        const int s = o / a.H, j = o - s * a.H;
        float acc = 0.f;
        for (int c = lc; c < a.C; c += 64) acc = __fmaf_rn(w1[j * a.C + c], s_ctx[s * a.C + c], acc);
        acc = vp_wave_sum(acc);
        if (lc == 0) {
            const float v = fmaxf(acc + a.b1[j], 0.f);
            s_hid[o] = v;
            a.hid[(size_t)b * a.nseg * a.H + o] = v;
        }
    }
    __syncthreads();
    // 3. m = sigmoid(W2 hid + b2)
    for (int o = wv; o < a.nseg * a.O; o += CAM_NT / 64) {
        const int s = o / a.O, j = o - s * a.O;
        float acc = 0.f;
        for (int c = lc; c < a.H; c += 64) acc = __fmaf_rn(w2[j * a.H + c], s_hid[s * a.H + c], acc);
        acc = vp_wave_sum(acc);
        if (lc == 0) {
            const float v = 1.f / (1.f + expf(-(acc + a.b2[j])));
            s_m[o] = v;
            a.m[(size_t)b * a.nseg * a.O + o] = v;
        }
    }
    __syncthreads();
    // 4. the gate
    const int O4 = a.O >> 2;
    for (int i = tid; i < a.T * O4; i += CAM_NT) {
        const int t = i / O4, c = (i - t * O4) * 4, s = t / a.seg_len;
        float v[4];
        vp_load4(a.y + ((size_t)b * a.T + t) * a.ldy + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= s_m[s * a.O + c + e];
        vp_store4(a.out + ((size_t)b * a.T + t) * a.ldo + c, v);
    }
}

struct CamBwdArgs {
    const float* g; int ldg;
    const float* y; int ldy;
    const float *hid, *m, *w1, *w2;
    int T, C, H, O, seg_len, nseg;
    float* dy; int lddy;
    float *dpre1, *dpre2;
    float* dh; int lddh;
    float* dyb;
    int w_lds;
};

__global__ __launch_bounds__(CAM_NT) void cam_gate_bwd_kernel(CamBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_m = sm;                                         // [nseg][O]
    float* s_dp2 = s_m + cam_al4(a.nseg * a.O);              // [nseg][O]
    float* s_dp1 = s_dp2 + cam_al4(a.nseg * a.O);            // [nseg][H]
    float* s_dctx = s_dp1 + cam_al4(a.nseg * a.H);           // [nseg][C]
    float* s_tot = s_dctx + cam_al4(a.nseg * a.C);           // [C]
    float* red = s_tot + cam_al4(a.C);                       // [2][row groups][O]
    float* red2 = red + 2 * CAM_RED;                         // [2][8][O]
    float* s_w1 = red2 + 16 * a.O;                           // [H][C]   (w_lds)
    float* s_w2 = s_w1 + a.H * a.C;                          // [O][H]
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *w1 = a.w1, *w2 = a.w2;
    if (a.w_lds) {
        for (int i = tid * 4; i < a.H * a.C; i += CAM_NT * 4) *reinterpret_cast<float4*>(s_w1 + i) = *reinterpret_cast<const float4*>(a.w1 + i);
        for (int i = tid * 4; i < a.O * a.H; i += CAM_NT * 4) *reinterpret_cast<float4*>(s_w2 + i) = *reinterpret_cast<const float4*>(a.w2 + i);
        w1 = s_w1; w2 = s_w2;
    }
    for (int i = tid; i < a.nseg * a.O; i += CAM_NT) s_m[i] = a.m[(size_t)b * a.nseg * a.O + i];
    __syncthreads();
    // 1. d y = g * m;  d m[s] = sum over the segment of g * y;  sum_t g per segment (-> the column sums of d y);  the sigmoid's backward
    const int OQ = a.O >> 2, RGn = CAM_NT / OQ;
    const int rgi = tid / OQ, q = tid - rgi * OQ;
    const bool act = rgi < RGn;
    float dysum = 0.f;
    for (int s = 0; s < a.nseg; ++s) {
        const int t0 = s * a.seg_len, t1 = min(a.T, t0 + a.seg_len);
        float accm[4] = {0.f, 0.f, 0.f, 0.f}, accg[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            float mv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) mv[e] = s_m[s * a.O + q * 4 + e];
            for (int t = t0 + rgi; t < t1; t += RGn) {
                const size_t r = (size_t)b * a.T + t;
                float gv[4], yv[4], d[4];
                vp_load4(a.g + r * a.ldg + q * 4, gv);
                vp_load4(a.y + r * a.ldy + q * 4, yv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d[e] = gv[e] * mv[e];
                    accm[e] = __fmaf_rn(gv[e], yv[e], accm[e]);
                    accg[e] += gv[e];
                }
                vp_store4(a.dy + r * a.lddy + q * 4, d);
            }
            vp_store4(red + rgi * a.O + q * 4, accm);
            vp_store4(red + CAM_RED + rgi * a.O + q * 4, accg);
        }
        __syncthreads();
        if (tid < 8 * a.O) {                                 // two levels, both in a fixed order
            const int k = tid / a.O, o = tid - k * a.O;
            float s0 = 0.f, s1 = 0.f;
            for (int r = k; r < RGn; r += 8) { s0 += red[r * a.O + o]; s1 += red[CAM_RED + r * a.O + o]; }
            red2[k * a.O + o] = s0;
            red2[8 * a.O + k * a.O + o] = s1;
        }
        __syncthreads();
        if (tid < a.O) {
            float dm = 0.f, gs = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { dm += red2[k * a.O + tid]; gs += red2[8 * a.O + k * a.O + tid]; }
            const float mv = s_m[s * a.O + tid];
            const float d2 = dm * mv * (1.f - mv);
            s_dp2[s * a.O + tid] = d2;
            a.dpre2[((size_t)b * a.nseg + s) * a.O + tid] = d2;
            dysum = __fmaf_rn(gs, mv, dysum);
        }
        // (the next segment's partials overwrite `red` only after its own barrier pair; red2 is re-written behind the first of them)
    }
    if (tid < a.O) a.dyb[(size_t)b * a.O + tid] = dysum;
    __syncthreads();
    // 2. through W2 and the ReLU
    for (int i = tid; i < a.nseg * a.H; i += CAM_NT) {
        const int s = i / a.H, j = i - s * a.H;
        float acc = 0.f;
        for (int o = 0; o < a.O; ++o) acc = __fmaf_rn(s_dp2[s * a.O + o], w2[o * a.H + j], acc);
        const float v = a.hid[(size_t)b * a.nseg * a.H + i] > 0.f ? acc : 0.f;
        s_dp1[i] = v;
        a.dpre1[(size_t)b * a.nseg * a.H + i] = v;
    }
    __syncthreads();
    // 3. through W1: d ctx
    for (int i = tid; i < a.nseg * a.C; i += CAM_NT) {
        const int s = i / a.C, c = i - s * a.C;
        float acc = 0.f;
        for (int j = 0; j < a.H; ++j) acc = __fmaf_rn(s_dp1[s * a.H + j], w1[j * a.C + c], acc);
        s_dctx[i] = acc;
    }
    __syncthreads();
    // 4. the context's share of d h: sum_s d ctx[s] / T + d ctx[seg(t)] / len(seg(t))   (the arithmetic of seg_ctx_bwd_kernel)
    if (tid < a.C) {
        float tot = 0.f;
        for (int s = 0; s < a.nseg; ++s) tot += s_dctx[s * a.C + tid];
        s_tot[tid] = tot / (float)a.T;
    }
    __syncthreads();
    const int QL = a.C >> 2;
    for (int i = tid; i < a.T * QL; i += CAM_NT) {
        const int t = i / QL, c = (i - t * QL) * 4, s = t / a.seg_len;
        const float len = (float)(min(a.T, (s + 1) * a.seg_len) - s * a.seg_len);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s_tot[c + e] + s_dctx[s * a.C + c + e] / len;
        vp_store4(a.dh + ((size_t)b * a.T + t) * a.lddh + c, v);
    }
}

struct CamWgArgs {
    const float *dpre1, *dpre2, *ctx, *hid, *dyb;
    int R, B, C, H, O;
    float *dw1, *db1, *dw2, *db2, *dbl;
};

// eight outputs per wave, the rows of each split over eight lanes (row r to lane r % 8) and folded by three shuffles: a fixed order
__global__ __launch_bounds__(256) void cam_gate_wgrad_kernel(CamWgArgs a) {
    const int lane = threadIdx.x & 63, oi = lane & 7, rs = lane >> 3;
    const long long i = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + oi;
    const long long n1 = (long long)a.H * a.C, n2 = (long long)a.O * a.H;
    const float *pa = nullptr, *pb = nullptr;
    int sa = 0, sb = 0, rows = 0;
    float* dst = nullptr;
    if (i < n1) {
        const int j = (int)(i / a.C), c = (int)(i - (long long)j * a.C);
        pa = a.dpre1 + j; sa = a.H; pb = a.ctx + c; sb = a.C; rows = a.R; dst = a.dw1 + i;
    } else if (i < n1 + n2) {
        const long long k = i - n1;
        const int o = (int)(k / a.H), j = (int)(k - (long long)o * a.H);
        pa = a.dpre2 + o; sa = a.O; pb = a.hid + j; sb = a.H; rows = a.R; dst = a.dw2 + k;
    } else if (i < n1 + n2 + a.H) {
        const int k = (int)(i - n1 - n2);
        pa = a.dpre1 + k; sa = a.H; rows = a.R; dst = a.db1 + k;
    } else if (i < n1 + n2 + a.H + a.O) {
        const int k = (int)(i - n1 - n2 - a.H);
        pa = a.dpre2 + k; sa = a.O; rows = a.R; dst = a.db2 + k;
    } else if (i < n1 + n2 + a.H + 2 * a.O && a.dbl) {
        const int k = (int)(i - n1 - n2 - a.H - a.O);
        pa = a.dyb + k; sa = a.O; rows = a.B; dst = a.dbl + k;
    }
    float acc = 0.f;
    if (pb) {
#pragma unroll 4
        for (int r = rs; r < rows; r += 8) acc = __fmaf_rn(pa[(size_t)r * sa], pb[(size_t)r * sb], acc);
    } else {
#pragma unroll 4
        for (int r = rs; r < rows; r += 8) acc += pa[(size_t)r * sa];
    }
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (rs == 0 && dst) *dst = acc;
}

// LDS floats of the two per-utterance kernels
size_t cam_fwd_lds(int nseg, int C, int H, int O) { return ((size_t)cam_al4(nseg * C) + cam_al4(nseg * H) + cam_al4(nseg * O) + CAM_RED) * sizeof(float); }
size_t cam_bwd_lds(int nseg, int C, int H, int O) {
    return ((size_t)2 * cam_al4(nseg * O) + cam_al4(nseg * H) + cam_al4(nseg * C) + cam_al4(C) + 2 * CAM_RED + 16 * O) * sizeof(float);
}
size_t cam_w_lds(int C, int H, int O) { return ((size_t)H * C + (size_t)O * H) * sizeof(float); }
constexpr size_t CAM_LDS_MAX = 128 * 1024;   // of the CU's 160 KB; above the default 64 KB of a launch: the function attribute is set once per device

int cam_lds_attr(vp_ctx* ctx) {
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cam_gate_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CAM_LDS_MAX));
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cam_gate_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CAM_LDS_MAX));
        attr_set = true;
    }
    return VP_OK;
}

bool cam_shapes_ok(int B, int T, int C, int H, int O, int seg_len) {
    return B > 0 && T > 0 && C > 0 && H > 0 && O > 0 && seg_len > 0;
}
// what the float4 / row-group layout of the two kernels takes
bool cam_layout_ok(int C, int O) { return (C & 3) == 0 && C <= 1024 && (O & 3) == 0 && O <= 128; }
bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int vp_cam_gate_fwd_f32(vp_ctx* ctx, const float* h, int ldh, const float* y, int ldy, const float* w1, const float* b1, const float* w2,
                        const float* b2, int B, int T, int C, int H, int O, int seg_len, float* ctx_out, float* hid_out, float* m_out,
                        float* out, int ldo, vp_stream stream) {
    if (!ctx || !h || !y || !w1 || !b1 || !w2 || !b2 || !ctx_out || !hid_out || !m_out || !out || !cam_shapes_ok(B, T, C, H, O, seg_len) ||
        ldh < C || ldy < O || ldo < O)
        VP_FAIL(ctx, VP_EINVAL, "cam_gate_fwd: bad arguments");
    const int nseg = (T + seg_len - 1) / seg_len;
    const size_t lds = cam_fwd_lds(nseg, C, H, O);
    if (!cam_layout_ok(C, O) || ((ldh | ldy | ldo) & 3) || !al16(h) || !al16(y) || !al16(out) || lds > CAM_LDS_MAX)
        VP_FAIL(ctx, VP_EUNSUP, "cam_gate_fwd: C %d / O %d / %d segments outside the fused kernel's layout, or rows not 16-byte aligned", C, O, nseg);
    const int w_lds = lds + cam_w_lds(C, H, O) <= CAM_LDS_MAX && al16(w1) && al16(w2);
    if (int rc = cam_lds_attr(ctx)) return rc;
    CamFwdArgs a{h, ldh, y, ldy, w1, b1, w2, b2, T, C, H, O, seg_len, nseg, ctx_out, hid_out, m_out, out, ldo, w_lds};
    hipLaunchKernelGGL(cam_gate_fwd_kernel, dim3(B), dim3(CAM_NT), lds + (w_lds ? cam_w_lds(C, H, O) : 0), (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "cam_gate_fwd");
    return VP_OK;
}

int vp_cam_gate_bwd_f32(vp_ctx* ctx, const float* g, int ldg, const float* y, int ldy, const float* hid, const float* m, const float* w1,
                        const float* w2, int B, int T, int C, int H, int O, int seg_len, float* dy, int lddy, float* dpre1, float* dpre2,
                        float* dh, int lddh, float* dyb, vp_stream stream) {
    if (!ctx || !g || !y || !hid || !m || !w1 || !w2 || !dy || !dpre1 || !dpre2 || !dh || !dyb || !cam_shapes_ok(B, T, C, H, O, seg_len) ||
        ldg < O || ldy < O || lddy < O || lddh < C)
        VP_FAIL(ctx, VP_EINVAL, "cam_gate_bwd: bad arguments");
    const int nseg = (T + seg_len - 1) / seg_len;
    const size_t lds = cam_bwd_lds(nseg, C, H, O);
    if (!cam_layout_ok(C, O) || ((ldg | ldy | lddy | lddh) & 3) || !al16(g) || !al16(y) || !al16(dy) || !al16(dh) || lds > CAM_LDS_MAX)
        VP_FAIL(ctx, VP_EUNSUP, "cam_gate_bwd: C %d / O %d / %d segments outside the fused kernel's layout, or rows not 16-byte aligned", C, O, nseg);
    const int w_lds = lds + cam_w_lds(C, H, O) <= CAM_LDS_MAX && al16(w1) && al16(w2);
    if (int rc = cam_lds_attr(ctx)) return rc;
    CamBwdArgs a{g, ldg, y, ldy, hid, m, w1, w2, T, C, H, O, seg_len, nseg, dy, lddy, dpre1, dpre2, dh, lddh, dyb, w_lds};
    hipLaunchKernelGGL(cam_gate_bwd_kernel, dim3(B), dim3(CAM_NT), lds + (w_lds ? cam_w_lds(C, H, O) : 0), (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "cam_gate_bwd");
    return VP_OK;
}

int vp_cam_gate_wgrad_f32(vp_ctx* ctx, const float* dpre1, const float* dpre2, const float* ctx_in, const float* hid, const float* dyb, int B,
                          int nseg, int C, int H, int O, float* dw1, float* db1, float* dw2, float* db2, float* dbl, vp_stream stream) {
    if (!ctx || !dpre1 || !dpre2 || !ctx_in || !hid || !dw1 || !db1 || !dw2 || !db2 || (dbl && !dyb) || B <= 0 || nseg <= 0 || C <= 0 || H <= 0 ||
        O <= 0)
        VP_FAIL(ctx, VP_EINVAL, "cam_gate_wgrad: bad arguments");
    CamWgArgs a{dpre1, dpre2, ctx_in, hid, dyb, B * nseg, B, C, H, O, dw1, db1, dw2, db2, dbl};
    const long long total = (long long)H * C + (long long)O * H + H + 2 * O;
    hipLaunchKernelGGL(cam_gate_wgrad_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "cam_gate_wgrad");
    return VP_OK;
}

}  // extern "C"
