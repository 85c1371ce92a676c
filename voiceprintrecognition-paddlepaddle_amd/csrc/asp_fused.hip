// Attentive statistics pooling, second half, as ONE kernel (bf16 engine):
//   logits = conv1x1(h)  (att -> C)      pooling.py:112  (self.conv)
//   attn   = softmax over time           pooling.py:115-121 (mask of ones)
//   mean, std = weighted statistics      pooling.py:122   -> pooled (B, 2C)
// The (B*T, C) f32 logits never exist: each wave owns 32 channels of one utterance, keeps its 8
// weight fragments (32 x att bf16) in registers, streams the utterance's frames through the matrix
// cores 16 at a time and folds every 16x16 logit tile into a per-lane online softmax
// (running max, sum p, sum p*x, sum p*x^2 with x centred on the plain time mean for conditioning).
// Lanes that share a channel merge with two xor-shuffles at the end; waves never need to talk.
// Roofline: HBM/L2 -- it must read x once (T*C*2 B per utterance = 0.92 MB) and h (T*att*2 B, L2-hot).
#include "common.h"

namespace {

constexpr int AF_ATT = 128;

struct AspFusedArgs {
    const bf16_t* h;        // (B*T, att)
    const bf16_t* w;        // [C][att]
    const float* bias;      // [C]
    const bf16_t* x;        // (B*T, ldx)
    const float* center;    // (B, ldc): time mean per channel
    float* pooled;          // (B, 2C)
    int ldx, ldc, T, C; float eps;
};

__device__ __forceinline__ void merge_state(float& m, float& s0, float& s1, float& s2, int off) {
    const float m2 = __shfl_xor(m, off), a0 = __shfl_xor(s0, off), a1 = __shfl_xor(s1, off), a2 = __shfl_xor(s2, off);
    const float M = fmaxf(m, m2);
    const float f1 = (m == -INFINITY) ? 0.f : expf(m - M);
    const float f2 = (m2 == -INFINITY) ? 0.f : expf(m2 - M);
    s0 = s0 * f1 + a0 * f2; s1 = s1 * f1 + a1 * f2; s2 = s2 * f1 + a2 * f2; m = M;
}

__global__ __launch_bounds__(256) void asp_fused_kernel(AspFusedArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 128 + wv * 32;
    if (c0 >= a.C) return;                                   // wave-uniform
    const size_t row0 = (size_t)b * a.T;

    // B operand: weight rows (channels) c0 + ni*16 + li, k chunk ks*4 + g
    bf16x8 wf[2][4];
    float bias[2], mu0[2];
    int ch[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ch[ni] = min(c0 + ni * 16 + li, a.C - 1);
        const bf16_t* wr = a.w + (size_t)ch[ni] * AF_ATT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ni][ks] = *reinterpret_cast<const bf16x8*>(wr + (ks * 4 + g) * 8);
        bias[ni] = a.bias[ch[ni]];
        mu0[ni] = a.center ? a.center[(size_t)b * a.ldc + ch[ni]] : 0.f;
    }
    float mx[2] = {-INFINITY, -INFINITY}, s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

    const int ntile = (a.T + 15) / 16;
    // loads of frame tile mt+1 are issued before the MFMAs / exps of tile mt (register double buffer)
    auto load_tile = [&](int mt, bf16x8 (&hf)[4], bf16_t (&xr)[2][4]) {
        const int ta = min(mt * 16 + li, a.T - 1);                    // A operand: frames mt*16 + li of h
        const bf16_t* hr = a.h + (row0 + ta) * AF_ATT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) hf[ks] = *reinterpret_cast<const bf16x8*>(hr + (ks * 4 + g) * 8);
        const int t0 = mt * 16 + g * 4;                                // result rows: frames t0 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t xo = (row0 + min(t0 + r, a.T - 1)) * a.ldx;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) xr[ni][r] = a.x[xo + ch[ni]];
        }
    };
    auto compute_tile = [&](int mt, const bf16x8 (&hf)[4], const bf16_t (&xr)[2][4]) {
        const int t0 = mt * 16 + g * 4;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[ks], wf[ni][ks], acc[ni], 0, 0, 0);
        // acc[ni][r] = logit(frame t0 + r, channel ch[ni]) - bias
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float e[4];
            float m4 = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (t0 + r < a.T) ? acc[ni][r] + bias[ni] : -INFINITY;
                m4 = fmaxf(m4, e[r]);
            }
            if (m4 > mx[ni]) {
                const float f = (mx[ni] == -INFINITY) ? 0.f : expf(mx[ni] - m4);
                s0[ni] *= f; s1[ni] *= f; s2[ni] *= f; mx[ni] = m4;
            }
            if (mx[ni] != -INFINITY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = expf(e[r] - mx[ni]);              // exp(-inf) = 0 for frames past T
                    const float xv = (float)xr[ni][r] - mu0[ni];
                    s0[ni] += p; s1[ni] += p * xv; s2[ni] += p * xv * xv;
                }
            }
        }
    };
    bf16x8 h0[4], h1[4];
    bf16_t x0[2][4], x1[2][4];
    load_tile(0, h0, x0);
    for (int mt = 0; mt < ntile; mt += 2) {
        const bool odd = mt + 1 < ntile;
        if (odd) load_tile(mt + 1, h1, x1);
        compute_tile(mt, h0, x0);
        if (odd) {
            if (mt + 2 < ntile) load_tile(mt + 2, h0, x0);
            compute_tile(mt + 1, h1, x1);
        }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        merge_state(mx[ni], s0[ni], s1[ni], s2[ni], 16);
        merge_state(mx[ni], s0[ni], s1[ni], s2[ni], 32);
        const int c = c0 + ni * 16 + li;
        if (g == 0 && c < a.C) {
            const float md = s1[ni] / s0[ni];
            const float var = s2[ni] / s0[ni] - md * md;
            a.pooled[(size_t)b * 2 * a.C + c] = mu0[ni] + md;
            a.pooled[(size_t)b * 2 * a.C + a.C + c] = sqrtf(fmaxf(var, a.eps));
        }
    }
}

}  // namespace

// VP_EUNSUP when the shape is not covered (caller falls back to the two-kernel path).
int vp_asp_fused_bf16(vp_ctx* ctx, const void* h, const void* w, const float* bias, const void* x, int ldx,
                      const float* center, int ldc, int B, int T, int C, int att, float eps, float* pooled,
                      hipStream_t st) {
    if (att != AF_ATT || B > 65535 || T < 1) return VP_EUNSUP;
    AspFusedArgs a;
    a.h = (const bf16_t*)h; a.w = (const bf16_t*)w; a.bias = bias; a.x = (const bf16_t*)x; a.center = center;
    a.pooled = pooled; a.ldx = ldx; a.ldc = ldc; a.T = T; a.C = C; a.eps = eps;
    hipLaunchKernelGGL(asp_fused_kernel, dim3((C + 127) / 128, B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "asp_fused");
    return VP_OK;
}
