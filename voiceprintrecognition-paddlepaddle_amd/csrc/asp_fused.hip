// Attentive statistics pooling, second half, as ONE kernel (bf16 engine):
//   logits = conv1x1(h)  (att -> C)      pooling.py:112  (self.conv)
//   attn   = softmax over time           pooling.py:115-121 (mask of ones)
//   mean, std = weighted statistics      pooling.py:122   -> pooled (B, 2C)
// The (B*T, C) f32 logits never exist: each wave owns 32 channels of one utterance, keeps its 8
// weight fragments (32 x att bf16) in registers, streams the utterance's frames (staged 16 at a time through LDS by
// coalesced loads) through the matrix cores and folds every 16x16 logit tile into a per-lane online softmax
// (running max, sum p, sum p*x, sum p*x^2 with x centred on the plain time mean for conditioning).
// Lanes that share a channel merge with two xor-shuffles at the end; waves never need to talk.
// Roofline: HBM/L2 -- it must read x once (T*C*2 B per utterance = 0.92 MB) and h (T*att*2 B, L2-hot).
#include "common.h"

namespace {

constexpr int AF_ATT = 128;
constexpr float AF_LOG2E = 1.4426950408889634f;

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
    const float f1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);       // maxima are kept in the log2 domain
    const float f2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - M);
    s0 = s0 * f1 + a0 * f2; s1 = s1 * f1 + a1 * f2; s2 = s2 * f1 + a2 * f2; m = M;
}

constexpr int AF_HROW = 256;          // bytes per staged h row (128 att bf16), 16-B chunks XOR-swizzled by row
constexpr int AF_XROW = 272;          // bytes per staged x row (128 channels bf16 + 16 pad: the 4 frame groups of a
                                      // result-layout read land on disjoint banks)

__global__ __launch_bounds__(256) void asp_fused_kernel(AspFusedArgs a) {
    __shared__ __attribute__((aligned(16))) char hs[2][16 * AF_HROW];
    __shared__ __attribute__((aligned(16))) char xs[2][16 * AF_XROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int cblk = blockIdx.x * 128;
    const int c0 = cblk + wv * 32;
    const size_t row0 = (size_t)b * a.T;

    // B operand: weight rows (channels) c0 + ni*16 + li, k chunk ks*4 + g
    bf16x8 wf[2][4];
    float biasl[2], mu0[2];
    int ch[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ch[ni] = min(c0 + ni * 16 + li, a.C - 1);
        const bf16_t* wr = a.w + (size_t)ch[ni] * AF_ATT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ni][ks] = *reinterpret_cast<const bf16x8*>(wr + (ks * 4 + g) * 8);
        biasl[ni] = a.bias[ch[ni]] * AF_LOG2E;
        mu0[ni] = a.center ? a.center[(size_t)b * a.ldc + ch[ni]] : 0.f;
    }
    float mx[2] = {-INFINITY, -INFINITY}, s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

    const int ntile = (a.T + 15) / 16;
    // Frame tile mt (16 frames): h (16 x 256 B, contiguous in memory) and this workgroup's x block (16 x 256 B) are each
    // ONE 16-byte load per thread, 16 lanes per row -- fully coalesced -- staged through LDS and re-read in the MFMA
    // operand / result layouts.  (Loading them directly in those layouts -- 2-byte and 16-row-strided accesses -- made
    // the vector-memory path the bottleneck: 160 us for 30 GFLOP.)
    const int srow = tid >> 4, schunk = tid & 15;
    const int xcol = min(cblk + schunk * 8, a.C - 8);                  // clamped: columns past C feed lanes that never store
    auto gload = [&](int mt, uint4& hv, uint4& xv) {
        const size_t t = row0 + min(mt * 16 + srow, a.T - 1);
        hv = *reinterpret_cast<const uint4*>(a.h + t * AF_ATT + schunk * 8);
        xv = *reinterpret_cast<const uint4*>(a.x + t * a.ldx + xcol);
    };
    auto swrite = [&](int buf, const uint4& hv, const uint4& xv) {
        *reinterpret_cast<uint4*>(hs[buf] + srow * AF_HROW + ((schunk ^ srow) << 4)) = hv;
        *reinterpret_cast<uint4*>(xs[buf] + srow * AF_XROW + (schunk << 4)) = xv;
    };
    auto compute_tile = [&](int mt, int buf) {
        const int t0 = mt * 16 + g * 4;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hs[buf] + li * AF_HROW + (((ks * 4 + g) ^ li) << 4));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf, wf[ni][ks], acc[ni], 0, 0, 0);
        }
        // acc[ni][r] = logit(frame t0 + r, channel ch[ni]) - bias.  The softmax runs in the log2 domain (logits pre-multiplied by
        // log2 e, v_exp_f32 directly) and without branches: the kernel is VALU-bound (~4 cycles per instruction, 4 waves per
        // SIMD), and expf's range reduction plus the divergent "new maximum" paths were half of its ~320 instructions per tile.
        const int tlim = a.T - t0;                                   // frames r < tlim exist
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (r < tlim) ? fmaf(acc[ni][r], AF_LOG2E, biasl[ni]) : -INFINITY;
            const float m4 = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
            const float mnew = fmaxf(fmaxf(mx[ni], m4), -1e30f);       // finite even when every frame so far is masked
            const float f = __builtin_amdgcn_exp2f(mx[ni] - mnew);      // exp2(-inf) = 0 on the first tile
            mx[ni] = mnew;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(e[r] - mnew);    // 0 for frames past T
                const bf16_t xb = *reinterpret_cast<const bf16_t*>(xs[buf] + (g * 4 + r) * AF_XROW + (wv * 32 + ni * 16 + li) * 2);
                const float xv = (float)xb - mu0[ni];
                const float t = pr * xv;
                p0 += pr; p1 += t; p2 = fmaf(t, xv, p2);
            }
            s0[ni] = fmaf(s0[ni], f, p0); s1[ni] = fmaf(s1[ni], f, p1); s2[ni] = fmaf(s2[ni], f, p2);
        }
    };
    uint4 hv, xv;
    gload(0, hv, xv);
    swrite(0, hv, xv);
    __syncthreads();
    for (int mt = 0; mt < ntile; ++mt) {
        const bool more = mt + 1 < ntile;
        if (more) gload(mt + 1, hv, xv);
        compute_tile(mt, mt & 1);
        if (more) swrite((mt + 1) & 1, hv, xv);
        __syncthreads();
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
    if (att != AF_ATT || B > 65535 || T < 1 || C < 8 || (C | ldx) & 7 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h)) & 15))
        return VP_EUNSUP;
    AspFusedArgs a;
    a.h = (const bf16_t*)h; a.w = (const bf16_t*)w; a.bias = bias; a.x = (const bf16_t*)x; a.center = center;
    a.pooled = pooled; a.ldx = ldx; a.ldc = ldc; a.T = T; a.C = C; a.eps = eps;
    hipLaunchKernelGGL(asp_fused_kernel, dim3((C + 127) / 128, B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "asp_fused");
    return VP_OK;
}
