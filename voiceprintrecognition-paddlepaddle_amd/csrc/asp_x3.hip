// Attentive statistics pooling, second half, in SPLIT PRECISION and as ONE kernel: the hl32 form of asp_fused.hip.
//   logits = conv1x1(h)  (att -> C)      pooling.py:112  (self.conv)
//   attn   = softmax over time           pooling.py:115-121 (mask of ones)
//   mean, std = weighted statistics      pooling.py:122   -> pooled (B, 2C)
// h (B*T, att) and x (B*T, ldx) are stored as split bf16 planes (vpmi.h: VP_HL32, value = hi + lo); the weights arrive as f32 and are
// split once per wave into register fragments; logits = h_hi w_hi + h_hi w_lo + h_lo w_hi on the bf16 matrix cores, f32 accumulate
// (conv_gemm_impl.h: x3_t) -- the f32 engine's (B*T, C) logits tensor (468 MB at 256 x 3 s, written and read back) never exists.
// Each wave owns 32 channels of one utterance and streams its frames 16 at a time: the h tile (16 x 512 B) is staged by coalesced
// 16-byte loads in the A-operand layout, the x tile is converted to f32 on its way into LDS (it is used as VALUES: p x, p x^2), and
// every 16 x 16 logit tile folds into a per-lane online softmax centred on the plain time mean.  Lanes that share a channel merge with
// two xor-shuffles at the end; waves never talk.  Roofline: HBM/L2 -- x once (T*C*4 B per utterance) + h per 128-channel block (L2-hot).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int AX_ATT = 128;
constexpr float AX_LOG2E = 1.4426950408889634f;
constexpr int AX_HROW = 640;          // bytes per staged h row: 4 groups x 128 B + 128 pad (odd multiple of 128: rows alternate bank halves)
// bytes per staged x row: NW x 32 channels f32 + 16 pad (the 4 frame groups of a result-layout read land on disjoint banks)
template <int NW> struct AxGeom { static constexpr int CB = NW * 32, XROW = CB * 4 + 16; };

struct AspX3Args {
    const char* h;          // hl32 (B*T, att)
    const float* w;         // [C][att] f32
    const float* bias;      // [C]
    const char* x;          // hl32 (B*T, ldx)
    const float* center;    // (B, ldc): time mean per channel, or NULL
    float* pooled;          // (B, 2C)
    int ldx, ldc, T, C; float eps;
};

__device__ __forceinline__ void ax_merge(float& m, float& s0, float& s1, float& s2, int off) {
    const float m2 = __shfl_xor(m, off), a0 = __shfl_xor(s0, off), a1 = __shfl_xor(s1, off), a2 = __shfl_xor(s2, off);
    const float M = fmaxf(m, m2);
    const float f1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);       // maxima are kept in the log2 domain
    const float f2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - M);
    s0 = s0 * f1 + a0 * f2; s1 = s1 * f1 + a1 * f2; s2 = s2 * f1 + a2 * f2; m = M;
}

// NW waves per workgroup = NW x 32 channels of one utterance per h tile (the h tensor is read once per channel block: 12 x 39 MB at
// C = 1536 with 128-channel blocks -- 40 % of the kernel's traffic by PMC; see the launcher for why 4 waves stay the default)
template <int NW>
__global__ __launch_bounds__(NW * 64) void asp_x3_kernel(AspX3Args a) {
    constexpr int CB = AxGeom<NW>::CB, AX_XROW = AxGeom<NW>::XROW;
    constexpr int TPR = NW * 4;                   // staging threads per frame row: 16 rows x TPR = the workgroup
    __shared__ __attribute__((aligned(16))) char hs[2][16 * AX_HROW];
    __shared__ __attribute__((aligned(16))) char xs[2][16 * AX_XROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int cblk = blockIdx.x * CB;
    const int c0 = cblk + wv * 32;
    const size_t row0 = (size_t)b * a.T;

    // B operand: weight rows (channels) c0 + ni*16 + li, k = ks*32 + g*8 .. +7, split into hi / lo fragments
    bf16x8 wh[2][4], wl[2][4];
    float biasl[2], mu0[2];
    int ch[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ch[ni] = min(c0 + ni * 16 + li, a.C - 1);
        const float* wr = a.w + (size_t)ch[ni] * AX_ATT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 p0 = *reinterpret_cast<const float4*>(wr + ks * 32 + g * 8);
            const float4 p1 = *reinterpret_cast<const float4*>(wr + ks * 32 + g * 8 + 4);
            const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bf16_t hh = (bf16_t)v[e];
                wh[ni][ks][e] = hh;
                wl[ni][ks][e] = (bf16_t)(v[e] - (float)hh);
            }
        }
        biasl[ni] = a.bias[ch[ni]] * AX_LOG2E;
        mu0[ni] = a.center ? a.center[(size_t)b * a.ldc + ch[ni]] : 0.f;
    }
    float mx[2] = {-INFINITY, -INFINITY}, s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

    const int ntile = (a.T + 15) / 16;
    // frame tile mt: thread (row = tid / TPR, q = tid % TPR) moves its share of the h row's 32 16-byte chunks (2 per thread with 4
    // waves, 1 with 8: group = chunk >> 3) and the 8 channels cblk + 8 q of the x row (16 B of the hi plane + 16 B of the lo plane)
    constexpr int HC = 32 / TPR;                  // h chunks per thread
    const int srow = tid / TPR, q = tid % TPR;
    const int xcol = min(cblk + q * 8, a.C - 8);                        // clamped: columns past C feed lanes that never store
    const int xgo = (xcol >> 5) * 128 + (xcol & 31) * 2;
    const size_t ldxb = (size_t)a.ldx * 4;
    auto gload = [&](int mt, uint4& h0, uint4& h1, uint4& xh, uint4& xl) {
        const size_t t = row0 + min(mt * 16 + srow, a.T - 1);
        const char* hp = a.h + t * (AX_ATT * 4) + q * (HC * 16);
        h0 = *reinterpret_cast<const uint4*>(hp);
        if constexpr (HC == 2) h1 = *reinterpret_cast<const uint4*>(hp + 16);
        const char* xp = a.x + t * ldxb + xgo;
        xh = *reinterpret_cast<const uint4*>(xp);
        xl = *reinterpret_cast<const uint4*>(xp + 64);
    };
    auto swrite = [&](int buf, const uint4& h0, const uint4& h1, const uint4& xh, const uint4& xl) {
        const int c0h = q * HC;                                         // first chunk of the row's 32
        char* hr = hs[buf] + srow * AX_HROW + (c0h >> 3) * 128;
        const int c = c0h & 7;
        *reinterpret_cast<uint4*>(hr + ((c ^ (srow & 7)) << 4)) = h0;
        if constexpr (HC == 2) *reinterpret_cast<uint4*>(hr + (((c + 1) ^ (srow & 7)) << 4)) = h1;
        const unsigned hw[4] = {xh.x, xh.y, xh.z, xh.w}, lw[4] = {xl.x, xl.y, xl.z, xl.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, hw[e] & 0xffff0000u) + __builtin_bit_cast(float, lw[e] & 0xffff0000u);
        }
        float* xr = reinterpret_cast<float*>(xs[buf] + srow * AX_XROW) + q * 8;
        *reinterpret_cast<float4*>(xr) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(xr + 4) = make_float4(v[4], v[5], v[6], v[7]);
    };
    auto compute_tile = [&](int mt, int buf) {
        const int t0 = mt * 16 + g * 4;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* hr = hs[buf] + li * AX_HROW + ks * 128;
            const bf16x8 hh = *reinterpret_cast<const bf16x8*>(hr + ((g ^ (li & 7)) << 4));
            const bf16x8 hl = *reinterpret_cast<const bf16x8*>(hr + (((4 + g) ^ (li & 7)) << 4));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hl, wh[ni][ks], acc[ni], 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh, wl[ni][ks], acc[ni], 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh, wh[ni][ks], acc[ni], 0, 0, 0);
            }
        }
        // acc[ni][r] = logit(frame t0 + r, channel ch[ni]) - bias; softmax in the log2 domain, branch-free (asp_fused.hip)
        const int tlim = a.T - t0;                                   // frames r < tlim exist
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (r < tlim) ? fmaf(acc[ni][r], AX_LOG2E, biasl[ni]) : -INFINITY;
            const float m4 = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
            const float mnew = fmaxf(fmaxf(mx[ni], m4), -1e30f);       // finite even when every frame so far is masked
            const float f = __builtin_amdgcn_exp2f(mx[ni] - mnew);      // exp2(-inf) = 0 on the first tile
            mx[ni] = mnew;
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(e[r] - mnew);    // 0 for frames past T
                const float xv = *reinterpret_cast<const float*>(xs[buf] + (g * 4 + r) * AX_XROW + (wv * 32 + ni * 16 + li) * 4) - mu0[ni];
                const float t = pr * xv;
                p0 += pr; p1 += t; p2 = fmaf(t, xv, p2);
            }
            s0[ni] = fmaf(s0[ni], f, p0); s1[ni] = fmaf(s1[ni], f, p1); s2[ni] = fmaf(s2[ni], f, p2);
        }
    };
    uint4 h0, h1 = uint4{0u, 0u, 0u, 0u}, xh, xl;
    gload(0, h0, h1, xh, xl);
    swrite(0, h0, h1, xh, xl);
    __syncthreads();
    for (int mt = 0; mt < ntile; ++mt) {
        const bool more = mt + 1 < ntile;
        if (more) gload(mt + 1, h0, h1, xh, xl);
        compute_tile(mt, mt & 1);
        if (more) swrite((mt + 1) & 1, h0, h1, xh, xl);
        __syncthreads();
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        ax_merge(mx[ni], s0[ni], s1[ni], s2[ni], 16);
        ax_merge(mx[ni], s0[ni], s1[ni], s2[ni], 32);
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

// VP_EUNSUP when the shape is not covered.
int vp_asp_fused_x3(vp_ctx* ctx, const void* h, const float* w, const float* bias, const void* x, int ldx, const float* center, int ldc,
                    int B, int T, int C, int att, float eps, float* pooled, hipStream_t st) {
    if (att != AX_ATT || B > 65535 || T < 1 || C < 32 || (C | ldx) & 31 ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(w)) & 15))
        return VP_EUNSUP;
    AspX3Args a;
    a.h = (const char*)h; a.w = w; a.bias = bias; a.x = (const char*)x; a.center = center;
    a.pooled = pooled; a.ldx = ldx; a.ldc = ldc; a.T = T; a.C = C; a.eps = eps;
    // 256-channel blocks (8 waves) halve the h re-reads but measured SLOWER at 256 x 3 s, C = 1536: 187 us against 150 us with 128-channel
    // blocks (a 512-thread workgroup with 53 KB of LDS leaves fewer independent workgroups per CU to hide the staging loads); kept for A/B
    static const bool wide = getenv("VPMI_ASPX3_256") != nullptr;
    if (C >= 512 && wide) hipLaunchKernelGGL(asp_x3_kernel<8>, dim3((C + 255) / 256, B), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(asp_x3_kernel<4>, dim3((C + 127) / 128, B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "asp_x3");
    return VP_OK;
}
