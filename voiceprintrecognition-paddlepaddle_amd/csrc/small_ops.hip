// Per-utterance (M = batch) pieces of the path: time-moment finalisation, exact-f32 dense layers on
// the f32 matrix cores, SE gating + residual, attention softmax + weighted statistics.
// All HBM/L2-bound or tiny; kept in f32 so that they add no error on top of the f32 reference.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

// ---------------------------------------------------------------- moments from conv partial sums
// SEBlock mean (ecapa_tdnn.py:78) and ASP global-context mean/std (pooling.py:90-104, mask of ones)
struct MomArgs {
    const float* psum; const float* psumsq; const float* shift; float* stats;
    int B, T, C, nseg, want_std; float eps;
    const float* scale;              // NULL: the sums are of y - shift (inference: BN folded into the conv); else of z with y = scale z + shift
};

__global__ __launch_bounds__(256) void moments_kernel(MomArgs a) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    const int t0 = (int)(((long long)b * a.T) / VP_CONV_BM);
    const int t1 = (int)(((long long)(b + 1) * a.T - 1) / VP_CONV_BM);
    float s1 = 0.f, s2 = 0.f;
    for (int tm = t0; tm <= t1; ++tm) {
        const int sg = b - (int)(((long long)tm * VP_CONV_BM) / a.T);
        const size_t idx = ((size_t)tm * a.nseg + sg) * a.C + c;
        s1 += a.psum[idx];
        if (a.want_std) s2 += a.psumsq[idx];
    }
    const float invT = 1.f / (float)a.T;
    const float md = s1 * invT;
    const float sh = a.shift ? a.shift[c] : 0.f;
    const float sc = a.scale ? a.scale[c] : 1.f;
    const int ld = a.want_std ? 2 * a.C : a.C;
    a.stats[(size_t)b * ld + c] = sh + sc * md;
    if (a.want_std) {
        const float var = (s2 * invT - md * md) * sc * sc;
        a.stats[(size_t)b * ld + a.C + c] = sqrtf(fmaxf(var, a.eps));
    }
}

// ---------------------------------------------------------------- dense f32 on v_mfma_f32_16x16x4_f32
// One 16x16 output tile per workgroup; the 4 waves split K and reduce through LDS (fixed order).
struct DenseArgs {
    const float* a; const float* w; const float* bias; const float* rowscale; const float* colscale; float* out;
    int lda, ldo, M, N, K, w_is_kn, act, kper, vec_ok, wvec_ok;
};

// K is split over the workgroup's waves (4, or 16 when K >= 1024: the M = batch-size layers of the heads are chains of
// dependent load -> MFMA round trips on a handful of workgroups, so the chain is cut 16 ways); partial tiles are summed in
// wave order.
__global__ __launch_bounds__(1024) void dense_f32_kernel(DenseArgs p) {
    __shared__ float red[16][16][17];
    const int nw = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int kbeg = wv * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int am = m0 + li, bn = n0 + li;
    const bool aok = am < p.M, bok = bn < p.N;
    const float* arow = p.a + (size_t)(aok ? am : 0) * p.lda;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // Aligned operands: four 16-wide k-steps of loads in flight before their MFMAs.  The loads are unconditional (addresses
    // clamped into range, values zeroed by a select): one dependent load -> MFMA round trip per k-step made the M = batch-size
    // layers of the heads pure latency chains (48 round trips for K = 3072).  Same MFMA order as the plain loop: same bits.
    const bool fast = p.vec_ok && (p.K & 3) == 0 && (p.w_is_kn || p.wvec_ok);
    int kk0 = kbeg;
    if (fast) {
        const float* wrow = p.w + (p.w_is_kn ? (size_t)(bok ? bn : 0) : (size_t)(bok ? bn : 0) * p.K);
        // one loop instance per weight layout: a layout branch inside the loop made hipcc sink the last k-step's loads behind
        // the MFMAs (load -> vmcnt(0) -> MFMA, four times over)
        auto run = [&](auto kn) {
            for (; kk0 < kend; kk0 += 64) {
                float4 ta[4], tb[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kk0 + 16 * u + 4 * g;
                    in[u] = k < kend;
                    const int kc = in[u] ? k : 0;
                    ta[u] = *reinterpret_cast<const float4*>(arow + kc);
                    if constexpr (decltype(kn)::value) {
                        tb[u].x = wrow[(size_t)kc * p.N]; tb[u].y = wrow[(size_t)(kc + 1) * p.N];
                        tb[u].z = wrow[(size_t)(kc + 2) * p.N]; tb[u].w = wrow[(size_t)(kc + 3) * p.N];
                    } else {
                        tb[u] = *reinterpret_cast<const float4*>(wrow + kc);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);             // all four k-steps' loads issue before the first MFMA
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ia = in[u] && aok, ib = in[u] && bok;
                    const float av[4] = {ia ? ta[u].x : 0.f, ia ? ta[u].y : 0.f, ia ? ta[u].z : 0.f, ia ? ta[u].w : 0.f};
                    const float bv[4] = {ib ? tb[u].x : 0.f, ib ? tb[u].y : 0.f, ib ? tb[u].z : 0.f, ib ? tb[u].w : 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
                }
            }
        };
        if (p.w_is_kn) run(std::true_type{});
        else run(std::false_type{});
    }
    for (int kk = kk0; kk < kend; kk += 16) {
        const int k = kk + 4 * g;
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (aok) {
            if (p.vec_ok && k + 3 < kend) {
                float4 t = *reinterpret_cast<const float4*>(arow + k);
                av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < kend) av[e] = arow[k + e];
            }
        }
        if (bok) {
            if (p.w_is_kn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < kend) bv[e] = p.w[(size_t)(k + e) * p.N + bn];
            } else {
                const float* wrow = p.w + (size_t)bn * p.K;
                if (p.wvec_ok && k + 3 < kend) {
                    float4 t = *reinterpret_cast<const float4*>(wrow + k);
                    bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < kend) bv[e] = wrow[k + e];
                }
            }
        }
        // lane group g feeds k = kk + 4g + e to instruction e for BOTH operands
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
    }
    // acc[r] = C[row = 4g + r (m)][col = li (n)]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][4 * g + r][li] = acc[r];
    __syncthreads();
    const int mm = (tid >> 4) & 15, nn = tid & 15;
    const int m = m0 + mm, n = n0 + nn;
    if (tid < 256 && m < p.M && n < p.N) {
        float v = red[0][mm][nn];
        for (int w = 1; w < nw; ++w) v += red[w][mm][nn];
        if (p.rowscale) v *= p.rowscale[m];
        if (p.colscale) v *= p.colscale[n];
        if (p.bias) v += p.bias[n];
        if (p.act == VP_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == VP_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        else if (p.act == VP_ACT_TANH) v = tanhf(v);
        p.out[(size_t)m * p.ldo + n] = v;
    }
}

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* x, int rows, int D, int ld, float eps, float* inv) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float v = x[(size_t)row * ld + d]; s += v * v; }
    s = vp_wave_sum(s);
    if (lane == 0) inv[row] = 1.f / fmaxf(sqrtf(s), eps);
}

// ---------------------------------------------------------------- SE gate + residual (ecapa_tdnn.py:82,142)
template <typename T>
struct SeArgs {
    const T* x; const float* s; const T* res; T* out;
    int ldx, xoff, ldr, roff, ldo, ooff, T_, C, relu; long long total;
    bf16_t* shadow; int lds_, soff;          // f32 flavour only: the same values once more as bf16 (the next GEMM's operand), or NULL
};

// bf16, x = the PRE-BatchNorm activation z of the conv in front of the SE block: h = bf16(z * bsc + bsh) formed on the fly (the values the
// BatchNorm apply pass would have stored), out = h * s[b] + res -- the training step's block output without a stored h
struct SeZArgs { const bf16_t* z; const float* bsc; const float* bsh; const float* s; const bf16_t* res; bf16_t* out;
                 int ldz, ldr, roff, ldo, ooff, T_, C; long long total; };
__global__ __launch_bounds__(256) void se_scale_residual_z16_kernel(SeZArgs a) {
    const int cv = a.C / 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < a.total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / cv;
        const int c = (int)(idx - m * cv) * 8;
        const int b = (int)(m / a.T_);
        const uint4 zr = *reinterpret_cast<const uint4*>(a.z + m * a.ldz + c);
        const uint4 rr = *reinterpret_cast<const uint4*>(a.res + m * a.ldr + a.roff + c);
        const bf16_t* ze = reinterpret_cast<const bf16_t*>(&zr);
        const bf16_t* re = reinterpret_cast<const bf16_t*>(&rr);
        float sp[8], sc[8], sh[8];                            // (C % 8 == 0 and 16-byte aligned vectors: host-checked)
        *reinterpret_cast<float4*>(sp) = *reinterpret_cast<const float4*>(a.s + (size_t)b * a.C + c);
        *reinterpret_cast<float4*>(sp + 4) = *reinterpret_cast<const float4*>(a.s + (size_t)b * a.C + c + 4);
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(a.bsc + c);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(a.bsc + c + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(a.bsh + c);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(a.bsh + c + 4);
        uint4 o;
        bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = (float)(bf16_t)__fmaf_rn((float)ze[e], sc[e], sh[e]);
            oe[e] = (bf16_t)__fmaf_rn(h, sp[e], (float)re[e]);
        }
        *reinterpret_cast<uint4*>(a.out + m * a.ldo + a.ooff + c) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void se_scale_residual_kernel(SeArgs<T> a) {
    constexpr int V = 16 / (int)sizeof(T);
    const int cv = a.C / V;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < a.total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / cv;
        const int c = (int)(idx - m * cv) * V;
        const int b = (int)(m / a.T_);
        uint4 xr = *reinterpret_cast<const uint4*>(a.x + m * a.ldx + a.xoff + c);
        uint4 rr = *reinterpret_cast<const uint4*>(a.res + m * a.ldr + a.roff + c);
        const T* xe = reinterpret_cast<const T*>(&xr);
        const T* re = reinterpret_cast<const T*>(&rr);
        const float* sp = a.s + (size_t)b * a.C + c;
        uint4 o;
        T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = __fmaf_rn(vp_to_f32(xe[e]), sp[e], vp_to_f32(re[e]));
            if (a.relu) v = fmaxf(v, 0.f);
            oe[e] = vp_from_f32<T>(v);
        }
        *reinterpret_cast<uint4*>(a.out + m * a.ldo + a.ooff + c) = o;
        if constexpr (sizeof(T) == 4) {
            if (a.shadow) {
                bf16x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = (bf16_t)vp_to_f32(oe[e]);
                *reinterpret_cast<bf16x4*>(a.shadow + m * a.lds_ + a.soff + c) = q;
            }
        }
    }
}

// the same on tensors stored as split bf16 planes (vpmi.h: VP_HL32): a thread takes 8 channels = 16 B of the hi plane + 16 B of the lo
// plane of their 32-channel group, works on hi + lo in f32 and splits the result again
struct SeHlArgs { const char* x; const float* s; const char* res; char* out; int ldx, xoff, ldr, roff, ldo, ooff, T_, C, relu; long long total; };
__global__ __launch_bounds__(256) void se_scale_residual_hl_kernel(SeHlArgs a) {
    const int cv = a.C / 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < a.total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / cv;
        const int c = (int)(idx - m * cv) * 8;
        const int b = (int)(m / a.T_);
        const int go = (c >> 5) * 128 + (c & 31) * 2;                       // (offsets are multiples of 32 channels: host-checked)
        const char* xp = a.x + (m * a.ldx + a.xoff) * 4 + go;
        const char* rp = a.res + (m * a.ldr + a.roff) * 4 + go;
        const uint4 xh = *reinterpret_cast<const uint4*>(xp), xl = *reinterpret_cast<const uint4*>(xp + 64);
        const uint4 rh = *reinterpret_cast<const uint4*>(rp), rl = *reinterpret_cast<const uint4*>(rp + 64);
        const unsigned xhw[4] = {xh.x, xh.y, xh.z, xh.w}, xlw[4] = {xl.x, xl.y, xl.z, xl.w};
        const unsigned rhw[4] = {rh.x, rh.y, rh.z, rh.w}, rlw[4] = {rl.x, rl.y, rl.z, rl.w};
        float sp[8];
        *reinterpret_cast<float4*>(sp) = *reinterpret_cast<const float4*>(a.s + (size_t)b * a.C + c);
        *reinterpret_cast<float4*>(sp + 4) = *reinterpret_cast<const float4*>(a.s + (size_t)b * a.C + c + 4);
        unsigned oh[4], ol[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = __builtin_bit_cast(float, xhw[e] << 16) + __builtin_bit_cast(float, xlw[e] << 16);
            const float x1 = __builtin_bit_cast(float, xhw[e] & 0xffff0000u) + __builtin_bit_cast(float, xlw[e] & 0xffff0000u);
            const float r0 = __builtin_bit_cast(float, rhw[e] << 16) + __builtin_bit_cast(float, rlw[e] << 16);
            const float r1 = __builtin_bit_cast(float, rhw[e] & 0xffff0000u) + __builtin_bit_cast(float, rlw[e] & 0xffff0000u);
            float v0 = __fmaf_rn(x0, sp[2 * e], r0), v1 = __fmaf_rn(x1, sp[2 * e + 1], r1);
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            const bf16_t h0 = (bf16_t)v0, h1 = (bf16_t)v1;
            const bf16_t l0 = (bf16_t)(v0 - (float)h0), l1 = (bf16_t)(v1 - (float)h1);
            oh[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            ol[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        char* op = a.out + (m * a.ldo + a.ooff) * 4 + go;
        *reinterpret_cast<uint4*>(op) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        *reinterpret_cast<uint4*>(op + 64) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
    }
}

// (B, T, F) f32 features -> the tapped conv's operand rows as split bf16 planes: out[b*T + t][j*F + c] = x[b][reflect(t - pad + j*dil)][c],
// columns [KW*F, Kp) zero -- blocks[0] of the split-precision ECAPA path (ecapa_tdnn.py:249: TDNNBlock(F -> C, k5)) then runs as a 1x1
// layer on the LDS-DMA ring kernel (csrc/conv_gemm256.hip) instead of the 128-wide tapped kernel.  A thread = 8 consecutive columns.
struct Im2colArgs { const float* x; char* out; int T, F, KW, dil, pad, Kp; long long total; };
__global__ __launch_bounds__(256) void im2col_hl_kernel(Im2colArgs a) {
    const int cv = a.Kp / 8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < a.total; idx += (long long)gridDim.x * 256) {
        const long long m = idx / cv;
        const int k0 = (int)(idx - m * cv) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k0 < a.KW * a.F) {                                          // (F % 8 == 0: the 8 columns share a tap)
            const int j = k0 / a.F, c = k0 - j * a.F;
            const long long b = m / a.T;
            const int t = (int)(m - b * a.T);
            int ts = t - a.pad + j * a.dil;
            ts = ts < 0 ? -ts : ts;
            ts = ts >= a.T ? 2 * (a.T - 1) - ts : ts;
            const float* src = a.x + ((size_t)b * a.T + ts) * a.F + c;
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 4);
        }
        unsigned oh[4], ol[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t h0 = (bf16_t)v[2 * e], h1 = (bf16_t)v[2 * e + 1];
            const bf16_t l0 = (bf16_t)(v[2 * e] - (float)h0), l1 = (bf16_t)(v[2 * e + 1] - (float)h1);
            oh[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            ol[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        char* op = a.out + (size_t)m * a.Kp * 4 + (k0 >> 5) * 128 + (k0 & 31) * 2;
        *reinterpret_cast<uint4*>(op) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        *reinterpret_cast<uint4*>(op + 64) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
    }
}

// ---------------------------------------------------------------- ASP softmax over time + weighted stats
// pooling.py:114-123: attn = softmax_t(logits); mean = sum attn x; std = sqrt(clip(sum attn (x-mean)^2, eps)).
// One workgroup = 64 channels of one utterance; the 4 waves split the frames, online softmax per
// lane, merged through LDS.  x is centred on `center` (the plain time mean) to keep the variance
// well conditioned in f32.
template <typename T>
struct AspArgs {
    const float* logits; const T* x; const float* center; float* pooled;
    int ldx, xoff, ldc, B, T_, C; float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void asp_softmax_stats_kernel(AspArgs<T> a) {
    __shared__ float sm[4][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const float mu0 = a.center ? a.center[(size_t)b * a.ldc + cc] : 0.f;
    float mx = -INFINITY, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const size_t row0 = (size_t)b * a.T_;
    for (int t = wv; t < a.T_; t += 4) {
        const float e = a.logits[(row0 + t) * a.C + cc];
        const float xv = vp_to_f32(a.x[(row0 + t) * a.ldx + a.xoff + cc]) - mu0;
        if (e > mx) {
            const float f = expf(mx - e);       // exp(-inf) = 0 on the first frame
            s0 = s0 * f + 1.f; s1 = s1 * f + xv; s2 = s2 * f + xv * xv; mx = e;
        } else {
            const float p = expf(e - mx);
            s0 += p; s1 += p * xv; s2 += p * xv * xv;
        }
    }
    sm[wv][0][lane] = mx; sm[wv][1][lane] = s0; sm[wv][2][lane] = s1; sm[wv][3][lane] = s2;
    __syncthreads();
    if (wv == 0 && ok) {
        float M = fmaxf(fmaxf(sm[0][0][lane], sm[1][0][lane]), fmaxf(sm[2][0][lane], sm[3][0][lane]));
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = sm[w][0][lane];
            const float f = (mw == -INFINITY) ? 0.f : expf(mw - M);
            t0 += sm[w][1][lane] * f; t1 += sm[w][2][lane] * f; t2 += sm[w][3][lane] * f;
        }
        const float md = t1 / t0;
        const float var = t2 / t0 - md * md;
        a.pooled[(size_t)b * 2 * a.C + c] = mu0 + md;
        a.pooled[(size_t)b * 2 * a.C + a.C + c] = sqrtf(fmaxf(var, a.eps));
    }
}

}  // namespace

static int se_scale_residual_impl(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, const float* s,
                                  const void* res, int ldr, int roff, void* out, int ldo, int ooff, int B, int T, int C,
                                  int relu, void* shadow, int ld_shadow, int shadow_off, hipStream_t st) {
    if (!ctx || !x || !s || !res || !out || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "se: bad arguments");
    if (shadow && (dtype != VP_F32 || (ld_shadow | shadow_off) & 3 || reinterpret_cast<uintptr_t>(shadow) & 7))
        VP_FAIL(ctx, VP_EINVAL, "se: the bf16 shadow goes with f32 tensors, ld / offset multiples of 4");
    const int V = dtype == VP_HL32 ? 32 : dtype == VP_BF16 ? 8 : 4;
    if (C % V || ldx % V || xoff % V || ldr % V || roff % V || ldo % V || ooff % V)
        VP_FAIL(ctx, VP_EINVAL, "se: C/ld/off must be multiples of %d", V);
    if (dtype == VP_HL32) {
        const long long tot = (long long)B * T * (C / 8);
        long long nb = (tot + 255) / 256;
        if (nb > 256 * 16) nb = 256 * 16;
        SeHlArgs a{(const char*)x, s, (const char*)res, (char*)out, ldx, xoff, ldr, roff, ldo, ooff, T, C, relu, tot};
        hipLaunchKernelGGL(se_scale_residual_hl_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
        VP_LAUNCH_CHECK(ctx, "se_scale_residual_hl");
        return VP_OK;
    }
    const long long total = (long long)B * T * (C / V);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == VP_BF16) {
        SeArgs<bf16_t> a{(const bf16_t*)x, s, (const bf16_t*)res, (bf16_t*)out, ldx, xoff, ldr, roff, ldo, ooff, T, C, relu, total, nullptr, 0, 0};
        hipLaunchKernelGGL(se_scale_residual_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else if (dtype == VP_F32) {
        SeArgs<float> a{(const float*)x, s, (const float*)res, (float*)out, ldx, xoff, ldr, roff, ldo, ooff, T, C, relu, total,
                        (bf16_t*)shadow, ld_shadow, shadow_off};
        hipLaunchKernelGGL(se_scale_residual_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        VP_FAIL(ctx, VP_EINVAL, "se: bad dtype");
    }
    VP_LAUNCH_CHECK(ctx, "se_scale_residual");
    return VP_OK;
}

int vp_se_scale_residual_ex(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, const float* s,
                            const void* res, int ldr, int roff, void* out, int ldo, int ooff, int B, int T, int C,
                            int relu, hipStream_t st) {
    return se_scale_residual_impl(ctx, dtype, x, ldx, xoff, s, res, ldr, roff, out, ldo, ooff, B, T, C, relu, nullptr, 0, 0, st);
}

int vp_im2col_hl32(vp_ctx* ctx, const float* x, int B, int T, int F, int KW, int dil, int pad, void* out, int Kp, hipStream_t st) {
    if (!ctx || !x || !out || B <= 0 || T <= 0 || F % 8 || Kp % 32 || Kp < KW * F || pad >= T || dil * (KW - 1) - pad >= T ||
        (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15)
        VP_FAIL(ctx, VP_EINVAL, "im2col_hl32: bad arguments");
    const long long total = (long long)B * T * (Kp / 8);
    long long nb = (total + 255) / 256;
    if (nb > 256 * 32) nb = 256 * 32;
    Im2colArgs a{x, (char*)out, T, F, KW, dil, pad, Kp, total};
    hipLaunchKernelGGL(im2col_hl_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "im2col_hl32");
    return VP_OK;
}

// mean / std over time straight from the activations (small T): stats[b] = [mean(C) | sqrt(max(E[(x-m)^2], eps))]
namespace {
template <typename T>
struct TmArgs { const T* x; float* stats; int ldx, Tn, C; float eps; int unbiased; };

template <typename T>
__global__ __launch_bounds__(256) void time_moments_kernel(TmArgs<T> a) {
    __shared__ float sm[2][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const T* xb = a.x + (size_t)b * a.Tn * a.ldx;
    const float c0 = vp_to_f32(xb[cc]);
    float s1 = 0.f, s2 = 0.f;
    for (int t = wv; t < a.Tn; t += 4) {
        const float v = vp_to_f32(xb[(size_t)t * a.ldx + cc]) - c0;
        s1 += v; s2 += v * v;
    }
    sm[0][wv][lane] = s1; sm[1][wv][lane] = s2;
    __syncthreads();
    if (wv == 0 && ok) {
        const float t1 = sm[0][0][lane] + sm[0][1][lane] + sm[0][2][lane] + sm[0][3][lane];
        const float t2 = sm[1][0][lane] + sm[1][1][lane] + sm[1][2][lane] + sm[1][3][lane];
        const float md = t1 / (float)a.Tn;
        a.stats[(size_t)b * 2 * a.C + c] = c0 + md;
        // ASP context: sqrt(clip(var_biased, eps)) (pooling.py:97-104); TSTP: sqrt(var_unbiased + eps) (pooling.py:141)
        const float ss = fmaxf(t2 - (float)a.Tn * md * md, 0.f);
        a.stats[(size_t)b * 2 * a.C + a.C + c] = a.unbiased ? sqrtf(ss / (float)(a.Tn > 1 ? a.Tn - 1 : 1) + a.eps)
                                                            : sqrtf(fmaxf(ss / (float)a.Tn, a.eps));
    }
}

// f32 activations (the training path), four channels per lane: 32 lanes x 8 frame groups, four frames' loads in flight
__global__ __launch_bounds__(256) void time_moments4_kernel(TmArgs<float> a) {
    __shared__ float sm[2][256][4];
    const int lc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int b = blockIdx.y, c4 = blockIdx.x * 32 + lc;
    const int C4 = a.C >> 2;
    const bool ok = c4 < C4;
    const int c = ok ? c4 * 4 : 0;
    const float* xb = a.x + (size_t)b * a.Tn * a.ldx + c;
    float c0[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    vp_load4(xb, c0);
    int t = rg;
    for (; t + 24 < a.Tn; t += 32) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vp_load4(xb + (size_t)(t + 8 * u) * a.ldx, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - c0[e]; s1[e] += d; s2[e] += d * d; }
    }
    for (; t < a.Tn; t += 8) {
        float v[4];
        vp_load4(xb + (size_t)t * a.ldx, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - c0[e]; s1[e] += d; s2[e] += d * d; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[0][threadIdx.x][e] = s1[e]; sm[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if (rg != 0 || !ok) return;
    float m[4], sd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < 8; ++r) { t1 += sm[0][r * 32 + lc][e]; t2 += sm[1][r * 32 + lc][e]; }
        const float md = t1 / (float)a.Tn;
        m[e] = c0[e] + md;
        const float ss = fmaxf(t2 - (float)a.Tn * md * md, 0.f);
        sd[e] = a.unbiased ? sqrtf(ss / (float)(a.Tn > 1 ? a.Tn - 1 : 1) + a.eps) : sqrtf(fmaxf(ss / (float)a.Tn, a.eps));
    }
    vp_store4(a.stats + (size_t)b * 2 * a.C + c, m);
    vp_store4(a.stats + (size_t)b * 2 * a.C + a.C + c, sd);
}
}  // namespace

namespace {
// Many positions per utterance (2-D feature maps): the frames are cut into chunks (grid = channel blocks x B x chunks), partial
// sums of (x - x[first frame]) and its square per chunk, then a finalize over the chunks in order.  part: [B][chunks][2][C].
struct TmChunkArgs { const float* x; float* part; int ldx, Tn, C, chunks, rows_per_chunk; };
__global__ __launch_bounds__(256) void time_moments_chunk_kernel(TmChunkArgs a) {
    __shared__ float sm[2][256][4];
    const int lc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int b = blockIdx.y, ch = blockIdx.z, c4 = blockIdx.x * 32 + lc;
    const int C4 = a.C >> 2;
    const bool ok = c4 < C4;
    const int c = ok ? c4 * 4 : 0;
    const float* xb = a.x + (size_t)b * a.Tn * a.ldx + c;
    const int t0 = ch * a.rows_per_chunk, t1 = min(a.Tn, t0 + a.rows_per_chunk);
    float c0[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    vp_load4(xb, c0);
    int t = t0 + rg;
    for (; t + 24 < t1; t += 32) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vp_load4(xb + (size_t)(t + 8 * u) * a.ldx, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - c0[e]; s1[e] += d; s2[e] += d * d; }
    }
    for (; t < t1; t += 8) {
        float v[4];
        vp_load4(xb + (size_t)t * a.ldx, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - c0[e]; s1[e] += d; s2[e] += d * d; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[0][threadIdx.x][e] = s1[e]; sm[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if (rg != 0 || !ok) return;
    float t1s[4], t2s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        t1s[e] = 0.f; t2s[e] = 0.f;
        for (int r = 0; r < 8; ++r) { t1s[e] += sm[0][r * 32 + lc][e]; t2s[e] += sm[1][r * 32 + lc][e]; }
    }
    float* p = a.part + ((size_t)b * a.chunks + ch) * 2 * a.C;
    vp_store4(p + c, t1s);
    vp_store4(p + a.C + c, t2s);
}
struct TmFinArgs { const float* x; const float* part; float* stats; int ldx, Tn, C, chunks, unbiased; float eps; };
__global__ __launch_bounds__(256) void time_moments_fin_kernel(TmFinArgs a) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.C) return;
    float t1 = 0.f, t2 = 0.f;
    for (int ch = 0; ch < a.chunks; ++ch) {
        const float* p = a.part + ((size_t)b * a.chunks + ch) * 2 * a.C;
        t1 += p[c]; t2 += p[a.C + c];
    }
    const float c0 = a.x[(size_t)b * a.Tn * a.ldx + c];
    const float md = t1 / (float)a.Tn;
    a.stats[(size_t)b * 2 * a.C + c] = c0 + md;
    const float ss = fmaxf(t2 - (float)a.Tn * md * md, 0.f);
    a.stats[(size_t)b * 2 * a.C + a.C + c] = a.unbiased ? sqrtf(ss / (float)(a.Tn > 1 ? a.Tn - 1 : 1) + a.eps) : sqrtf(fmaxf(ss / (float)a.Tn, a.eps));
}
static void tm_geometry(int B, int T, int C, int& chunks, int& rpc) {
    const int cblocks = (C / 4 + 31) / 32;
    long long ch = 2048 / ((long long)B * cblocks > 0 ? (long long)B * cblocks : 1);
    if (ch < 1) ch = 1;
    long long r = (T + ch - 1) / ch;
    if (r < 64) r = 64;
    rpc = (int)r;
    chunks = (T + rpc - 1) / rpc;
}
}  // namespace

extern "C" size_t vp_time_stats_workspace_bytes(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    int chunks, rpc;
    tm_geometry(B, T, C, chunks, rpc);
    return (size_t)B * chunks * 2 * C * sizeof(float) + 256;
}

// f32 (B, T, C) with MANY frames per utterance (ResNetSE / ERes2Net feature maps as (B, T*F', C)): [mean | std] with the frames spread over
// ~2048 workgroups.  C % 4 == 0, ldx % 4 == 0, 16-byte aligned; else VP_EUNSUP (callers use vp_time_stats_f32).
extern "C" int vp_time_stats_ws_f32(vp_ctx* ctx, const float* x, int ldx, int B, int T, int C, float eps, int unbiased, float* stats, void* ws,
                                    size_t ws_bytes, vp_stream stream) {
    if (!ctx || !x || !stats || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "time_stats_ws: bad arguments");
    if (((C | ldx) & 3) || ((uintptr_t)x & 15)) return VP_EUNSUP;
    if (!ws || ws_bytes < vp_time_stats_workspace_bytes(B, T, C)) VP_FAIL(ctx, VP_EWORKSPACE, "time_stats_ws: workspace too small");
    int chunks, rpc;
    tm_geometry(B, T, C, chunks, rpc);
    hipStream_t st = (hipStream_t)stream;
    TmChunkArgs a{x, (float*)ws, ldx, T, C, chunks, rpc};
    hipLaunchKernelGGL(time_moments_chunk_kernel, dim3((C / 4 + 31) / 32, B, chunks), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "time_moments_chunk");
    TmFinArgs f{x, (const float*)ws, stats, ldx, T, C, chunks, unbiased, eps};
    hipLaunchKernelGGL(time_moments_fin_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, f);
    VP_LAUNCH_CHECK(ctx, "time_moments_fin");
    return VP_OK;
}

int vp_time_moments(vp_ctx* ctx, int dtype, const void* x, int ldx, int B, int T, int C, float eps, int unbiased,
                    float* stats, hipStream_t st) {
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "time_moments: batch too large");
    dim3 grid((C + 63) / 64, B);
    if (dtype != VP_BF16 && ((C | ldx) & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)stats & 15) == 0) {
        TmArgs<float> a{(const float*)x, stats, ldx, T, C, eps, unbiased};
        hipLaunchKernelGGL(time_moments4_kernel, dim3((C / 4 + 31) / 32, B), dim3(256), 0, st, a);
    } else if (dtype == VP_BF16) {
        TmArgs<bf16_t> a{(const bf16_t*)x, stats, ldx, T, C, eps, unbiased};
        hipLaunchKernelGGL(time_moments_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    } else {
        TmArgs<float> a{(const float*)x, stats, ldx, T, C, eps, unbiased};
        hipLaunchKernelGGL(time_moments_kernel<float>, grid, dim3(256), 0, st, a);
    }
    VP_LAUNCH_CHECK(ctx, "time_moments");
    return VP_OK;
}

namespace {
// out[r, ooff + c] = x[r, xoff + c]  (column block copy between row-major tensors; 4 elements per thread)
template <typename T>
struct CopyArgs { const T* x; T* y; int ldx, xoff, ldy, yoff, C4; long long total; };
template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(CopyArgs<T> a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long r = i / a.C4;
        const int c = (int)(i - r * a.C4) * 4;
        float v[4];
        vp_load4(a.x + r * a.ldx + a.xoff + c, v);
        vp_store4(a.y + r * a.ldy + a.yoff + c, v);
    }
}
// AFF output (eres2net.py:48-51) from t = tanh(local_att(cat(x, y))):  x * (1 + t) + y * (1 - t)
template <typename T>
struct AffArgs { const T* t; const T* x; const T* y; T* o; int ldt, ldx, xoff, ldy, yoff, ldo, ooff, C4; long long total; };
template <typename T>
__global__ __launch_bounds__(256) void aff_combine_kernel(AffArgs<T> a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long r = i / a.C4;
        const int c = (int)(i - r * a.C4) * 4;
        float t[4], x[4], y[4], o[4];
        vp_load4(a.t + r * a.ldt + c, t);
        vp_load4(a.x + r * a.ldx + a.xoff + c, x);
        vp_load4(a.y + r * a.ldy + a.yoff + c, y);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = x[e] * (1.f + t[e]) + y[e] * (1.f - t[e]);
        vp_store4(a.o + r * a.ldo + a.ooff + c, o);
    }
}
unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b));
}
}  // namespace

int vp_copy_cols(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, void* y, int ldy, int yoff, long long rows, int C,
                 hipStream_t st) {
    if ((C | ldx | xoff | ldy | yoff) & 3) VP_FAIL(ctx, VP_EINVAL, "copy_cols: widths / offsets must be multiples of 4");
    const long long total = rows * (C / 4);
    if (dtype == VP_BF16) {
        CopyArgs<bf16_t> a{(const bf16_t*)x, (bf16_t*)y, ldx, xoff, ldy, yoff, C / 4, total};
        hipLaunchKernelGGL(copy_cols_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, a);
    } else {
        CopyArgs<float> a{(const float*)x, (float*)y, ldx, xoff, ldy, yoff, C / 4, total};
        hipLaunchKernelGGL(copy_cols_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, a);
    }
    VP_LAUNCH_CHECK(ctx, "copy_cols");
    return VP_OK;
}

int vp_aff_combine(vp_ctx* ctx, int dtype, const void* t, int ldt, const void* x, int ldx, int xoff, const void* y, int ldy,
                   int yoff, void* o, int ldo, int ooff, long long rows, int C, hipStream_t st) {
    if ((C | ldt | ldx | xoff | ldy | yoff | ldo | ooff) & 3) VP_FAIL(ctx, VP_EINVAL, "aff_combine: widths / offsets must be multiples of 4");
    const long long total = rows * (C / 4);
    if (dtype == VP_BF16) {
        AffArgs<bf16_t> a{(const bf16_t*)t, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)o, ldt, ldx, xoff, ldy, yoff, ldo, ooff, C / 4, total};
        hipLaunchKernelGGL(aff_combine_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, a);
    } else {
        AffArgs<float> a{(const float*)t, (const float*)x, (const float*)y, (float*)o, ldt, ldx, xoff, ldy, yoff, ldo, ooff, C / 4, total};
        hipLaunchKernelGGL(aff_combine_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, a);
    }
    VP_LAUNCH_CHECK(ctx, "aff_combine");
    return VP_OK;
}

// ---------------------------------------------------------------- SE gate in one launch
// s[b, :] = sigmoid(W2 relu(W1 mean_t(y[b]) + b1) + b2), the mean taken from the producing conv's fused column sums
// (SEBlock, ecapa_tdnn.py:69-82; SELayer, resnet_se.py:48-63).  One workgroup per utterance: the three dependent steps
// (finalise the mean, C -> H, H -> C) were three launches of ~5 + 10 + 10 us on a (B, 512) problem.
namespace {
struct SeGateArgs {
    const float* psum; const float* shift; const float* w1; const float* b1; const float* w2; const float* b2; float* out;
    int B, T, C, H, nseg;
};

// out[n] = act(bias[n] + sum_k in[k] W[k][n]) for n < N; `in`, `out`, `part` (256 floats) live in LDS.  W is [K][N]
// (input-major: consecutive threads read consecutive float4 column groups, the remaining threads split K); the k loop is
// unrolled so every thread keeps 16 independent 16-byte loads in flight -- a dependent load-reduce chain per output
// made this kernel 237 us on a 512-channel block.
constexpr int SE_UPW = 1;             // utterances per workgroup: every workgroup streams both weight matrices once

// out[u][n] = act(bias[n] + sum_k in[u][k] W[k][n]) for SE_UPW utterances; `in` (stride ldi), `out` (stride ldo), `part` in LDS
__device__ __forceinline__ void se_matvec(const float* in, int ldi, int K, const float* W, int N, const float* bias, int sigmoid,
                                          float* out, int ldo, float* part) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nv = N >> 2;                            // float4 column groups (N % 4 == 0, N <= 1024: host-checked)
    const int ns = nt / nv;                           // K slices worked on in parallel (1024 threads: 32 / 8 slices at 128 / 512 outputs)
    const int v = tid % nv, sl = tid / nv;
    if (sl < ns) {
        const int k0 = (int)((long long)K * sl / ns), k1 = (int)((long long)K * (sl + 1) / ns);
        float4 acc[SE_UPW];
#pragma unroll
        for (int u = 0; u < SE_UPW; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wp = W + 4 * v;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(wp + (size_t)k * N);
#pragma unroll
            for (int u = 0; u < SE_UPW; ++u) {
                const float x = in[u * ldi + k];
                acc[u].x += x * w.x; acc[u].y += x * w.y; acc[u].z += x * w.z; acc[u].w += x * w.w;
            }
        }
#pragma unroll
        for (int u = 0; u < SE_UPW; ++u) *reinterpret_cast<float4*>(part + (u * ns + sl) * N + 4 * v) = acc[u];
    }
    __syncthreads();
    for (int i = tid; i < SE_UPW * N; i += nt) {
        const int u = i / N, n = i - u * N;
        float acc = bias ? bias[n] : 0.f;
        for (int q = 0; q < ns; ++q) acc += part[(u * ns + q) * N + n];
        out[u * ldo + n] = sigmoid ? 1.f / (1.f + __expf(-acc)) : fmaxf(acc, 0.f);
    }
}

// 1024 threads: the two matvecs are chains of dependent 16-byte weight loads (64 per thread at 256 threads); four times the
// threads cut the chains to 16.
__global__ __launch_bounds__(1024) void se_gate_kernel(SeGateArgs a) {
    extern __shared__ float sm[];        // mean[UPW][C] | h[UPW][H] | s[UPW][C] | part[UPW][4 * blockDim.x]
    float* mean = sm;
    float* h = mean + SE_UPW * a.C;
    float* sg = h + SE_UPW * a.H;
    float* part = sg + SE_UPW * a.C;
    const int b0 = blockIdx.x * SE_UPW;
    // time mean from the conv's partial sums: (B*T positions) / 128 tiles per utterance -- 3 for a 3 s ECAPA utterance, 186 for a
    // ResNetSE stage-1 map.  The tiles of a channel are split over blockDim / C threads (a lone thread walking 186 dependent
    // loads made this kernel 32 us per call there), partials through LDS, fixed order.
    {
        const int groups = max(1, (int)blockDim.x / (SE_UPW * a.C));
        const int i = threadIdx.x % (SE_UPW * a.C), gq = threadIdx.x / (SE_UPW * a.C);
        const int u = i / a.C, c = i - u * a.C;
        const int b = min(b0 + u, a.B - 1);
        const int t0 = (int)(((long long)b * a.T) / VP_CONV_BM);
        const int t1 = (int)(((long long)(b + 1) * a.T - 1) / VP_CONV_BM);
        for (int i2 = i; i2 < SE_UPW * a.C; i2 += blockDim.x) {      // C > blockDim: several channels per thread (groups == 1)
            const int u2 = i2 / a.C, c2 = i2 - u2 * a.C;
            float s1 = 0.f;
            if (gq < groups) {
#pragma unroll 4
                for (int tm = t0 + gq; tm <= t1; tm += groups) {
                    const int seg = b - (int)(((long long)tm * VP_CONV_BM) / a.T);
                    s1 += a.psum[((size_t)tm * a.nseg + seg) * a.C + c2];
                }
            }
            if (groups == 1) mean[i2] = (a.shift ? a.shift[c2] : 0.f) + s1 / (float)a.T;
            else if (gq < groups) part[gq * (SE_UPW * a.C) + i2] = s1;
            (void)u2;
        }
        if (groups > 1) {
            __syncthreads();
            if (gq == 0) {
                float s1 = 0.f;
                for (int q = 0; q < groups; ++q) s1 += part[q * (SE_UPW * a.C) + i];
                mean[i] = (a.shift ? a.shift[c] : 0.f) + s1 / (float)a.T;
            }
        }
        (void)u;
    }
    __syncthreads();
    se_matvec(mean, a.C, a.C, a.w1, a.H, a.b1, 0, h, a.H, part);
    __syncthreads();
    se_matvec(h, a.H, a.H, a.w2, a.C, a.b2, 1, sg, a.C, part);
    __syncthreads();
    for (int i = threadIdx.x; i < SE_UPW * a.C; i += blockDim.x) {
        const int u = i / a.C;
        if (b0 + u < a.B) a.out[(size_t)(b0 + u) * a.C + (i - u * a.C)] = sg[i];
    }
}
}  // namespace

// w1 [C][H], w2 [H][C]: input-major (Paddle Linear layout; Conv1D weights are transposed at pack time)
int vp_se_gate(vp_ctx* ctx, const float* psum, const float* shift, int B, int T, int C, int H, const float* w1, const float* b1,
               const float* w2, const float* b2, float* out, hipStream_t st) {
    if (!psum || !w1 || !w2 || !out || B <= 0 || T <= 0 || C <= 0 || H <= 0) VP_FAIL(ctx, VP_EINVAL, "se_gate: bad arguments");
    if ((C | H) & 3 || C > 1024 || H > 1024) VP_FAIL(ctx, VP_EUNSUP, "se_gate: C %d / H %d (multiples of 4, <= 1024)", C, H);
    constexpr int SE_THREADS = 1024;
    const size_t smem = (size_t)SE_UPW * (2 * C + H + 4 * SE_THREADS) * sizeof(float);
    if (smem > 64 * 1024) VP_FAIL(ctx, VP_EUNSUP, "se_gate: %d channels do not fit the LDS budget", C);
    SeGateArgs a{psum, shift, w1, b1, w2, b2, out, B, T, C, H, vp_conv1d_nseg(T)};
    hipLaunchKernelGGL(se_gate_kernel, dim3((B + SE_UPW - 1) / SE_UPW), dim3(SE_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "se_gate");
    return VP_OK;
}

int vp_dense_f32_ex(vp_ctx* ctx, const float* a, int lda, const float* w, int w_is_kn, const float* bias,
                    const float* rowscale, const float* colscale, int M, int N, int K, int act, float* out,
                    int ldo, hipStream_t st) {
    if (!a || !w || !out || M <= 0 || N <= 0 || K <= 0) VP_FAIL(ctx, VP_EINVAL, "dense: bad arguments");
    DenseArgs p;
    p.a = a; p.w = w; p.bias = bias; p.rowscale = rowscale; p.colscale = colscale; p.out = out;
    p.lda = lda; p.ldo = ldo; p.M = M; p.N = N; p.K = K; p.w_is_kn = w_is_kn; p.act = act;
    const int nw = K >= 1024 ? 16 : 4;
    p.kper = ((K + nw - 1) / nw + 15) / 16 * 16;
    p.vec_ok = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
    p.wvec_ok = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    dim3 grid((N + 15) / 16, (M + 15) / 16);
    if (grid.y > 65535) VP_FAIL(ctx, VP_EINVAL, "dense: M too large");
    hipLaunchKernelGGL(dense_f32_kernel, grid, dim3(64 * nw), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "dense_f32");
    return VP_OK;
}

int vp_row_inv_norm(vp_ctx* ctx, const float* x, int rows, int D, int ld, float eps, float* inv, hipStream_t st) {
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, D, ld, eps, inv);
    VP_LAUNCH_CHECK(ctx, "row_inv_norm");
    return VP_OK;
}

extern "C" {

// training: the conv's fused sums are of z = ReLU(conv + bias), the layer output is y = scale z + shift (batch-statistics BatchNorm):
// stats[b] = [mean_t y | sqrt(max(var_t y, eps))] without a pass over y (SEBlock's squeeze, ecapa_tdnn.py:66-71; ASP's context, pooling.py:97-104)
int vp_moments_finalize_affine(vp_ctx* ctx, const float* psum, const float* psumsq, const float* scale, const float* shift,
                               int B, int T, int C, float eps, int want_std, float* stats, vp_stream stream) {
    if (!ctx || !psum || !stats || !scale || !shift || (want_std && !psumsq) || B <= 0 || T <= 0 || C <= 0)
        VP_FAIL(ctx, VP_EINVAL, "moments: bad arguments");
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "moments: batch too large");
    MomArgs a;
    a.psum = psum; a.psumsq = psumsq; a.shift = shift; a.scale = scale; a.stats = stats; a.B = B; a.T = T; a.C = C;
    a.nseg = vp_conv1d_nseg(T); a.want_std = want_std; a.eps = eps;
    hipLaunchKernelGGL(moments_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "moments");
    return VP_OK;
}

int vp_moments_finalize(vp_ctx* ctx, const float* psum, const float* psumsq, const float* shift,
                        int B, int T, int C, float eps, int want_std, float* stats, vp_stream stream) {
    if (!ctx || !psum || !stats || (want_std && !psumsq) || B <= 0 || T <= 0 || C <= 0)
        VP_FAIL(ctx, VP_EINVAL, "moments: bad arguments");
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "moments: batch too large");
    MomArgs a;
    a.psum = psum; a.psumsq = psumsq; a.shift = shift; a.scale = nullptr; a.stats = stats; a.B = B; a.T = T; a.C = C;
    a.nseg = vp_conv1d_nseg(T); a.want_std = want_std; a.eps = eps;
    hipLaunchKernelGGL(moments_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "moments");
    return VP_OK;
}

int vp_dense_f32(vp_ctx* ctx, const float* a, int lda, const float* w, int w_is_kn, const float* bias,
                 int M, int N, int K, int act, float* out, int ldo, vp_stream stream) {
    if (!ctx) return VP_EINVAL;
    return vp_dense_f32_ex(ctx, a, lda, w, w_is_kn, bias, nullptr, nullptr, M, N, K, act, out, ldo, (hipStream_t)stream);
}

// f32 tensors, plus the result once more as bf16 at shadow[m * ld_shadow + shadow_off + c] -- the operand of the GEMMs that consume the
// block output under mixed precision (next block's tdnn1, the MFA concatenation), written by the pass that produces it
int vp_se_scale_residual_shadow(vp_ctx* ctx, const float* x, int ldx, int xoff, const float* s, const float* res, int ldr, int roff,
                                float* out, int ldo, int ooff, void* shadow, int ld_shadow, int shadow_off, int B, int T, int C,
                                vp_stream stream) {
    return se_scale_residual_impl(ctx, VP_F32, x, ldx, xoff, s, res, ldr, roff, out, ldo, ooff, B, T, C, 0, shadow, ld_shadow, shadow_off,
                                  (hipStream_t)stream);
}

int vp_se_scale_residual_z16(vp_ctx* ctx, const void* z, int ldz, const float* bn_scale, const float* bn_shift, const float* s, const void* res,
                             int ldr, int roff, void* out, int ldo, int ooff, int B, int T, int C, vp_stream stream) {
    if (!ctx || !z || !bn_scale || !bn_shift || !s || !res || !out || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "se_z16: bad arguments");
    if (C % 8 || ldz % 8 || ldr % 8 || roff % 8 || ldo % 8 || ooff % 8 ||
        ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(s) |
          reinterpret_cast<uintptr_t>(bn_scale) | reinterpret_cast<uintptr_t>(bn_shift)) & 15))
        VP_FAIL(ctx, VP_EINVAL, "se_z16: C / ld / offsets must be multiples of 8, tensors 16-byte aligned");
    const long long total = (long long)B * T * (C / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    SeZArgs a{(const bf16_t*)z, bn_scale, bn_shift, s, (const bf16_t*)res, (bf16_t*)out, ldz, ldr, roff, ldo, ooff, T, C, total};
    hipLaunchKernelGGL(se_scale_residual_z16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "se_scale_residual_z16");
    return VP_OK;
}

int vp_se_scale_residual(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, const float* s,
                         const void* res, int ldr, int roff, void* out, int ldo, int ooff,
                         int B, int T, int C, vp_stream stream) {
    return vp_se_scale_residual_ex(ctx, dtype, x, ldx, xoff, s, res, ldr, roff, out, ldo, ooff, B, T, C, 0, (hipStream_t)stream);
}

}  // extern "C"

namespace {
// f32 x (the training engine), T <= 8 NT: a thread keeps its share of an utterance's logits and x in registers -- all loads issued up front
// (the streaming kernel above has two loads in flight per thread and a divergent branch per frame: 4.2 TB/s) -- then max, then the sums.
// TL: the logits as stored -- float, or bf16 under enable_amp (a.logits reinterpreted: what Paddle's O1 hands the softmax is the bf16
// output of the logits conv, cast up)
template <int NT, typename TL = float, typename TX = float>
__global__ __launch_bounds__(512) void asp_softmax_stats_reg_kernel(AspArgs<float> a) {
    const TL* __restrict__ lg = reinterpret_cast<const TL*>(a.logits);
    const TX* __restrict__ xg = reinterpret_cast<const TX*>(a.x);
    __shared__ float sm[4][8][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const float mu0 = a.center ? a.center[(size_t)b * a.ldc + cc] : 0.f;
    const size_t row0 = (size_t)b * a.T_;
    float ev[NT], xv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = min(rg + 8 * i, a.T_ - 1);
        ev[i] = vp_to_f32(lg[(row0 + t) * a.C + cc]);
        xv[i] = vp_to_f32(xg[(row0 + t) * a.ldx + a.xoff + cc]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NT; ++i) if (rg + 8 * i < a.T_) mx = fmaxf(mx, ev[i]);
    sm[0][rg][lane] = mx;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) mx = fmaxf(mx, sm[0][q][lane]);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (rg + 8 * i < a.T_) {
            const float p = expf(ev[i] - mx), d = xv[i] - mu0;
            s0 += p; s1 += p * d; s2 += p * d * d;
        }
    }
    sm[1][rg][lane] = s0; sm[2][rg][lane] = s1; sm[3][rg][lane] = s2;
    __syncthreads();
    if (rg == 0 && ok) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { t0 += sm[1][q][lane]; t1 += sm[2][q][lane]; t2 += sm[3][q][lane]; }
        const float md = t1 / t0;
        const float var = t2 / t0 - md * md;
        a.pooled[(size_t)b * 2 * a.C + c] = mu0 + md;
        a.pooled[(size_t)b * 2 * a.C + a.C + c] = sqrtf(fmaxf(var, a.eps));
    }
}
}  // namespace

int vp_asp_softmax_stats_ex(vp_ctx* ctx, int dtype, const float* logits, const void* x, int ldx, int xoff,
                            const float* center, int ldc, int B, int T, int C, float eps, float* pooled, hipStream_t st) {
    if (!ctx || !logits || !x || !pooled || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "asp: bad arguments");
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "asp: batch too large");
    dim3 grid((C + 63) / 64, B);
    if (dtype == VP_BF16) {
        AspArgs<bf16_t> a{logits, (const bf16_t*)x, center, pooled, ldx, xoff, ldc, B, T, C, eps};
        hipLaunchKernelGGL(asp_softmax_stats_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    } else if (dtype == VP_F32) {
        AspArgs<float> a{logits, (const float*)x, center, pooled, ldx, xoff, ldc, B, T, C, eps};
        static const bool plain = getenv("VPMI_ASP_STATS_PLAIN") != nullptr;          // A/B switch
        if (T <= 160 && !plain) hipLaunchKernelGGL(asp_softmax_stats_reg_kernel<20>, grid, dim3(512), 0, st, a);
        else if (T <= 320 && !plain) hipLaunchKernelGGL(asp_softmax_stats_reg_kernel<40>, grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL(asp_softmax_stats_kernel<float>, grid, dim3(256), 0, st, a);
    } else {
        VP_FAIL(ctx, VP_EINVAL, "asp: bad dtype");
    }
    VP_LAUNCH_CHECK(ctx, "asp_softmax_stats");
    return VP_OK;
}

extern "C" {

int vp_asp_softmax_stats(vp_ctx* ctx, int dtype, const float* logits, const void* x, int ldx, int xoff,
                         int B, int T, int C, float eps, float* pooled, vp_stream stream) {
    return vp_asp_softmax_stats_ex(ctx, dtype, logits, x, ldx, xoff, nullptr, 0, B, T, C, eps, pooled, (hipStream_t)stream);
}

// f32 x, logits stored as bf16 (the training engine under enable_amp; T <= 320, else VP_EUNSUP: the caller keeps f32 logits)
int vp_asp_softmax_stats_l16(vp_ctx* ctx, const void* logits_bf16, const void* x, int x_dtype, int ldx, int xoff, int B, int T, int C, float eps,
                             float* pooled, vp_stream stream) {
    if (!ctx || !logits_bf16 || !x || !pooled || B <= 0 || T <= 0 || C <= 0 || B > 65535 || (x_dtype != VP_F32 && x_dtype != VP_BF16))
        VP_FAIL(ctx, VP_EINVAL, "asp_l16: bad arguments");
    if (T > 320) return VP_EUNSUP;
    AspArgs<float> a{(const float*)logits_bf16, (const float*)x, nullptr, pooled, ldx, xoff, 0, B, T, C, eps};
    const dim3 grid((C + 63) / 64, B);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == VP_BF16) {
        if (T <= 160) hipLaunchKernelGGL((asp_softmax_stats_reg_kernel<20, bf16_t, bf16_t>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((asp_softmax_stats_reg_kernel<40, bf16_t, bf16_t>), grid, dim3(512), 0, st, a);
    } else if (T <= 160) hipLaunchKernelGGL((asp_softmax_stats_reg_kernel<20, bf16_t>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((asp_softmax_stats_reg_kernel<40, bf16_t>), grid, dim3(512), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "asp_softmax_stats_l16");
    return VP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- f32 -> bf16 cast (model entry when the
// caller hands over f32 features without the fused bf16 twin the Fbank kernel can emit)
namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* x, bf16_t* y, long long n) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        if (i + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            bf16x4 o;
            o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
            *reinterpret_cast<bf16x4*>(y + i) = o;
        } else {
            for (long long j = i; j < n; ++j) y[j] = (bf16_t)x[j];
        }
    }
}
}  // namespace

extern "C" int vp_cast_f32_bf16(vp_ctx* ctx, const float* x, void* y, long long n, vp_stream stream) {
    if (!ctx || !x || !y || n <= 0) VP_FAIL(ctx, VP_EINVAL, "cast: bad arguments");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7)) VP_FAIL(ctx, VP_EINVAL, "cast: misaligned");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
    VP_LAUNCH_CHECK(ctx, "cast_bf16");
    return VP_OK;
}
