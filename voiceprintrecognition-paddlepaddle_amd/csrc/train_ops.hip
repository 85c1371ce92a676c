// Training-side kernels (f32 engine): conv weight gradient, BatchNorm with batch statistics (forward finalise /
// apply, backward reduce / apply with the ReLU mask), column sums, Adam.
//
// Reference: what paddle's autograd + optimiser run for PPVectorTrainer.__train_epoch (ppvector/trainer.py:202-274):
// Conv1D backward (models/utils.py:65-93 layers), BatchNorm1D in train mode (utils.py:96-119: batch statistics over
// (B, T); running = 0.9 running + 0.1 batch), ReLU, and Adam with coupled L2 (optimizer/__init__.py:12-18,
// configs/*.yml weight_decay 1e-6).  Activations are position-major (M = B*T rows, C contiguous) like everywhere else;
// every reduction is two-stage and fixed-order (no atomics): results are bit-reproducible run to run.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- conv weight gradient
// dW[n][j*Cin + c] = sum_{b,t} dz[b, t, n] * x[b, src(t, j), c],  src = the forward's tap index (reflect / zero / none).
// One workgroup = a 64 (n) x 64 (k-column) tile of dW over a slice of the rows; 4 waves of 32 x 32; v_mfma_f32_16x16x4_f32
// with dz^T as the A operand and the shifted x rows as B -- both operands are read row-wise (16 lanes = 64 contiguous
// bytes), the reduction dimension (rows) is the MFMA k.  Partials [S][Cout][K] are summed by wgrad_reduce_kernel.
struct WgradArgs {
    const float* x; const float* dz; float* part;
    int ldx, xoff, lddz, M, N, K, Cin, T_in, T_out, dilation, stride, pad_left, pad_mode, rows_per_split;
    int F_in, F_out, KF, stride_f, pad_f;        // 2-D convs (zero padding): rows are (b, t, f), taps (kt, kf); F_in = F_out = KF = 1 for 1-D
    // batched launch (amp kernel): blockIdx.z = conv * splits + split; conv c reads x + c * xb, dz + c * dzb (elements of their dtype)
    int splits; long long xb, dzb;
};

// LDS-tiled: per 32-row chunk the workgroup stages dz[32][64 n] and the tap-shifted x[32][64 k-columns] with 16-byte loads
// (16 lanes per row, prefetched one chunk ahead in registers), then every wave reads its MFMA operands from LDS
// (row stride 80 floats: the four k-rows of a fragment land on disjoint banks).  The first version fetched every
// fragment straight from global memory, 4 bytes per lane: 53 TFLOP/s on the 1536 x 1536 layer.
constexpr int WG_RC = 32;              // rows per staged chunk
constexpr int WG_LD = 80;              // floats per staged row (64 + 16 pad)

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float dzs[WG_RC * WG_LD];
    __shared__ __attribute__((aligned(16))) float xs[WG_RC * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int wn = wv & 1, wk = wv >> 1;
    const int nb = blockIdx.y * 64, kb = blockIdx.x * 64;
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    // staging role: rows srow and srow + 16 of the chunk, 4 consecutive columns starting at scol
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    const int ncol = nb + scol;                       // dz column (4 consecutive n; N % 4 == 0 -> all or none valid)
    const bool nvalid = ncol < a.N;
    const int kcol = kb + scol;                       // k-column = tap * Cin + c (Cin % 4 == 0: the 4 stay inside one tap)
    const bool kvalid = kcol < a.K;
    const int j = kvalid ? kcol / a.Cin : 0;
    const int cc = kvalid ? kcol - j * a.Cin : 0;
    const int kt = j / a.KF;
    const int tapoff = kt * a.dilation - a.pad_left, tapf = (j - kt * a.KF) - a.pad_f;
    const bool vec_ok = (a.Cin & 3) == 0 && ((a.ldx | a.xoff) & 3) == 0 && (a.lddz & 3) == 0;

    float4 rdz[2], rx[2];
    auto gload = [&](int mbase) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = mbase + srow + 16 * h;
            rdz[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            rx[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m >= m_end) continue;
            if (nvalid) rdz[h] = *reinterpret_cast<const float4*>(a.dz + (size_t)m * a.lddz + ncol);
            if (kvalid) {
                const int bt = m / a.F_out, f = m - bt * a.F_out;
                const int b = bt / a.T_out, t = bt - b * a.T_out;
                const int traw = t * a.stride + tapoff, fs = f * a.stride_f + tapf;
                int ts = traw;
                bool ok = fs >= 0 && fs < a.F_in;
                if (a.pad_mode == VP_PAD_REFLECT) {
                    ts = ts < 0 ? -ts : ts;
                    ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                } else {
                    ok = ok && traw >= 0 && traw < a.T_in;
                }
                if (ok) rx[h] = *reinterpret_cast<const float4*>(a.x + (((size_t)b * a.T_in + ts) * a.F_in + fs) * a.ldx + a.xoff + cc);
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(dzs + (srow + 16 * h) * WG_LD + scol) = rdz[h];
            *reinterpret_cast<float4*>(xs + (srow + 16 * h) * WG_LD + scol) = rx[h];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!vec_ok) return;                              // host refuses these shapes (kept so the kernel cannot misread)
    gload(m_begin);
    for (int mb = m_begin; mb < m_end; mb += WG_RC) {
        swrite();
        __syncthreads();
        gload(mb + WG_RC);                            // rows past m_end come back as zeros
#pragma unroll
        for (int r = 0; r < WG_RC; r += 4) {
            float av[2], bv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                av[q] = dzs[(r + kk) * WG_LD + wn * 32 + q * 16 + i];
                bv[q] = xs[(r + kk) * WG_LD + wk * 32 + q * 16 + i];
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p], bv[q], acc[p][q], 0, 0, 0);
        }
        __syncthreads();
    }
    const int n0 = nb + wn * 32, k0 = kb + wk * 32;
    float* out = a.part + (size_t)blockIdx.z * a.N * a.K;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = k0 + q * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + p * 16 + kk * 4 + r;
                if (n < a.N && col < a.K) out[(size_t)n * a.K + col] = acc[p][q][r];
            }
        }
}

// Mixed precision (vp_conv1d_desc.mfma_bf16): the same contraction on the bf16 matrix cores.  One workgroup = a 128 (n) x 128
// (k-column) tile of dW over a slice of the rows, 4 waves of 64 x 64 (v_mfma_f32_16x16x32_bf16, f32 accumulate).  The reduction
// index is the ROW, along which both operands are strided in memory: a thread loads 8 rows x 4 consecutive columns (16-byte
// loads, a half-wave = 512 contiguous bytes of a row), rounds to bf16 and writes the four 8-row column pieces TRANSPOSED into
// LDS ([column][64 rows], 128 B per column), where a fragment is one ds_read_b128.  The 16-byte chunk q of column R sits at
// position q ^ L(R), L(R) = ((R >> 1) & 7) ^ ((R >> 4) & 1): conflict-free for the transposed writes (8 lanes = columns 4
// apart) AND for the fragment reads (searched exhaustively over XOR-linear maps against the ds_read_b128 lane groups).
constexpr int WA_T = 128;              // tile edge of dW
constexpr int WA_RC = 64;              // rows per staged chunk = two MFMA k-steps
__device__ __forceinline__ int wa_L(int R) { return ((R >> 1) & 7) ^ ((R >> 4) & 1); }

// BF: x and dz are already bf16 in memory (the wide layers' operands, see ConvBlock): 8-byte loads of four bf16, no rounding here.
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

// X3 (f32 operands only): split precision -- each operand is staged as TWO transposed bf16 planes (hi = bf16(v), lo = bf16(v - hi)) and
// the contraction is lo*hi + hi*lo + hi*hi (conv_gemm_impl.h: x3_t): the weight gradient of the 'float32x3' training mode, ~2^-17 per
// product.  Twice the LDS (128 KB: one workgroup per CU) and three MFMAs per fragment pair.
template <bool BF, bool X3 = false>
__global__ __launch_bounds__(256, X3 ? 1 : 2) void conv_wgrad_amp_kernel(WgradArgs a) {
    static_assert(!(BF && X3), "split precision takes f32 operands");
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr int STG = PLANES * 2 * WA_T * 128;                       // bytes per stage
    extern __shared__ __attribute__((aligned(16))) char wsm[];        // [2 stages][dz^T 16 KB | x^T 16 KB] (x3: + the two lo planes)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int wn = wv & 1, wk = wv >> 1;
    const int nb = blockIdx.y * WA_T, kb = blockIdx.x * WA_T;
    const int zb = blockIdx.z / a.splits, sp = blockIdx.z - zb * a.splits;
    const int m_begin = sp * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    // staging role: rows 8 rg .. 8 rg + 7 of the chunk, columns 4 cg .. 4 cg + 3 of the tile
    const int cg = tid & 31, rg = tid >> 5;
    const int ncol = nb + 4 * cg;
    const bool nvalid = ncol < a.N;
    const int kcol = kb + 4 * cg;                     // k-column = tap * Cin + c (Cin % 4 == 0: the 4 stay inside one tap)
    const bool kvalid = kcol < a.K;
    const int j = kvalid ? kcol / a.Cin : 0;
    const int cc = kvalid ? kcol - j * a.Cin : 0;
    const int kt = j / a.KF;
    const int tapoff = kt * a.dilation - a.pad_left, tapf = (j - kt * a.KF) - a.pad_f;
    // 1x1 convs with identical input / output geometry (the wide layers): source row = output row
    const bool simple = a.K == a.Cin && a.stride == 1 && a.pad_left == 0 && a.T_in == a.T_out && a.F_in == 1 && a.F_out == 1;

    using ld_t = typename std::conditional<BF, uint2, float4>::type;
    using el_t = typename std::conditional<BF, bf16_t, float>::type;
    const el_t* __restrict__ gdz = reinterpret_cast<const el_t*>(a.dz) + zb * a.dzb;
    const el_t* __restrict__ gx = reinterpret_cast<const el_t*>(a.x) + zb * a.xb;
    ld_t rdz[8], rx[8];
    auto gload = [&](int mbase) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = mbase + 8 * rg + r;
            rdz[r] = ld_t{};
            rx[r] = ld_t{};
            if (m >= m_end) continue;
            if (nvalid) rdz[r] = *reinterpret_cast<const ld_t*>(gdz + (size_t)m * a.lddz + ncol);
            if (!kvalid) continue;
            if (simple) {
                rx[r] = *reinterpret_cast<const ld_t*>(gx + (size_t)m * a.ldx + a.xoff + cc);
            } else {
                const int bt = m / a.F_out, f = m - bt * a.F_out;
                const int b = bt / a.T_out, t = bt - b * a.T_out;
                const int traw = t * a.stride + tapoff, fs = f * a.stride_f + tapf;
                int ts = traw;
                bool ok = fs >= 0 && fs < a.F_in;
                if (a.pad_mode == VP_PAD_REFLECT) {
                    ts = ts < 0 ? -ts : ts;
                    ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                } else {
                    ok = ok && traw >= 0 && traw < a.T_in;
                }
                if (ok) rx[r] = *reinterpret_cast<const ld_t*>(gx + (((size_t)b * a.T_in + ts) * a.F_in + fs) * a.ldx + a.xoff + cc);
            }
        }
    };
    auto swrite = [&](int s) {
        char* dzs = wsm + s * STG;
        char* xs = dzs + WA_T * 128;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int R = 4 * cg + e;
            const int pos = (rg ^ wa_L(R)) << 4;
            if constexpr (BF) {
                u16x8 vd, vx;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned wd = e < 2 ? rdz[r].x : rdz[r].y, wx = e < 2 ? rx[r].x : rx[r].y;
                    vd[r] = (unsigned short)((e & 1) ? (wd >> 16) : (wd & 0xffffu));
                    vx[r] = (unsigned short)((e & 1) ? (wx >> 16) : (wx & 0xffffu));
                }
                *reinterpret_cast<u16x8*>(dzs + R * 128 + pos) = vd;
                *reinterpret_cast<u16x8*>(xs + R * 128 + pos) = vx;
            } else {
                bf16x8 vd, vx;
                [[maybe_unused]] bf16x8 ld, lx;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float d = e == 0 ? rdz[r].x : (e == 1 ? rdz[r].y : (e == 2 ? rdz[r].z : rdz[r].w));
                    const float xv = e == 0 ? rx[r].x : (e == 1 ? rx[r].y : (e == 2 ? rx[r].z : rx[r].w));
                    vd[r] = (bf16_t)d;
                    vx[r] = (bf16_t)xv;
                    if constexpr (X3) {
                        ld[r] = (bf16_t)(d - (float)vd[r]);
                        lx[r] = (bf16_t)(xv - (float)vx[r]);
                    }
                }
                *reinterpret_cast<bf16x8*>(dzs + R * 128 + pos) = vd;
                *reinterpret_cast<bf16x8*>(xs + R * 128 + pos) = vx;
                if constexpr (X3) {
                    *reinterpret_cast<bf16x8*>(dzs + 2 * WA_T * 128 + R * 128 + pos) = ld;
                    *reinterpret_cast<bf16x8*>(xs + 2 * WA_T * 128 + R * 128 + pos) = lx;
                }
            }
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    gload(m_begin);
    int s = 0;
    for (int mb = m_begin; mb < m_end; mb += WA_RC) {
        swrite(s);
        __syncthreads();
        gload(mb + WA_RC);                            // rows past m_end come back as zeros
        const char* dzs = wsm + s * STG;
        const char* xs = dzs + WA_T * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[4];
            [[maybe_unused]] bf16x8 al[4], bl[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int R = wn * 64 + p * 16 + i;
                af[p] = *reinterpret_cast<const bf16x8*>(dzs + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
                if constexpr (X3) al[p] = *reinterpret_cast<const bf16x8*>(dzs + 2 * WA_T * 128 + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = wk * 64 + q * 16 + i;
                bf[q] = *reinterpret_cast<const bf16x8*>(xs + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
                if constexpr (X3) bl[q] = *reinterpret_cast<const bf16x8*>(xs + 2 * WA_T * 128 + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (X3) {
                        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[p], bf[q], acc[p][q], 0, 0, 0);
                        acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p], bl[q], acc[p][q], 0, 0, 0);
                    }
                    acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p], bf[q], acc[p][q], 0, 0, 0);
                }
        }
        s ^= 1;
    }
    const int n0 = nb + wn * 64, k0 = kb + wk * 64;
    float* out = a.part + (size_t)blockIdx.z * a.N * a.K;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = k0 + q * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + p * 16 + g * 4 + r;
                if (n < a.N && col < a.K) out[(size_t)n * a.K + col] = acc[p][q][r];
            }
        }
}

// The same contraction for a NARROW layer: N <= 64 outputs (the Res2Net chunk convs: 64 x 192; the 32- and 64-channel 2-D convs of
// ResNetSE / ERes2Net on their largest feature maps).  The square 128 x 128 tile above spends up to 3/4 of its MFMAs and LDS traffic on
// output rows that do not exist there; here a workgroup takes all N <= 64 outputs x 256 k-columns (blockIdx.x), the four waves 64
// k-columns each.  LDS: 2 stages x (dz^T 64 x 128 B | x^T 256 x 128 B).  BF: bf16 operands in memory (1-D only, as the caller has them);
// else f32 operands rounded on the way in, 1-D or 2-D taps as in the kernel above.
__device__ __forceinline__ uint2 wg_pack4(float4 v) {
    bf16x4 o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    return __builtin_bit_cast(uint2, o);
}
template <bool BF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_n64_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    constexpr int STAGE = (64 + 256) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int kb = blockIdx.x * 256;
    const int zb = blockIdx.z / a.splits, sp = blockIdx.z - zb * a.splits;
    const int m_begin = sp * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int cg = tid & 31, rg = tid >> 5;
    const bool nvalid = cg < 16 && 4 * cg < a.N;
    int cc[2], tapoff[2], tapf[2];
    bool kvalid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kcol = kb + 128 * u + 4 * cg;
        kvalid[u] = kcol < a.K;
        const int j = kvalid[u] ? kcol / a.Cin : 0;
        cc[u] = kvalid[u] ? kcol - j * a.Cin : 0;
        const int kt = j / a.KF;
        tapoff[u] = kt * a.dilation - a.pad_left;
        tapf[u] = (j - kt * a.KF) - a.pad_f;
    }
    uint2 rdz[8], rx[2][8];
    auto gload = [&](int mbase) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = mbase + 8 * rg + r;
            rdz[r] = uint2{0u, 0u}; rx[0][r] = uint2{0u, 0u}; rx[1][r] = uint2{0u, 0u};
            if (m >= m_end) continue;
            if (nvalid) {
                if constexpr (BF) rdz[r] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.dz) + zb * a.dzb + (size_t)m * a.lddz + 4 * cg);
                else rdz[r] = wg_pack4(*reinterpret_cast<const float4*>(a.dz + (size_t)m * a.lddz + 4 * cg));
            }
            const int bt = m / a.F_out, f = m - bt * a.F_out;
            const int b = bt / a.T_out, t = bt - b * a.T_out;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!kvalid[u]) continue;
                const int traw = t * a.stride + tapoff[u], fs = f * a.stride_f + tapf[u];
                int ts = traw;
                bool ok = fs >= 0 && fs < a.F_in;
                if (a.pad_mode == VP_PAD_REFLECT) {
                    ts = ts < 0 ? -ts : ts;
                    ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                } else {
                    ok = ok && traw >= 0 && traw < a.T_in;
                }
                if (!ok) continue;
                const size_t o = (((size_t)b * a.T_in + ts) * a.F_in + fs) * a.ldx + a.xoff + cc[u];
                if constexpr (BF) rx[u][r] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.x) + zb * a.xb + o);
                else rx[u][r] = wg_pack4(*reinterpret_cast<const float4*>(a.x + o));
            }
        }
    };
    auto put = [&](char* base, int R, int e, const uint2 (&v)[8]) {
        u16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned w = e < 2 ? v[r].x : v[r].y;
            o[r] = (unsigned short)((e & 1) ? (w >> 16) : (w & 0xffffu));
        }
        *reinterpret_cast<u16x8*>(base + R * 128 + ((rg ^ wa_L(R)) << 4)) = o;
    };
    auto swrite = [&](int s) {
        char* dzs = wsm + s * STAGE;
        char* xs = dzs + 64 * 128;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (cg < 16) put(dzs, 4 * cg + e, e, rdz);
            put(xs, 4 * cg + e, e, rx[0]);
            put(xs, 128 + 4 * cg + e, e, rx[1]);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    gload(m_begin);
    int s = 0;
    for (int mb = m_begin; mb < m_end; mb += WA_RC) {
        swrite(s);
        __syncthreads();
        gload(mb + WA_RC);
        const char* dzs = wsm + s * STAGE;
        const char* xs = dzs + 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int R = p * 16 + i;
                af[p] = *reinterpret_cast<const bf16x8*>(dzs + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = wv * 64 + q * 16 + i;
                bf[q] = *reinterpret_cast<const bf16x8*>(xs + R * 128 + (((ks * 4 + g) ^ wa_L(R)) << 4));
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p], bf[q], acc[p][q], 0, 0, 0);
        }
        s ^= 1;
    }
    float* out = a.part + (size_t)blockIdx.z * a.N * a.K;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kb + wv * 64 + q * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = p * 16 + g * 4 + r;
                if (n < a.N && col < a.K) out[(size_t)n * a.K + col] = acc[p][q][r];
            }
        }
}

// out[i] = sum_k part[k][i]: 16 outputs x 16 partial-lanes per workgroup, fixed order (one thread walking all S partials
// serially took 33 us per call -- 3.6 ms of a 39 ms training step over ~110 calls)
// (KW > 1: the weight gradient leaves in the model's (Cout, Cin, KW) layout: partial index (o, tap, c) -> (o, c, tap))
// (2-D convs, KF > 1: the partial's tap index is kt * KF + kf -- the forward kernel's order -- the model's is kf * KT + kt)
__device__ __forceinline__ long long oik_index(long long idx, int Cin, int KW, int KF = 1) {
    if (KW <= 1) return idx;
    const long long o = idx / ((long long)KW * Cin);
    const int r = (int)(idx - o * KW * Cin);
    int j = r / Cin;
    const int c = r - j * Cin;
    if (KF > 1) j = (j % KF) * (KW / KF) + j / KF;
    return (o * Cin + c) * KW + j;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* part, int S, long long n, float* out, int Cin, int KW, int KF) {
    __shared__ float sm[16][17];
    part += (size_t)blockIdx.y * S * n;               // batched weight gradients: one (S, n) slab and one output per blockIdx.y
    out += (size_t)blockIdx.y * n;
    const int ol = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long long idx = (long long)blockIdx.x * 16 + ol;
    float s = 0.f;
    if (idx < n) {
        int k = pl;
        for (; k + 48 < S; k += 64) {             // four partial rows per trip, their loads issued together, summed in row order
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = part[(size_t)(k + 16 * u) * n + idx];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; k < S; k += 16) s += part[(size_t)k * n + idx];
    }
    sm[pl][ol] = s;
    __syncthreads();
    if (pl == 0 && idx < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k][ol];
        out[oik_index(idx, Cin, KW, KF)] = t;
    }
}

// Many outputs (a weight gradient of >= 32 768 elements): 64 consecutive outputs per workgroup, the S partial rows dealt to the four
// waves (row k to wave k % 4, four loads in flight per lane) and the four wave sums added in wave order -- a fixed order.  (One thread
// walking all S rows of its output: 0.97 TB/s on a 33 MB partial slab, 37 us per call and 3-5 % of the 2-D backbones' training steps.)
__global__ __launch_bounds__(256) void sum_partials_wide_kernel(const float* part, int S, long long n, float* out, int Cin, int KW, int KF) {
    __shared__ float sm[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    part += (size_t)blockIdx.y * S * n;
    out += (size_t)blockIdx.y * n;
    float s = 0.f;
    if (i < n) {
        int k = w;
        for (; k + 12 < S; k += 16) {
            const float v0 = part[(size_t)k * n + i], v1 = part[(size_t)(k + 4) * n + i];
            const float v2 = part[(size_t)(k + 8) * n + i], v3 = part[(size_t)(k + 12) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < S; k += 4) s += part[(size_t)k * n + i];
    }
    if (w) sm[w - 1][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) out[oik_index(i, Cin, KW, KF)] = ((s + sm[0][lane]) + sm[1][lane]) + sm[2][lane];
}

// wp[o][tap * Cin + c] = w[o][c][tap] (the forward kernel's weight panel) and w2[c][(KW - 1 - tap) * Cout + o] = w[o][c][tap]
// (the data-gradient conv's: taps reversed, channel roles swapped) from the model's (Cout, Cin, KW) tensor in one launch.
// 2-D (KF > 1): the model stores (Cout, Cin, KF, KT), tap kf * KT + kt; the kernels' tap order is kt * KF + kf (KW = KF * KT taps).
__global__ __launch_bounds__(256) void weight_layouts_kernel(const float* w, int Cout, int Cin, int KW, int KF, float* wp, float* w2) {
    const long long n = (long long)Cout * Cin * KW;
    const int KT = KW / KF;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int o = (int)(i / ((long long)Cin * KW));
        const int r = (int)(i - (long long)o * Cin * KW);
        const int c = r / KW;
        int j = r - c * KW;
        if (KF > 1) j = (j % KT) * KF + j / KT;
        const float v = w[i];
        if (wp) wp[((size_t)o * KW + j) * Cin + c] = v;
        if (w2) w2[((size_t)c * KW + (KW - 1 - j)) * Cout + o] = v;
    }
}

// ---------------------------------------------------------------------------------------------- bf16 weight panels, all layers at once
// For up to WPREP_MAX f32 matrices w_e (rows_e x cols_e, row pitch ld_e: a column slice of a wider tensor is allowed): the bf16 copy
// w16_e [rows][cols] (the forward GEMM's panel) and the bf16 TRANSPOSE wt16_e [cols][rows] (the data-gradient GEMM's panel) in ONE
// launch -- what a mixed-precision step used to do with two conversion launches per wide layer (auto_cast casts every conv's weight per
// call in the reference, trainer.py:209-213).  64 x 64 tiles through LDS: both outputs leave as 128-byte rows.
constexpr int WPREP_MAX = 24;
struct WPrepArgs {
    const float* w[WPREP_MAX]; bf16_t* w16[WPREP_MAX]; bf16_t* wt16[WPREP_MAX];
    int rows[WPREP_MAX], cols[WPREP_MAX], ld[WPREP_MAX], tile0[WPREP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void weight_prep_kernel(WPrepArgs a) {
    __shared__ float sm[64][65];
    int e = 0;
    while (e + 1 < a.n && (int)blockIdx.x >= a.tile0[e + 1]) ++e;
    const int t = blockIdx.x - a.tile0[e];
    const int rows = a.rows[e], cols = a.cols[e], ld = a.ld[e];
    const int tc = (cols + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const float* __restrict__ w = a.w[e];
    bf16_t* __restrict__ w16 = a.w16[e];
    bf16_t* __restrict__ wt = a.wt16[e];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ly; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + lx;
        const float v = (r < rows && c < cols) ? w[(size_t)r * ld + c] : 0.f;
        sm[i][lx] = v;
        if (w16 && r < rows && c < cols) w16[(size_t)r * cols + c] = (bf16_t)v;
    }
    if (!wt) return;
    __syncthreads();
#pragma unroll 4
    for (int i = ly; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + lx;
        if (c < cols && r < rows) wt[(size_t)c * rows + r] = (bf16_t)sm[lx][i];
    }
}

// ---------------------------------------------------------------------------------------------- column sums over rows
// part[chunk][which][c]: which 0 = sum_m a[m][c], 1 = sum_m a[m][c] * (b[m][c] - bmean[c]) * bscale[c]  (second only when b)
struct ColSumArgs { const float* a; const float* b; const float* bmean; const float* bscale; float* part; int lda, ldb, M, C, rows_per_chunk; };

__global__ __launch_bounds__(256) void col_sums_kernel(ColSumArgs p) {
    __shared__ float sm[2][4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lc;
    const int m0 = blockIdx.y * p.rows_per_chunk, m1 = min(p.M, m0 + p.rows_per_chunk);
    float s1 = 0.f, s2 = 0.f;
    if (c < p.C) {
        const float mu = p.b ? p.bmean[c] : 0.f, sc = p.b ? p.bscale[c] : 0.f;
        for (int m = m0 + rg; m < m1; m += 4) {
            const float av = p.a[(size_t)m * p.lda + c];
            s1 += av;
            if (p.b) s2 += av * (p.b[(size_t)m * p.ldb + c] - mu) * sc;
        }
    }
    sm[0][rg][lc] = s1; sm[1][rg][lc] = s2;
    __syncthreads();
    if (rg == 0 && c < p.C) {
        float* o = p.part + (size_t)blockIdx.y * 2 * p.C;
        o[c] = sm[0][0][lc] + sm[0][1][lc] + sm[0][2][lc] + sm[0][3][lc];
        o[p.C + c] = sm[1][0][lc] + sm[1][1][lc] + sm[1][2][lc] + sm[1][3][lc];
    }
}

// Four channels per lane (C and the row pitches multiples of 4, 16-byte aligned bases): a wave-instruction moves 1 KB of a row
// instead of 256 B, and four rows' loads are issued before the first add.  Workgroup = CL column lanes x (256 / CL) row groups
// over rows_per_chunk rows; CL = 64 / 32 / 16 by width so that 64-channel tensors (the Res2 convs) still fill the lanes.
#ifndef VP_BNBWD_ROWS
// rows in flight per thread of the BatchNorm-backward passes.  Round 5 A/B (tools/build_variant.sh u8 -DVP_BNBWD_ROWS=8, one box, one
// session): 8 rows 8.043 ms per ECAPA B = 256 training step, 4 rows 8.011 ms -- the passes are not short of loads in flight; 4 it stays
#define VP_BNBWD_ROWS 4
#endif
struct ColSum4Args { const float* a; const float* b; const float* bmean; const float* bscale; float* part; int lda, ldb, M, C4, rows_per_chunk, cl_shift;
                     const float* ms; const float* mh;      // optional: a counts only where b * ms + mh > 0 (a ReLU BEHIND the BatchNorm, resnet_se.py:72-74)
                     // UTT: the summed tensor is a * us[b] + um[b] * inv_t per utterance b = row / T -- the SE block's input gradient
                     // dh = dout * s + dmean / T (ecapa_tdnn.py:50-82 backward) formed on the fly, never stored
                     const float* us; const float* um; int T; float inv_t;
                     // UTT == 2: the summed tensor is a + us[b] + um[b] * bf16(b * xsc + xsh) -- the gradient that reaches a TDNNBlock's output y = BN(z)
                     // through its consumer's context statistics [mean_t y | std_t y] (pooling.py:97-104), affine in y per utterance and
                     // channel: us = alpha, um = beta of vp_time_stats_bwd_coeffs; y re-formed from the bf16 z the pass reads anyway
                     const float* xsc; const float* xsh;
                     float mask_hi; };     // with ms / mh: a also counts only where b * ms + mh < mask_hi (0: no upper bound) -- Hardtanh(0, 20) behind the BatchNorm (eres2net.py)

template <int UTT>
__device__ __forceinline__ void utt_affine4(float (&v)[4], const float* us, const float* um, int T, float inv_t, int C, int m, int c) {
    if constexpr (UTT == 1) {
        const size_t o = (size_t)(m / T) * C + c;
        float sv[4], dv[4];
        vp_load4(us + o, sv); vp_load4(um + o, dv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fmaf_rn(v[e], sv[e], dv[e] * inv_t);
    }
}

// UTT == 2: v += alpha[b] + beta[b] * bf16(z * xsc + xsh)
template <int UTT>
__device__ __forceinline__ void ctx_affine4(float (&v)[4], const float (&zv)[4], const float* ua, const float* ub, const float (&xsc)[4],
                                            const float (&xsh)[4], int T, int C, int m, int c) {
    if constexpr (UTT == 2) {
        const size_t o = (size_t)(m / T) * C + c;
        float al[4], be[4];
        vp_load4(ua + o, al); vp_load4(ub + o, be);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __fmaf_rn(be[e], (float)(bf16_t)__fmaf_rn(zv[e], xsc[e], xsh[e]), al[e]);
    }
}

template <bool HASB, typename TB = float, int UTT = 0>
__global__ __launch_bounds__(256) void col_sums4_kernel(ColSum4Args p) {
    const TB* __restrict__ gb = reinterpret_cast<const TB*>(p.b);
    __shared__ float sm[2][256][4];
    const int CL = 1 << p.cl_shift, RG = 256 >> p.cl_shift;
    const int lc = threadIdx.x & (CL - 1), rg = threadIdx.x >> p.cl_shift;
    const int c4 = blockIdx.x * CL + lc, c = c4 * 4;
    const int m0 = blockIdx.y * p.rows_per_chunk, m1 = min(p.M, m0 + p.rows_per_chunk);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c4 < p.C4) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
        if (HASB) { vp_load4(p.bmean + c, mu); vp_load4(p.bscale + c, sc); }
        float ms[4] = {0.f, 0.f, 0.f, 0.f}, mh[4] = {1.f, 1.f, 1.f, 1.f};          // (no mask: 0 * b + 1 > 0 always)
        if (HASB && p.ms) { vp_load4(p.ms + c, ms); vp_load4(p.mh + c, mh); }
        float xsc[4] = {0.f, 0.f, 0.f, 0.f}, xsh[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (UTT == 2) { vp_load4(p.xsc + c, xsc); vp_load4(p.xsh + c, xsh); }
        int m = m0 + rg;
        // VP_BNBWD_ROWS rows per trip, their loads issued together, summed in row order
        constexpr int U = UTT ? 4 : VP_BNBWD_ROWS;              // (the per-utterance operands cost the registers of four rows)
        for (; m + (U - 1) * RG < m1; m += U * RG) {
            float av[U][4], bv[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vp_load4(p.a + (size_t)(m + u * RG) * p.lda + c, av[u]);
                if (HASB) vp_load4(gb + (size_t)(m + u * RG) * p.ldb + c, bv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                utt_affine4<UTT>(av[u], p.us, p.um, p.T, p.inv_t, p.C4 * 4, m + u * RG, c);
                if constexpr (HASB) ctx_affine4<UTT>(av[u], bv[u], p.us, p.um, xsc, xsh, p.T, p.C4 * 4, m + u * RG, c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = av[u][e];
                    if constexpr (HASB) {
                        const float mv = (float)bv[u][e] * ms[e] + mh[e];
                        g = (mv > 0.f && (p.mask_hi == 0.f || mv < p.mask_hi)) ? g : 0.f;
                    }
                    s1[e] += g;
                    if (HASB) s2[e] += g * (bv[u][e] - mu[e]) * sc[e];
                }
        }
        for (; m < m1; m += RG) {
            float av[4], bv[4];
            vp_load4(p.a + (size_t)m * p.lda + c, av);
            if (HASB) vp_load4(gb + (size_t)m * p.ldb + c, bv);
            utt_affine4<UTT>(av, p.us, p.um, p.T, p.inv_t, p.C4 * 4, m, c);
            if constexpr (HASB) ctx_affine4<UTT>(av, bv, p.us, p.um, xsc, xsh, p.T, p.C4 * 4, m, c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = av[e];
                if constexpr (HASB) {
                    const float mv = (float)bv[e] * ms[e] + mh[e];
                    g = (mv > 0.f && (p.mask_hi == 0.f || mv < p.mask_hi)) ? g : 0.f;
                }
                s1[e] += g;
                if (HASB) s2[e] += g * (bv[e] - mu[e]) * sc[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[0][threadIdx.x][e] = s1[e]; sm[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    if (rg == 0 && c4 < p.C4) {
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) { t1[e] += sm[0][r * CL + lc][e]; t2[e] += sm[1][r * CL + lc][e]; }
        float* o = p.part + (size_t)blockIdx.y * 8 * p.C4;
        vp_store4(o + c, t1); vp_store4(o + 4 * p.C4 + c, t2);
    }
}

// ---------------------------------------------------------------------------------------------- BatchNorm, batch statistics
// from the producing conv's fused sums: mean / biased variance over all M rows, the folded affine for the apply pass,
// the saved mean / inverse std for backward, and the running statistics (Paddle: running = mom * running + (1 - mom) * batch).
struct BnFinArgs {
    const float* psum; const float* psumsq; int nparts; int M, C;
    const float* gamma; const float* beta; float* run_mean; float* run_var; float momentum, eps;
    float* mean; float* invstd; float* scale; float* shift;
    const unsigned* fault;      // the context's grid-barrier bail-out word: while it is set the running statistics are left untouched
};

// One workgroup = 16 channels x 16 part-lanes: the nparts partial rows (tiles x segments: ~1200 at B = 256 x 3 s) are
// summed 16 at a time with independent loads, then across the 16 lanes in fixed order.  (One thread per channel walking all
// the parts serially took 336 us per call -- 10 ms of a 54 ms training step.)
__global__ __launch_bounds__(256) void bn_train_finalize_kernel(BnFinArgs a) {
    __shared__ float sm[2][16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < a.C) {
        // four partial rows per trip with their eight loads issued together (one load latency per trip instead of per row: the
        // ~75 rows per thread made this 25 us per call), summed in row order
        int k = pl;
        for (; k + 48 < a.nparts; k += 64) {
            float p[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { p[u] = a.psum[(size_t)(k + 16 * u) * a.C + c]; q[u] = a.psumsq[(size_t)(k + 16 * u) * a.C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += p[u]; s2 += q[u]; }
        }
        for (; k < a.nparts; k += 16) { s1 += a.psum[(size_t)k * a.C + c]; s2 += a.psumsq[(size_t)k * a.C + c]; }
    }
    sm[0][pl][cl] = s1; sm[1][pl][cl] = s2;
    __syncthreads();
    if (pl != 0 || c >= a.C) return;
    s1 = 0.f; s2 = 0.f;
    for (int k = 0; k < 16; ++k) { s1 += sm[0][k][cl]; s2 += sm[1][k][cl]; }
    const float mu = s1 / (float)a.M;
    const float var = fmaxf(s2 / (float)a.M - mu * mu, 0.f);
    const float is = rsqrtf(var + a.eps);
    a.mean[c] = mu; a.invstd[c] = is;
    const float g = a.gamma ? a.gamma[c] : 1.f, be = a.beta ? a.beta[c] : 0.f;
    a.scale[c] = g * is;
    a.shift[c] = be - mu * g * is;
    // a fused training kernel upstream gave up at its grid barrier: this layer's batch statistics come from incomplete sums.  The step is
    // dropped by the optimiser kernels; the PERSISTENT state written here (the running statistics) must not see it either
    if (a.fault && __hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (a.run_mean) a.run_mean[c] = a.momentum * a.run_mean[c] + (1.f - a.momentum) * mu;
    if (a.run_var) a.run_var[c] = a.momentum * a.run_var[c] + (1.f - a.momentum) * var;
}

// y = z * scale + shift
template <typename TZ, typename TY = float>
__global__ __launch_bounds__(256) void affine_rows_kernel(const TZ* z, int ldz, const float* scale, const float* shift, long long M,
                                                          int C4, TY* y, int ldy, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * C4; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        float v[4], s[4], h[4];
        vp_load4(z + m * ldz + c, v); vp_load4(scale + c, s); vp_load4(shift + c, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = __fmaf_rn(v[e], s[e], h[e]);
            if (relu) v[e] = fmaxf(v[e], 0.f);
            if (relu == 2) v[e] = fminf(v[e], 20.f);          // Hardtanh(0, 20)
        }
        vp_store4(y + m * ldy + c, v);
    }
}

// y = z * scale + shift into a channel slice of a wider tensor (ldy), and aux = y + add (add: another slice of a wider tensor):
// a Res2Net chunk's output written where the concatenation wants it, and the next chunk's input y_i + x_{i+1} from the same pass
struct AffAuxArgs { const float* z; const float* scale; const float* shift; float* y; const float* add; float* aux;
                    int ldz, ldy, ld_add, ld_aux, C4; long long M; };
__global__ __launch_bounds__(256) void affine_rows_aux_kernel(AffAuxArgs a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.M * a.C4; i += (long long)gridDim.x * 256) {
        const long long m = i / a.C4;
        const int c = (int)(i - m * a.C4) * 4;
        float v[4], s[4], h[4];
        vp_load4(a.z + m * a.ldz + c, v); vp_load4(a.scale + c, s); vp_load4(a.shift + c, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * s[e] + h[e];
        vp_store4(a.y + m * a.ldy + c, v);
        if (a.aux) {
            float ad[4];
            vp_load4(a.add + m * a.ld_add + c, ad);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ad[e];
            vp_store4(a.aux + m * a.ld_aux + c, v);
        }
    }
}

// dz = [z > 0] * gamma * invstd * (dy - sum_dy / M - zhat * sum_dy_zhat / M),  zhat = (z - mean) * invstd
// (BatchNorm backward through y = BN(z), then the ReLU that produced z; relu_mask = 0 skips the mask)
struct BnBwdArgs {
    const float* dy; const float* z; const float* mean; const float* invstd; const float* gamma; const float* sums;   // sums [2][C]
    float* dz; int lddy, ldz, lddz, C4, relu_mask; long long M;
    const float* ms; const float* mh;       // optional: d y counts only where z * ms + mh > 0 (a ReLU BEHIND the BatchNorm)
    const float* us; const float* um; int T; float inv_t;      // bn_relu_bwd_dbias_kernel<.., UTT>: d y = dy * us[b] + um[b] * inv_t, b = row / T
    const float* xsc; const float* xsh;                        // UTT == 2: d y = dy + us[b] + um[b] * bf16(z * xsc + xsh) (see ColSum4Args)
    float mask_hi;                                             // with ms / mh: ... and z * ms + mh < mask_hi (0: no upper bound)
    int accumulate;                                            // bn_relu_bwd_kernel: dz += (a DenseNet block's shared gradient buffer, campplus.py:168-171)
};

__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(BnBwdArgs a) {
    const float invM = 1.f / (float)a.M;
    const int C = a.C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.M * a.C4; i += (long long)gridDim.x * 256) {
        const long long m = i / a.C4;
        const int c = (int)(i - m * a.C4) * 4;
        float dy[4], z[4], mu[4], is[4], g[4], s1[4], s2[4], o[4];
        vp_load4(a.dy + m * a.lddy + c, dy); vp_load4(a.z + m * a.ldz + c, z);
        vp_load4(a.mean + c, mu); vp_load4(a.invstd + c, is); vp_load4(a.sums + c, s1); vp_load4(a.sums + C + c, s2);
        if (a.gamma) vp_load4(a.gamma + c, g); else { g[0] = g[1] = g[2] = g[3] = 1.f; }
        if (a.ms) {
            float ms[4], mh[4];
            vp_load4(a.ms + c, ms); vp_load4(a.mh + c, mh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mv = z[e] * ms[e] + mh[e];
                dy[e] = (mv > 0.f && (a.mask_hi == 0.f || mv < a.mask_hi)) ? dy[e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float zh = (z[e] - mu[e]) * is[e];
            const float v = g[e] * is[e] * (dy[e] - s1[e] * invM - zh * s2[e] * invM);
            o[e] = (a.relu_mask && !(z[e] > 0.f)) ? 0.f : v;
        }
        if (a.accumulate) {
            float old[4];
            vp_load4(a.dz + m * a.lddz + c, old);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += old[e];
        }
        vp_store4(a.dz + m * a.lddz + c, o);
    }
}

// The same dz, laid out like col_sums4_kernel so that a lane keeps its four channels: the column sums of dz (the bias gradient of
// the conv in front of the ReLU) come out of the pass that writes dz instead of a second read of it.  part[chunk][C].
struct BnBwdSumArgs { BnBwdArgs b; float* part; int M, rows_per_chunk, cl_shift; };

// TO = bf16_t: dz leaves as bf16 (b.dz reinterpreted, lddz in elements) -- the operand the wide layers' data- and weight-gradient
// GEMMs read (they would round it to bf16 anyway); the column sums are of the unrounded values.
template <typename TO, typename TZ = float, int UTT = 0>
__global__ __launch_bounds__(256) void bn_relu_bwd_dbias_kernel(BnBwdSumArgs p) {
    __shared__ float sm[256][4];
    const BnBwdArgs& a = p.b;
    TO* __restrict__ gdz = reinterpret_cast<TO*>(a.dz);
    const TZ* __restrict__ gz = reinterpret_cast<const TZ*>(a.z);
    const int CL = 1 << p.cl_shift, RG = 256 >> p.cl_shift;
    const int lc = threadIdx.x & (CL - 1), rg = threadIdx.x >> p.cl_shift;
    const int c4 = blockIdx.x * CL + lc, c = c4 * 4, C = a.C4 * 4;
    const int m0 = blockIdx.y * p.rows_per_chunk, m1 = min(p.M, m0 + p.rows_per_chunk);
    const float invM = 1.f / (float)a.M;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c4 < a.C4) {
        float mu[4], is[4], g[4], s1[4], s2[4];
        vp_load4(a.mean + c, mu); vp_load4(a.invstd + c, is); vp_load4(a.sums + c, s1); vp_load4(a.sums + C + c, s2);
        if (a.gamma) vp_load4(a.gamma + c, g); else { g[0] = g[1] = g[2] = g[3] = 1.f; }
        float xsc[4] = {0.f, 0.f, 0.f, 0.f}, xsh[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (UTT == 2) { vp_load4(a.xsc + c, xsc); vp_load4(a.xsh + c, xsh); }
        auto one = [&](const float* dy, const float* z, float* o) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float zh = (z[e] - mu[e]) * is[e];
                const float v = g[e] * is[e] * (dy[e] - s1[e] * invM - zh * s2[e] * invM);
                o[e] = (a.relu_mask && !(z[e] > 0.f)) ? 0.f : v;
                acc[e] += o[e];
            }
        };
        int m = m0 + rg;
        constexpr int U = UTT ? 4 : VP_BNBWD_ROWS;               // (rows in flight per thread: see col_sums4_kernel)
        for (; m + (U - 1) * RG < m1; m += U * RG) {
            float dy[U][4], z[U][4], o[4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vp_load4(a.dy + (size_t)(m + u * RG) * a.lddy + c, dy[u]);
                vp_load4(gz + (size_t)(m + u * RG) * a.ldz + c, z[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                utt_affine4<UTT>(dy[u], a.us, a.um, a.T, a.inv_t, C, m + u * RG, c);
                ctx_affine4<UTT>(dy[u], z[u], a.us, a.um, xsc, xsh, a.T, C, m + u * RG, c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { one(dy[u], z[u], o); vp_store4(gdz + (size_t)(m + u * RG) * a.lddz + c, o); }
        }
        for (; m < m1; m += RG) {
            float dy[4], z[4], o[4];
            vp_load4(a.dy + (size_t)m * a.lddy + c, dy); vp_load4(gz + (size_t)m * a.ldz + c, z);
            utt_affine4<UTT>(dy, a.us, a.um, a.T, a.inv_t, C, m, c);
            ctx_affine4<UTT>(dy, z, a.us, a.um, xsc, xsh, a.T, C, m, c);
            one(dy, z, o);
            vp_store4(gdz + (size_t)m * a.lddz + c, o);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (rg == 0 && c4 < a.C4) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += sm[r * CL + lc][e];
        vp_store4(p.part + (size_t)blockIdx.y * C + c, t);
    }
}

// ---------------------------------------------------------------------------------------------- Adam (coupled L2)
// paddle.optimizer.Adam(weight_decay=L2Decay-style float): g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
// keep = 1 - lr * coeff for paddle.optimizer.AdamW's DECOUPLED decay (the parameter shrinks first, the moments never see the decay);
// 1 for Adam.
// fault: the context's grid-barrier bail-out word (csrc/res2_train.hip).  A fused training kernel whose grid barrier timed out has
// produced statistics from incomplete sums: the update of that step is DROPPED on the device -- the host polls the word only every
// few steps (a D2H sync per step would stall the launch pipeline) and must not find the bad step in the weights or Adam's moments.
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                                                   float eps, float wd, float c1, float c2, float gscale, float keep, const unsigned* fault) {
    if (fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pv = p[i];
        const float gr = g[i] * gscale + wd * pv;
        const float mn = b1 * m[i] + (1.f - b1) * gr;
        const float vn = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mn; v[i] = vn;
        p[i] = pv * keep - lr * (mn / c1) / (sqrtf(vn / c2) + eps);
    }
}

// paddle.optimizer.Momentum (and SGD = momentum 0): g += wd * p (L2Decay-style float);  vel = mu * vel + g;
// p -= lr * vel, or with use_nesterov p -= lr * (g + mu * vel)
__global__ __launch_bounds__(256) void momentum_kernel(float* p, const float* g, float* vel, long long n, float lr, float mu, float wd,
                                                       float gscale, int nesterov, const unsigned* fault) {
    if (fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pv = p[i];
        const float gr = g[i] * gscale + wd * pv;
        const float vn = mu * vel[i] + gr;
        vel[i] = vn;
        p[i] = pv - lr * (nesterov ? gr + mu * vn : vn);
    }
}

// ---------------------------------------------------------------------------------------------- gradient packing
// dst[off_i .. off_i + n_i) = src_i (or zeros when src_i is NULL) for up to 128 segments per launch: the parameters' gradient
// tensors, as autograd left them, into the optimiser's flat gradient buffer.  (With .grad bound to views of the flat buffer
// autograd accumulates in place: one add_ launch per parameter, 148 per ECAPA step -- a seventh of all launches.)
constexpr int PACK_MAX = 128;
struct PackArgs { const float* src[PACK_MAX]; long long off[PACK_MAX]; long long n[PACK_MAX]; float* dst; };

__global__ __launch_bounds__(256) void pack_segments_kernel(const PackArgs a) {
    const int sgm = blockIdx.y;
    const float* __restrict__ src = a.src[sgm];
    float* __restrict__ dst = a.dst + a.off[sgm];
    const long long n = a.n[sgm];
    // 16 bytes per lane where the segment allows it (the 9.4 MB MFA weight gradient took 63 us at 4 bytes per lane on 64 workgroups)
    if (src && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src ? src[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------- per-utterance pieces (ASP)
// out[b][c] = sum_t a[b, t, c]   (gradient of a per-utterance bias; one workgroup = 64 channels of one utterance)
template <typename TA = float>
__global__ __launch_bounds__(256) void utt_sums_kernel(const TA* a, int lda, int T, int C, float* out) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    float s = 0.f;
    if (c < C)
        for (int t = rg; t < T; t += 4) s += vp_to_f32(a[((size_t)b * T + t) * lda + c]);
    sm[rg][lc] = s;
    __syncthreads();
    if (rg == 0 && c < C) out[(size_t)b * C + c] = sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc];
}

// Backward of stats[b] = [mean_t x | sqrt(max(var_biased_t x, eps))] (pooling.py:97-104 with a mask of ones):
// dx[b,t,c] = dmean / T + [var > eps] * dstd / std * (x - mean) / T
struct TsBwdArgs { const float* x; const float* stats; const float* dstats; float* dx; int ldx, lddx, T, C4; float eps; long long total; int unbiased;
                   const float* add; int ldadd; };     // add: other gradients of x summed in the same pass (may alias dx)
template <typename TX = float>
__global__ __launch_bounds__(256) void time_stats_bwd_kernel(TsBwdArgs a) {
    const int C = a.C4 * 4;
    const float invT = 1.f / (float)a.T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long m = i / a.C4;
        const int c = (int)(i - m * a.C4) * 4;
        const long long b = m / a.T;
        float x[4], mu[4], sd[4], dm[4], ds[4], o[4];
        vp_load4(reinterpret_cast<const TX*>(a.x) + m * a.ldx + c, x);
        vp_load4(a.stats + b * 2 * C + c, mu); vp_load4(a.stats + b * 2 * C + C + c, sd);
        vp_load4(a.dstats + b * 2 * C + c, dm); vp_load4(a.dstats + b * 2 * C + C + c, ds);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            // sd = sqrt(var_unbiased + eps).  eps = 0 (CAM++'s statistics pooling, campplus.py:24-30) and a channel that is constant over an
            // utterance's frames (a ReLU output that is zero throughout) give sd = 0: the term is 0 / 0.  Its limit -- and what autograd
            // frameworks return for d sqrt(var) at var = 0 (torch masks it; round 5: this NaN reached Adam ~200 steps into a CAM++ run and
            // killed the model, tools/train_dynamics_ab.py) -- is no contribution from the std
            o[e] = a.unbiased ? dm[e] * invT + (sd[e] > 0.f ? ds[e] / sd[e] * (x[e] - mu[e]) / (float)(a.T > 1 ? a.T - 1 : 1) : 0.f)
                              : dm[e] * invT + ((sd[e] * sd[e] > a.eps) ? ds[e] / sd[e] * (x[e] - mu[e]) * invT : 0.f);
        if (a.add) {
            float ad[4];
            vp_load4(a.add + m * a.ldadd + c, ad);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += ad[e];
        }
        vp_store4(a.dx + m * a.lddx + c, o);
    }
}

// The same gradient as per-utterance coefficients: dx[b,t,c] = alpha[b,c] + beta[b,c] * x[b,t,c] with beta = [var > eps] dstd / (std T),
// alpha = dmean / T - beta * mean -- what the BatchNorm-backward passes of the producing layer add to their d y on the fly
// (ColSum4Args UTT == 2) instead of a pass over the (B*T, C) tensor.  ab: [2][B][C]
__global__ __launch_bounds__(256) void time_stats_bwd_coeffs_kernel(const float* stats, const float* dstats, int B, int C, int T, float eps, float* ab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float invT = 1.f / (float)T;
    const float mu = stats[(size_t)b * 2 * C + c], sd = stats[(size_t)b * 2 * C + C + c];
    const float dm = dstats[(size_t)b * 2 * C + c], ds = dstats[(size_t)b * 2 * C + C + c];
    const float be = (sd * sd > eps) ? ds / sd * invT : 0.f;
    ab[i] = __fmaf_rn(-be, mu, dm * invT);
    ab[(size_t)B * C + i] = be;
}

// Backward of attentive statistics (pooling.py:114-123): alpha = softmax_t(e), mu = sum alpha x, sd = sqrt(max(sum alpha (x-mu)^2, eps)).
//   dv = [sd^2 > eps] dsd / (2 sd);  dalpha_t = dmu x_t + dv (x_t - mu)^2;  S = sum_t alpha_t dalpha_t
//   de_t = alpha_t (dalpha_t - S);   dx_t = alpha_t (dmu + 2 dv (x_t - mu))
// One workgroup = 64 channels of one utterance, three passes over its frames (max; normaliser and S; outputs).
struct AsBwdArgs { const float* e; const float* x; const float* pooled; const float* dpooled; float* de; float* dx; int ldx, lddx, T, C; float eps; };
// TE = bf16_t: d e leaves as bf16 (a.de reinterpreted) -- the operand the logits conv's two backward GEMMs would round it to anyway
template <typename TE>
__global__ __launch_bounds__(256) void attn_stats_bwd_kernel(AsBwdArgs a) {
    __shared__ float sm[2][4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const float* eb = a.e + (size_t)b * a.T * a.C + cc;
    const float* xb = a.x + (size_t)b * a.T * a.ldx + cc;
    const float mu = a.pooled[(size_t)b * 2 * a.C + cc], sd = a.pooled[(size_t)b * 2 * a.C + a.C + cc];
    const float dmu = a.dpooled[(size_t)b * 2 * a.C + cc], dsd = a.dpooled[(size_t)b * 2 * a.C + a.C + cc];
    const float dv = (sd * sd > a.eps) ? dsd / (2.f * sd) : 0.f;
    float mx = -INFINITY;
    for (int t = rg; t < a.T; t += 4) mx = fmaxf(mx, eb[(size_t)t * a.C]);
    sm[0][rg][lc] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0][0][lc], sm[0][1][lc]), fmaxf(sm[0][2][lc], sm[0][3][lc]));
    __syncthreads();
    float z = 0.f, u = 0.f;
    for (int t = rg; t < a.T; t += 4) {
        const float p = expf(eb[(size_t)t * a.C] - mx);
        const float xv = xb[(size_t)t * a.ldx], d = xv - mu;
        z += p; u += p * (dmu * xv + dv * d * d);
    }
    sm[0][rg][lc] = z; sm[1][rg][lc] = u;
    __syncthreads();
    z = sm[0][0][lc] + sm[0][1][lc] + sm[0][2][lc] + sm[0][3][lc];
    u = sm[1][0][lc] + sm[1][1][lc] + sm[1][2][lc] + sm[1][3][lc];
    const float S = u / z;
    if (!ok) return;
    for (int t = rg; t < a.T; t += 4) {
        const float al = expf(eb[(size_t)t * a.C] - mx) / z;
        const float xv = xb[(size_t)t * a.ldx], d = xv - mu;
        reinterpret_cast<TE*>(a.de)[((size_t)b * a.T + t) * a.C + c] = (TE)(al * (dmu * xv + dv * d * d - S));
        a.dx[((size_t)b * a.T + t) * a.lddx + c] = al * (dmu + 2.f * dv * d);
    }
}

// The same with the thread's share of e and x kept in REGISTERS between the passes: one workgroup = 64 channels of one utterance,
// 512 threads = 8 frame groups, NT = ceil(T / 8) <= 40 values of each per thread (T <= 320).  The plain kernel above re-reads both
// slices from HBM in its second and third pass (PMC: 2.35 GB read per launch at B = 256 for 0.94 GB of inputs -- 552 us, the largest
// kernel of the backward pass bar the weight gradients); here every input byte is read once.  (An LDS-resident variant -- 76 KB per
// workgroup, two per CU -- measured SLOWER than the plain kernel: 8 waves per CU cannot keep enough loads in flight.)
template <typename TE, int NT, typename TL = float, typename TX = float>
__global__ __launch_bounds__(512) void attn_stats_bwd_reg_kernel(AsBwdArgs a) {
    __shared__ float sm[2][8][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    const bool ok = c < a.C;
    const int cc = ok ? c : 0;
    const TL* eb = reinterpret_cast<const TL*>(a.e) + (size_t)b * a.T * a.C + cc;        // TL = bf16: the logits as the forward stored them
    const TX* xb = reinterpret_cast<const TX*>(a.x) + (size_t)b * a.T * a.ldx + cc;     // TX = bf16: x as its producer stored it
    const float mu = a.pooled[(size_t)b * 2 * a.C + cc], sd = a.pooled[(size_t)b * 2 * a.C + a.C + cc];
    const float dmu = a.dpooled[(size_t)b * 2 * a.C + cc], dsd = a.dpooled[(size_t)b * 2 * a.C + a.C + cc];
    const float dv = (sd * sd > a.eps) ? dsd / (2.f * sd) : 0.f;
    float ev[NT], xv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {                  // (unconditional loads on a clamped frame; the uses below are predicated)
        const int t = min(rg + 8 * i, a.T - 1);
        ev[i] = vp_to_f32(eb[(size_t)t * a.C]);
        xv[i] = vp_to_f32(xb[(size_t)t * a.ldx]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NT; ++i) if (rg + 8 * i < a.T) mx = fmaxf(mx, ev[i]);
    sm[0][rg][lc] = mx;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) mx = fmaxf(mx, sm[0][q][lc]);
    __syncthreads();
    float z = 0.f, u = 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const float p = rg + 8 * i < a.T ? expf(ev[i] - mx) : 0.f;
        const float d = xv[i] - mu;
        ev[i] = p;                                  // exp(e - max): the third pass needs nothing else of e
        z += p; u += p * (dmu * xv[i] + dv * d * d);
    }
    sm[0][rg][lc] = z; sm[1][rg][lc] = u;
    __syncthreads();
    z = 0.f; u = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { z += sm[0][q][lc]; u += sm[1][q][lc]; }
    const float S = u / z;
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = rg + 8 * i;
        if (t < a.T) {
            const float al = ev[i] / z;
            const float d = xv[i] - mu;
            reinterpret_cast<TE*>(a.de)[((size_t)b * a.T + t) * a.C + c] = (TE)(al * (dmu * xv[i] + dv * d * d - S));
            a.dx[((size_t)b * a.T + t) * a.lddx + c] = al * (dmu + 2.f * dv * d);
        }
    }
}

template <typename TE>
static int launch_attn_stats_bwd(vp_ctx* ctx, const AsBwdArgs& a, int B, hipStream_t st) {
    (void)ctx;
    const dim3 grid((a.C + 63) / 64, B);
    static const bool plain = getenv("VPMI_ASB_PLAIN") != nullptr;      // A/B switch, read once per process
    if (a.T <= 160 && !plain) hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<TE, 20>), grid, dim3(512), 0, st, a);
    else if (a.T <= 320 && !plain) hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<TE, 40>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(attn_stats_bwd_kernel<TE>, grid, dim3(256), 0, st, a);
    return VP_OK;
}

// Adjoint of reflect padding: dxp (B, T + 2p, C) is the gradient w.r.t. the reflect-padded input (from the zero-padded
// "full" data-gradient conv); frames 1..p and T-1-p..T-2 also receive their mirror images' gradients.
struct FoldArgs { const float* dxp; float* dx; int T, p, C4; long long total;
                  int lddx; const float* add; int ld_add; float* sum; };     // dx rows lddx apart; sum = folded + add (dense), when add
__global__ __launch_bounds__(256) void reflect_fold_kernel(FoldArgs a) {
    const int Tp = a.T + 2 * a.p;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long m = i / a.C4;
        const int c = (int)(i - m * a.C4) * 4;
        const long long b = m / a.T;
        const int t = (int)(m - b * a.T);
        const float* base = a.dxp + (size_t)b * Tp * a.C4 * 4 + c;
        float v[4], w[4];
        vp_load4(base + (size_t)(t + a.p) * a.C4 * 4, v);
        if (t >= 1 && t <= a.p) {
            vp_load4(base + (size_t)(a.p - t) * a.C4 * 4, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += w[e];
        }
        if (t <= a.T - 2 && t >= a.T - 1 - a.p) {
            vp_load4(base + (size_t)(2 * (a.T - 1) - t + a.p) * a.C4 * 4, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += w[e];
        }
        vp_store4(a.dx + m * a.lddx + c, v);
        if (a.add) {
            vp_load4(a.add + m * a.ld_add + c, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += w[e];
            vp_store4(a.sum + m * a.C4 * 4 + c, v);
        }
    }
}

// Backward of out[b,t,c] = x[b,t,c] * s[b,c] (+ res): dx = dy * s;  ds[b,c] = sum_t dy * x.  One workgroup = 64 channels
// of one utterance.
__global__ __launch_bounds__(256) void scale_rows_bwd_kernel(const float* dy, const float* x, const float* s, int T, int C, float* dx,
                                                             float* ds) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    float acc = 0.f;
    if (c < C) {
        const float sv = s[(size_t)b * C + c];
        for (int t = rg; t < T; t += 4) {
            const size_t o = ((size_t)b * T + t) * C + c;
            const float g = dy[o];
            acc += g * x[o];
            dx[o] = g * sv;
        }
    }
    sm[rg][lc] = acc;
    __syncthreads();
    if (rg == 0 && c < C) ds[(size_t)b * C + c] = sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc];
}

// The same for utterances of MANY positions (ResNetSE / ERes2Net feature maps: T x F' = 10^4 positions of 32-256 channels): the kernel
// above gives one workgroup per (utterance, 64 channels) -- 32 workgroups walking 19 072 rows each on the stage-1 maps, 574 us per call
// and a quarter of the ResNetSE training step.  Here the positions are cut into chunks (grid = chunks x B), four channels per lane,
// partial sums per chunk reduced by sum_partials (fixed order).  part: [B][chunks][C].
struct SrbArgs { const float* dy; const float* x; const float* s; float* dx; float* part; int T, C4, chunks, rows_per_chunk, cl_shift; };
__global__ __launch_bounds__(256) void scale_rows_bwd_chunk_kernel(SrbArgs a) {
    __shared__ float sm[256][4];
    const int CL = 1 << a.cl_shift, RG = 256 >> a.cl_shift;
    const int lc = threadIdx.x & (CL - 1), rg = threadIdx.x >> a.cl_shift;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * a.rows_per_chunk, p1 = min(a.T, p0 + a.rows_per_chunk);
    const int C = a.C4 * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (lc < a.C4) {
        float sv[4];
        vp_load4(a.s + (size_t)b * C + 4 * lc, sv);
        int p = p0 + rg;
        for (; p + 3 * RG < p1; p += 4 * RG) {                   // four rows of loads in flight
            float g[4][4], xv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = ((size_t)b * a.T + p + u * RG) * C + 4 * lc;
                vp_load4(a.dy + o, g[u]); vp_load4(a.x + o, xv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += g[u][e] * xv[u][e]; o4[e] = g[u][e] * sv[e]; }
                vp_store4(a.dx + ((size_t)b * a.T + p + u * RG) * C + 4 * lc, o4);
            }
        }
        for (; p < p1; p += RG) {
            const size_t o = ((size_t)b * a.T + p) * C + 4 * lc;
            float g[4], xv[4], o4[4];
            vp_load4(a.dy + o, g); vp_load4(a.x + o, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += g[e] * xv[e]; o4[e] = g[e] * sv[e]; }
            vp_store4(a.dx + o, o4);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (rg == 0 && lc < a.C4) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += sm[q * CL + lc][e];
        vp_store4(a.part + ((size_t)b * a.chunks + ch) * C + 4 * lc, t);
    }
}

// The SE gate's backward in two passes (see SEBlockFn in train/functions.py): ds[b,c] = sum_t dy * x first -- the squeeze path's
// gradient dm[b,c] (through the two dense layers) depends on it -- then dx = dy * s[b,c] + dm[b,c] / T in one write of dx, instead
// of dx = dy * s, a separate mean-backward tensor and their sum.  Four channels per lane; 32 lanes x 8 frame groups per utterance.
// AFF: x is the PRE-BatchNorm activation z (bf16); the SE block's input h = bf16(z * bsc + bsh) is formed on the fly -- the same values
// the BatchNorm apply pass (vp_affine_rows_b16_b16) would have stored
template <bool AFF>
__device__ __forceinline__ void bn_affine4_b16(float (&v)[4], const float (&sc)[4], const float (&sh)[4]) {
    if constexpr (AFF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)(bf16_t)__fmaf_rn(v[e], sc[e], sh[e]);
    }
}

template <typename TX = float, bool AFF = false>
__global__ __launch_bounds__(256) void utt_dot4_kernel(const float* dy, const TX* x, int T, int C, float* ds, const float* bsc = nullptr,
                                                       const float* bsh = nullptr) {
    __shared__ float sm[256][4];
    const int lc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int b = blockIdx.y, c4 = blockIdx.x * 32 + lc;
    const bool ok = c4 < (C >> 2);
    const int c = ok ? c4 * 4 : 0;
    const float* gb = dy + (size_t)b * T * C + c;
    const TX* xb = x + (size_t)b * T * C + c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (AFF) { vp_load4(bsc + c, sc); vp_load4(bsh + c, sh); }
    int t = rg;
    for (; t + 24 < T; t += 32) {
        float g[4][4], v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { vp_load4(gb + (size_t)(t + 8 * u) * C, g[u]); vp_load4(xb + (size_t)(t + 8 * u) * C, v[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) bn_affine4_b16<AFF>(v[u], sc, sh);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += g[u][e] * v[u][e];
    }
    for (; t < T; t += 8) {
        float g[4], v[4];
        vp_load4(gb + (size_t)t * C, g); vp_load4(xb + (size_t)t * C, v);
        bn_affine4_b16<AFF>(v, sc, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += g[e] * v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (rg != 0 || !ok) return;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += sm[r * 32 + lc][e];
    vp_store4(ds + (size_t)b * C + c, o);
}

__global__ __launch_bounds__(256) void scale_shift_rows4_kernel(const float* dy, const float* s, const float* dm, int T, int C4, float inv_t,
                                                                long long total, float* dx) {
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        const long long b = m / T;
        float g[4], sv[4], d[4], o[4];
        vp_load4(dy + m * C + c, g); vp_load4(s + b * C + c, sv); vp_load4(dm + b * C + c, d);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fmaf_rn(g[e], sv[e], d[e] * inv_t);       // (the form utt_affine4 evaluates: bit-identical A/B)
        vp_store4(dx + m * C + c, o);
    }
}

// up[b, t, f, :] = dz[b, t / s, f / s, :] when t and f are multiples of s (and in range), else 0: the zero-insertion that turns
// the data gradient of a stride-s conv into a stride-1 conv over `up` (T_in x F_in positions).
struct ZiArgs { const float* dz; float* up; int T_out, F_out, T_in, F_in, st, sf, C4; long long total; };
__global__ __launch_bounds__(256) void zero_insert_kernel(ZiArgs a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long pos = i / a.C4;
        const int c = (int)(i - pos * a.C4) * 4;
        const long long bt = pos / a.F_in;
        const int f = (int)(pos - bt * a.F_in);
        const long long b = bt / a.T_in;
        const int t = (int)(bt - b * a.T_in);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (t % a.st == 0 && f % a.sf == 0 && t / a.st < a.T_out && f / a.sf < a.F_out)
            vp_load4(a.dz + (((size_t)b * a.T_out + t / a.st) * a.F_out + f / a.sf) * a.C4 * 4 + c, v);
        vp_store4(a.up + pos * a.C4 * 4 + c, v);
    }
}

// dz = [y > 0] * dy   (ReLU backward from its output)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* dy, const float* y, long long n4, float* dz) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float g[4], v[4];
        vp_load4(dy + i * 4, g); vp_load4(y + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
        vp_store4(dz + i * 4, g);
    }
}

// Backward of the AFF output o = x (1 + t) + y (1 - t) (eres2net.py:48-51, t = tanh(local_att)): dx = g (1 + t), dy = g (1 - t), dt = g (x - y)
__global__ __launch_bounds__(256) void aff_combine_bwd_kernel(const float* g, const float* t, const float* x, const float* y, long long n4,
                                                              float* dx, float* dy, float* dt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float gv[4], tv[4], xv[4], yv[4], a[4], b[4], c[4];
        vp_load4(g + i * 4, gv); vp_load4(t + i * 4, tv); vp_load4(x + i * 4, xv); vp_load4(y + i * 4, yv);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = gv[e] * (1.f + tv[e]); b[e] = gv[e] * (1.f - tv[e]); c[e] = gv[e] * (xv[e] - yv[e]); }
        vp_store4(dx + i * 4, a); vp_store4(dy + i * 4, b); vp_store4(dt + i * 4, c);
    }
}

// ---------------------------------------------------------------------------------------------- CAM++ context (campplus.py:88-106)
// ctx[b, s, c] = mean_t x[b, t, c] + mean_{t in segment s} x[b, t, c]  (100-frame segments, the last one over its valid frames);
// backward: dx[b, t, c] = sum_s dctx[b, s, c] / T + dctx[b, seg(t), c] / len(seg(t)).  One workgroup = 64 channels of one utterance.
__global__ __launch_bounds__(256) void seg_ctx_kernel(const float* x, int T, int C, int seg_len, int nseg, float* ctx) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    const bool ok = c < C;
    float tot = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const int t0 = s * seg_len, t1 = min(T, t0 + seg_len);
        float acc = 0.f;
        if (ok) for (int t = t0 + rg; t < t1; t += 4) acc += x[((size_t)b * T + t) * C + c];
        sm[rg][lc] = acc;
        __syncthreads();
        const float ssum = sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc];
        __syncthreads();
        tot += ssum;
        if (rg == 0 && ok) ctx[((size_t)b * nseg + s) * C + c] = ssum / (float)(t1 - t0);
    }
    if (rg == 0 && ok)
        for (int s = 0; s < nseg; ++s) ctx[((size_t)b * nseg + s) * C + c] += tot / (float)T;
}

__global__ __launch_bounds__(256) void seg_ctx_bwd_kernel(const float* dctx, int T, int C, int seg_len, int nseg, float* dx) {
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    if (c >= C) return;
    float tot = 0.f;
    for (int s = 0; s < nseg; ++s) tot += dctx[((size_t)b * nseg + s) * C + c];
    tot /= (float)T;
    for (int t = rg; t < T; t += 4) {
        const int s = t / seg_len;
        const int len = min(T, (s + 1) * seg_len) - s * seg_len;
        dx[((size_t)b * T + t) * C + c] = tot + dctx[((size_t)b * nseg + s) * C + c] / (float)len;
    }
}

// out[b, t, c] = y[b, t, c] * m[b, seg(t), c];  backward: dy = g * m,  dm[b, s, c] = sum_{t in s} g * y
__global__ __launch_bounds__(256) void seg_scale_kernel(const float* y, const float* m, int T, int C4, int seg_len, int nseg, long long total,
                                                        float* out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C4;
        const int c = (int)(i - r * C4) * 4;
        const long long b = r / T;
        const int t = (int)(r - b * T);
        float v[4], g[4];
        vp_load4(y + r * C4 * 4 + c, v);
        vp_load4(m + (b * nseg + t / seg_len) * C4 * 4 + c, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= g[e];
        vp_store4(out + r * C4 * 4 + c, v);
    }
}

__global__ __launch_bounds__(256) void seg_scale_bwd_kernel(const float* g, const float* y, const float* m, int T, int C, int seg_len, int nseg,
                                                            float* dy, float* dm) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + lc;
    const bool ok = c < C;
    for (int s = 0; s < nseg; ++s) {
        const int t0 = s * seg_len, t1 = min(T, t0 + seg_len);
        const float mv = ok ? m[((size_t)b * nseg + s) * C + c] : 0.f;
        float acc = 0.f;
        if (ok)
            for (int t = t0 + rg; t < t1; t += 4) {
                const size_t o = ((size_t)b * T + t) * C + c;
                const float gv = g[o];
                acc += gv * y[o];
                dy[o] = gv * mv;
            }
        sm[rg][lc] = acc;
        __syncthreads();
        if (rg == 0 && ok) dm[((size_t)b * nseg + s) * C + c] = sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc];
        __syncthreads();
    }
}

// dz = dy * (1 - y^2)   (tanh backward from its output)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* dy, const float* y, long long n4, float* dz, int sigmoid) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float g[4], v[4];
        vp_load4(dy + i * 4, g); vp_load4(y + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d;
            if (sigmoid == 2) d = v[e] > 0.f ? 1.f : 0.f;
            else if (sigmoid == 3) d = (v[e] > 0.f && v[e] < 20.f) ? 1.f : 0.f;
            else if (sigmoid == 4) { const float sg = 1.f / (1.f + expf(-v[e])); d = sg * (1.f + v[e] * (1.f - sg)); }   // v = the INPUT
            else d = sigmoid ? v[e] * (1.f - v[e]) : 1.f - v[e] * v[e];
            g[e] *= d;
        }
        vp_store4(dz + i * 4, g);
    }
}
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* x, long long n4, float* y, int sigmoid) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float v[4];
        vp_load4(x + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (sigmoid == 2) v[e] = fmaxf(v[e], 0.f);
            else if (sigmoid == 3) v[e] = fminf(fmaxf(v[e], 0.f), 20.f);
            else if (sigmoid == 4) v[e] = v[e] / (1.f + expf(-v[e]));
            else v[e] = sigmoid ? 1.f / (1.f + expf(-v[e])) : tanhf(v[e]);
        }
        vp_store4(y + i * 4, v);
    }
}

int act_kind(int act) {
    return act == VP_ACT_SIGMOID ? 1 : act == VP_ACT_RELU ? 2 : act == VP_ACT_HARDTANH20 ? 3 : act == VP_ACT_SILU ? 4 : 0;
}

unsigned grid1d(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b > 256 * 64 ? 256 * 64 : (b < 1 ? 1 : b));
}

}  // namespace

// many outputs: one thread per output (coalesced over the outputs); few outputs, many partials: 16 x 16 per workgroup
static bool getenv_once(const char* name) {       // A/B switches: read at first use (the few names used are distinct call sites)
    static const bool v = getenv(name) != nullptr;
    return v;
}

static void launch_sum_partials(const float* part, int S, long long n, float* out, hipStream_t st, int Cin = 1, int KW = 1, int batch = 1,
                                int KF = 1) {
    if (n >= 32768) hipLaunchKernelGGL(sum_partials_wide_kernel, dim3((unsigned)((n + 63) / 64), batch), dim3(256), 0, st, part, S, n, out, Cin, KW, KF);
    else hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 15) / 16), batch), dim3(256), 0, st, part, S, n, out, Cin, KW, KF);
}

extern "C" {

size_t vp_conv1d_wgrad_workspace_bytes(const vp_conv1d_desc* d) {
    if (!d || d->Cout <= 0 || d->KW <= 0 || d->Cin <= 0) return 0;
    const long long M = (long long)d->B * d->T_out * ((d->KF > 1 || d->F_in > 1 || d->F_out > 1) ? d->F_out : 1);
    const int K = d->KW * d->Cin;
    const int tiles = ((d->Cout + 63) / 64) * ((K + 63) / 64);
    int S = 2048 / tiles;
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    if ((long long)S * 64 > M) S = (int)((M + 63) / 64);
    // the 256-tile kernel of the wide 1x1 layers (wgrad_tr.hip) splits the rows differently
    if (d->KW == 1 && (d->Cout % 256 == 0 || d->Cout == 128) && (K % 256 == 0 || K == 128) && M >= 32) {
        const int s2 = vp_wgrad_tr256_splits(M, d->Cout, K);
        if (s2 > S) S = s2;
    }
    return (size_t)S * d->Cout * K * sizeof(float) + 256;
}

// d: the FORWARD conv's descriptor (x / ldx / xoff and the geometry; w, y, epilogue fields ignored).  dz (B*T_out, lddz) f32.
static int wgrad_impl(vp_ctx* ctx, const vp_conv1d_desc* d, const float* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                      vp_stream stream, bool oik, int nbatch = 1, long long x_bstride = 0, long long dz_bstride = 0) {
    if (!ctx || !d || !d->x || !dz || !dW) VP_FAIL(ctx, VP_EINVAL, "wgrad: null argument");
    const bool bf_in = d->dtype_in == VP_BF16;          // x AND dz bf16 in memory (vp_conv1d_wgrad_bf16_oik): bf16 matrix cores only
    if (d->dtype_in != VP_F32 && !bf_in) VP_FAIL(ctx, VP_EUNSUP, "wgrad: f32 or bf16 tensors");
    const bool two_d = d->KF > 1 || d->F_in > 1 || d->F_out > 1;
    if (two_d && (d->KF < 1 || d->KW % d->KF || d->F_in < 1 || d->F_out < 1 || d->stride_f < 1 || d->pad_mode != VP_PAD_ZERO))
        VP_FAIL(ctx, VP_EINVAL, "wgrad: bad 2-D geometry (zero padding only)");
    if (d->B <= 0 || d->T_in <= 0 || d->T_out <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KW <= 0 || d->stride <= 0 || d->dilation <= 0)
        VP_FAIL(ctx, VP_EINVAL, "wgrad: bad shape");
    const size_t need = vp_conv1d_wgrad_workspace_bytes(d) * (size_t)nbatch;
    if (!ws || ws_bytes < need) VP_FAIL(ctx, VP_EWORKSPACE, "wgrad: workspace %zu < %zu", ws_bytes, need);
    if (nbatch < 1 || (nbatch > 1 && !(d->dtype_in == VP_BF16))) VP_FAIL(ctx, VP_EINVAL, "wgrad: batched launches take bf16 operands");
    const long long M = (long long)d->B * d->T_out * (two_d ? d->F_out : 1);
    if (M > 0x7fffffffLL / 2) VP_FAIL(ctx, VP_EINVAL, "wgrad: too many rows");
    const int K = d->KW * d->Cin;
    const int tn = (d->Cout + 63) / 64, tk = (K + 63) / 64;
    int S = 2048 / (tn * tk);
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    if (nbatch > 1 && S > 512 / nbatch) S = 512 / nbatch > 1 ? 512 / nbatch : 1;      // the batch fills the chip: fewer, longer row splits (less to reduce)
    if ((long long)S * 64 > M) S = (int)((M + 63) / 64);
    const bool x3 = d->mfma_bf16 == 2 && !bf_in;       // split precision (f32 operands): the 128 x 128 kernel, one workgroup per CU
    if ((d->mfma_bf16 || bf_in) && nbatch == 1) {
        // The 128 x 128 mixed-precision kernel keeps two workgroups per CU resident.  A grid of 1.5 rounds of them costs two rounds of
        // time with half-empty CUs in the second (a 3 x 3 conv of 32 channels: 3 tiles x 256 splits = 768 workgroups on 512 slots): take the
        // largest split count that fills WHOLE rounds instead (3 x 170 = 510: the same rows in one round, and a third fewer partial rows).
        static int cus_dev[64] = {};
        int& cus = cus_dev[ctx->device & 63];
        if (!cus && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) cus = 256;
        const long long slots = (x3 ? 1LL : 2LL) * cus;
        const long long tiles128 = (long long)((K + 127) / 128) * ((d->Cout + 127) / 128);
        const long long W = tiles128 * S;
        if (W > slots) {
            const long long S2 = slots * (W / slots) / tiles128;
            if (S2 >= 1 && S2 < S) S = (int)S2;
        }
    }
    if ((d->Cin | d->Cout | d->ldx | d->xoff | lddz) & 3) VP_FAIL(ctx, VP_EINVAL, "wgrad: Cin / Cout / ldx / xoff / lddz must be multiples of 4");
    int rps = (int)((M + S - 1) / S);
    const int rq = (d->mfma_bf16 || bf_in) ? 64 : 32;             // rows per staged chunk of the kernel
    rps = (rps + rq - 1) / rq * rq;
    S = (int)((M + rps - 1) / rps);
    WgradArgs a;
    a.x = (const float*)d->x; a.dz = dz; a.part = (float*)ws;
    a.ldx = d->ldx; a.xoff = d->xoff; a.lddz = lddz; a.M = (int)M; a.N = d->Cout; a.K = K; a.Cin = d->Cin;
    a.T_in = d->T_in; a.T_out = d->T_out; a.dilation = d->dilation; a.stride = d->stride; a.pad_left = d->pad_left;
    a.pad_mode = d->pad_mode; a.rows_per_split = rps;
    a.F_in = two_d ? d->F_in : 1; a.F_out = two_d ? d->F_out : 1; a.KF = two_d ? d->KF : 1;
    a.stride_f = two_d ? d->stride_f : 1; a.pad_f = two_d ? d->pad_f : 0;
    a.splits = S; a.xb = x_bstride; a.dzb = dz_bstride;
    hipStream_t st = (hipStream_t)stream;
    static const bool no_tr256 = getenv("VPMI_WGRAD_TR256_OFF") != nullptr;      // A/B switch (tools/train_probe.py)
    const bool simple = K == d->Cin && d->stride == 1 && d->pad_left == 0 && d->T_in == d->T_out && !two_d;
    int tr = VP_EUNSUP;
    if (bf_in && simple && nbatch == 1 && !no_tr256) {
        int s2 = 0;
        tr = vp_wgrad_tr256_bf16(ctx, d->x, d->ldx, d->xoff, dz, lddz, M, d->Cout, K, (float*)ws, &s2, st);
        if (tr == VP_OK) S = s2;
        else if (tr != VP_EUNSUP) return tr;
    }
    if (tr == VP_OK) {
    } else if (d->mfma_bf16 || bf_in) {
        constexpr int smem = 2 * 2 * WA_T * 128;
        static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
        if (!attr_set) {
            VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_amp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_amp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        const dim3 grid((K + WA_T - 1) / WA_T, (d->Cout + WA_T - 1) / WA_T, S * nbatch);
        // narrow layer with bf16 operands in memory (the batched Res2Net chunk convs): its own tile shape.  The f32-operand flavour of the
        // kernel (2-D taps too: the 32- / 64-channel convs of ResNetSE / ERes2Net / the FCM head) is built but NOT dispatched: measured
        // slower than the square tile there (ResNetSE step 22.7 -> 23.8 ms, ERes2Net 31.4 -> 33.9 ms, CAM++ 30.1 -> 31.8 ms;
        // VPMI_WGRAD_NARROW_F32=1 dispatches it for study)
        if (!x3 && d->Cout <= 64 && K > 128 && ((bf_in && !two_d && d->stride == 1) || (!bf_in && getenv_once("VPMI_WGRAD_NARROW_F32")))) {
            constexpr int smem64 = 2 * (64 + 256) * 128;
            static bool attr64_dev[64] = {};
            bool& attr64 = attr64_dev[ctx->device & 63];
            if (!attr64) {
                VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_n64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem64));
                VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_n64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem64));
                attr64 = true;
            }
            const dim3 g64((K + 255) / 256, 1, S * nbatch);
            if (bf_in) hipLaunchKernelGGL(conv_wgrad_n64_kernel<true>, g64, dim3(256), smem64, st, a);
            else hipLaunchKernelGGL(conv_wgrad_n64_kernel<false>, g64, dim3(256), smem64, st, a);
        } else if (bf_in) hipLaunchKernelGGL(conv_wgrad_amp_kernel<true>, grid, dim3(256), smem, st, a);
        else if (x3) {
            static bool attr3_dev[64] = {};
            bool& attr3 = attr3_dev[ctx->device & 63];
            if (!attr3) {
                VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_amp_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem));
                attr3 = true;
            }
            hipLaunchKernelGGL((conv_wgrad_amp_kernel<false, true>), grid, dim3(256), 2 * smem, st, a);
        } else hipLaunchKernelGGL(conv_wgrad_amp_kernel<false>, grid, dim3(256), smem, st, a);
    } else {
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tk, tn, S), dim3(256), 0, st, a);
    }
    VP_LAUNCH_CHECK(ctx, "conv_wgrad");
    const long long n = (long long)d->Cout * K;
    launch_sum_partials((const float*)ws, S, n, dW, st, d->Cin, oik ? d->KW : 1, nbatch, (oik && two_d) ? d->KF : 1);
    VP_LAUNCH_CHECK(ctx, "wgrad_reduce");
    return VP_OK;
}

int vp_conv1d_wgrad_f32(vp_ctx* ctx, const vp_conv1d_desc* d, const float* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                        vp_stream stream) {
    return wgrad_impl(ctx, d, dz, lddz, dW, ws, ws_bytes, stream, false);
}

// the same gradient in the model's own (Cout, Cin, KW) layout (KW = KT * KF taps for the 2-D convs): no permute copy afterwards
int vp_conv1d_wgrad_oik_f32(vp_ctx* ctx, const vp_conv1d_desc* d, const float* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                            vp_stream stream) {
    return wgrad_impl(ctx, d, dz, lddz, dW, ws, ws_bytes, stream, true);
}

// x (d->x, d->dtype_in == VP_BF16) and dz both bf16 in memory; dW f32 in the model's (Cout, Cin, KW) layout
int vp_conv1d_wgrad_bf16_oik(vp_ctx* ctx, const vp_conv1d_desc* d, const void* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                             vp_stream stream) {
    if (!d || d->dtype_in != VP_BF16) VP_FAIL(ctx, VP_EINVAL, "wgrad_bf16: the descriptor's dtype_in must be bf16");
    return wgrad_impl(ctx, d, (const float*)dz, lddz, dW, ws, ws_bytes, stream, true);
}

// nbatch convs of identical geometry in ONE launch: conv c reads x + c * x_bstride and dz + c * dz_bstride (bf16 elements) and writes
// dW + c * Cout * Cin * KW; ws = nbatch x vp_conv1d_wgrad_workspace_bytes(d).  (The seven chunk convs of a Res2Net block.)
int vp_conv1d_wgrad_bf16_oik_batched(vp_ctx* ctx, const vp_conv1d_desc* d, const void* dz, int lddz, float* dW, int nbatch, long long x_bstride,
                                     long long dz_bstride, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!d || d->dtype_in != VP_BF16) VP_FAIL(ctx, VP_EINVAL, "wgrad_bf16: the descriptor's dtype_in must be bf16");
    return wgrad_impl(ctx, d, (const float*)dz, lddz, dW, ws, ws_bytes, stream, true, nbatch, x_bstride, dz_bstride);
}

int vp_conv_weight_layouts_f32(vp_ctx* ctx, const float* w, int Cout, int Cin, int KW, float* wp, float* w2, vp_stream stream) {
    if (!ctx || !w || (!wp && !w2) || Cout <= 0 || Cin <= 0 || KW <= 0) VP_FAIL(ctx, VP_EINVAL, "weight_layouts: bad arguments");
    hipLaunchKernelGGL(weight_layouts_kernel, dim3(grid1d((long long)Cout * Cin * KW)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KW, 1,
                       wp, w2);
    VP_LAUNCH_CHECK(ctx, "weight_layouts");
    return VP_OK;
}

int vp_conv2d_weight_layouts_f32(vp_ctx* ctx, const float* w, int Cout, int Cin, int KF, int KT, float* wp, float* w2, vp_stream stream) {
    if (!ctx || !w || (!wp && !w2) || Cout <= 0 || Cin <= 0 || KF <= 0 || KT <= 0) VP_FAIL(ctx, VP_EINVAL, "weight_layouts_2d: bad arguments");
    hipLaunchKernelGGL(weight_layouts_kernel, dim3(grid1d((long long)Cout * Cin * KF * KT)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                       KF * KT, KF, wp, w2);
    VP_LAUNCH_CHECK(ctx, "weight_layouts_2d");
    return VP_OK;
}

int vp_prep_weights_bf16(vp_ctx* ctx, const void* const* w, void* const* w16, void* const* wt16, const int* rows, const int* cols,
                         const int* ld, int n, vp_stream stream) {
    if (!ctx || !w || !w16 || !wt16 || !rows || !cols || !ld || n <= 0) VP_FAIL(ctx, VP_EINVAL, "prep_weights: bad arguments");
    for (int i0 = 0; i0 < n; i0 += WPREP_MAX) {
        WPrepArgs a;
        memset(&a, 0, sizeof(a));
        const int m = n - i0 < WPREP_MAX ? n - i0 : WPREP_MAX;
        int tiles = 0;
        for (int i = 0; i < m; ++i) {
            const int k = i0 + i;
            if (!w[k] || (!w16[k] && !wt16[k]) || rows[k] <= 0 || cols[k] <= 0 || ld[k] < cols[k]) VP_FAIL(ctx, VP_EINVAL, "prep_weights: bad entry %d", k);
            a.w[i] = (const float*)w[k]; a.w16[i] = (bf16_t*)w16[k]; a.wt16[i] = (bf16_t*)wt16[k];
            a.rows[i] = rows[k]; a.cols[i] = cols[k]; a.ld[i] = ld[k]; a.tile0[i] = tiles;
            tiles += ((rows[k] + 63) / 64) * ((cols[k] + 63) / 64);
        }
        a.tile0[m] = tiles; a.n = m;
        hipLaunchKernelGGL(weight_prep_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
        VP_LAUNCH_CHECK(ctx, "weight_prep");
    }
    return VP_OK;
}

// rows per chunk / chunks / lane split of the four-channels-per-lane kernels: ~1024 workgroups, at most 512 chunks
static void colsum4_geometry(long long M, int C4, int& cl_shift, int& colblocks, int& rpc, int& chunks) {
    cl_shift = C4 >= 64 ? 6 : (C4 >= 32 ? 5 : 4);
    const int CL = 1 << cl_shift, RG = 256 >> cl_shift;
    colblocks = (C4 + CL - 1) / CL;
    long long ch = 1024 / colblocks;              // (more chunks cost more in the partial-sum reduce than they gain here)
    if (ch < 1) ch = 1;
    if (ch > 512) ch = 512;
    long long r = (M + ch - 1) / ch;
    if (r < 4 * RG) r = 4 * RG;
    rpc = (int)r;
    chunks = (int)((M + r - 1) / r);
}

size_t vp_col_sums_workspace_bytes(long long M, int C) {
    (void)M;
    return (size_t)1024 * 2 * C * sizeof(float) + 256;
}

// sums [2][C]: sum_m a[m][c] and (when b) sum_m a[m][c] * (b[m][c] - bmean[c]) * bscale[c]
static int col_sums_impl(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                         const float* mask_scale, const float* mask_shift, long long M, int C, float* sums, void* ws, size_t ws_bytes,
                         vp_stream stream, float mask_hi = 0.f);

int vp_col_sums_f32(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                    long long M, int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream) {
    return col_sums_impl(ctx, a, lda, b, ldb, bmean, bscale, nullptr, nullptr, M, C, sums, ws, ws_bytes, stream);
}

// BatchNorm -> ReLU units (resnet_se.py:72-74, eres2net.py): the ReLU's backward folded into the two BatchNorm-backward passes.  a (= d y)
// counts only where the unit's output b * mask_scale + mask_shift (the BatchNorm affine of the saved pre-BN tensor b) was positive -- the
// same expression the forward's vp_affine_rows_f32 evaluates.  C % 4 == 0, 16-byte aligned; else VP_EUNSUP.
int vp_col_sums_masked_f32(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                           const float* mask_scale, const float* mask_shift, float mask_hi, long long M, int C, float* sums, void* ws,
                           size_t ws_bytes, vp_stream stream) {
    if (!b || !mask_scale || !mask_shift || mask_hi < 0.f) VP_FAIL(ctx, VP_EINVAL, "col_sums_masked: bad arguments");
    if (((C | lda | ldb) & 3) || (((uintptr_t)a | (uintptr_t)b) & 15)) return VP_EUNSUP;
    return col_sums_impl(ctx, a, lda, b, ldb, bmean, bscale, mask_scale, mask_shift, M, C, sums, ws, ws_bytes, stream, mask_hi);
}

static int col_sums_impl(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                         const float* mask_scale, const float* mask_shift, long long M, int C, float* sums, void* ws, size_t ws_bytes,
                         vp_stream stream, float mask_hi) {
    if (!ctx || !a || !sums || M <= 0 || C <= 0 || (b && (!bmean || !bscale))) VP_FAIL(ctx, VP_EINVAL, "col_sums: bad arguments");
    if (!ws || ws_bytes < vp_col_sums_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "col_sums: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    long long chunks;
    if (M > 0x7fffffffLL) VP_FAIL(ctx, VP_EINVAL, "col_sums: more than 2^31 rows");
    if (((C | lda | (b ? ldb : 0)) & 3) == 0 && (((uintptr_t)a | (uintptr_t)(b ? b : a)) & 15) == 0) {
        int cl_shift, colblocks, rpc, ch;
        colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, ch);
        chunks = ch;
        ColSum4Args p{a, b, bmean, bscale, (float*)ws, lda, ldb, (int)M, C / 4, rpc, cl_shift, mask_scale, mask_shift};
        p.mask_hi = mask_hi;
        if (b) hipLaunchKernelGGL(col_sums4_kernel<true>, dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(col_sums4_kernel<false>, dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    } else {
        chunks = (M + 511) / 512;
        if (chunks > 1024) chunks = 1024;
        const int rpc = (int)((M + chunks - 1) / chunks);
        chunks = (M + rpc - 1) / rpc;
        ColSumArgs p{a, b, bmean, bscale, (float*)ws, lda, ldb, (int)M, C, rpc};
        hipLaunchKernelGGL(col_sums_kernel, dim3((C + 63) / 64, (unsigned)chunks), dim3(256), 0, st, p);
    }
    VP_LAUNCH_CHECK(ctx, "col_sums");
    launch_sum_partials((const float*)ws, (int)chunks, (long long)2 * C, sums, st);
    VP_LAUNCH_CHECK(ctx, "col_sums_reduce");
    return VP_OK;
}

// the same two sums with b stored as bf16 (ldb in elements): the BatchNorm-backward reductions over a bf16 pre-BN activation
int vp_col_sums_f32_b16(vp_ctx* ctx, const float* a, int lda, const void* b, int ldb, const float* bmean, const float* bscale, long long M,
                        int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !a || !b || !bmean || !bscale || !sums || M <= 0 || M > 0x7fffffffLL || C <= 0 || (C | lda | ldb) & 3 ||
        ((uintptr_t)a & 15) || ((uintptr_t)b & 7))
        VP_FAIL(ctx, VP_EINVAL, "col_sums_b16: bad arguments");
    if (!ws || ws_bytes < vp_col_sums_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "col_sums_b16: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    ColSum4Args p{a, (const float*)b, bmean, bscale, (float*)ws, lda, ldb, (int)M, C / 4, rpc, cl_shift, nullptr, nullptr, nullptr, nullptr, 1, 0.f};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((col_sums4_kernel<true, bf16_t>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16");
    launch_sum_partials((const float*)ws, chunks, (long long)2 * C, sums, st);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16_reduce");
    return VP_OK;
}

// the same sums of a * us[b] + um[b] / T (b = row / T) against bf16 b: the BatchNorm-backward reductions of the conv in front of an SE
// block, with the SE block's input gradient dh = dout * s + dmean / T formed on the fly (vp_scale_shift_rows_f32 never runs)
int vp_col_sums_f32_b16_utt(vp_ctx* ctx, const float* a, int lda, const float* utt_scale, const float* utt_shift, int T, const void* b, int ldb,
                            const float* bmean, const float* bscale, long long M, int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !a || !b || !bmean || !bscale || !sums || !utt_scale || !utt_shift || T <= 0 || M <= 0 || M % T || M > 0x7fffffffLL || C <= 0 ||
        (C | lda | ldb) & 3 || (((uintptr_t)a | (uintptr_t)utt_scale | (uintptr_t)utt_shift) & 15) || ((uintptr_t)b & 7))
        VP_FAIL(ctx, VP_EINVAL, "col_sums_b16_utt: bad arguments");
    if (!ws || ws_bytes < vp_col_sums_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "col_sums_b16_utt: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    ColSum4Args p{a, (const float*)b, bmean, bscale, (float*)ws, lda, ldb, (int)M, C / 4, rpc, cl_shift, nullptr, nullptr, utt_scale, utt_shift, T,
                  1.f / (float)T};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((col_sums4_kernel<true, bf16_t, 1>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16_utt");
    launch_sum_partials((const float*)ws, chunks, (long long)2 * C, sums, st);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16_utt_reduce");
    return VP_OK;
}

int vp_bn_train_finalize(vp_ctx* ctx, const float* psum, const float* psumsq, int nparts, long long M, int C, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                         float* invstd, float* scale, float* shift, vp_stream stream) {
    if (!ctx || !psum || !psumsq || nparts <= 0 || M <= 0 || C <= 0 || !mean || !invstd || !scale || !shift)
        VP_FAIL(ctx, VP_EINVAL, "bn_finalize: bad arguments");
    BnFinArgs a{psum, psumsq, nparts, (int)M, C, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, vp_fault_word(ctx)};
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "bn_train_finalize");
    return VP_OK;
}

int vp_affine_rows_f32(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y,
                       int ldy, int relu, vp_stream stream) {
    if (!ctx || !z || !scale || !shift || !y || M <= 0 || C <= 0 || (C | ldz | ldy) & 3) VP_FAIL(ctx, VP_EINVAL, "affine_rows: bad arguments");
    hipLaunchKernelGGL(affine_rows_kernel<float>, dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, z, ldz, scale, shift, M, C / 4, y, ldy, relu);
    VP_LAUNCH_CHECK(ctx, "affine_rows");
    return VP_OK;
}

// z stored as bf16 (the wide mixed-precision layers keep the conv output in the low precision, as Paddle's O1 does): y f32
int vp_affine_rows_b16_f32(vp_ctx* ctx, const void* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y,
                           int ldy, int relu, vp_stream stream) {
    if (!ctx || !z || !scale || !shift || !y || M <= 0 || C <= 0 || (C | ldz | ldy) & 3) VP_FAIL(ctx, VP_EINVAL, "affine_rows_b16: bad arguments");
    hipLaunchKernelGGL(affine_rows_kernel<bf16_t>, dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, ldz, scale,
                       shift, M, C / 4, y, ldy, relu);
    VP_LAUNCH_CHECK(ctx, "affine_rows_b16");
    return VP_OK;
}

// z AND y bf16: a layer output whose only consumers read it as bf16 (ECAPA's MFA output under enable_amp: ASP's GEMM operand and statistics)
int vp_affine_rows_b16_b16(vp_ctx* ctx, const void* z, int ldz, const float* scale, const float* shift, long long M, int C, void* y,
                           int ldy, int relu, vp_stream stream) {
    if (!ctx || !z || !scale || !shift || !y || M <= 0 || C <= 0 || (C | ldz | ldy) & 3) VP_FAIL(ctx, VP_EINVAL, "affine_rows_b16_b16: bad arguments");
    hipLaunchKernelGGL((affine_rows_kernel<bf16_t, bf16_t>), dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, ldz,
                       scale, shift, M, C / 4, (bf16_t*)y, ldy, relu);
    VP_LAUNCH_CHECK(ctx, "affine_rows_b16_b16");
    return VP_OK;
}

// z f32, y bf16 (a layer outside the wide set whose only consumers read bf16: ECAPA's block 0 under enable_amp)
int vp_affine_rows_f32_b16(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, void* y,
                           int ldy, int relu, vp_stream stream) {
    if (!ctx || !z || !scale || !shift || !y || M <= 0 || C <= 0 || (C | ldz | ldy) & 3) VP_FAIL(ctx, VP_EINVAL, "affine_rows_f32_b16: bad arguments");
    hipLaunchKernelGGL((affine_rows_kernel<float, bf16_t>), dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, z, ldz, scale, shift, M,
                       C / 4, (bf16_t*)y, ldy, relu);
    VP_LAUNCH_CHECK(ctx, "affine_rows_f32_b16");
    return VP_OK;
}

int vp_bn_relu_bwd_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                       const float* gamma, const float* sums, long long M, int C, int relu_mask, float* dz, int lddz,
                       vp_stream stream) {
    if (!ctx || !dy || !z || !mean || !invstd || !sums || !dz || M <= 0 || C <= 0 || (C | lddy | ldz | lddz) & 3)
        VP_FAIL(ctx, VP_EINVAL, "bn_relu_bwd: bad arguments");
    BnBwdArgs a{dy, z, mean, invstd, gamma, sums, dz, lddy, ldz, lddz, C / 4, relu_mask, M, nullptr, nullptr};
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd");
    return VP_OK;
}

// BatchNorm backward with a ReLU BEHIND the BatchNorm folded in (see vp_col_sums_masked_f32): d y counts only where z * mask_scale + mask_shift > 0
int vp_bn_relu_bwd_masked_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                              const float* gamma, const float* sums, const float* mask_scale, const float* mask_shift, float mask_hi,
                              long long M, int C, float* dz, int lddz, int accumulate, vp_stream stream) {
    if (!ctx || !dy || !z || !mean || !invstd || !sums || !dz || !mask_scale || !mask_shift || mask_hi < 0.f || M <= 0 || C <= 0 ||
        (C | lddy | ldz | lddz) & 3)
        VP_FAIL(ctx, VP_EINVAL, "bn_relu_bwd_masked: bad arguments");
    BnBwdArgs a{dy, z, mean, invstd, gamma, sums, dz, lddy, ldz, lddz, C / 4, 0, M, mask_scale, mask_shift};
    a.mask_hi = mask_hi;
    a.accumulate = accumulate != 0;
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_masked");
    return VP_OK;
}

size_t vp_bn_relu_bwd_dbias_workspace_bytes(long long M, int C) {
    (void)M;
    return (size_t)1024 * C * sizeof(float) + 256;
}

static int bn_bwd_dbias_impl(vp_ctx* ctx, const float* dy, int lddy, const void* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz, bool dz_bf16,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream, bool z_bf16 = false);

int vp_bn_relu_bwd_dbias_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, float* dz, int lddz,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream) {
    return bn_bwd_dbias_impl(ctx, dy, lddy, z, ldz, mean, invstd, gamma, sums, M, C, relu_mask, dz, lddz, false, dbias, ws, ws_bytes, stream);
}

// the same with dz written as bf16 (lddz in elements): see bn_relu_bwd_dbias_kernel
int vp_bn_relu_bwd_dbias_bf16out(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                                 const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz,
                                 float* dbias, void* ws, size_t ws_bytes, vp_stream stream) {
    return bn_bwd_dbias_impl(ctx, dy, lddy, z, ldz, mean, invstd, gamma, sums, M, C, relu_mask, dz, lddz, true, dbias, ws, ws_bytes, stream);
}

// z AND dz bf16: the wide layers with the pre-BN activation kept in the low precision
int vp_bn_relu_bwd_dbias_b16(vp_ctx* ctx, const float* dy, int lddy, const void* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream) {
    return bn_bwd_dbias_impl(ctx, dy, lddy, z, ldz, mean, invstd, gamma, sums, M, C, relu_mask, dz, lddz, true, dbias, ws, ws_bytes, stream, true);
}

// alpha / beta of the context-statistics gradient (time_stats_bwd_coeffs_kernel): ab [2][B][C]
int vp_time_stats_bwd_coeffs(vp_ctx* ctx, const float* stats, const float* dstats, int B, int T, int C, float eps, float* ab, vp_stream stream) {
    if (!ctx || !stats || !dstats || !ab || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "time_stats_bwd_coeffs: bad arguments");
    hipLaunchKernelGGL(time_stats_bwd_coeffs_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, dstats, B, C, T, eps, ab);
    VP_LAUNCH_CHECK(ctx, "time_stats_bwd_coeffs");
    return VP_OK;
}

// The two BatchNorm-backward passes of a TDNNBlock whose output y = BN(z) is consumed by a pooling layer's context statistics: d y =
// dy + alpha[b] + beta[b] * bf16(z * bn_scale + bn_shift) (vp_time_stats_bwd_coeffs), y re-formed from the bf16 z the passes read anyway --
// vp_time_stats_bwd_add_x16's pass over the (B*T, C) tensors never runs.  sums: [2][C]
int vp_col_sums_f32_b16_ctx(vp_ctx* ctx, const float* a, int lda, const float* alpha, const float* beta, int T, const float* bn_scale,
                            const float* bn_shift, const void* b, int ldb, const float* bmean, const float* bscale, long long M, int C, float* sums,
                            void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !a || !b || !bmean || !bscale || !sums || !alpha || !beta || !bn_scale || !bn_shift || T <= 0 || M <= 0 || M % T ||
        M > 0x7fffffffLL || C <= 0 || (C | lda | ldb) & 3 || (((uintptr_t)a | (uintptr_t)alpha | (uintptr_t)beta) & 15) || ((uintptr_t)b & 7))
        VP_FAIL(ctx, VP_EINVAL, "col_sums_b16_ctx: bad arguments");
    if (!ws || ws_bytes < vp_col_sums_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "col_sums_b16_ctx: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    ColSum4Args p{a, (const float*)b, bmean, bscale, (float*)ws, lda, ldb, (int)M, C / 4, rpc, cl_shift, nullptr, nullptr, alpha, beta, T, 0.f,
                  bn_scale, bn_shift};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((col_sums4_kernel<true, bf16_t, 2>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16_ctx");
    launch_sum_partials((const float*)ws, chunks, (long long)2 * C, sums, st);
    VP_LAUNCH_CHECK(ctx, "col_sums_b16_ctx_reduce");
    return VP_OK;
}

int vp_bn_relu_bwd_dbias_b16_ctx(vp_ctx* ctx, const float* dy, int lddy, const float* alpha, const float* beta, int T, const float* bn_scale,
                                 const float* bn_shift, const void* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                 const float* sums, long long M, int C, int relu_mask, void* dz, int lddz, float* dbias, void* ws,
                                 size_t ws_bytes, vp_stream stream) {
    if (!ctx || !dy || !z || !mean || !invstd || !sums || !dz || !dbias || !alpha || !beta || !bn_scale || !bn_shift || T <= 0 || M <= 0 ||
        M % T || M > 0x7fffffffLL || C <= 0 || (C | lddy | ldz | lddz) & 3 || (((uintptr_t)dy | (uintptr_t)alpha | (uintptr_t)beta) & 15) ||
        (((uintptr_t)z | (uintptr_t)dz) & 7))
        VP_FAIL(ctx, VP_EINVAL, "bn_relu_bwd_dbias_ctx: bad arguments");
    if (!ws || ws_bytes < vp_bn_relu_bwd_dbias_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "bn_relu_bwd_dbias_ctx: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    BnBwdSumArgs p{{dy, (const float*)z, mean, invstd, gamma, sums, (float*)dz, lddy, ldz, lddz, C / 4, relu_mask, M, nullptr, nullptr,
                    alpha, beta, T, 0.f, bn_scale, bn_shift}, (float*)ws, (int)M, rpc, cl_shift};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_relu_bwd_dbias_kernel<bf16_t, bf16_t, 2>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias_ctx");
    launch_sum_partials((const float*)ws, chunks, (long long)C, dbias, st);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias_ctx_reduce");
    return VP_OK;
}

// z AND dz bf16, d y = dy * utt_scale[b] + utt_shift[b] / T (b = row / T) formed on the fly: BatchNorm + ReLU backward of the conv in
// front of an SE block straight from the block's OUTPUT gradient (see vp_col_sums_f32_b16_utt)
int vp_bn_relu_bwd_dbias_b16_utt(vp_ctx* ctx, const float* dy, int lddy, const float* utt_scale, const float* utt_shift, int T, const void* z,
                                 int ldz, const float* mean, const float* invstd, const float* gamma, const float* sums, long long M, int C,
                                 int relu_mask, void* dz, int lddz, float* dbias, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !dy || !z || !mean || !invstd || !sums || !dz || !dbias || !utt_scale || !utt_shift || T <= 0 || M <= 0 || M % T ||
        M > 0x7fffffffLL || C <= 0 || (C | lddy | ldz | lddz) & 3 || (((uintptr_t)dy | (uintptr_t)utt_scale | (uintptr_t)utt_shift) & 15) ||
        (((uintptr_t)z | (uintptr_t)dz) & 7))
        VP_FAIL(ctx, VP_EINVAL, "bn_relu_bwd_dbias_utt: bad arguments");
    if (!ws || ws_bytes < vp_bn_relu_bwd_dbias_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "bn_relu_bwd_dbias_utt: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    BnBwdSumArgs p{{dy, (const float*)z, mean, invstd, gamma, sums, (float*)dz, lddy, ldz, lddz, C / 4, relu_mask, M, nullptr, nullptr,
                    utt_scale, utt_shift, T, 1.f / (float)T}, (float*)ws, (int)M, rpc, cl_shift};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_relu_bwd_dbias_kernel<bf16_t, bf16_t, 1>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias_utt");
    launch_sum_partials((const float*)ws, chunks, (long long)C, dbias, st);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias_utt_reduce");
    return VP_OK;
}

static int bn_bwd_dbias_impl(vp_ctx* ctx, const float* dy, int lddy, const void* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz, bool dz_bf16,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream, bool z_bf16) {
    if (!ctx || !dy || !z || !mean || !invstd || !sums || !dz || !dbias || M <= 0 || M > 0x7fffffffLL || C <= 0 ||
        (C | lddy | ldz | lddz) & 3 || ((uintptr_t)dy & 15) || ((uintptr_t)z & (z_bf16 ? 7 : 15)) || ((uintptr_t)dz & (dz_bf16 ? 7 : 15)) ||
        (z_bf16 && !dz_bf16))
        VP_FAIL(ctx, VP_EINVAL, "bn_relu_bwd_dbias: bad arguments");
    if (!ws || ws_bytes < vp_bn_relu_bwd_dbias_workspace_bytes(M, C)) VP_FAIL(ctx, VP_EWORKSPACE, "bn_relu_bwd_dbias: workspace too small");
    int cl_shift, colblocks, rpc, chunks;
    colsum4_geometry(M, C / 4, cl_shift, colblocks, rpc, chunks);
    BnBwdSumArgs p{{dy, (const float*)z, mean, invstd, gamma, sums, (float*)dz, lddy, ldz, lddz, C / 4, relu_mask, M, nullptr, nullptr}, (float*)ws, (int)M, rpc, cl_shift};
    hipStream_t st = (hipStream_t)stream;
    if (z_bf16) hipLaunchKernelGGL((bn_relu_bwd_dbias_kernel<bf16_t, bf16_t>), dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    else if (dz_bf16) hipLaunchKernelGGL(bn_relu_bwd_dbias_kernel<bf16_t>, dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(bn_relu_bwd_dbias_kernel<float>, dim3(colblocks, (unsigned)chunks), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias");
    launch_sum_partials((const float*)ws, chunks, (long long)C, dbias, st);
    VP_LAUNCH_CHECK(ctx, "bn_relu_bwd_dbias_reduce");
    return VP_OK;
}

int vp_adam_step_f32(vp_ctx* ctx, float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, vp_stream stream) {
    if (!ctx || !param || !grad || !m || !v || n <= 0 || step < 1) VP_FAIL(ctx, VP_EINVAL, "adam: bad arguments");
    const float c1 = 1.f - powf(beta1, (float)step), c2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, c1, c2, grad_scale, 1.f, vp_fault_word(ctx));
    VP_LAUNCH_CHECK(ctx, "adam");
    return VP_OK;
}

int vp_adamw_step_f32(vp_ctx* ctx, float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1, float beta2,
                      float eps, float coeff, int step, float grad_scale, vp_stream stream) {
    if (!ctx || !param || !grad || !m || !v || n <= 0 || step < 1) VP_FAIL(ctx, VP_EINVAL, "adamw: bad arguments");
    const float c1 = 1.f - powf(beta1, (float)step), c2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr, beta1, beta2, eps,
                       0.f, c1, c2, grad_scale, 1.f - lr * coeff, vp_fault_word(ctx));
    VP_LAUNCH_CHECK(ctx, "adamw");
    return VP_OK;
}

int vp_momentum_step_f32(vp_ctx* ctx, float* param, const float* grad, float* velocity, long long n, float lr, float momentum,
                         float weight_decay, int use_nesterov, float grad_scale, vp_stream stream) {
    if (!ctx || !param || !grad || !velocity || n <= 0) VP_FAIL(ctx, VP_EINVAL, "momentum: bad arguments");
    hipLaunchKernelGGL(momentum_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, param, grad, velocity, n, lr, momentum,
                       weight_decay, grad_scale, use_nesterov, vp_fault_word(ctx));
    VP_LAUNCH_CHECK(ctx, "momentum");
    return VP_OK;
}

int vp_pack_segments_f32(vp_ctx* ctx, const void* const* srcs, const long long* offs, const long long* sizes, int n, float* dst,
                         vp_stream stream) {
    if (!ctx || !srcs || !offs || !sizes || !dst || n <= 0) VP_FAIL(ctx, VP_EINVAL, "pack_segments: bad arguments");
    for (int i0 = 0; i0 < n; i0 += PACK_MAX) {
        PackArgs a;
        const int cnt = n - i0 < PACK_MAX ? n - i0 : PACK_MAX;
        long long big = 0;
        for (int i = 0; i < cnt; ++i) {
            if (offs[i0 + i] < 0 || sizes[i0 + i] < 0) VP_FAIL(ctx, VP_EINVAL, "pack_segments: negative offset / size");
            a.src[i] = (const float*)srcs[i0 + i]; a.off[i] = offs[i0 + i]; a.n[i] = sizes[i0 + i];
            if (sizes[i0 + i] > big) big = sizes[i0 + i];
        }
        a.dst = dst;
        long long bx = (big + 4 * 256 - 1) / (4 * 256);
        if (bx < 1) bx = 1;
        if (bx > 512) bx = 512;
        hipLaunchKernelGGL(pack_segments_kernel, dim3((unsigned)bx, cnt), dim3(256), 0, (hipStream_t)stream, a);
        VP_LAUNCH_CHECK(ctx, "pack_segments");
    }
    return VP_OK;
}

int vp_utt_sums_b16(vp_ctx* ctx, const void* a_bf16, int lda, int B, int T, int C, float* out, vp_stream stream) {
    if (!ctx || !a_bf16 || !out || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "utt_sums_b16: bad arguments");
    hipLaunchKernelGGL(utt_sums_kernel<bf16_t>, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a_bf16, lda, T, C, out);
    VP_LAUNCH_CHECK(ctx, "utt_sums_b16");
    return VP_OK;
}

int vp_utt_sums_f32(vp_ctx* ctx, const float* a, int lda, int B, int T, int C, float* out, vp_stream stream) {
    if (!ctx || !a || !out || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "utt_sums: bad arguments");
    hipLaunchKernelGGL(utt_sums_kernel<float>, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, a, lda, T, C, out);
    VP_LAUNCH_CHECK(ctx, "utt_sums");
    return VP_OK;
}

int vp_time_stats_f32(vp_ctx* ctx, const float* x, int ldx, int B, int T, int C, float eps, int unbiased, float* stats, vp_stream stream) {
    if (!ctx || !x || !stats || B <= 0 || T <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "time_stats: bad arguments");
    return vp_time_moments(ctx, VP_F32, x, ldx, B, T, C, eps, unbiased, stats, (hipStream_t)stream);
}

int vp_time_stats_bwd_f32(vp_ctx* ctx, const float* x, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                          int unbiased, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !x || !stats || !dstats || !dx || B <= 0 || T <= 0 || C <= 0 || (C | ldx | lddx) & 3) VP_FAIL(ctx, VP_EINVAL, "time_stats_bwd: bad arguments");
    TsBwdArgs a{x, stats, dstats, dx, ldx, lddx, T, C / 4, eps, (long long)B * T * (C / 4), unbiased, nullptr, 0};
    hipLaunchKernelGGL(time_stats_bwd_kernel<float>, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "time_stats_bwd");
    return VP_OK;
}

// dx = add + (the gradient through the statistics): the other consumers' gradients of x folded into this pass (add may be dx)
int vp_time_stats_bwd_add_f32(vp_ctx* ctx, const float* x, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                              int unbiased, const float* add, int ldadd, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !x || !stats || !dstats || !dx || !add || B <= 0 || T <= 0 || C <= 0 || (C | ldx | lddx | ldadd) & 3)
        VP_FAIL(ctx, VP_EINVAL, "time_stats_bwd_add: bad arguments");
    TsBwdArgs a{x, stats, dstats, dx, ldx, lddx, T, C / 4, eps, (long long)B * T * (C / 4), unbiased, add, ldadd};
    hipLaunchKernelGGL(time_stats_bwd_kernel<float>, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "time_stats_bwd_add");
    return VP_OK;
}

// the same with x stored as bf16 (ldx in elements)
int vp_time_stats_bwd_add_x16(vp_ctx* ctx, const void* x_bf16, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                              int unbiased, const float* add, int ldadd, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !x_bf16 || !stats || !dstats || !dx || !add || B <= 0 || T <= 0 || C <= 0 || (C | ldx | lddx | ldadd) & 3)
        VP_FAIL(ctx, VP_EINVAL, "time_stats_bwd_add_x16: bad arguments");
    TsBwdArgs a{(const float*)x_bf16, stats, dstats, dx, ldx, lddx, T, C / 4, eps, (long long)B * T * (C / 4), unbiased, add, ldadd};
    hipLaunchKernelGGL(time_stats_bwd_kernel<bf16_t>, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "time_stats_bwd_add_x16");
    return VP_OK;
}

int vp_attn_stats_bwd_f32(vp_ctx* ctx, const float* e, const float* x, int ldx, const float* pooled, const float* dpooled, int B, int T,
                          int C, float eps, float* de, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !e || !x || !pooled || !dpooled || !de || !dx || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "attn_stats_bwd: bad arguments");
    AsBwdArgs a{e, x, pooled, dpooled, de, dx, ldx, lddx, T, C, eps};
    { const int rc = launch_attn_stats_bwd<float>(ctx, a, B, (hipStream_t)stream); if (rc) return rc; }
    VP_LAUNCH_CHECK(ctx, "attn_stats_bwd");
    return VP_OK;
}

// d e written as bf16 ((B*T, C) dense): mixed precision, the logits conv's backward GEMMs then read bf16 operands
// d e bf16 AND e stored as bf16 (vp_asp_softmax_stats_l16's logits); T <= 320, else VP_EUNSUP
int vp_attn_stats_bwd_e16(vp_ctx* ctx, const void* e_bf16, const void* x, int x_dtype, int ldx, const float* pooled, const float* dpooled, int B,
                          int T, int C, float eps, void* de_bf16, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !e_bf16 || !x || !pooled || !dpooled || !de_bf16 || !dx || B <= 0 || T <= 0 || C <= 0 || B > 65535 ||
        (x_dtype != VP_F32 && x_dtype != VP_BF16))
        VP_FAIL(ctx, VP_EINVAL, "attn_stats_bwd_e16: bad arguments");
    if (T > 320) return VP_EUNSUP;
    AsBwdArgs a{(const float*)e_bf16, (const float*)x, pooled, dpooled, (float*)de_bf16, dx, ldx, lddx, T, C, eps};
    const dim3 grid((C + 63) / 64, B);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == VP_BF16) {
        if (T <= 160) hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<bf16_t, 20, bf16_t, bf16_t>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<bf16_t, 40, bf16_t, bf16_t>), grid, dim3(512), 0, st, a);
    } else if (T <= 160) hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<bf16_t, 20, bf16_t>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((attn_stats_bwd_reg_kernel<bf16_t, 40, bf16_t>), grid, dim3(512), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "attn_stats_bwd_e16");
    return VP_OK;
}

int vp_attn_stats_bwd_de16(vp_ctx* ctx, const float* e, const float* x, int ldx, const float* pooled, const float* dpooled, int B, int T,
                           int C, float eps, void* de, float* dx, int lddx, vp_stream stream) {
    if (!ctx || !e || !x || !pooled || !dpooled || !de || !dx || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "attn_stats_bwd: bad arguments");
    AsBwdArgs a{e, x, pooled, dpooled, (float*)de, dx, ldx, lddx, T, C, eps};
    { const int rc = launch_attn_stats_bwd<bf16_t>(ctx, a, B, (hipStream_t)stream); if (rc) return rc; }
    VP_LAUNCH_CHECK(ctx, "attn_stats_bwd_de16");
    return VP_OK;
}

int vp_act_f32(vp_ctx* ctx, int act, const float* x, long long n, float* y, vp_stream stream) {
    if (!ctx || !x || !y || n <= 0 || n & 3 || act < VP_ACT_RELU || act > VP_ACT_SILU) VP_FAIL(ctx, VP_EINVAL, "act: bad arguments");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid1d(n / 4)), dim3(256), 0, (hipStream_t)stream, x, n / 4, y, act_kind(act));
    VP_LAUNCH_CHECK(ctx, "act");
    return VP_OK;
}

int vp_act_bwd_f32(vp_ctx* ctx, int act, const float* dy, const float* y, long long n, float* dz, vp_stream stream) {
    if (!ctx || !dy || !y || !dz || n <= 0 || n & 3 || act < VP_ACT_RELU || act > VP_ACT_SILU) VP_FAIL(ctx, VP_EINVAL, "act_bwd: bad arguments");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid1d(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, n / 4, dz, act_kind(act));
    VP_LAUNCH_CHECK(ctx, "act_bwd");
    return VP_OK;
}

int vp_reflect_fold_f32(vp_ctx* ctx, const float* dxp, int B, int T, int pad, int C, float* dx, vp_stream stream) {
    if (!ctx || !dxp || !dx || B <= 0 || T <= 0 || pad < 0 || pad >= T || C <= 0 || C & 3) VP_FAIL(ctx, VP_EINVAL, "reflect_fold: bad arguments");
    FoldArgs a{dxp, dx, T, pad, C / 4, (long long)B * T * (C / 4), C, nullptr, 0, nullptr};
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "reflect_fold");
    return VP_OK;
}

// the same fold written into a channel slice of a wider gradient tensor (rows lddx apart), and sum = folded + add (dense (B*T, C))
// when add is given: a Res2Net chunk's input gradient goes to its slice of d x, and, summed with the concatenation's gradient
// of the previous chunk's output, becomes that chunk's output gradient (ecapa_tdnn.py:36-46), in the pass that folds
int vp_reflect_fold_into_f32(vp_ctx* ctx, const float* dxp, int B, int T, int pad, int C, float* dx, int lddx, const float* add, int ld_add,
                             float* sum, vp_stream stream) {
    if (!ctx || !dxp || !dx || B <= 0 || T <= 0 || pad < 0 || pad >= T || C <= 0 || (C | lddx | ld_add) & 3 || (add && !sum) ||
        (((uintptr_t)dx | (uintptr_t)add | (uintptr_t)sum) & 15))
        VP_FAIL(ctx, VP_EINVAL, "reflect_fold_into: bad arguments");
    FoldArgs a{dxp, dx, T, pad, C / 4, (long long)B * T * (C / 4), lddx, add, ld_add, sum};
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "reflect_fold_into");
    return VP_OK;
}

int vp_affine_rows_aux_f32(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y, int ldy,
                           const float* add, int ld_add, float* aux, int ld_aux, vp_stream stream) {
    if (!ctx || !z || !scale || !shift || !y || M <= 0 || C <= 0 || (C | ldz | ldy) & 3 || (aux && (!add || (ld_add | ld_aux) & 3)) ||
        (((uintptr_t)z | (uintptr_t)y | (uintptr_t)add | (uintptr_t)aux) & 15))
        VP_FAIL(ctx, VP_EINVAL, "affine_rows_aux: bad arguments");
    AffAuxArgs a{z, scale, shift, y, add, aux, ldz, ldy, ld_add, ld_aux, C / 4, M};
    hipLaunchKernelGGL(affine_rows_aux_kernel, dim3(grid1d(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "affine_rows_aux");
    return VP_OK;
}

static void srb_geometry(int B, int T, int C, int& cl_shift, int& chunks, int& rpc) {
    const int C4 = C / 4;
    cl_shift = 0;
    while ((1 << cl_shift) < C4 && cl_shift < 8) ++cl_shift;
    const int RG = 256 >> cl_shift;
    long long ch = 2048 / (B > 0 ? B : 1);                      // ~2048 workgroups
    if (ch < 1) ch = 1;
    long long r = (T + ch - 1) / ch;
    if (r < 8LL * RG) r = 8LL * RG;                              // at least two four-row trips per thread
    rpc = (int)r;
    chunks = (T + rpc - 1) / rpc;
}

size_t vp_scale_rows_bwd_workspace_bytes(int B, int T, int C) {
    if (B <= 0 || T <= 0 || C <= 0 || (C & 3) || C > 1024) return 0;
    int cl, chunks, rpc;
    srb_geometry(B, T, C, cl, chunks, rpc);
    return (size_t)B * chunks * C * sizeof(float) + 256;
}

// out = x * s[b] (+ res) backward with the positions of an utterance spread over many workgroups (ws from vp_scale_rows_bwd_workspace_bytes;
// C % 4 == 0, C <= 1024, 16-byte aligned tensors -- otherwise, or for few positions, callers use vp_scale_rows_bwd_f32)
int vp_scale_rows_bwd_ws_f32(vp_ctx* ctx, const float* dy, const float* x, const float* s, int B, int T, int C, float* dx, float* ds, void* ws,
                             size_t ws_bytes, vp_stream stream) {
    if (!ctx || !dy || !x || !s || !dx || !ds || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "scale_rows_bwd: bad arguments");
    if ((C & 3) || C > 1024 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)s | (uintptr_t)dx) & 15)) return VP_EUNSUP;
    const size_t need = vp_scale_rows_bwd_workspace_bytes(B, T, C);
    if (!ws || ws_bytes < need) VP_FAIL(ctx, VP_EWORKSPACE, "scale_rows_bwd: workspace %zu < %zu", ws_bytes, need);
    int cl, chunks, rpc;
    srb_geometry(B, T, C, cl, chunks, rpc);
    SrbArgs a{dy, x, s, dx, (float*)ws, T, C / 4, chunks, rpc, cl};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scale_rows_bwd_chunk_kernel, dim3(chunks, B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "scale_rows_bwd_chunk");
    launch_sum_partials((const float*)ws, chunks, C, ds, st, 1, 1, B);
    VP_LAUNCH_CHECK(ctx, "scale_rows_bwd_reduce");
    return VP_OK;
}

int vp_scale_rows_bwd_f32(vp_ctx* ctx, const float* dy, const float* x, const float* s, int B, int T, int C, float* dx, float* ds,
                          vp_stream stream) {
    if (!ctx || !dy || !x || !s || !dx || !ds || B <= 0 || T <= 0 || C <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "scale_rows_bwd: bad arguments");
    hipLaunchKernelGGL(scale_rows_bwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, dy, x, s, T, C, dx, ds);
    VP_LAUNCH_CHECK(ctx, "scale_rows_bwd");
    return VP_OK;
}

int vp_utt_dot_f32(vp_ctx* ctx, const float* dy, const float* x, int B, int T, int C, float* ds, vp_stream stream) {
    if (!ctx || !dy || !x || !ds || B <= 0 || T <= 0 || C <= 0 || (C & 3) || B > 65535 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)ds) & 15))
        VP_FAIL(ctx, VP_EINVAL, "utt_dot: bad arguments");
    hipLaunchKernelGGL(utt_dot4_kernel<float>, dim3((C / 4 + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, dy, x, T, C, ds);
    VP_LAUNCH_CHECK(ctx, "utt_dot");
    return VP_OK;
}

// x stored as bf16 (the SE block's input kept in the low precision: its gate's gradient ds[b,c] = sum_t dy * x)
int vp_utt_dot_x16(vp_ctx* ctx, const float* dy, const void* x_bf16, int B, int T, int C, float* ds, vp_stream stream) {
    if (!ctx || !dy || !x_bf16 || !ds || B <= 0 || T <= 0 || C <= 0 || (C & 3) || B > 65535 || (((uintptr_t)dy | (uintptr_t)ds) & 15) || ((uintptr_t)x_bf16 & 7))
        VP_FAIL(ctx, VP_EINVAL, "utt_dot_x16: bad arguments");
    hipLaunchKernelGGL(utt_dot4_kernel<bf16_t>, dim3((C / 4 + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)x_bf16, T, C, ds);
    VP_LAUNCH_CHECK(ctx, "utt_dot_x16");
    return VP_OK;
}

// ds[b][c] = sum_t dy * bf16(z * bn_scale + bn_shift): the SE gate's gradient over the PRE-BatchNorm activation of the conv in front of
// it (bf16), the BatchNorm apply pass folded into the read (the block's input h is never stored)
int vp_utt_dot_z16(vp_ctx* ctx, const float* dy, const void* z_bf16, const float* bn_scale, const float* bn_shift, int B, int T, int C,
                   float* ds, vp_stream stream) {
    if (!ctx || !dy || !z_bf16 || !bn_scale || !bn_shift || !ds || B <= 0 || T <= 0 || C <= 0 || (C & 3) || B > 65535 ||
        (((uintptr_t)dy | (uintptr_t)ds | (uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) || ((uintptr_t)z_bf16 & 7))
        VP_FAIL(ctx, VP_EINVAL, "utt_dot_z16: bad arguments");
    hipLaunchKernelGGL((utt_dot4_kernel<bf16_t, true>), dim3((C / 4 + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)z_bf16, T, C,
                       ds, bn_scale, bn_shift);
    VP_LAUNCH_CHECK(ctx, "utt_dot_z16");
    return VP_OK;
}

int vp_scale_shift_rows_f32(vp_ctx* ctx, const float* dy, const float* s, const float* dm, int B, int T, int C, float* dx, vp_stream stream) {
    if (!ctx || !dy || !s || !dm || !dx || B <= 0 || T <= 0 || C <= 0 || (C & 3) ||
        (((uintptr_t)dy | (uintptr_t)s | (uintptr_t)dm | (uintptr_t)dx) & 15))
        VP_FAIL(ctx, VP_EINVAL, "scale_shift_rows: bad arguments");
    const long long total = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(scale_shift_rows4_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, dy, s, dm, T, C / 4, 1.f / (float)T,
                       total, dx);
    VP_LAUNCH_CHECK(ctx, "scale_shift_rows");
    return VP_OK;
}

int vp_zero_insert_2d_f32(vp_ctx* ctx, const float* dz, int B, int T_out, int F_out, int C, int T_in, int F_in, int stride_t, int stride_f,
                          float* up, vp_stream stream) {
    if (!ctx || !dz || !up || B <= 0 || T_out <= 0 || F_out <= 0 || T_in <= 0 || F_in <= 0 || stride_t < 1 || stride_f < 1 || C <= 0 || C & 3)
        VP_FAIL(ctx, VP_EINVAL, "zero_insert: bad arguments");
    ZiArgs a{dz, up, T_out, F_out, T_in, F_in, stride_t, stride_f, C / 4, (long long)B * T_in * F_in * (C / 4)};
    hipLaunchKernelGGL(zero_insert_kernel, dim3(grid1d(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "zero_insert");
    return VP_OK;
}

int vp_relu_bwd_f32(vp_ctx* ctx, const float* dy, const float* y, long long n, float* dz, vp_stream stream) {
    if (!ctx || !dy || !y || !dz || n <= 0 || n & 3) VP_FAIL(ctx, VP_EINVAL, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid1d(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, n / 4, dz);
    VP_LAUNCH_CHECK(ctx, "relu_bwd");
    return VP_OK;
}

int vp_aff_combine_f32(vp_ctx* ctx, const float* t, const float* x, const float* y, long long rows, int C, float* out, vp_stream stream) {
    if (!ctx || !t || !x || !y || !out || rows <= 0 || C <= 0 || C & 3) VP_FAIL(ctx, VP_EINVAL, "aff_combine: bad arguments");
    return vp_aff_combine(ctx, VP_F32, t, C, x, C, 0, y, C, 0, out, C, 0, rows, C, (hipStream_t)stream);
}

int vp_aff_combine_bwd_f32(vp_ctx* ctx, const float* g, const float* t, const float* x, const float* y, long long n, float* dx, float* dy,
                           float* dt, vp_stream stream) {
    if (!ctx || !g || !t || !x || !y || !dx || !dy || !dt || n <= 0 || n & 3) VP_FAIL(ctx, VP_EINVAL, "aff_combine_bwd: bad arguments");
    hipLaunchKernelGGL(aff_combine_bwd_kernel, dim3(grid1d(n / 4)), dim3(256), 0, (hipStream_t)stream, g, t, x, y, n / 4, dx, dy, dt);
    VP_LAUNCH_CHECK(ctx, "aff_combine_bwd");
    return VP_OK;
}

int vp_seg_ctx_f32(vp_ctx* ctx, const float* x, int B, int T, int C, int seg_len, float* out, vp_stream stream) {
    if (!ctx || !x || !out || B <= 0 || T <= 0 || C <= 0 || seg_len <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "seg_ctx: bad arguments");
    hipLaunchKernelGGL(seg_ctx_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, T, C, seg_len, (T + seg_len - 1) / seg_len, out);
    VP_LAUNCH_CHECK(ctx, "seg_ctx");
    return VP_OK;
}

int vp_seg_ctx_bwd_f32(vp_ctx* ctx, const float* dctx, int B, int T, int C, int seg_len, float* dx, vp_stream stream) {
    if (!ctx || !dctx || !dx || B <= 0 || T <= 0 || C <= 0 || seg_len <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "seg_ctx_bwd: bad arguments");
    hipLaunchKernelGGL(seg_ctx_bwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, dctx, T, C, seg_len, (T + seg_len - 1) / seg_len, dx);
    VP_LAUNCH_CHECK(ctx, "seg_ctx_bwd");
    return VP_OK;
}

int vp_seg_scale_f32(vp_ctx* ctx, const float* y, const float* m, int B, int T, int C, int seg_len, float* out, vp_stream stream) {
    if (!ctx || !y || !m || !out || B <= 0 || T <= 0 || C <= 0 || C & 3 || seg_len <= 0) VP_FAIL(ctx, VP_EINVAL, "seg_scale: bad arguments");
    const long long total = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(seg_scale_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, y, m, T, C / 4, seg_len, (T + seg_len - 1) / seg_len, total, out);
    VP_LAUNCH_CHECK(ctx, "seg_scale");
    return VP_OK;
}

int vp_seg_scale_bwd_f32(vp_ctx* ctx, const float* g, const float* y, const float* m, int B, int T, int C, int seg_len, float* dy, float* dm,
                         vp_stream stream) {
    if (!ctx || !g || !y || !m || !dy || !dm || B <= 0 || T <= 0 || C <= 0 || seg_len <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "seg_scale_bwd: bad arguments");
    hipLaunchKernelGGL(seg_scale_bwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, g, y, m, T, C, seg_len, (T + seg_len - 1) / seg_len, dy, dm);
    VP_LAUNCH_CHECK(ctx, "seg_scale_bwd");
    return VP_OK;
}

}  // extern "C"
