// conv1d as an implicit GEMM on the gfx950 matrix cores, with the TDNN epilogue fused in.
//
// Replaces per launch: Conv1d.forward (ppvector/models/utils.py:65-93) / nn.Conv1D (tdnn.py:13-21)
// -> ReLU -> BatchNorm1d eval (utils.py:96-119, :147-148), the Res2Net hand-off x_{i+1} + y_i
// (ecapa_tdnn.py:36-47) and the time sums SEBlock / ASP need (ecapa_tdnn.py:69-78, pooling.py:97-104).
//
// Layout: activations are frame-major (B*T, C) -- K (channels) contiguous for BOTH operands, so a
// dilated tap is just a row shift (reflect / zero handled in the row index), never an im2col.
//   M = B*T_out rows,  N = Cout,  K = KW*Cin  (k = tap*Cin + channel).
// Tile: 128 x BN (BN = 128 | 64) per 256-thread workgroup (4 waves), K staged 128 B per row per
// stage (64 bf16 / 32 f32), double-buffered LDS with an XOR swizzle of the 16-B chunk index by
// (row & 7) -- conflict-free for the ds_write_b128 staging and the ds_read_b128 fragment reads.
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16 path) or v_mfma_f32_16x16x4_f32 (exact-f32 path), weights as
// the A operand and activations as the B operand so each lane ends up with 4 CONSECUTIVE output
// channels of one frame: 8/16-byte epilogue loads and stores, float4 parameter reads.
// Roofline: MFMA-bound (dense contraction); algorithmic flops = 2*M*N*K per launch.
#include "common.h"

namespace {

constexpr int BM = VP_CONV_BM;
constexpr int ROWB = 128;        // bytes of K per tile row per stage
constexpr int NSEG_MAX = 8;

struct ConvArgs {
    const void* x; const void* w;
    const float* bias; const float* rowbias; const float* bn_scale; const float* bn_shift;
    void* y; void* y2; const void* add_in; void* aux; float* psum; float* psumsq;
    unsigned x_bytes, w_bytes;
    int ldx, xoff, ldy, yoff, ldy2, y2off, ysplit, ld_add, add_off, ld_aux, aux_off;
    int M, N, K, KC, cpt, KT;
    int T_in, T_out, dilation, stride, pad_left, pad_mode, act, act2;
    int tiles_m, tiles_n, nseg, group_m;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float4 lo, hi; };

__device__ __forceinline__ void load_frag(const char* tile, int row, int ks, int g, Frag<bf16_t>& f) {
    const int c = ks * 4 + g;
    f.v = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((c ^ (row & 7)) << 4));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<float>& f) {
    const int c = 2 * g;
    f.lo = *reinterpret_cast<const float4*>(tile + row * ROWB + ((c ^ (row & 7)) << 4));
    f.hi = *reinterpret_cast<const float4*>(tile + row * ROWB + (((c + 1) ^ (row & 7)) << 4));
}
// D[n][m] += sum_k W[n][k] * X[m][k]: weights are the A operand (row = lane & 15 -> n), activations
// the B operand (col = lane & 15 -> m); result register r of lane l = (n = (l >> 4) * 4 + r, m = l & 15).
__device__ __forceinline__ void mma(const Frag<bf16_t>& w, const Frag<bf16_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, x.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<float>& w, const Frag<float>& x, f32x4& c) {
    // lane group g holds k = 8g .. 8g+7 of the 32-wide stage for BOTH operands; instruction e
    // contracts the four k = 8g + e, the eight instructions cover the stage (exact f32 fma chain).
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.x, x.lo.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.y, x.lo.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.z, x.lo.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.w, x.lo.w, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.x, x.hi.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.y, x.hi.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.z, x.hi.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.w, x.hi.w, c, 0, 0, 0);
}

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<bf16_t> { typedef bf16x4 type; };

__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

template <typename TI, typename TO, int BN, bool KW1>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvArgs a) {
    constexpr int EPC = 16 / (int)sizeof(TI);        // elements per 16-B chunk
    constexpr int KSTEPS = (8 * EPC) / 32;           // MFMA k-steps per stage: bf16 2, f32 1
    constexpr int WN = BN / 64;                      // waves along N (64 columns each)
    constexpr int WM = 4 / WN;
    constexpr int MI = BM / (WM * 16);
    constexpr int NI = 4;
    constexpr int BROWS = BN / 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int li = lane & 15, g = lane >> 4;

    // Block -> tile map. (1) XCD-aware, bijective: block b runs on XCD b % 8, so each XCD gets a
    // contiguous run of the tile order and keeps its own L2 working set.  (2) Grouped order inside
    // the run: GM consecutive M-tiles x all N-tiles form a group, M fastest -- the ~64 workgroups an
    // XCD runs at once then share GM activation panels and a few weight panels instead of streaming
    // the whole weight matrix once per M-tile (measured on the MFA GEMM: 2.1 GB of L2 misses for a
    // 0.47 GB problem with N-fastest order).
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int gsz = a.group_m * a.tiles_n;
    const int grp = swz / gsz, rem = swz - grp * gsz;
    const int gm = min(a.group_m, a.tiles_m - grp * a.group_m);      // M-tiles in this (maybe last) group
    const int tn = rem / gm;
    const int tm = grp * a.group_m + (rem - tn * gm);
    const int m0 = tm * BM, n0 = tn * BN;

    // Operands are fetched with buffer loads through wave-uniform resource descriptors: a 32-bit
    // byte offset per lane, and anything that must read as zero (rows past M, channels past K, zero
    // padding, weight rows past N) gets an out-of-range offset -- the hardware returns 0, no branch,
    // no select on the loaded value, so the loads stay in flight across the MFMA block.
    constexpr unsigned ES = sizeof(TI);
    constexpr unsigned OOB = 0xfffffff0u;      // every dword of the 16-B access is >= num_records, no wrap
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

    // global -> LDS staging assignment: 16-B chunk cc of rows r0 + 32 i
    const int cc = tid & 7, r0 = tid >> 3;
    const int pw = (cc ^ (r0 & 7)) << 4;
    unsigned rowoff[4], woff[BROWS];
    int tpos[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / a.T_out;
        const int t = mm - b * a.T_out;
        rowoff[i] = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.ldx + (unsigned)a.xoff) * ES : OOB;
        tpos[i] = t * a.stride - a.pad_left;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + r0 + 32 * i;
        woff[i] = n < a.N ? (unsigned)n * (unsigned)a.K * ES : OOB;
    }
    const bool zero_pad = a.pad_mode == VP_PAD_ZERO;
    const unsigned ldxb = (unsigned)a.ldx * ES;
    // 1x1 convolutions (most of the flops): the source row of every staged row is fixed, only the
    // K offset moves -- hoist the whole row address out of the K loop.
    unsigned rowfix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int traw = tpos[i];
        int ts = traw < 0 ? -traw : traw;
        ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
        const bool inr = traw >= 0 && traw < a.T_in;
        rowfix[i] = (rowoff[i] != OOB && (inr || !zero_pad)) ? rowoff[i] + (unsigned)ts * ldxb : OOB;
    }

    // TWO register stages: the loads of K-stage k+2 are issued while stage k is computed and are
    // written to LDS at the end of stage k+1, so a fetch has two stages of MFMA work to land
    // (one stage was not enough to cover HBM/L2 latency with only two workgroups per CU).
    u32x4 ra0[4], rb0[BROWS], ra1[4], rb1[BROWS];

    auto gload = [&](int kt, u32x4 (&ra)[4], u32x4 (&rb)[BROWS]) {
        const int q = kt * 8 + cc;
        const bool kv = q < a.KC;
        const unsigned kb = (unsigned)q * 16u;
        if constexpr (KW1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = kv && rowfix[i] != OOB;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? rowfix[i] + kb : OOB, 0, 0);
            }
        } else {
            const int j = q / a.cpt;
            const unsigned cb = (unsigned)(q - j * a.cpt) * 16u;      // byte offset of the chunk in its tap
            const int tj = j * a.dilation;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int traw = tpos[i] + tj;
                int ts = traw < 0 ? -traw : traw;                      // reflect (identity when in range)
                ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                const bool inr = traw >= 0 && traw < a.T_in;
                const bool ok = kv && rowoff[i] != OOB && (inr || !zero_pad);
                const unsigned off = rowoff[i] + (unsigned)ts * ldxb + cb;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? off : OOB, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const bool ok = kv && woff[i] != OOB;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, ok ? woff[i] + kb : OOB, 0, 0);
        }
    };
    auto swrite = [&](int s, const u32x4 (&ra)[4], const u32x4 (&rb)[BROWS]) {
        char* As = smem + s * STAGE;
        char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(As + (r0 + 32 * i) * ROWB + pw) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * i) * ROWB + pw) = rb[i];
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) {
        const char* As = smem + s * STAGE;
        const char* Bs = As + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<TI> xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) load_frag(As, wm * (MI * 16) + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) load_frag(Bs, wn * 64 + ni * 16 + li, ks, g, wf[ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(wf[ni], xf[mi], acc[mi][ni]);
        }
    };

    // The loads inside the loop are UNCONDITIONAL (a stage past K is all out-of-range offsets: zeros,
    // no memory traffic): with a condition around them hipcc can no longer count the outstanding
    // loads and falls back to vmcnt(0) before every LDS write, which kills the two-stage distance.
    const int KT = a.KT;
    gload(0, ra0, rb0);
    gload(1, ra1, rb1);
    swrite(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        // even stage kt on LDS[0]; set0 <- stage kt+2; set1 (stage kt+1) -> LDS[1]
        gload(kt + 2, ra0, rb0);
        compute(0);
        swrite(1, ra1, rb1);
        __syncthreads();
        if (kt + 1 >= KT) break;
        // odd stage kt+1 on LDS[1]; set1 <- stage kt+3; set0 (stage kt+2) -> LDS[0]
        gload(kt + 3, ra1, rb1);
        compute(1);
        swrite(0, ra0, rb0);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    TO* __restrict__ Y = static_cast<TO*>(a.y);
    TO* __restrict__ Y2 = static_cast<TO*>(a.y2);
    const TO* __restrict__ ADD = static_cast<const TO*>(a.add_in);
    TO* __restrict__ AUX = static_cast<TO*>(a.aux);
    const int bfirst = m0 / a.T_out;
    int rowm[MI], rowb[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        rowm[mi] = m0 + wm * (MI * 16) + mi * 16 + li;
        rowb[mi] = (rowm[mi] < a.M ? rowm[mi] : (a.M - 1)) / a.T_out;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int nb = n0 + wn * 64 + ni * 16 + g * 4;
        const bool nvalid = nb < a.N;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f}, sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvalid) {
            if (a.bias) load4(a.bias + nb, bias4);
            if (a.bn_scale) load4(a.bn_scale + nb, sc4);
            if (a.bn_shift) load4(a.bn_shift + nb, sh4);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = rowm[mi];
            const bool ok = nvalid && m < a.M;
            float v[4];
            float rbias[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok && a.rowbias) load4(a.rowbias + (size_t)rowb[mi] * a.N + nb, rbias);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[mi][ni][r] + bias4[r] + rbias[r];
                if (a.act == VP_ACT_RELU) t = fmaxf(t, 0.f);
                t = t * sc4[r] + sh4[r];
                if (a.act2 == VP_ACT_TANH) t = tanhf(t);
                v[r] = t;
            }
            if (ok) {
                store4(Y + (size_t)m * a.ldy + a.yoff + nb, v);
                if (nb < a.ysplit) store4(Y2 + (size_t)m * a.ldy2 + a.y2off + nb, v);
                if (AUX) {
                    float ad[4];
                    load4(ADD + (size_t)m * a.ld_add + a.add_off + nb, ad);
                    float s4[4] = {v[0] + ad[0], v[1] + ad[1], v[2] + ad[2], v[3] + ad[3]};
                    store4(AUX + (size_t)m * a.ld_aux + a.aux_off + nb, s4);
                }
            }
            // keep (y - shift) for the column sums; zero for rows / columns outside the problem
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = ok ? (v[r] - sh4[r]) : 0.f;
        }
    }
    if (a.psum) {
        // per (M-tile, utterance segment) column sums, deterministic: lanes -> waves -> workgroup
        float* red = reinterpret_cast<float*>(smem);          // [2][WM][NSEG_MAX][BN]
        for (int s = 0; s < a.nseg; ++s) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const float d = (rowb[mi] - bfirst == s) ? acc[mi][ni][r] : 0.f;
                        s1 += d;
                        s2 += d * d;
                    }
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        s1 += __shfl_xor(s1, o);
                        s2 += __shfl_xor(s2, o);
                    }
                    if (li == 0) {
                        const int col = wn * 64 + ni * 16 + g * 4 + r;
                        red[((0 * WM + wm) * NSEG_MAX + s) * BN + col] = s1;
                        red[((1 * WM + wm) * NSEG_MAX + s) * BN + col] = s2;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            for (int s = 0; s < a.nseg; ++s) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    s1 += red[((0 * WM + w) * NSEG_MAX + s) * BN + tid];
                    s2 += red[((1 * WM + w) * NSEG_MAX + s) * BN + tid];
                }
                const size_t o = ((size_t)tm * a.nseg + s) * a.N + n0 + tid;
                a.psum[o] = s1;
                if (a.psumsq) a.psumsq[o] = s2;
            }
        }
    }
}

template <typename TI, typename TO, int BN, bool KW1>
int launch1(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<TI, TO, BN, KW1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<TI, TO, BN, KW1>), dim3(a.tiles_m * a.tiles_n), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_gemm");
    return VP_OK;
}

template <typename TI, typename TO, int BN>
int launch(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    return a.KC == a.cpt ? launch1<TI, TO, BN, true>(ctx, a, st) : launch1<TI, TO, BN, false>(ctx, a, st);
}

}  // namespace

extern "C" {

int vp_conv1d_tiles_m(int B, int T_out) { return (int)(((long long)B * T_out + BM - 1) / BM); }
int vp_conv1d_nseg(int T_out) { return T_out > 0 ? (BM - 1) / T_out + 2 : 0; }

int vp_conv1d_fwd(vp_ctx* ctx, const vp_conv1d_desc* d, vp_stream stream) {
    if (!ctx || !d || !d->x || !d->w || !d->y) VP_FAIL(ctx, VP_EINVAL, "conv1d: null argument");
    if ((d->dtype_in != VP_F32 && d->dtype_in != VP_BF16) || (d->dtype_out != VP_F32 && d->dtype_out != VP_BF16))
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad dtype");
    if (d->dtype_in == VP_F32 && d->dtype_out == VP_BF16) VP_FAIL(ctx, VP_EUNSUP, "conv1d: f32 -> bf16 not built");
    const int epc = d->dtype_in == VP_BF16 ? 8 : 4;
    if (d->B <= 0 || d->T_in <= 0 || d->T_out <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->dilation <= 0)
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad shape");
    if (d->Cin % epc || d->ldx % epc || d->xoff % epc) VP_FAIL(ctx, VP_EINVAL, "conv1d: Cin/ldx/xoff must be multiples of %d", epc);
    if (d->Cout % 4 || d->ldy % 4 || d->yoff % 4) VP_FAIL(ctx, VP_EINVAL, "conv1d: Cout/ldy/yoff must be multiples of 4");
    if (d->ysplit) {
        if (!d->y2 || d->ysplit % 4 || d->ldy2 % 4 || d->y2off % 4 || d->ysplit > d->Cout)
            VP_FAIL(ctx, VP_EINVAL, "conv1d: bad y2 split");
    }
    const size_t es = d->dtype_in == VP_BF16 ? 2 : 4;
    const unsigned long long xbytes = ((unsigned long long)d->B * d->T_in - 1) * d->ldx * es + (d->xoff + d->Cin) * es;
    const unsigned long long wbytes = (unsigned long long)d->Cout * d->KW * d->Cin * es;
    if (xbytes >= 0xffffff00ull || wbytes >= 0xffffff00ull)
        VP_FAIL(ctx, VP_EUNSUP, "conv1d: operand larger than 4 GiB (32-bit buffer offsets)");
    if (d->aux && (!d->add_in || d->ld_add % 4 || d->add_off % 4 || d->ld_aux % 4 || d->aux_off % 4))
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad aux/add_in");
    const int span = d->dilation * (d->KW - 1);
    if (d->pad_mode == VP_PAD_NONE) {
        if ((d->T_out - 1) * d->stride + span > d->T_in - 1)
            VP_FAIL(ctx, VP_EINVAL, "conv1d: un-padded window leaves the input (T_in %d, T_out %d)", d->T_in, d->T_out);
        if (d->pad_left != 0) VP_FAIL(ctx, VP_EINVAL, "conv1d: pad_left with PAD_NONE");
    } else if (d->pad_mode == VP_PAD_REFLECT) {
        const int right = (d->T_out - 1) * d->stride - d->pad_left + span - (d->T_in - 1);
        if (d->pad_left >= d->T_in || right >= d->T_in) VP_FAIL(ctx, VP_EINVAL, "conv1d: reflect pad >= T_in");
    } else if (d->pad_mode != VP_PAD_ZERO) {
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad pad_mode");
    }
    if ((long long)d->B * d->T_out > 0x7fffffffLL / 2) VP_FAIL(ctx, VP_EINVAL, "conv1d: B*T too large");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.rowbias = d->rowbias;
    a.x_bytes = (unsigned)xbytes; a.w_bytes = (unsigned)wbytes; a.y2 = d->y2;
    a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.y = d->y; a.add_in = d->add_in; a.aux = d->aux;
    a.psum = d->psum; a.psumsq = d->psumsq;
    a.ldx = d->ldx; a.xoff = d->xoff; a.ldy2 = d->ldy2; a.y2off = d->y2off; a.ysplit = d->ysplit;
    a.ldy = d->ldy; a.yoff = d->yoff; a.ld_add = d->ld_add; a.add_off = d->add_off; a.ld_aux = d->ld_aux;
    a.aux_off = d->aux_off;
    a.M = d->B * d->T_out; a.N = d->Cout; a.K = d->KW * d->Cin; a.cpt = d->Cin / epc; a.KC = a.K / epc;
    a.KT = (a.KC + 7) / 8;
    a.T_in = d->T_in; a.T_out = d->T_out; a.dilation = d->dilation; a.stride = d->stride; a.pad_left = d->pad_left;
    a.pad_mode = d->pad_mode; a.act = d->act; a.act2 = d->act2;
    const int bn = d->Cout <= 64 ? 64 : 128;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + bn - 1) / bn;
    a.nseg = vp_conv1d_nseg(d->T_out);
    a.group_m = 96 / a.tiles_n;
    if (a.group_m < 1) a.group_m = 1;
    if (a.group_m > 16) a.group_m = 16;
    if (d->psum && a.nseg > NSEG_MAX) VP_FAIL(ctx, VP_EUNSUP, "conv1d: T_out %d too short for fused time sums", d->T_out);
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype_in == VP_BF16 && d->dtype_out == VP_BF16)
        return bn == 64 ? launch<bf16_t, bf16_t, 64>(ctx, a, st) : launch<bf16_t, bf16_t, 128>(ctx, a, st);
    if (d->dtype_in == VP_BF16 && d->dtype_out == VP_F32)
        return bn == 64 ? launch<bf16_t, float, 64>(ctx, a, st) : launch<bf16_t, float, 128>(ctx, a, st);
    return bn == 64 ? launch<float, float, 64>(ctx, a, st) : launch<float, float, 128>(ctx, a, st);
}

}  // extern "C"
