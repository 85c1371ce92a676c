// Res2Net chain of one SE-Res2 block in SPLIT PRECISION, fused: the hl32 form of res2_chain.hip.
//
// Replaces Res2NetBlock.forward (ppvector/models/ecapa_tdnn.py:36-47): y_0 = x_0, y_1 = f_1(x_1), y_j = f_j(x_j + y_{j-1}),
// f_j = TDNNBlock(w -> w, k3, dilation d) = BN(ReLU(conv(reflect-pad))), concat -- on tensors stored as split bf16 planes (vpmi.h:
// VP_HL32, value = hi + lo), every product as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with f32 accumulation (~2^-16 per product:
// the arithmetic that carries the reference's 1e-4 score tolerance, conv_gemm_impl.h: x3_t / hl_t).
//
// What differs from the bf16 kernel: an activation row is 256 B (64 channels x (hi + lo)), so a whole 3 s utterance with both ping-pong
// buffers and the weights (48 KB per conv) does not fit the CU's 160 KB.  The utterance is therefore cut into `nsplit` time segments,
// one workgroup each, with a halo of H = nconv * dil frames on every interior cut: conv i needs frames within dil of its outputs, so
// after i convs a segment's values are exact on [lo + i dil, hi - i dil) -- still covering its own frames after the last conv.  The halo
// frames are recomputed by both neighbours (9 - 19 % more MFMA work at T = 298) and never stored.  LDS: act[2][2 groups][TP][128 B]
// + wts[3 taps][2 groups][64][128 B] (single buffer: a tap's region takes the next conv's weights as soon as every wave holds the
// tap's fragments in registers) + the per-channel terms.
// A 32-channel group of a row is 128 B = [32 hi | 32 lo]: exactly the bf16 kernels' K-stage row, so LDS-DMA pieces (8 rows x 128 B, XOR
// swizzle on the source chunk), fragment reads (chunk g = hi, chunk 4 + g = lo) and the coalesced copy-out are those of res2_chain.hip.
// Per conv and wave: taps outermost (16 weight fragments live), up to two 16-frame tiles, 72 MFMAs per tile.
#include "common.h"

namespace {

constexpr int RX_W = 64;
constexpr int RX_THREADS = 512;
constexpr int RX_WAVES = RX_THREADS / 64;
constexpr int RX_ROUNDS = 2;                 // 16-frame tiles per wave: TP <= 256
constexpr int RX_WT_BYTES = 3 * 2 * RX_W * 128;          // 49,152
constexpr int RX_TP_MAX = 208;               // 2 x 2 x 208 x 128 + 49,152 + 5,376 = 161,024 B
typedef __attribute__((address_space(3))) void* rx_lds_ptr;

struct Res2X3Args {
    const char* t1;          // hl32 (B*T, C): tdnn1 output, x_j = channels [j*64, (j+1)*64)
    char* r2;                // hl32 (B*T, C): concat of y_j (slice 0 written by tdnn1's epilogue)
    const char* w[VP_MAX_RES2];         // hl32 [64][192], k = tap*64 + c
    const float* bias[VP_MAX_RES2];
    const float* scale[VP_MAX_RES2];
    const float* shift[VP_MAX_RES2];
    int T, C, nconv, dil, TP, Tseg, H;
    unsigned t1_bytes;
};

__device__ __forceinline__ int rx_reflect(int t, int T) {
    t = t < 0 ? -t : t;
    return t >= T ? 2 * (T - 1) - t : t;
}
__device__ __forceinline__ float rx_val(unsigned h, unsigned l, int odd) {
    return odd ? __builtin_bit_cast(float, h & 0xffff0000u) + __builtin_bit_cast(float, l & 0xffff0000u)
               : __builtin_bit_cast(float, h << 16) + __builtin_bit_cast(float, l << 16);
}

__global__ __launch_bounds__(RX_THREADS, 1) void res2_x3_kernel(Res2X3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TP = a.TP;
    const int grp_bytes = TP * 128, act_bytes = 2 * grp_bytes;
    char* act0 = smem;
    char* wts = smem + 2 * act_bytes;
    float* prm = reinterpret_cast<float*>(wts + RX_WT_BYTES);            // [nconv][3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, seg = blockIdx.x;
    // this segment owns frames [own0, own1) and holds frames [lo, hi) of the utterance
    const int own0 = seg * a.Tseg, own1 = min(a.T, own0 + a.Tseg);
    const int lo = max(0, own0 - a.H), hi = min(a.T, own1 + a.H);
    const int rows = hi - lo;
    const size_t row0 = (size_t)b * a.T;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.t1), 0, a.t1_bytes, 0x00020000);
    const unsigned ldb = (unsigned)a.C * 4u;

    const int drow = lane >> 3;
    const unsigned dchunk = (unsigned)((lane & 7) ^ drow) << 4;
    // slice `sl` of the segment's frames -> act buffer `dst`: pieces (group, 8 rows), round-robin over the waves
    auto dma_x = [&](int sl, char* dst) {
        const int np = TP / 8;
        for (int p = wv; p < 2 * np; p += RX_WAVES) {
            const int gq = p >= np ? 1 : 0, pr = p - gq * np;
            const int r = pr * 8 + drow;
            const unsigned off = r < rows ? (unsigned)(row0 + lo + r) * ldb + (unsigned)(sl * 256 + gq * 128) + dchunk : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (rx_lds_ptr)(dst + gq * grp_bytes + pr * 1024), 16, off, 0, 0, 0);
        }
    };
    // weights of conv j, tap `tap` -> wts as [tap * 2 + group][n][128 B]: 16 pieces of 8 rows per tap, 2 per wave
    auto dma_w = [&](int j, int tap) {
        const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w[j]), 0, RX_WT_BYTES, 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = wv + u * RX_WAVES;              // 0..15
            const int q = tap * 2 + (p >> 3), nb = (p & 7) * 8;       // q = tap * 2 + group
            const unsigned off = (unsigned)((nb + drow) * 768 + q * 128) + dchunk;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (rx_lds_ptr)(wts + (q * 64 + nb) * 128), 16, off, 0, 0, 0);
        }
    };

    dma_x(1, act0);
    dma_w(0, 0); dma_w(0, 1); dma_w(0, 2);
    for (int i = tid; i < a.nconv * 192; i += RX_THREADS) {
        const int j = i / 192, k = i - j * 192, which = k >> 6, n = k & 63;
        prm[i] = which == 0 ? a.bias[j][n] : (which == 1 ? a.scale[j][n] : a.shift[j][n]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int ntile = TP / 16;
    const int sw_h = g, sw_l = 4 + g;
    for (int j = 0; j < a.nconv; ++j) {
        char* ain = act0 + (j & 1) * act_bytes;
        char* aout = act0 + ((j + 1) & 1) * act_bytes;
        const bool has_next = j + 1 < a.nconv;
        if (has_next) dma_x(j + 2, aout);                               // lands under this conv's MFMAs
        f32x4 acc[RX_ROUNDS][4];
#pragma unroll
        for (int r = 0; r < RX_ROUNDS; ++r)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[r][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            bf16x8 wh[4][2], wl[4][2];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ni * 16 + li;
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    const char* wr = wts + ((tap * 2 + gq) * 64 + n) * 128;
                    wh[ni][gq] = *reinterpret_cast<const bf16x8*>(wr + ((sw_h ^ (n & 7)) << 4));
                    wl[ni][gq] = *reinterpret_cast<const bf16x8*>(wr + ((sw_l ^ (n & 7)) << 4));
                }
            }
            if (has_next) {
                // every wave holds this tap's weights in registers behind the barrier: the tap's LDS region is dead and takes the NEXT
                // conv's weights of the same tap now, under this conv's MFMAs (a single weight buffer; loading all 48 KB after the
                // MFMAs left ~1.5 us of L2 latency exposed per conv).  Raw s_barrier: __syncthreads() would also wait for the x DMA in flight
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_w(j + 1, tap);
            }
#pragma unroll
            for (int r = 0; r < RX_ROUNDS; ++r) {
                const int mt = wv + r * RX_WAVES;
                if (mt < ntile) {                                        // wave-uniform
                    const int tl = min(mt * 16 + li, rows - 1);          // rows past the segment compute on its last row: discarded
                    int ts = rx_reflect(lo + tl + (tap - 1) * a.dil, a.T) - lo;
                    ts = min(max(ts, 0), rows - 1);                      // a source outside the held frames only feeds halo outputs
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        const char* xr = ain + gq * grp_bytes + ts * 128;
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(xr + ((sw_h ^ (ts & 7)) << 4));
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(xr + ((sw_l ^ (ts & 7)) << 4));
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            acc[r][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni][gq], xh, acc[r][ni], 0, 0, 0);
                            acc[r][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni][gq], xl, acc[r][ni], 0, 0, 0);
                            acc[r][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni][gq], xh, acc[r][ni], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // own DMAs landed; after the barrier everyone's have, and every wave is done reading `ain` and the weights
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* pj = prm + j * 192;
#pragma unroll
        for (int r = 0; r < RX_ROUNDS; ++r) {
            const int mt = wv + r * RX_WAVES;
            if (mt < ntile) {
                const int t = mt * 16 + li;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int nb = ni * 16 + g * 4;
                    const float4 bb = *reinterpret_cast<const float4*>(pj + nb);
                    const float4 ss = *reinterpret_cast<const float4*>(pj + 64 + nb);
                    const float4 hh = *reinterpret_cast<const float4*>(pj + 128 + nb);
                    float v[4];
                    v[0] = fmaxf(acc[r][ni][0] + bb.x, 0.f) * ss.x + hh.x;
                    v[1] = fmaxf(acc[r][ni][1] + bb.y, 0.f) * ss.y + hh.y;
                    v[2] = fmaxf(acc[r][ni][2] + bb.z, 0.f) * ss.z + hh.z;
                    v[3] = fmaxf(acc[r][ni][3] + bb.w, 0.f) * ss.w + hh.w;
                    // channel nb: group nb >> 5, hi chunk (nb & 31) >> 3, byte (nb & 7) * 2 of row t; lo chunk = hi chunk + 4
                    const int gq = nb >> 5, ch = (nb & 31) >> 3, sub = (nb & 7) * 2;
                    const int ph = gq * grp_bytes + t * 128 + ((ch ^ (t & 7)) << 4) + sub;
                    const int pl = gq * grp_bytes + t * 128 + (((4 + ch) ^ (t & 7)) << 4) + sub;
                    auto put = [&](char* buf, const float (&u)[4]) {
                        unsigned short h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bf16_t xh = (bf16_t)u[e];
                            const bf16_t xl = (bf16_t)(u[e] - (float)xh);
                            h[e] = __builtin_bit_cast(unsigned short, xh);
                            l[e] = __builtin_bit_cast(unsigned short, xl);
                        }
                        *reinterpret_cast<uint2*>(buf + ph) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                        *reinterpret_cast<uint2*>(buf + pl) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    };
                    put(ain, v);                                       // y_{j+1}: staged in the dead input buffer
                    if (has_next) {                                    // next input = y_{j+1} + x_{j+2}, in place (f32 sum, then split)
                        const uint2 xh = *reinterpret_cast<const uint2*>(aout + ph), xl = *reinterpret_cast<const uint2*>(aout + pl);
                        const float s[4] = {v[0] + rx_val(xh.x, xl.x, 0), v[1] + rx_val(xh.x, xl.x, 1),
                                            v[2] + rx_val(xh.y, xl.y, 0), v[3] + rx_val(xh.y, xl.y, 1)};
                        put(aout, s);
                    }
                }
            }
        }
        __syncthreads();
        // y_{j+1} of the segment's OWN frames out: 16 lanes per frame, 256 contiguous bytes (both groups, un-swizzled)
        {
            char* ybase = a.r2 + (row0 + lo) * (size_t)ldb + (size_t)(j + 1) * 256;
            const int r0 = own0 - lo, r1 = own1 - lo;
            for (int i = r0 * 16 + tid; i < r1 * 16; i += RX_THREADS) {
                const int t = i >> 4, c = i & 15, gq = c >> 3, cc = c & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(ain + gq * grp_bytes + t * 128 + ((cc ^ (t & 7)) << 4));
                *reinterpret_cast<uint4*>(ybase + (size_t)t * ldb + c * 16) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next conv's weights
        __syncthreads();                                                // `ain` is the next conv's DMA target
    }
}

// fewest time segments whose frames (own + halos) fit the LDS; nsplit = 0 when none does
void rx_plan(int T, int H, int& nsplit, int& Tseg, int& TP) {
    nsplit = 0;
    for (int ns = 1; ns <= 64 && !nsplit; ++ns) {                      // (20 s of audio = 2 000 frames: 14 segments at dilation 4)
        const int ts = ((T + ns - 1) / ns + 15) / 16 * 16;              // own frames per segment, whole tiles
        if (ns > 1 && (ns - 1) * ts >= T) break;                        // an empty last segment: not a useful cut
        int need = 0;
        for (int s = 0; s < ns; ++s) {
            const int o0 = s * ts, o1 = T < o0 + ts ? T : o0 + ts;
            const int l = o0 - H > 0 ? o0 - H : 0, h = o1 + H < T ? o1 + H : T;
            need = h - l > need ? h - l : need;
        }
        const int tp = (need + 15) / 16 * 16;
        if (tp <= RX_TP_MAX && tp / 16 <= RX_WAVES * RX_ROUNDS) { nsplit = ns; Tseg = ts; TP = tp; }
    }
}

}  // namespace

// would vp_res2_chain_x3 take this chain?  (the ECAPA driver decides on its fast path before the first launch)
bool vp_res2_chain_x3_ok(const vp_tdnn_layer* layers, int nconv, int B, int T, int C, int width) {
    if (width != RX_W || nconv < 1 || nconv > VP_MAX_RES2 || T < 2 || C % 32 || B < 1 || B > 65535 ||
        (unsigned long long)B * T * C * 4 >= 0xffffff00ull)
        return false;                                               // (32-bit buffer offsets into t1)
    const int dil = layers[0].dil;
    for (int j = 0; j < nconv; ++j) {
        const vp_tdnn_layer& L = layers[j];
        if (L.kw != 3 || L.cin != RX_W || L.cout != RX_W || L.dil != dil || !L.bias || !L.bn_scale || !L.bn_shift || !L.w_hl) return false;
    }
    int nsplit, Tseg, TP;
    rx_plan(T, nconv * dil, nsplit, Tseg, TP);
    return nsplit > 0 && dil < T && dil < Tseg;
}

// hl32 in (t1) / hl32 out (r2); weights hl32 [64][192] (vp_tdnn_layer.w_hl).  VP_EUNSUP when the shape is not covered.
int vp_res2_chain_x3(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C, int width,
                     hipStream_t st) {
    if (width != RX_W || nconv < 1 || nconv > VP_MAX_RES2 || T < 2 || B > 65535 || C % 32) return VP_EUNSUP;
    const int dil = layers[0].dil;
    const int H = nconv * dil;
    int nsplit = 0, Tseg = 0, TP = 0;
    rx_plan(T, H, nsplit, Tseg, TP);
    if (!nsplit) return VP_EUNSUP;
    Res2X3Args a;
    memset(&a, 0, sizeof(a));
    for (int j = 0; j < nconv; ++j) {
        const vp_tdnn_layer& L = layers[j];
        if (L.kw != 3 || L.cin != RX_W || L.cout != RX_W || L.dil != dil || !L.bias || !L.bn_scale || !L.bn_shift || !L.w_hl)
            return VP_EUNSUP;
        a.w[j] = (const char*)L.w_hl; a.bias[j] = L.bias; a.scale[j] = L.bn_scale; a.shift[j] = L.bn_shift;
    }
    if (dil >= T || dil >= Tseg) return VP_EUNSUP;
    a.t1 = (const char*)t1; a.r2 = (char*)r2; a.T = T; a.C = C; a.nconv = nconv; a.dil = dil; a.TP = TP; a.Tseg = Tseg; a.H = H;
    const unsigned long long t1b = (unsigned long long)B * T * C * 4;
    if (t1b >= 0xffffff00ull || ((reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(r2)) & 15)) return VP_EUNSUP;
    a.t1_bytes = (unsigned)t1b;
    const size_t smem = (size_t)4 * TP * 128 + RX_WT_BYTES + (size_t)nconv * 192 * 4;
    static bool attr_dev[64] = {};
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(res2_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(res2_x3_kernel, dim3(nsplit, B), dim3(RX_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "res2_x3");
    return VP_OK;
}
