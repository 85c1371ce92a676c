// Res2Net chain of one SE-Res2 block as ONE kernel: one workgroup per utterance.
//
// Replaces Res2NetBlock.forward (ppvector/models/ecapa_tdnn.py:36-47): y_0 = x_0, y_1 = f_1(x_1),
// y_j = f_j(x_j + y_{j-1}), f_j = TDNNBlock(w -> w, k3, dilation d) = BN(ReLU(conv(reflect-pad))),
// then concat.  As separate launches this is 7 serially dependent M x 64 x 192 GEMMs per block (21 per
// forward, launch-shaped: 19 us each on MI355X).  Here the whole chain runs out of LDS:
//   - the utterance's current input (T x w bf16) lives in LDS, ping-ponged with the next input
//     (y_j + x_{j+1}); reflect padding is an LDS row index;
//   - ALL global traffic is row-coalesced: x_{j+1} and the next conv's weights arrive by LDS-DMA (8 rows x 128 B per
//     wave-instruction, XOR swizzle applied on the source chunk) straight into the buffers they are used from, and
//     y_j leaves through the input buffer the conv has just finished reading, 16 B per lane.  The first version did
//     its x loads / y stores in the MFMA accumulator layout (8 B per lane, 16 rows per instruction): the texture
//     path handled those a lane at a time -- 4 + 3.4 us of an 11 us conv whose compute is 3.7 us (measured by
//     ablation, tools/res2_timing.py).
// Per conv: DMAs issued -> 3 rounds of 24 MFMAs per wave -> barrier (DMAs landed, input dead) -> epilogue (y into the
// dead input buffer, y + x added in place into the next input) -> barrier -> coalesced copy-out -> barrier.
// Used when the buffers fit in LDS and each wave owns <= 3 frame tiles (T <= 384 at w = 64); longer utterances fall
// back to the per-conv path.
#include "common.h"

namespace {

constexpr int R2_W = 64;            // chunk width (channels per Res2 scale slice)
constexpr int R2_K = 3 * R2_W;      // K of one conv
constexpr int R2_THREADS = 512;
constexpr int R2_WAVES = R2_THREADS / 64;
constexpr int R2_ROUNDS = 3;        // frame tiles (16 frames) per wave: T <= 16 * 8 * 3 = 384
typedef __attribute__((address_space(3))) void* r2_lds_ptr;

struct Res2Args {
    const bf16_t* t1;        // (B*T, C): tdnn1 output, x_j = columns [j*w, (j+1)*w)
    bf16_t* r2;              // (B*T, C): concat of y_j (slice 0 already written by tdnn1's epilogue)
    const bf16_t* w[VP_MAX_RES2];       // [64][192] bf16, k = tap*64 + c
    const float* bias[VP_MAX_RES2];
    const float* scale[VP_MAX_RES2];
    const float* shift[VP_MAX_RES2];
    int T, C, nconv, dil, TP;          // TP = T rounded up to 16
    unsigned t1_bytes;
#ifdef VP_TIMING
    unsigned long long* dbg;
#endif
};

__device__ __forceinline__ int reflect_idx(int t, int T) {
    t = t < 0 ? -t : t;
    return t >= T ? 2 * (T - 1) - t : t;
}

__global__ __launch_bounds__(R2_THREADS, 1) void res2_chain_kernel(Res2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: act[2][TP][128 B] | wts[2][3 taps][64][128 B] | prm[nconv][3][64] f32
    const int act_bytes = a.TP * 128;
    char* act0 = smem;
    char* wt0 = smem + 2 * act_bytes;
    constexpr int WT_BYTES = R2_W * R2_K * 2;     // 24576
#ifdef VP_TIMING
    unsigned long long stamps[32];
    int nst = 0;
    stamps[nst++] = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const size_t row0 = (size_t)b * a.T;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.t1), 0, a.t1_bytes, 0x00020000);

    // LDS-DMA lane mapping: a wave-instruction lands 8 rows x 128 B linearly; lane l -> row l >> 3, slot l & 7, and the
    // slot holds source chunk slot ^ row (the swizzle every ds_read_b128 below undoes).
    const int drow = lane >> 3;
    const unsigned dchunk = (unsigned)((lane & 7) ^ drow) << 4;
    // x_{slice} rows [8 p, 8 p + 8) of this utterance -> act buffer `dst`; pieces p = wv, wv + 8, ...
    auto dma_x = [&](int slice, char* dst) {
        const unsigned colb = (unsigned)slice * (R2_W * 2) + dchunk;
        for (int p = wv; p < a.TP / 8; p += R2_WAVES) {
            const int t = p * 8 + drow;
            const unsigned off = t < a.T ? (unsigned)((row0 + t) * a.C * 2) + colb : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (r2_lds_ptr)(dst + p * 1024), 16, off, 0, 0, 0);
        }
    };
    // weights of conv j -> wts buffer `dst` as [tap][n][128 B]; 24 pieces of (tap, 8 rows), 3 per wave
    auto dma_w = [&](int j, char* dst) {
        const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w[j]), 0, WT_BYTES, 0x00020000);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int p = wv + u * R2_WAVES;              // 0..23
            const int tap = p >> 3, nb = (p & 7) * 8;
            const unsigned off = (unsigned)((nb + drow) * (R2_K * 2) + tap * 128) + dchunk;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (r2_lds_ptr)(dst + tap * 8192 + nb * 128), 16, off, 0, 0, 0);
        }
    };

    dma_x(1, act0);
    dma_w(0, wt0);
    const int ntile = a.TP / 16;
    float* prm = reinterpret_cast<float*>(wt0 + 2 * WT_BYTES);          // [nconv][3][64]
    for (int i = tid; i < a.nconv * 192; i += R2_THREADS) {
        const int j = i / 192, k = i - j * 192, which = k >> 6, n = k & 63;
        prm[i] = which == 0 ? a.bias[j][n] : (which == 1 ? a.scale[j][n] : a.shift[j][n]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef VP_TIMING
    stamps[nst++] = wall_clock64();
#endif

    for (int j = 0; j < a.nconv; ++j) {
        char* ain = act0 + (j & 1) * act_bytes;
        char* aout = act0 + ((j + 1) & 1) * act_bytes;
        const char* wl = wt0 + (j & 1) * WT_BYTES;
        const bool has_next = j + 1 < a.nconv;
        if (has_next) {                                               // land under this conv's MFMAs
            dma_x(j + 2, aout);
            dma_w(j + 1, wt0 + ((j + 1) & 1) * WT_BYTES);
        }
        // weight fragments for the 4 N-tiles x 3 taps x 2 k-steps: registers, reused for every frame
        bf16x8 wf[4][6];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = ni * 16 + li;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int c = (s & 1) * 4 + g;                        // chunk within the tap: ks*4 + g
                wf[ni][s] = *reinterpret_cast<const bf16x8*>(wl + (s >> 1) * 8192 + n * 128 + ((c ^ (n & 7)) << 4));
            }
        }
        f32x4 acc[R2_ROUNDS][4];
#pragma unroll
        for (int r = 0; r < R2_ROUNDS; ++r) {
            const int mt = wv + r * R2_WAVES;
            const int t = min(mt * 16 + li, a.TP - 1);                // tiles past the utterance compute on row TP-1: discarded
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[r][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                int ts = reflect_idx(t + (tap - 1) * a.dil, a.T);
                ts = min(max(ts, 0), a.TP - 1);                        // rows >= T only feed discarded outputs
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int c = ks * 4 + g;
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(ain + ts * 128 + ((c ^ (ts & 7)) << 4));
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[r][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][tap * 2 + ks], xf, acc[r][ni], 0, 0, 0);
                }
            }
        }
        // own DMAs landed; after the barrier everyone's have, and every wave is done reading `ain`
#ifdef VP_TIMING
        if (nst < 32) stamps[nst++] = wall_clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef VP_TIMING
        if (nst < 32) stamps[nst++] = wall_clock64();
#endif
        const float* pj = prm + j * 192;
#pragma unroll
        for (int r = 0; r < R2_ROUNDS; ++r) {
            const int mt = wv + r * R2_WAVES;
            if (mt < ntile) {                                          // wave-uniform
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
                    // channel nb lives in 16-B chunk nb/8, byte (nb % 8) * 2 of row t
                    const int pos = t * 128 + (((nb >> 3) ^ (t & 7)) << 4) + (nb & 7) * 2;
                    bf16x4 o;
                    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                    *reinterpret_cast<bf16x4*>(ain + pos) = o;         // y_{j+1}: staged in the dead input buffer
                    if (has_next) {                                    // next input = y_{j+1} + x_{j+2}, in place
                        const bf16x4 xn = *reinterpret_cast<const bf16x4*>(aout + pos);
                        bf16x4 s;
                        s[0] = (bf16_t)(v[0] + (float)xn[0]); s[1] = (bf16_t)(v[1] + (float)xn[1]);
                        s[2] = (bf16_t)(v[2] + (float)xn[2]); s[3] = (bf16_t)(v[3] + (float)xn[3]);
                        *reinterpret_cast<bf16x4*>(aout + pos) = s;
                    }
                }
            }
        }
        __syncthreads();
#ifdef VP_TIMING
        if (nst < 32) stamps[nst++] = wall_clock64();
#endif
        // y_{j+1} out: 8 lanes per frame, 128 contiguous bytes
        {
            bf16_t* ybase = a.r2 + row0 * a.C + (j + 1) * R2_W;
            for (int i = tid; i < a.T * 8; i += R2_THREADS) {
                const int t = i >> 3, c = i & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(ain + t * 128 + ((c ^ (t & 7)) << 4));
                *reinterpret_cast<uint4*>(ybase + (size_t)t * a.C + c * 8) = v;
            }
        }
        __syncthreads();                                              // `ain` is the next conv's DMA target
#ifdef VP_TIMING
        if (nst < 32) stamps[nst++] = wall_clock64();
#endif
    }
#ifdef VP_TIMING
    if (a.dbg && tid == 0)
        for (int i = 0; i < 32; ++i) a.dbg[(size_t)b * 32 + i] = i < nst ? stamps[i] : 0;
#endif
}

}  // namespace

#ifdef VP_TIMING
static unsigned long long* g_res2_dbg = nullptr;
extern "C" void vp_dbg_res2_buffer(void* p) { g_res2_dbg = (unsigned long long*)p; }
#endif

// Returns VP_EUNSUP when the shape does not fit this kernel (caller falls back to per-conv launches).
int vp_res2_chain_bf16(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T,
                       int C, int width, hipStream_t st) {
    if (width != R2_W || nconv < 1 || nconv > VP_MAX_RES2) return VP_EUNSUP;
    const int TP = (T + 15) / 16 * 16;
    const size_t smem = (size_t)2 * TP * 128 + 2 * (size_t)R2_W * R2_K * 2 + (size_t)nconv * 192 * 4;
    if (smem > 160 * 1024 || T < 2 || TP / 16 > R2_WAVES * R2_ROUNDS) return VP_EUNSUP;
    const int dil = layers[0].dil;
    Res2Args a;
    memset(&a, 0, sizeof(a));
    for (int j = 0; j < nconv; ++j) {
        const vp_tdnn_layer& L = layers[j];
        if (L.kw != 3 || L.cin != R2_W || L.cout != R2_W || L.dil != dil || !L.bias || !L.bn_scale || !L.bn_shift)
            return VP_EUNSUP;
        a.w[j] = (const bf16_t*)L.w; a.bias[j] = L.bias; a.scale[j] = L.bn_scale; a.shift[j] = L.bn_shift;
    }
    if (dil >= T) return VP_EUNSUP;
    a.t1 = (const bf16_t*)t1; a.r2 = (bf16_t*)r2; a.T = T; a.C = C; a.nconv = nconv; a.dil = dil; a.TP = TP;
    const unsigned long long t1b = (unsigned long long)B * T * C * 2;
    if (t1b >= 0xffffff00ull || (C * 2) % 16 || (reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(r2)) & 15) return VP_EUNSUP;
    a.t1_bytes = (unsigned)t1b;
#ifdef VP_TIMING
    a.dbg = g_res2_dbg;
#endif
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(res2_chain_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(res2_chain_kernel, dim3(B), dim3(R2_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "res2_chain");
    return VP_OK;
}
