// Res2Net chain of one SE-Res2 block as ONE kernel: one workgroup per utterance.
//
// Replaces Res2NetBlock.forward (ppvector/models/ecapa_tdnn.py:36-47): y_0 = x_0, y_1 = f_1(x_1),
// y_j = f_j(x_j + y_{j-1}), f_j = TDNNBlock(w -> w, k3, dilation d) = BN(ReLU(conv(reflect-pad))),
// then concat.  As separate launches this is 7 serially dependent M x 64 x 192 GEMMs per block (21 per
// forward, launch-shaped: 19 us each on MI355X).  Here the whole chain runs out of LDS:
//   - the utterance's current input (T x w bf16) lives in LDS, ping-ponged with the next input
//     (y_j + x_{j+1}) that the epilogue of conv j writes; reflect padding is an LDS row index;
//   - the 64 x 192 weights of conv j sit in LDS (double-buffered: conv j+1's weights are fetched while
//     conv j runs) and each wave keeps its 24 weight fragments in registers across all its frames;
//   - y_j goes to global memory once (slice j of the concat buffer), x_{j+1} is read once.
// MFMA-bound in principle (2*T*64*192 flop per conv) but short: what it removes is 7 launches,
// 7 HBM/L2 round trips of the activations and the aux buffers.  Used when T*w*2 bytes * 2 + weights
// fit in LDS and each wave owns <= 3 frame tiles (T <= 384 at w = 64); longer utterances fall back
// to the per-conv path.
#include "common.h"

namespace {

constexpr int R2_W = 64;            // chunk width (channels per Res2 scale slice)
constexpr int R2_K = 3 * R2_W;      // K of one conv
constexpr int R2_THREADS = 512;
constexpr int R2_WAVES = R2_THREADS / 64;
constexpr int R2_ROUNDS = 3;        // frame tiles (16 frames) per wave: T <= 16 * 8 * 3 = 384

struct Res2Args {
    const bf16_t* t1;        // (B*T, C): tdnn1 output, x_j = columns [j*w, (j+1)*w)
    bf16_t* r2;              // (B*T, C): concat of y_j (slice 0 already written by tdnn1's epilogue)
    const bf16_t* w[VP_MAX_RES2];       // [64][192] bf16, k = tap*64 + c
    const float* bias[VP_MAX_RES2];
    const float* scale[VP_MAX_RES2];
    const float* shift[VP_MAX_RES2];
    int T, C, nconv, dil, TP;          // TP = T rounded up to 16
};

__device__ __forceinline__ int reflect_idx(int t, int T) {
    t = t < 0 ? -t : t;
    return t >= T ? 2 * (T - 1) - t : t;
}

__global__ __launch_bounds__(R2_THREADS, 1) void res2_chain_kernel(Res2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: act[2][TP][128 B] | wts[2][64][384 B]
    const int act_bytes = a.TP * 128;
    char* act0 = smem;
    char* wt0 = smem + 2 * act_bytes;
    constexpr int WT_BYTES = R2_W * R2_K * 2;     // 24576

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const size_t row0 = (size_t)b * a.T;

    // 64 rows x 24 chunks of 16 B = 1536 chunks = 3 per thread; chunk c of row n is stored at chunk
    // (c ^ (n & 7)) within its group of 8.  Fetch (global -> registers) and store (registers -> LDS)
    // are split so the fetch of conv j+1's weights flies under conv j's MFMAs.
    uint4 wpre[3];
    auto fetch_weights = [&](int j) {
        const char* src = reinterpret_cast<const char*>(a.w[j]);
#pragma unroll
        for (int u = 0; u < 3; ++u) wpre[u] = *reinterpret_cast<const uint4*>(src + (size_t)(tid + u * R2_THREADS) * 16);
    };
    auto store_weights = [&](int buf) {
        char* dst = wt0 + buf * WT_BYTES;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + u * R2_THREADS;
            const int n = i / 24, c = i - n * 24;
            *reinterpret_cast<uint4*>(dst + n * 384 + (((c & ~7) | ((c ^ n) & 7)) << 4)) = wpre[u];
        }
    };

    // stage x_1 into act[0] (rows >= T zero-filled), weights of conv 0 into wts[0]
    fetch_weights(0);
#pragma unroll 4
    for (int i = tid; i < a.TP * 8; i += R2_THREADS) {
        const int t = i >> 3, c = i & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < a.T) v = *reinterpret_cast<const uint4*>(a.t1 + (row0 + t) * a.C + R2_W + c * 8);
        *reinterpret_cast<uint4*>(act0 + t * 128 + ((c ^ (t & 7)) << 4)) = v;
    }
    store_weights(0);

    const int ntile = a.TP / 16;
    // per-conv epilogue parameters live in LDS (read as float4 per N-tile) to keep VGPRs for prefetch
    float* prm = reinterpret_cast<float*>(wt0 + 2 * WT_BYTES);          // [nconv][3][64]
    for (int i = tid; i < a.nconv * 192; i += R2_THREADS) {
        const int j = i / 192, k = i - j * 192, which = k >> 6, n = k & 63;
        prm[i] = which == 0 ? a.bias[j][n] : (which == 1 ? a.scale[j][n] : a.shift[j][n]);
    }
    // x_{j+2} rows of this wave's frames, prefetched ONE CONV AHEAD (global latency ~ a whole conv)
    bf16x4 xcur[R2_ROUNDS][4], xnxt[R2_ROUNDS][4];
    auto fetch_x = [&](int j, bf16x4 (&dst)[R2_ROUNDS][4]) {           // x chunk used by conv j's epilogue
        const int col = (j + 2) * R2_W;
#pragma unroll
        for (int r = 0; r < R2_ROUNDS; ++r) {
            const int t = min((wv + r * R2_WAVES) * 16 + li, a.T - 1);
            const bf16_t* xp = a.t1 + (row0 + t) * a.C + col;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) dst[r][ni] = *reinterpret_cast<const bf16x4*>(xp + ni * 16 + g * 4);
        }
    };
    if (a.nconv > 1) fetch_x(0, xcur);
    __syncthreads();

    for (int j = 0; j < a.nconv; ++j) {
        const char* ain = act0 + (j & 1) * act_bytes;
        char* aout = act0 + ((j + 1) & 1) * act_bytes;
        const char* wl = wt0 + (j & 1) * WT_BYTES;
        const bool has_next = j + 1 < a.nconv;
        if (has_next) fetch_weights(j + 1);                           // in flight under this conv's MFMAs
        if (j + 2 < a.nconv) fetch_x(j + 1, xnxt);
        // weight fragments for the 4 N-tiles x 3 taps x 2 k-steps: registers, reused for every frame
        bf16x8 wf[4][6];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = ni * 16 + li;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int c = s * 4 + g;                              // chunk = tap*8 + ks*4 + g
                wf[ni][s] = *reinterpret_cast<const bf16x8*>(wl + n * 384 + (((c & ~7) | ((c ^ n) & 7)) << 4));
            }
        }
        const float* pj = prm + j * 192;
        const int slice = (j + 1) * R2_W;                             // y_{j+1} in the reference's numbering
#pragma unroll
        for (int r = 0; r < R2_ROUNDS; ++r) {
            const int mt = wv + r * R2_WAVES;
            if (mt >= ntile) break;                                    // wave-uniform
            const int t = mt * 16 + li;                                // this lane's frame (B operand column)
            f32x4 acc[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][tap * 2 + ks], xf, acc[ni], 0, 0, 0);
                }
            }
            // epilogue: lane holds channels nb..nb+3 of frame t for each N-tile
            if (t < a.T) {
                bf16_t* yrow = a.r2 + (row0 + t) * a.C + slice;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int nb = ni * 16 + g * 4;
                    const float4 bb = *reinterpret_cast<const float4*>(pj + nb);
                    const float4 ss = *reinterpret_cast<const float4*>(pj + 64 + nb);
                    const float4 hh = *reinterpret_cast<const float4*>(pj + 128 + nb);
                    float v[4];
                    v[0] = fmaxf(acc[ni][0] + bb.x, 0.f) * ss.x + hh.x;
                    v[1] = fmaxf(acc[ni][1] + bb.y, 0.f) * ss.y + hh.y;
                    v[2] = fmaxf(acc[ni][2] + bb.z, 0.f) * ss.z + hh.z;
                    v[3] = fmaxf(acc[ni][3] + bb.w, 0.f) * ss.w + hh.w;
                    bf16x4 o;
                    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                    *reinterpret_cast<bf16x4*>(yrow + nb) = o;
                    if (has_next) {
                        const bf16x4 xn = xcur[r][ni];
                        bf16x4 s;
                        s[0] = (bf16_t)(v[0] + (float)xn[0]); s[1] = (bf16_t)(v[1] + (float)xn[1]);
                        s[2] = (bf16_t)(v[2] + (float)xn[2]); s[3] = (bf16_t)(v[3] + (float)xn[3]);
                        // channel nb lives in 16-B chunk nb/8, byte (nb % 8) * 2 of row t
                        *reinterpret_cast<bf16x4*>(aout + t * 128 + (((nb >> 3) ^ (t & 7)) << 4) + (nb & 7) * 2) = s;
                    }
                }
            }
        }
        if (has_next) store_weights((j + 1) & 1);
#pragma unroll
        for (int r = 0; r < R2_ROUNDS; ++r)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) xcur[r][ni] = xnxt[r][ni];
        __syncthreads();
    }
}

}  // namespace

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
    static bool attr_set = false;
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(res2_chain_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(res2_chain_kernel, dim3(B), dim3(R2_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "res2_chain");
    return VP_OK;
}
