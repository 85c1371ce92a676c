// Class-tiled cosine classifier + AAM-softmax cross-entropy: no (B, C) tensor.
//
// Replaces, fused, SpeakerIdentification.forward 'Cosine' (ppvector/models/fc.py:41-53) + AAMLoss.forward
// (ppvector/loss/aamloss.py:28-47) for the evaluation / forward value: the unfused path (head.hip) writes the (B, C) f32 cosine
// matrix and the loss kernel reads it back -- 102 MB each way for BASELINE configs[4] (200 000 classes x 128 utterances per GPU),
// and four launches around it (row norms, column norms, GEMM, loss).  Here a workgroup owns a tile of 64 classes:
//   W tile (D x 64 f32, read ONCE from HBM: the whole head is streamed exactly once) -> LDS, its column norms from the same bytes;
//   per block of 64 utterances: embeddings -> LDS (+ row norms), cos = (E W) * rinv * cinv on the f32 matrix cores
//   (v_mfma_f32_16x16x4_f32: the exact-f32 arithmetic of the unfused path), margin on the target column, scale, and the block's
//   per-row online-softmax partials (max, sum exp, sum of logits) over its 64 classes -> part[3][B][tiles].
// A second small kernel merges the partials per row (log-sum-exp, label smoothing) and a third takes the mean.
// Roofline: f32 MFMA (2 B C D flops at 157 TFLOP/s) for large B, else HBM (4 D C bytes of W); 128 x 200 000 x 192: 9.8 GFLOP = 63 us.
#include "common.h"

#include <math.h>

namespace {

constexpr int HT_CT = 64;            // classes per tile
constexpr int HT_RB = 64;            // utterances per block
constexpr int HT_DMAX = 256;         // embedding width supported (192 in every shipped config)
constexpr int HT_SW = HT_CT + 2;     // LDS strides = 2 mod 32 banks: see the operand reads below
typedef __attribute__((ext_vector_type(4))) float v4f;

struct HeadTileArgs {
    const float* emb; const float* W; const long long* labels;
    float* part;                     // [3][B][tiles]: max, sum exp(out - max), sum out
    float* tgt;                      // [B] scaled, margined target logit
    float* cinv;                     // [C] column inverse norms (by-product; NULL = not wanted)
    int B, D, C, tiles, SE;          // SE = D + 2
    float cos_m, sin_m, th, mmm, scale; int easy;
    const float* mt;
};

// sum / max over the 16 lanes of a DPP row; every lane ends with the result
__device__ __forceinline__ float row16_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, false));
    return x;
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, false)));
    return x;
}

__global__ __launch_bounds__(256) void head_tile_fwd_kernel(HeadTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                   // [D][HT_SW]
    float* Es = Ws + a.D * HT_SW;                                 // [HT_RB][SE]
    float* cinv_s = Es + HT_RB * a.SE;                            // [64]
    float* rinv_s = cinv_s + HT_CT;                               // [64]
    float* red = rinv_s + HT_RB;                                  // [3][4 waves][64 rows]
    int* lab_s = reinterpret_cast<int*>(red + 3 * 4 * 64);        // [64] labels of the row block (-1 past B)
    if (a.mt) { a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, c0 = tile * HT_CT;
    // ---- W tile: rows k, 64 consecutive classes = 256 contiguous bytes per row; thread t: class chunk (t & 15) * 4, rows t >> 4, + 16, ...
    {
        const int cq = (tid & 15) * 4, k0 = tid >> 4;
        for (int k = k0; k < a.D; k += 16) {
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            const float* src = a.W + (size_t)k * a.C + c0 + cq;
            if (c0 + cq + 3 < a.C && ((a.C & 3) == 0)) v = *reinterpret_cast<const v4f*>(src);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c0 + cq + e < a.C) v[e] = src[e];
            }
            float* dst = Ws + k * HT_SW + cq;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
    }
    __syncthreads();
    // column norms of the tile: thread t -> class t & 63, quarter t >> 6 of the rows; fixed-order 4-way sum
    {
        const int c = tid & 63, q = tid >> 6;
        float s = 0.f;
        for (int k = q; k < a.D; k += 4) { const float v = Ws[k * HT_SW + c]; s += v * v; }
        red[q * 64 + c] = s;
        __syncthreads();
        if (tid < 64) {
            const float t = red[c] + red[64 + c] + red[128 + c] + red[192 + c];
            const float inv = 1.f / fmaxf(sqrtf(t), 1e-12f);
            cinv_s[c] = inv;
            if (a.cinv && c0 + c < a.C) a.cinv[c0 + c] = inv;
        }
    }
    // wave wv owns the 16 classes [16 wv, 16 wv + 16) of the tile, against every row block
    const int cw = wv * 16;
    // small heads (few class tiles): the row blocks are spread over blockIdx.y as well (the W tile is then read once per row split)
    for (int b0 = blockIdx.y * HT_RB; b0 < a.B; b0 += gridDim.y * HT_RB) {
        __syncthreads();                                          // previous block's Es / red readers are done
        if (tid < HT_RB) lab_s[tid] = b0 + tid < a.B ? (int)a.labels[b0 + tid] : -1;
        // ---- embeddings of 64 utterances: thread t: k chunk (t % (D/4)) ... plain strided copy, 16-byte loads
        {
            const int per_row = a.D >> 2;
            for (int i = tid; i < HT_RB * per_row; i += 256) {
                const int r = i / per_row, kq = (i - r * per_row) * 4;
                v4f v = v4f{0.f, 0.f, 0.f, 0.f};
                if (b0 + r < a.B) v = *reinterpret_cast<const v4f*>(a.emb + (size_t)(b0 + r) * a.D + kq);
                float* dst = Es + r * a.SE + kq;
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        }
        __syncthreads();
        {
            const int r = tid & 63, q = tid >> 6;
            float s = 0.f;
            for (int k = q; k < a.D; k += 4) { const float v = Es[r * a.SE + k]; s += v * v; }
            red[q * 64 + r] = s;
            __syncthreads();
            if (tid < 64) rinv_s[r] = 1.f / fmaxf(sqrtf(red[r] + red[64 + r] + red[128 + r] + red[192 + r]), 1e-12f);
            __syncthreads();
        }
        // ---- cos tile: C[row = utterance][col = class]; A = E (lane: row li, k g), B = W (lane: k g, class li)
        v4f acc[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = v4f{0.f, 0.f, 0.f, 0.f};
        const float* wp = Ws + g * HT_SW + cw + li;               // banks g 2 + li: two-way at worst
        const float* ep = Es + li * a.SE + g;                     // banks li 2 + g: conflict-free
        const int se16 = 16 * a.SE;
        for (int k = 0; k < a.D; k += 16) {                       // four k-steps of operands in flight ahead of their MFMAs (D % 16 handled below)
            float bv[4], av[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 4 * u < a.D ? k + 4 * u : 0;   // past D: re-read step 0 and multiply by zero
                bv[u] = k + 4 * u < a.D ? wp[kk * HT_SW] : 0.f;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) av[u][mi] = ep[mi * se16 + kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mi], bv[u], acc[mi], 0, 0, 0);
        }
        // ---- epilogue: lane holds rows mi 16 + g 4 + r, class cw + li
        const int c = c0 + cw + li;
        const bool cvalid = c < a.C;
        const float ci = cinv_s[cw + li];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mi * 16 + g * 4 + r;
                const int b = b0 + row;
                const int y = lab_s[row];
                const float cs = acc[mi][r] * rinv_s[row] * ci;
                float o = cs;
                if (c == y) {
                    const float sine = sqrtf(1.f - cs * cs);
                    const float phi = cs * a.cos_m - sine * a.sin_m;
                    o = a.easy ? (cs > 0.f ? phi : cs) : (cs > a.th ? phi : cs - a.mmm);
                }
                o *= a.scale;
                if (c == y) a.tgt[b] = o;
                const float ov = cvalid ? o : -INFINITY;
                const float m = row16_max(ov);                    // over this wave's 16 classes
                const float s = row16_sum(cvalid ? expf(o - m) : 0.f);
                const float so = row16_sum(cvalid ? o : 0.f);
                if (li == 0) { red[(0 * 4 + wv) * 64 + row] = m; red[(1 * 4 + wv) * 64 + row] = s; red[(2 * 4 + wv) * 64 + row] = so; }
            }
        }
        __syncthreads();
        if (tid < 64 && b0 + tid < a.B) {                         // merge the four waves' 16-class partials (fixed order)
            const int row = tid;
            float M = fmaxf(fmaxf(red[0 * 64 + row], red[1 * 64 + row]), fmaxf(red[2 * 64 + row], red[3 * 64 + row]));
            float S = 0.f, O = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw = red[(0 * 4 + w) * 64 + row];
                S += mw == -INFINITY ? 0.f : red[(1 * 4 + w) * 64 + row] * expf(mw - M);
                O += red[(2 * 4 + w) * 64 + row];
            }
            const size_t o = (size_t)(b0 + row) * a.tiles + tile;
            a.part[o] = M;
            a.part[(size_t)a.B * a.tiles + o] = S;
            a.part[2 * (size_t)a.B * a.tiles + o] = O;
        }
    }
}

// per row: merge the tiles' partials -> lse, loss row
__global__ __launch_bounds__(256) void head_tile_merge_kernel(const float* part, const float* tgt, int B, int C, int tiles, float ls,
                                                              float* lse_out, float* row_loss) {
    __shared__ float sm[3][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* pm = part + (size_t)b * tiles;
    const float* ps = part + (size_t)B * tiles + (size_t)b * tiles;
    const float* po = part + 2 * (size_t)B * tiles + (size_t)b * tiles;
    float mx = -INFINITY, se = 0.f, so = 0.f;
    for (int t = tid; t < tiles; t += 256) {
        const float m = pm[t], s = ps[t];
        so += po[t];
        if (m > mx) { se = se * expf(mx - m) + s; mx = m; }
        else if (m != -INFINITY) se += s * expf(m - mx);
    }
    const float wmx = vp_wave_max(mx);
    se = vp_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
    so = vp_wave_sum(so);
    if (lane == 0) { sm[0][wv] = wmx; sm[1][wv] = se; sm[2][wv] = so; }
    __syncthreads();
    if (tid == 0) {
        const float M = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
        float S = 0.f, O = 0.f;
        for (int w = 0; w < 4; ++w) {
            S += (sm[0][w] == -INFINITY) ? 0.f : sm[1][w] * expf(sm[0][w] - M);
            O += sm[2][w];
        }
        const float lse = M + logf(S);
        if (lse_out) lse_out[b] = lse;
        row_loss[b] = (1.f - ls) * (lse - tgt[b]) + ls * (lse - O / (float)C);
    }
}

__global__ __launch_bounds__(256) void head_mean_kernel(const float* v, int n, float* out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = vp_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sm[0] + sm[1] + sm[2] + sm[3]) / (float)n;
}

}  // namespace

extern "C" {

size_t vp_cosine_aam_tiled_workspace_bytes(int B, int D, int C) {
    (void)D;
    const size_t tiles = (size_t)(C + HT_CT - 1) / HT_CT;
    return vp_align_up(3 * (size_t)B * tiles * 4, 256) + vp_align_up((size_t)B * 4, 256);
}

int vp_cosine_aam_tiled_fwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                            float scale, float label_smoothing, int easy_margin, float* loss, float* row_loss, float* lse,
                            float* cinv, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !labels || !loss || !row_loss || B <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_aam_tiled: bad arguments");
    if (D < 4 || D > HT_DMAX || (D & 3)) VP_FAIL(ctx, VP_EUNSUP, "cosine_aam_tiled: embedding width %d (multiples of 4 up to %d)", D, HT_DMAX);
    if (!ws || ws_bytes < vp_cosine_aam_tiled_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_aam_tiled: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    HeadTileArgs a;
    a.emb = emb; a.W = W; a.labels = (const long long*)labels;
    a.tiles = (C + HT_CT - 1) / HT_CT;
    a.part = (float*)ws;
    a.tgt = (float*)((char*)ws + vp_align_up(3 * (size_t)B * a.tiles * 4, 256));
    a.cinv = cinv;
    a.B = B; a.D = D; a.C = C; a.SE = D + 2;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.easy = easy_margin; a.mt = ctx->margin_table;
    const int smem = (D * HT_SW + HT_RB * (D + 2) + HT_CT + HT_RB + 3 * 4 * 64 + HT_RB) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(head_tile_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (HT_DMAX * HT_SW + HT_RB * (HT_DMAX + 2) + HT_CT + HT_RB + 3 * 4 * 64 + HT_RB) * 4));
        attr_set = true;
    }
    const int rblocks = (B + HT_RB - 1) / HT_RB;
    int ysplit = 1;
    while (ysplit < rblocks && a.tiles * ysplit < 192) ++ysplit;
    hipLaunchKernelGGL(head_tile_fwd_kernel, dim3(a.tiles, ysplit), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "head_tile_fwd");
    hipLaunchKernelGGL(head_tile_merge_kernel, dim3(B), dim3(256), 0, st, a.part, a.tgt, B, C, a.tiles, label_smoothing, lse, row_loss);
    VP_LAUNCH_CHECK(ctx, "head_tile_merge");
    hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
    VP_LAUNCH_CHECK(ctx, "head_mean");
    return VP_OK;
}

}  // extern "C"
