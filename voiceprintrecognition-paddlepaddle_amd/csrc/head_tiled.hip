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
constexpr int HT_RB = 64;            // utterances per block (backward kernel)
constexpr int HT_RF = 32;            // utterances per block of the forward kernel: 79 KB of LDS, two workgroups per CU
constexpr int HT_DMAX = 256;         // embedding width supported (192 in every shipped config)
constexpr int HT_SW = HT_CT + 2;     // LDS strides = 2 mod 32 banks: see the operand reads below
typedef __attribute__((ext_vector_type(4))) float v4f;

struct HeadTileArgs {
    const float* emb; const float* W; const long long* labels;
    float* part;                     // [5][B][tiles]: max, sum exp(out - max), sum out, best cosine, its class (as float bits of an int)
    float* tgt;                      // [B] scaled, margined target logit
    float* cinv;                     // [C] column inverse norms (by-product; NULL = not wanted)
    const float* rinv;               // [B] row inverse norms of the embeddings (vp_row_inv_norm ahead of the launch)
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


typedef __attribute__((ext_vector_type(4))) unsigned int ht_u32x4;
constexpr unsigned HT_OOB = 0xfffffff0u;

// W tile (D rows k, 64 classes from c0) -> Ws[k][HT_SW].  Thread t: class chunk (t & 15) * 4 of rows t >> 4, + 16, ...: all loads
// are unconditional buffer loads issued together (a conditional load compiles to a branch + vmcnt(0) per load: twelve serialised
// HBM round trips per thread made the first version of these kernels 40 us per tile); what must read as zero -- classes past C --
// is an out-of-range offset.  C % 4 != 0 (never with the shipped heads) takes scalar loads.
template <int NR>                    // rows per thread = ceil(D / 16), compile-time bound on the loads in flight
__device__ __forceinline__ void ht_load_w(const float* W, int D, int C, int c0, int tid, float* Ws) {
    const int cq = (tid & 15) * 4, k0 = tid >> 4;
    if ((C & 3) == 0) {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (unsigned)((size_t)D * C * 4), 0x00020000);
        const bool cok = c0 + cq < C;
        ht_u32x4 v[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int k = k0 + 16 * i;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, (cok && k < D) ? (unsigned)(((size_t)k * C + c0 + cq) * 4) : HT_OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int k = k0 + 16 * i;
            if (k < D) {
                float* dst = Ws + k * HT_SW + cq;
                const ht_u32x4 t = v[i];
                dst[0] = __builtin_bit_cast(float, (unsigned)t[0]); dst[1] = __builtin_bit_cast(float, (unsigned)t[1]);
                dst[2] = __builtin_bit_cast(float, (unsigned)t[2]); dst[3] = __builtin_bit_cast(float, (unsigned)t[3]);
            }
        }
    } else {
        for (int k = k0; k < D; k += 16)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[k * HT_SW + cq + e] = c0 + cq + e < C ? W[(size_t)k * C + c0 + cq + e] : 0.f;
    }
}

// embeddings of utterances [b0, b0 + 64) -> Es[r][SE]; rows past B read as zero (out-of-range offsets)
template <int NE, int ROWS = HT_RB>  // NE: 16-byte chunks per thread = ceil(ROWS * D / 4 / 256)
__device__ __forceinline__ void ht_issue_e(const float* emb, int B, int D, int b0, int tid, ht_u32x4 (&v)[NE]) {
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(emb), 0, (unsigned)((size_t)B * D * 4), 0x00020000);
    const int per_row = D >> 2, total = ROWS * per_row;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / per_row, kq = (idx - r * per_row) * 4;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, (idx < total && b0 + r < B) ? (unsigned)(((size_t)(b0 + r) * D + kq) * 4) : HT_OOB, 0, 0);
    }
}
template <int NE, int ROWS = HT_RB>
__device__ __forceinline__ void ht_commit_e(int D, int SE, int tid, float* Es, const ht_u32x4 (&v)[NE]) {
    const int per_row = D >> 2, total = ROWS * per_row;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int idx = tid + 256 * i;
        if (idx < total) {
            const int r = idx / per_row, kq = (idx - r * per_row) * 4;
            float* dst = Es + r * SE + kq;
            const ht_u32x4 t = v[i];
            dst[0] = __builtin_bit_cast(float, (unsigned)t[0]); dst[1] = __builtin_bit_cast(float, (unsigned)t[1]);
            dst[2] = __builtin_bit_cast(float, (unsigned)t[2]); dst[3] = __builtin_bit_cast(float, (unsigned)t[3]);
        }
    }
}
template <int NE, int ROWS = HT_RB>
__device__ __forceinline__ void ht_load_e(const float* emb, int B, int D, int SE, int b0, int tid, float* Es) {
    ht_u32x4 v[NE];
    ht_issue_e<NE, ROWS>(emb, B, D, b0, tid, v);
    ht_commit_e<NE, ROWS>(D, SE, tid, Es, v);
}

__global__ __launch_bounds__(256, 2) void head_tile_fwd_kernel(HeadTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                   // [D][HT_SW]
    float* Es = Ws + a.D * HT_SW;                                 // [HT_RF][SE]
    float* cinv_s = Es + HT_RF * a.SE;                            // [64]
    float* rinv_s = cinv_s + HT_CT;                               // [HT_RF]
    float* red = rinv_s + HT_RF;                                  // [5][4 waves][HT_RF rows] (and [4][64] / [8][HT_RF] for the norms)
    int* lab_s = reinterpret_cast<int*>(red + 5 * 4 * HT_RF);     // [HT_RF] labels of the row block (-1 past B)
    if (a.mt) { a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, c0 = tile * HT_CT;
    ht_load_w<HT_DMAX / 16>(a.W, a.D, a.C, c0, tid, Ws);
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
    // the embeddings of a row block are fetched into registers while the previous block is multiplied (they come from L2: every
    // class tile reads the same B x D floats), and their inverse norms were taken once, ahead of the launch
    constexpr int NE = HT_RF * HT_DMAX / 4 / 256;
    ht_u32x4 ev[NE];
    if ((int)(blockIdx.y * HT_RF) < a.B) ht_issue_e<NE, HT_RF>(a.emb, a.B, a.D, blockIdx.y * HT_RF, tid, ev);
    for (int b0 = blockIdx.y * HT_RF; b0 < a.B; b0 += gridDim.y * HT_RF) {
        __syncthreads();                                          // previous block's Es / red readers are done
        if (tid < HT_RF) {
            lab_s[tid] = b0 + tid < a.B ? (int)a.labels[b0 + tid] : -1;
            rinv_s[tid] = b0 + tid < a.B ? a.rinv[b0 + tid] : 0.f;
        }
        ht_commit_e<NE, HT_RF>(a.D, a.SE, tid, Es, ev);
        __syncthreads();
        if (b0 + (int)gridDim.y * HT_RF < a.B) ht_issue_e<NE, HT_RF>(a.emb, a.B, a.D, b0 + gridDim.y * HT_RF, tid, ev);
        // ---- cos tile: C[row = utterance][col = class]; A = E (lane: row li, k g), B = W (lane: k g, class li)
        constexpr int MI = HT_RF / 16;
        v4f acc[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = v4f{0.f, 0.f, 0.f, 0.f};
        const float* wp = Ws + g * HT_SW + cw + li;               // banks g 2 + li: two-way at worst
        const float* ep = Es + li * a.SE + g;                     // banks li 2 + g: conflict-free
        const int se16 = 16 * a.SE;
        for (int k = 0; k < a.D; k += 16) {                       // four k-steps of operands in flight ahead of their MFMAs (D % 16 handled below)
            float bv[4], av[4][MI];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + 4 * u < a.D ? k + 4 * u : 0;   // past D: re-read step 0 and multiply by zero
                bv[u] = k + 4 * u < a.D ? wp[kk * HT_SW] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[u][mi] = ep[mi * se16 + kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mi], bv[u], acc[mi], 0, 0, 0);
        }
        // ---- epilogue: lane holds rows mi 16 + g 4 + r, class cw + li
        const int c = c0 + cw + li;
        const bool cvalid = c < a.C;
        const float ci = cinv_s[cw + li];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
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
                // prediction = argmax of the UN-margined cosine (what trainer.py:233-236 takes from outputs["logits"]); first index wins ties
                const float cv = cvalid ? cs : -INFINITY;
                const float bc = row16_max(cv);
                const float bi = -row16_max(cv == bc ? -(float)c : -INFINITY);     // smallest class index holding the maximum (exact below 2^24)
                if (li == 0) {
                    red[(0 * 4 + wv) * HT_RF + row] = m; red[(1 * 4 + wv) * HT_RF + row] = s; red[(2 * 4 + wv) * HT_RF + row] = so;
                    red[(3 * 4 + wv) * HT_RF + row] = bc; red[(4 * 4 + wv) * HT_RF + row] = bi;
                }
            }
        }
        __syncthreads();
        if (tid < HT_RF && b0 + tid < a.B) {                      // merge the four waves' 16-class partials (fixed order)
            const int row = tid;
            float M = fmaxf(fmaxf(red[0 * HT_RF + row], red[1 * HT_RF + row]), fmaxf(red[2 * HT_RF + row], red[3 * HT_RF + row]));
            float S = 0.f, O = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw = red[(0 * 4 + w) * HT_RF + row];
                S += mw == -INFINITY ? 0.f : red[(1 * 4 + w) * HT_RF + row] * expf(mw - M);
                O += red[(2 * 4 + w) * HT_RF + row];
            }
            float BC = -INFINITY, BI = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {                           // waves hold ascending class ranges: strict > keeps the first maximum
                const float v = red[(3 * 4 + w) * HT_RF + row];
                if (v > BC) { BC = v; BI = red[(4 * 4 + w) * HT_RF + row]; }
            }
            const size_t o = (size_t)(b0 + row) * a.tiles + tile;
            const size_t st = (size_t)a.B * a.tiles;
            a.part[o] = M;
            a.part[st + o] = S;
            a.part[2 * st + o] = O;
            a.part[3 * st + o] = BC;
            a.part[4 * st + o] = BI;
        }
    }
}

// per row: merge the tiles' partials -> lse, loss row
__global__ __launch_bounds__(256) void head_tile_merge_kernel(const float* part, const float* tgt, int B, int C, int tiles, float ls,
                                                              float* lse_out, float* row_loss, int* pred) {
    __shared__ float sm[5][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* pm = part + (size_t)b * tiles;
    const float* ps = part + (size_t)B * tiles + (size_t)b * tiles;
    const float* po = part + 2 * (size_t)B * tiles + (size_t)b * tiles;
    const float* pc = part + 3 * (size_t)B * tiles + (size_t)b * tiles;
    const float* pi = part + 4 * (size_t)B * tiles + (size_t)b * tiles;
    float mx = -INFINITY, se = 0.f, so = 0.f, bc = -INFINITY, bi = 0.f;
    for (int t = tid; t < tiles; t += 256) {
        const float m = pm[t], s = ps[t];
        so += po[t];
        if (m > mx) { se = se * expf(mx - m) + s; mx = m; }
        else if (m != -INFINITY) se += s * expf(m - mx);
        const float v = pc[t];
        if (v > bc) { bc = v; bi = pi[t]; }                       // ascending tiles per thread: the first maximum stays
    }
    // best cosine over the workgroup: value first, then the smallest class index among equal values
    const float wbc = vp_wave_max(bc);
    const float wbi = -vp_wave_max(bc == wbc ? -bi : -INFINITY);
    if (lane == 0) { sm[3][wv] = wbc; sm[4][wv] = wbi; }
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
        if (pred) {
            float BC = -INFINITY, BI = 0.f;
            for (int w = 0; w < 4; ++w)
                if (sm[3][w] > BC || (sm[3][w] == BC && sm[4][w] < BI)) { BC = sm[3][w]; BI = sm[4][w]; }
            pred[b] = (int)BI;
        }
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


// ------------------------------------------------------------------------------------------------ backward, class-tiled
// d loss / d emb and d loss / d W without the (B, C) cosine and gradient matrices (unfused: both are written and re-read, 2 x 102 MB
// at 200 000 x 128, and W is streamed four times).  A workgroup walks class tiles t = blockIdx.x, + gridDim.x, ...; per tile, per
// block of 64 utterances (B <= 128: the d emb accumulators live in registers), three products on the f32 matrix cores:
//   1. cos = (E W) rinv cinv  (recomputed)  ->  G' = gs / B * scale * (softmax - q) * d margin / d cos * cinv, from the forward's
//      per-row log-sum-exp; G' -> LDS (64 x 64)
//   2. dwn'[k][c] += sum_b E[b][k] (rinv[b] G'[b][c])        contraction over utterances; 48 accumulator registers per lane
//   3. dxn[b][k]  += sum_c G'[b][c] W[k][c]                   contraction over the tile's classes; 96 accumulator registers per lane
// After the tile's row blocks: dW[k][c] = dwn' - W cinv^2 sum_k' W dwn' (the column-normalisation backward; a class column lives in one
// wave), written once, final.  After the workgroup's tiles: its dxn partial -> workspace; a second kernel adds the partials in fixed
// order and applies the row-normalisation backward.  LDS strides are = 2 mod 32 banks: every operand read of the three products
// is conflict-free or two-way.
constexpr int HT_SG = HT_CT + 2;

struct HeadBwdArgs {
    const float* emb; const float* W; const long long* labels; const float* lse;
    float* dW; float* dxn_part;      // [gridDim.x][B][D]
    int B, D, C, tiles, SE;
    float cos_m, sin_m, th, mmm, scale, ls, gscale; int easy;
    const float* mt;
};

template <int NKB>                   // D = 16 NKB (12: the 192-wide embeddings of every shipped config)
__global__ __launch_bounds__(256) void head_tile_bwd_kernel(HeadBwdArgs a) {
    constexpr int D = 16 * NKB;
    constexpr int SE = D + 2;
    constexpr int KPW = NKB / 4;     // k-blocks of dxn per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ws = reinterpret_cast<float*>(smem);                   // [D][HT_SW]
    float* Es = Ws + D * HT_SW;                                   // [64][SE]
    float* Gs = Es + HT_RB * SE;                                  // [64][HT_SG]
    float* cinv_s = Gs + HT_RB * HT_SG;                           // [64]
    float* rinv_s = cinv_s + HT_CT;                               // [64]
    float* lse_s = rinv_s + HT_RB;                                // [64]
    float* red = lse_s + HT_RB;                                   // [4][64]
    int* lab_s = reinterpret_cast<int*>(red + 4 * 64);            // [64]
    if (a.mt) { a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int cw = wv * 16;
    const float kk = a.gscale / (float)a.B * a.scale;
    const float qoff = a.ls / (float)a.C;
    const int nrb = (a.B + HT_RB - 1) / HT_RB;                    // 1 or 2 (host-checked B <= 128)
    v4f dxn[2][4][KPW];                                           // [row block][16-row block][k-block of this wave]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int kb = 0; kb < KPW; ++kb) dxn[rb][mi][kb] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int c0 = tile * HT_CT;
        __syncthreads();                                          // the previous tile's readers of Ws / cinv_s are done
        ht_load_w<NKB>(a.W, D, a.C, c0, tid, Ws);
        __syncthreads();
        {
            const int c = tid & 63, q = tid >> 6;
            float s = 0.f;
            for (int k = q; k < D; k += 4) { const float v = Ws[k * HT_SW + c]; s += v * v; }
            red[q * 64 + c] = s;
            __syncthreads();
            if (tid < 64) cinv_s[c] = 1.f / fmaxf(sqrtf(red[c] + red[64 + c] + red[128 + c] + red[192 + c]), 1e-12f);
        }
        v4f dwn[NKB];                                             // rows k-block kb, columns = this wave's 16 classes
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) dwn[kb] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb < nrb) {
                const int b0 = rb * HT_RB;
                __syncthreads();                                  // previous block's Es / Gs readers are done
                if (tid < HT_RB) {
                    const bool ok = b0 + tid < a.B;
                    lab_s[tid] = ok ? (int)a.labels[b0 + tid] : -1;
                    lse_s[tid] = ok ? a.lse[b0 + tid] : 0.f;
                }
                ht_load_e<HT_RB * D / 4 / 256>(a.emb, a.B, D, SE, b0, tid, Es);
                __syncthreads();
                {
                    const int r = tid & 63, q = tid >> 6;
                    float s = 0.f;
                    for (int k = q; k < D; k += 4) { const float v = Es[r * SE + k]; s += v * v; }
                    red[q * 64 + r] = s;
                    __syncthreads();
                    if (tid < 64) rinv_s[r] = 1.f / fmaxf(sqrtf(red[r] + red[64 + r] + red[128 + r] + red[192 + r]), 1e-12f);
                    __syncthreads();
                }
                // ---- 1: cos tile (rows = utterances, columns = this wave's 16 classes)
                v4f acc[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi] = v4f{0.f, 0.f, 0.f, 0.f};
                {
                    const float* wp = Ws + g * HT_SW + cw + li;
                    const float* ep = Es + li * SE + g;
                    for (int k = 0; k < D; k += 16) {
                        float bv[4], av[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            bv[u] = wp[(k + 4 * u) * HT_SW];
#pragma unroll
                            for (int mi = 0; mi < 4; ++mi) av[u][mi] = ep[mi * 16 * SE + k + 4 * u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mi], bv[u], acc[mi], 0, 0, 0);
                    }
                }
                // ---- G' into LDS
                {
                    const int c = c0 + cw + li;
                    const bool cvalid = c < a.C;
                    const float ci = cinv_s[cw + li];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = mi * 16 + g * 4 + r;
                            const int y = lab_s[row];
                            const float cs = acc[mi][r] * rinv_s[row] * ci;
                            float o = cs, dm = 1.f;
                            if (c == y) {
                                const float sine = sqrtf(1.f - cs * cs);
                                const float phi = cs * a.cos_m - sine * a.sin_m;
                                const bool use_phi = a.easy ? (cs > 0.f) : (cs > a.th);
                                o = use_phi ? phi : (a.easy ? cs : cs - a.mmm);
                                if (use_phi) dm = a.cos_m + cs * a.sin_m / sine;
                            }
                            o *= a.scale;
                            const float q = qoff + (c == y ? 1.f - a.ls : 0.f);
                            const float gv = (cvalid && y >= 0) ? kk * (expf(o - lse_s[row]) - q) * dm * ci : 0.f;
                            Gs[row * HT_SG + cw + li] = gv;
                        }
                }
                __syncthreads();
                // ---- 2: dwn'[k][c] += sum_b E[b][k] (rinv[b] G'[b][c]);  A: lane (k li, b g), B: lane (b g, class li)
                {
                    const float* ap = Es + g * SE + li;           // + b 4-step * SE, + kb 16
                    const float* bp = Gs + g * HT_SG + cw + li;
                    for (int b = 0; b < HT_RB; b += 4) {
                        const float bv = bp[b * HT_SG] * rinv_s[b + g];
                        float av[NKB];
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) av[kb] = ap[b * SE + kb * 16];
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) dwn[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb], bv, dwn[kb], 0, 0, 0);
                    }
                }
                // ---- 3: dxn[b][k] += sum_c G'[b][c] W[k][c];  A: lane (b li, c g), B: lane (c g, k li); this wave: k-blocks KPW wv ..
                {
                    const float* ap = Gs + li * HT_SG + g;
                    const float* bp = Ws + (wv * KPW * 16 + li) * HT_SW + g;
                    for (int c = 0; c < HT_CT; c += 4) {
                        float av[4], bv[KPW];
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) av[mi] = ap[mi * 16 * HT_SG + c];
#pragma unroll
                        for (int kb = 0; kb < KPW; ++kb) bv[kb] = bp[kb * 16 * HT_SW + c];
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                            for (int kb = 0; kb < KPW; ++kb)
                                dxn[rb][mi][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[kb], dxn[rb][mi][kb], 0, 0, 0);
                    }
                }
            }
        }
        // ---- dW tile: lane holds dwn'[k = kb 16 + g 4 + r][class cw + li]
        {
            float t = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) t += Ws[(kb * 16 + g * 4 + r) * HT_SW + cw + li] * dwn[kb][r];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            const float ci = cinv_s[cw + li];
            t *= ci * ci;
            const int c = c0 + cw + li;
            if (c < a.C) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = kb * 16 + g * 4 + r;
                        a.dW[(size_t)k * a.C + c] = dwn[kb][r] - Ws[k * HT_SW + cw + li] * t;
                    }
            }
        }
    }
    // ---- this workgroup's dxn partial: lane holds dxn[b = rb 64 + mi 16 + g 4 + r][k = (wv KPW + kb) 16 + li]
    float* out = a.dxn_part + (size_t)blockIdx.x * a.B * D;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int kb = 0; kb < KPW; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = rb * HT_RB + mi * 16 + g * 4 + r;
                    if (b < a.B) out[(size_t)b * D + (wv * KPW + kb) * 16 + li] = dxn[rb][mi][kb][r];
                }
}

// d emb[b][:] = rinv (dxn - xn <xn, dxn>),  dxn = sum over the workgroups' partials (fixed order)
__global__ __launch_bounds__(256) void head_dx_finish_kernel(const float* emb, const float* part, int nparts, int B, int D, float* demb) {
    __shared__ float sm[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float x = 0.f, gsum = 0.f;
    if (tid < D) {
        x = emb[(size_t)b * D + tid];
        for (int p = 0; p < nparts; ++p) gsum += part[((size_t)p * B + b) * D + tid];
    }
    float n2 = vp_wave_sum(x * x), dt = vp_wave_sum(x * gsum);
    if ((tid & 63) == 0) { sm[tid >> 6] = n2; }
    __syncthreads();
    const float nn = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    if ((tid & 63) == 0) { sm[tid >> 6] = dt; }
    __syncthreads();
    const float dot = sm[0] + sm[1] + sm[2] + sm[3];
    const float rinv = 1.f / fmaxf(sqrtf(nn), 1e-12f);
    if (tid < D) demb[(size_t)b * D + tid] = rinv * (gsum - x * rinv * rinv * dot);
}

}  // namespace

extern "C" {

size_t vp_cosine_aam_tiled_workspace_bytes(int B, int D, int C) {
    (void)D;
    const size_t tiles = (size_t)(C + HT_CT - 1) / HT_CT;
    return vp_align_up(5 * (size_t)B * tiles * 4, 256) + 2 * vp_align_up((size_t)B * 4, 256);       // partials, target logits, row inverse norms
}

int vp_cosine_aam_tiled_fwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                            float scale, float label_smoothing, int easy_margin, float* loss, float* row_loss, float* lse,
                            float* cinv, int* pred, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !labels || !loss || !row_loss || B <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_aam_tiled: bad arguments");
    if (C >= (1 << 24)) VP_FAIL(ctx, VP_EUNSUP, "cosine_aam_tiled: %d classes (class indices travel as exact f32)", C);
    // W and emb are addressed through 32-bit buffer offsets (D = 192: up to ~5.59 M classes)
    if ((unsigned long long)D * C * 4 >= 0xfffffff0ull || (unsigned long long)B * D * 4 >= 0xfffffff0ull)
        VP_FAIL(ctx, VP_EUNSUP, "cosine_aam_tiled: W (%d x %d) or emb larger than 4 GiB", D, C);
    if (D < 4 || D > HT_DMAX || (D & 3)) VP_FAIL(ctx, VP_EUNSUP, "cosine_aam_tiled: embedding width %d (multiples of 4 up to %d)", D, HT_DMAX);
    if (!ws || ws_bytes < vp_cosine_aam_tiled_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_aam_tiled: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    HeadTileArgs a;
    a.emb = emb; a.W = W; a.labels = (const long long*)labels;
    a.tiles = (C + HT_CT - 1) / HT_CT;
    a.part = (float*)ws;
    a.tgt = (float*)((char*)ws + vp_align_up(5 * (size_t)B * a.tiles * 4, 256));
    a.cinv = cinv;
    float* rinv = (float*)((char*)a.tgt + vp_align_up((size_t)B * 4, 256));
    { const int rc = vp_row_inv_norm(ctx, emb, B, D, D, 1e-12f, rinv, st); if (rc != VP_OK) return rc; }
    a.rinv = rinv;
    a.B = B; a.D = D; a.C = C; a.SE = D + 2;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.easy = easy_margin; a.mt = ctx->margin_table;
    const int smem = (D * HT_SW + HT_RF * (D + 2) + HT_CT + HT_RF + 5 * 4 * HT_RF + HT_RF) * 4;
    static bool attr_set[64] = {};                                 // the attribute is per DEVICE: a process that drives several GPUs sets it on each
    bool& attr_dev = attr_set[ctx->device & 63];
    if (!attr_dev) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(head_tile_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (HT_DMAX * HT_SW + HT_RF * (HT_DMAX + 2) + HT_CT + HT_RF + 5 * 4 * HT_RF + HT_RF) * 4));
        attr_dev = true;
    }
    const int rblocks = (B + HT_RF - 1) / HT_RF;
    int ysplit = 1;
    while (ysplit < rblocks && a.tiles * ysplit < 192) ++ysplit;
    hipLaunchKernelGGL(head_tile_fwd_kernel, dim3(a.tiles, ysplit), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "head_tile_fwd");
    hipLaunchKernelGGL(head_tile_merge_kernel, dim3(B), dim3(256), 0, st, a.part, a.tgt, B, C, a.tiles, label_smoothing, lse, row_loss, pred);
    VP_LAUNCH_CHECK(ctx, "head_tile_merge");
    hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
    VP_LAUNCH_CHECK(ctx, "head_mean");
    return VP_OK;
}

size_t vp_cosine_aam_tiled_bwd_workspace_bytes(int B, int D, int C) {
    const size_t tiles = (size_t)(C + HT_CT - 1) / HT_CT;
    const size_t nwg = tiles < 256 ? tiles : 256;
    return vp_cosine_aam_tiled_workspace_bytes(B, D, C) + vp_align_up((size_t)B * 4, 256) * 2 + vp_align_up(nwg * (size_t)B * D * 4, 256);
}

/* forward value (optional) + both gradients, class-tiled: B <= 128, D == 192; otherwise VP_EUNSUP (callers fall back to
 * vp_cosine_aam_ce_bwd) */
int vp_cosine_aam_tiled_bwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                            float scale, float label_smoothing, int easy_margin, float grad_scale, float* demb, float* dW, float* loss,
                            int* pred, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !labels || !demb || !dW || B <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_aam_tiled_bwd: bad arguments");
    if (B > 2 * HT_RB || D != 192) return VP_EUNSUP;
    if (!ws || ws_bytes < vp_cosine_aam_tiled_bwd_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_aam_tiled_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)ws;
    const size_t fw = vp_cosine_aam_tiled_workspace_bytes(B, D, C);
    void* fws = p; p += fw;
    float* row_loss = (float*)p; p += vp_align_up((size_t)B * 4, 256);
    float* lse = (float*)p; p += vp_align_up((size_t)B * 4, 256);
    float* part = (float*)p;
    // forward pass: per-row log-sum-exp (and the loss value when asked for; the merge always writes row losses)
    float* loss_dst = loss ? loss : row_loss;                     // a scratch slot when the caller does not want the value
    int rc = vp_cosine_aam_tiled_fwd(ctx, emb, W, labels, B, D, C, margin, scale, label_smoothing, easy_margin, loss_dst, row_loss, lse,
                                     nullptr, pred, fws, fw, stream);
    if (rc) return rc;
    HeadBwdArgs a;
    a.emb = emb; a.W = W; a.labels = (const long long*)labels; a.lse = lse; a.dW = dW; a.dxn_part = part;
    a.B = B; a.D = D; a.C = C; a.tiles = (C + HT_CT - 1) / HT_CT; a.SE = D + 2;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.ls = label_smoothing; a.gscale = grad_scale; a.easy = easy_margin; a.mt = ctx->margin_table;
    const int nwg = a.tiles < 256 ? a.tiles : 256;
    const int smem = (D * HT_SW + HT_RB * (D + 2) + HT_RB * HT_SG + 64 * 3 + 4 * 64 + 64) * 4;
    static bool attr_set[64] = {};
    bool& attr_dev = attr_set[ctx->device & 63];
    if (!attr_dev) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(head_tile_bwd_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (192 * HT_SW + HT_RB * 194 + HT_RB * HT_SG + 64 * 3 + 4 * 64 + 64) * 4));
        attr_dev = true;
    }
    hipLaunchKernelGGL(head_tile_bwd_kernel<12>, dim3(nwg), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "head_tile_bwd");
    hipLaunchKernelGGL(head_dx_finish_kernel, dim3(B), dim3(256), 0, st, emb, part, nwg, B, D, demb);
    VP_LAUNCH_CHECK(ctx, "head_dx_finish");
    return VP_OK;
}

}  // extern "C"
