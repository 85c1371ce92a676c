// conv GEMM kernel template (included by conv_gemm_bf16.hip / conv_gemm_f32.hip).
// Design notes: conv_gemm.hip.
#pragma once
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int BM = VP_CONV_BM;
constexpr int ROWB = 128;        // bytes of K per tile row per stage
constexpr int NSEG_MAX = 8;

// loader / prologue flavour of an instantiation
constexpr int MODE_TAPS = 0;     // 1-D conv, KW taps along time (reflect / zero / none)
constexpr int MODE_1X1 = 1;      // KW == 1: source row fixed, address hoisted out of the K loop
constexpr int MODE_2D = 2;       // 2-D conv over (time, freq), zero padding, stride on freq
constexpr int MODE_1X1_PRO = 3;  // 1x1 with BN-affine + ReLU applied to the INPUT channels while staging

struct ConvArgs {
    const void* x; const void* w;
    const float* bias; const float* rowbias; const float* bn_scale; const float* bn_shift;
    const float* pro_scale; const float* pro_shift; const float* gate;
    void* y; void* y2; const void* add_in; void* aux; const void* res; float* psum; float* psumsq;
    unsigned x_bytes, w_bytes;
    int ldx, xoff, ldy, yoff, ldy2, y2off, ysplit, ld_add, add_off, ld_aux, aux_off, ld_res, res_off;
    int M, N, K, KC, cpt, KT, Cin;
    int Kw;                 // elements per WEIGHT row (= K, or K rounded up to a multiple of 32 for weights given as split bf16 planes: x3w_t)
    int T_in, T_out, dilation, stride, pad_left, pad_mode, act, act2;
    int F_in, F_out, KF, stride_f, pad_f;
    int gate_len, gate_nseg;
    int tiles_m, tiles_n, nseg, group_m;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// amp_t: f32 tensors in memory, ROUNDED TO bf16 on their way into LDS -- bf16 MFMA with f32 accumulation and f32 outputs
// (the mixed-precision training engine: paddle.amp.auto_cast O1 runs conv / matmul in low precision and keeps the rest f32,
// trainer.py:209-229).  A stage is 32 k (one v_mfma_f32_16x16x32_bf16 step) = 64-byte LDS rows.
struct amp_t { float v; };
// x3_t: f32 tensors in memory, SPLIT into two bf16 terms on their way into LDS -- x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (both round-to-nearest-even; x - hi is exact in f32) -- and contracted as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with f32
// accumulation: the product of two f32 numbers to ~2^-16 relative (the dropped lo*lo term and lo's own rounding), against 2^-8 for a
// single bf16 pass, at one third of the bf16 MFMA rate (833 TFLOP/s dense on MI355X, 5.3x the exact-f32 MFMA rate).  This is the
// engine that carries the reference's 1e-4 score tolerance at trained weights.  A stage is 32 k: a 128-byte
// LDS row = [32 hi | 32 lo], so the hi / lo fragments are the ks = 0 / ks = 1 reads of the bf16 layout (same swizzle, conflict-free).
struct x3_t { float v; };
// hl_t: a tensor STORED as split bf16 planes ("hl32"): a row of C channels (C % 32 == 0) is C / 32 groups of 128 bytes, each
// [32 x bf16 hi | 32 x bf16 lo] of its 32 channels, value = hi + lo -- 4 bytes per element like f32, so byte offsets / leading
// dimensions are those of an f32 tensor, and a 128-byte LDS row of a K-stage is one group: the layout x3_t builds while staging is
// what an hl_t operand already has in memory (no split arithmetic in the consumer, LDS-DMA-able: conv_gemm256.hip).  Producers split
// once per output element in their epilogue.  As an OUTPUT type the epilogue stores hi / lo halves instead of f32 words.
struct hl_t { float v; };
// x3w_t: x3_t whose WEIGHTS arrive already split (hl32 planes, [Cout][Kw], Kw = K rounded up to a multiple of 32 with zero columns:
// vp_tdnn_layer.w_hl) -- the activations are split while staging, the weight chunks are copied as they are.  Halves the split arithmetic
// and the 8-byte LDS writes of a stage (the staging of x3_t is VALU-bound: ablation in profiles/r06_x3_staging_ablation.log).
struct x3w_t { float v; };
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float4 lo, hi; };
template <> struct Frag<amp_t> { bf16x8 v; };
template <> struct Frag<x3_t> { bf16x8 hi, lo; };
template <> struct Frag<hl_t> { bf16x8 hi, lo; };
template <> struct Frag<x3w_t> { bf16x8 hi, lo; };
constexpr int ROWB_AMP = 64;
// 64-byte rows: the ds_read_b128 lane groups ({0-3,12-15,20-27}, ...) are conflict-free with the 16-B chunk g stored at
// position g ^ ((-(row >> 2)) & 3)
__device__ __forceinline__ int amp_pos(int row, int g) { return (g ^ ((0 - (row >> 2)) & 3)) << 4; }

__device__ __forceinline__ void load_frag(const char* tile, int row, int ks, int g, Frag<bf16_t>& f) {
    const int c = ks * 4 + g;
    f.v = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((c ^ (row & 7)) << 4));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<float>& f) {
    const int c = 2 * g;
    f.lo = *reinterpret_cast<const float4*>(tile + row * ROWB + ((c ^ (row & 7)) << 4));
    f.hi = *reinterpret_cast<const float4*>(tile + row * ROWB + (((c + 1) ^ (row & 7)) << 4));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<amp_t>& f) {
    f.v = *reinterpret_cast<const bf16x8*>(tile + row * ROWB_AMP + amp_pos(row, g));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<x3_t>& f) {
    f.hi = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((g ^ (row & 7)) << 4));
    f.lo = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + (((4 + g) ^ (row & 7)) << 4));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<x3w_t>& f) {
    f.hi = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((g ^ (row & 7)) << 4));
    f.lo = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + (((4 + g) ^ (row & 7)) << 4));
}
__device__ __forceinline__ void load_frag(const char* tile, int row, int /*ks*/, int g, Frag<hl_t>& f) {
    f.hi = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((g ^ (row & 7)) << 4));
    f.lo = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + (((4 + g) ^ (row & 7)) << 4));
}
// D[n][m] += sum_k W[n][k] * X[m][k]: weights are the A operand (row = lane & 15 -> n), activations
// the B operand (col = lane & 15 -> m); result register r of lane l = (n = (l >> 4) * 4 + r, m = l & 15).
__device__ __forceinline__ void mma(const Frag<bf16_t>& w, const Frag<bf16_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, x.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<amp_t>& w, const Frag<amp_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, x.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<x3_t>& w, const Frag<x3_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, x.hi, c, 0, 0, 0);      // the two cross terms, then the leading one
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.hi, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<x3w_t>& w, const Frag<x3w_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, x.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.hi, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<hl_t>& w, const Frag<hl_t>& x, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, x.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.hi, c, 0, 0, 0);
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

// value -> (hi, lo) bf16 pair of the split representation
__device__ __forceinline__ void hl_split(float x, unsigned short& h, unsigned short& l) {
    const bf16_t xh = (bf16_t)x;
    const bf16_t xl = (bf16_t)(x - (float)xh);
    h = __builtin_bit_cast(unsigned short, xh);
    l = __builtin_bit_cast(unsigned short, xl);
}
// 4 consecutive channels `col` .. `col + 3` (col % 4 == 0) of an hl32 row starting at byte address `row`
__device__ __forceinline__ void hl_store4(char* row, int col, const float v[4]) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hl_split(v[e], h[e], l[e]);
    char* p = row + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(p + 64) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}
__device__ __forceinline__ void hl_load4(const char* row, int col, float v[4]) {
    const char* p = row + (col >> 5) * 128 + (col & 31) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + 64);
    v[0] = __builtin_bit_cast(float, h.x << 16) + __builtin_bit_cast(float, l.x << 16);
    v[1] = __builtin_bit_cast(float, h.x & 0xffff0000u) + __builtin_bit_cast(float, l.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, h.y << 16) + __builtin_bit_cast(float, l.y << 16);
    v[3] = __builtin_bit_cast(float, h.y & 0xffff0000u) + __builtin_bit_cast(float, l.y & 0xffff0000u);
}
// epilogue accessors: tensor `base` of element type T, `off` = element offset of the row's channel 0 (row * ld + channel offset),
// `col` = channel within the row slice.  Plain types: base[off + col ..]; hl_t: the row's groups (off % 32 == 0, host-checked).
template <typename T> __device__ __forceinline__ void st4(T* base, size_t off, int col, const float v[4]) { store4(base + off + col, v); }
template <> __device__ __forceinline__ void st4<hl_t>(hl_t* base, size_t off, int col, const float v[4]) {
    hl_store4(reinterpret_cast<char*>(base) + off * 4, col, v);
}
template <typename T> __device__ __forceinline__ void ld4(const T* base, size_t off, int col, float v[4]) { load4(base + off + col, v); }
template <> __device__ __forceinline__ void ld4<hl_t>(const hl_t* base, size_t off, int col, float v[4]) {
    hl_load4(reinterpret_cast<const char*>(base) + off * 4, col, v);
}

// input prologue on one staged 16-byte chunk: x <- relu(x * s + h), per channel
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], bf16_t) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float lo = __builtin_bit_cast(float, v[w] << 16);
        const float hi = __builtin_bit_cast(float, v[w] & 0xffff0000u);
        const bf16_t a = (bf16_t)fmaxf(lo * s[2 * w] + h[2 * w], 0.f);
        const bf16_t b = (bf16_t)fmaxf(hi * s[2 * w + 1] + h[2 * w + 1], 0.f);
        o[w] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    }
    return o;
}
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], float);
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], amp_t) {
    return prologue_chunk(v, s, h, float{});
}
// four f32 -> four bf16 (round to nearest even), packed into 8 bytes
__device__ __forceinline__ uint2 amp_pack(u32x4 v) {
    const unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
    const bf16_t b0 = (bf16_t)__builtin_bit_cast(float, a0), b1 = (bf16_t)__builtin_bit_cast(float, a1);
    const bf16_t b2 = (bf16_t)__builtin_bit_cast(float, a2), b3 = (bf16_t)__builtin_bit_cast(float, a3);
    uint2 o;
    o.x = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
    o.y = (unsigned)__builtin_bit_cast(unsigned short, b2) | ((unsigned)__builtin_bit_cast(unsigned short, b3) << 16);
    return o;
}
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], x3_t) {
    return prologue_chunk(v, s, h, float{});
}
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&)[8], const float (&)[8], hl_t) { return v; }   // (never dispatched)
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], x3w_t) { return prologue_chunk(v, s, h, float{}); }
// four f32 -> their bf16 hi terms and bf16 lo terms (x = hi + lo up to 2^-16 relative), 8 bytes each
__device__ __forceinline__ void x3_split(u32x4 v, uint2& hi, uint2& lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = v[e];              // NOT bit_cast(v[e]): on a vector-element lvalue hipcc reads element 0
        const float x = __builtin_bit_cast(float, u);
        const bf16_t xh = (bf16_t)x;
        const bf16_t xl = (bf16_t)(x - (float)xh);
        h[e] = __builtin_bit_cast(unsigned short, xh);
        l[e] = __builtin_bit_cast(unsigned short, xl);
    }
    hi.x = (unsigned)h[0] | ((unsigned)h[1] << 16); hi.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
    lo.x = (unsigned)l[0] | ((unsigned)l[1] << 16); lo.y = (unsigned)l[2] | ((unsigned)l[3] << 16);
}
__device__ __forceinline__ u32x4 prologue_chunk(u32x4 v, const float (&s)[8], const float (&h)[8], float) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned u = v[w];          // NOT bit_cast(v[w]): on a vector-element lvalue hipcc reads element 0
        o[w] = __builtin_bit_cast(unsigned, fmaxf(__builtin_bit_cast(float, u) * s[w] + h[w], 0.f));
    }
    return o;
}

template <typename TI, typename TO, int BN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvArgs a) {
    constexpr int EPC = 16 / (int)sizeof(TI);        // elements per 16-B chunk
    constexpr int KSTEPS = (8 * EPC) / 32;           // MFMA k-steps per stage: bf16 2, f32 1
    constexpr int NI = BN >= 64 ? 4 : BN / 16;       // 16-column MFMA tiles per wave
    constexpr int WN = BN / (NI * 16);               // waves along N
    constexpr int WM = 4 / WN;
    constexpr int MI = BM / (WM * 16);
    constexpr int WCOLS = NI * 16;
    constexpr int BROWS = BN / 32;
    constexpr bool AMP = std::is_same<TI, amp_t>::value;
    constexpr bool WHL = std::is_same<TI, x3w_t>::value;           // weights given as split planes: copied, not split
    constexpr bool X3 = std::is_same<TI, x3_t>::value || WHL;
    constexpr int RB = AMP ? ROWB_AMP : ROWB;            // bytes per tile row per stage in LDS
    constexpr int STAGE = (BM + BN) * RB;
    constexpr bool PRO = MODE == MODE_1X1_PRO;
    constexpr bool ONE = MODE == MODE_1X1 || MODE == MODE_1X1_PRO;
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
    // x3: chunk cc (4 f32) becomes 8 bytes of the hi plane (16-B chunk cc >> 1 of the row) and 8 bytes of the lo plane (chunk 4 + (cc >> 1))
    const int pw = AMP ? ((((cc >> 1) ^ ((0 - (r0 >> 2)) & 3)) << 4) + (cc & 1) * 8)
                 : X3  ? ((((cc >> 1) ^ (r0 & 7)) << 4) + (cc & 1) * 8)
                       : ((cc ^ (r0 & 7)) << 4);
    [[maybe_unused]] const int pw_lo = (((4 + (cc >> 1)) ^ (r0 & 7)) << 4) + (cc & 1) * 8;
    // x3: an 8-byte LDS write retires 16 lanes (= two tile rows of 8 chunks) per cycle onto 32 banks = 128 bytes.  The hi halves of rows r
    // and r + 1 sit in the SAME 64-byte half of their 128-byte rows (the swizzle moves a row's hi plane to the other half only with bit 2
    // of r): a 2-way conflict on every write (measured: SQ_LDS_BANK_CONFLICT 0.30-0.33 of the LDS cycles, the two write planes 120 of 366 us
    // on a 512 x 1536 x 76288 layer).  Odd rows therefore write their LO plane first and their HI plane second: each 16-lane group then
    // covers all 32 banks once.
    [[maybe_unused]] const bool x3_swap = (r0 & 1) != 0;
    [[maybe_unused]] const int pw_a = x3_swap ? pw_lo : pw, pw_b = x3_swap ? pw : pw_lo;
    const bool zero_pad = a.pad_mode == VP_PAD_ZERO;
    const unsigned ldxb = (unsigned)a.ldx * ES;
    // per staged row: rowoff = byte offset of (utterance b, frame 0 [, freq 0]); tpos / fpos = the
    // un-padded source position of tap 0; rowfix = the full row offset when there is a single tap
    unsigned rowoff[4], rowfix[4], woff[BROWS];
    int tpos[4], fpos[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < a.M;
        int mm = ok ? m : 0;
        int f = 0;
        if constexpr (MODE == MODE_2D) {
            const int bt = mm / a.F_out;
            f = mm - bt * a.F_out;
            mm = bt;
        }
        const int b = mm / a.T_out;
        const int t = mm - b * a.T_out;
        tpos[i] = t * a.stride - a.pad_left;
        fpos[i] = f * a.stride_f - a.pad_f;
        if constexpr (MODE == MODE_2D)
            rowoff[i] = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.F_in * (unsigned)a.ldx + (unsigned)a.xoff) * ES : OOB;
        else
            rowoff[i] = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.ldx + (unsigned)a.xoff) * ES : OOB;
        const int traw = tpos[i];
        int ts = traw < 0 ? -traw : traw;
        ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
        const bool inr = traw >= 0 && traw < a.T_in;
        rowfix[i] = (rowoff[i] != OOB && (inr || !zero_pad)) ? rowoff[i] + (unsigned)ts * ldxb : OOB;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + r0 + 32 * i;
        woff[i] = n < a.N ? (unsigned)n * (unsigned)a.Kw * ES : OOB;
    }

    // TWO register stages: the loads of K-stage k+2 are issued while stage k is computed and are
    // written to LDS at the end of stage k+1, so a fetch has two stages of MFMA work to land.
    constexpr int NPRO = PRO ? 8 : 1;
    u32x4 ra0[4], rb0[BROWS], ra1[4], rb1[BROWS];
    float ps0[NPRO], ph0[NPRO], ps1[NPRO], ph1[NPRO];

    auto gload = [&](int kt, u32x4 (&ra)[4], u32x4 (&rb)[BROWS], float (&ps)[NPRO], float (&ph)[NPRO]) {
        const int q = kt * 8 + cc;
        const bool kv = q < a.KC;
        const unsigned kb = (unsigned)q * 16u;
        if constexpr (ONE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = kv && rowfix[i] != OOB;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? rowfix[i] + kb : OOB, 0, 0);
            }
            if constexpr (PRO) {
                const int c0 = min(q * EPC, a.Cin - EPC);               // clamp: K-tail chunks meet zero weights anyway
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.pro_scale + c0 + e);
                    const float4 h4 = *reinterpret_cast<const float4*>(a.pro_shift + c0 + e);
                    ps[e] = s4.x; ps[e + 1] = s4.y; ps[e + 2] = s4.z; ps[e + 3] = s4.w;
                    ph[e] = h4.x; ph[e + 1] = h4.y; ph[e + 2] = h4.z; ph[e + 3] = h4.w;
                }
            }
        } else if constexpr (MODE == MODE_2D) {
            const int j = q / a.cpt;
            const unsigned cb = (unsigned)(q - j * a.cpt) * 16u;
            const int kt_ = j / a.KF;
            const int kf = j - kt_ * a.KF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = tpos[i] + kt_ * a.dilation;
                const int fs = fpos[i] + kf;
                const bool ok = kv && rowoff[i] != OOB && ts >= 0 && ts < a.T_in && fs >= 0 && fs < a.F_in;
                const unsigned off = rowoff[i] + ((unsigned)ts * (unsigned)a.F_in + (unsigned)fs) * ldxb + cb;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? off : OOB, 0, 0);
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
        const bool kvw = WHL ? q * EPC < a.Kw : kv;                 // (split-plane weight rows are zero-padded to whole 32-element groups)
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const bool ok = kvw && woff[i] != OOB;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, ok ? woff[i] + kb : OOB, 0, 0);
        }
    };
    auto swrite = [&](int s, const u32x4 (&ra)[4], const u32x4 (&rb)[BROWS], const float (&ps)[NPRO],
                      const float (&ph)[NPRO]) {
        char* As = smem + s * STAGE;
        char* Bs = As + BM * RB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 v = ra[i];
            if constexpr (PRO) {
                float s8[8], h8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { s8[e] = ps[e % NPRO]; h8[e] = ph[e % NPRO]; }
                v = prologue_chunk(v, s8, h8, TI{});
            }
            if constexpr (AMP) *reinterpret_cast<uint2*>(As + (r0 + 32 * i) * RB + pw) = amp_pack(v);
            else if constexpr (X3) {
                uint2 hi, lo;
                x3_split(v, hi, lo);
                *reinterpret_cast<uint2*>(As + (r0 + 32 * i) * RB + pw_a) = x3_swap ? lo : hi;
                *reinterpret_cast<uint2*>(As + (r0 + 32 * i) * RB + pw_b) = x3_swap ? hi : lo;
            } else *reinterpret_cast<u32x4*>(As + (r0 + 32 * i) * RB + pw) = v;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            if constexpr (AMP) *reinterpret_cast<uint2*>(Bs + (r0 + 32 * i) * RB + pw) = amp_pack(rb[i]);
            else if constexpr (WHL) *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * i) * RB + ((cc ^ (r0 & 7)) << 4)) = rb[i];
            else if constexpr (X3) {
                uint2 hi, lo;
                x3_split(rb[i], hi, lo);
                *reinterpret_cast<uint2*>(Bs + (r0 + 32 * i) * RB + pw_a) = x3_swap ? lo : hi;
                *reinterpret_cast<uint2*>(Bs + (r0 + 32 * i) * RB + pw_b) = x3_swap ? hi : lo;
            } else *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * i) * RB + pw) = rb[i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) {
        const char* As = smem + s * STAGE;
        const char* Bs = As + BM * RB;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            Frag<TI> xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) load_frag(As, wm * (MI * 16) + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) load_frag(Bs, wn * WCOLS + ni * 16 + li, ks, g, wf[ni]);
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
    gload(0, ra0, rb0, ps0, ph0);
    gload(1, ra1, rb1, ps1, ph1);
    swrite(0, ra0, rb0, ps0, ph0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        // even stage kt on LDS[0]; set0 <- stage kt+2; set1 (stage kt+1) -> LDS[1]
        gload(kt + 2, ra0, rb0, ps0, ph0);
        compute(0);
        swrite(1, ra1, rb1, ps1, ph1);
        __syncthreads();
        if (kt + 1 >= KT) break;
        // odd stage kt+1 on LDS[1]; set1 <- stage kt+3; set0 (stage kt+2) -> LDS[0]
        gload(kt + 3, ra1, rb1, ps1, ph1);
        compute(1);
        swrite(0, ra0, rb0, ps0, ph0);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // y = act2( bn( act( acc + bias + rowbias ) ) * gate + res );   aux = y + add_in
    // y goes out through a wave-private LDS slab (the staging buffers are free now): the accumulator layout would store
    // 8 bytes per lane, 16 rows per instruction -- the vector-memory path handles that a lane at a time -- while the slab
    // is copied out 16 bytes per lane, whole rows contiguous.  Row stride = row bytes + 16: conflict-free ds_write_b64.
    constexpr int SLAB_ROW = WCOLS * (int)sizeof(TO) + 16;
    char* slab = smem + wv * (MI * 16 * SLAB_ROW);
    TO* __restrict__ Y = static_cast<TO*>(a.y);
    TO* __restrict__ Y2 = static_cast<TO*>(a.y2);
    const TO* __restrict__ ADD = static_cast<const TO*>(a.add_in);
    const TO* __restrict__ RES = static_cast<const TO*>(a.res);
    TO* __restrict__ AUX = static_cast<TO*>(a.aux);
    const int bfirst = m0 / a.T_out;
    int rowm[MI], rowb[MI], rowg[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        rowm[mi] = m0 + wm * (MI * 16) + mi * 16 + li;
        const int mc = rowm[mi] < a.M ? rowm[mi] : (a.M - 1);
        rowb[mi] = mc / a.T_out;
        rowg[mi] = a.gate ? rowb[mi] * a.gate_nseg + (mc - rowb[mi] * a.T_out) / a.gate_len : 0;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int nb = n0 + wn * WCOLS + ni * 16 + g * 4;
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
            float rbias[4] = {0.f, 0.f, 0.f, 0.f}, gt[4] = {1.f, 1.f, 1.f, 1.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok && a.rowbias) load4(a.rowbias + (size_t)rowb[mi] * a.N + nb, rbias);
            if (ok && a.gate) load4(a.gate + (size_t)rowg[mi] * a.N + nb, gt);
            if (ok && RES) ld4(RES, (size_t)m * a.ld_res + a.res_off, nb, rs);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[mi][ni][r] + bias4[r] + rbias[r];
                if (a.act == VP_ACT_RELU) t = fmaxf(t, 0.f);
                t = (t * sc4[r] + sh4[r]) * gt[r] + rs[r];
                if (a.act2 == VP_ACT_TANH) t = vp_tanh_for<TO>(t);
                else if (a.act2 == VP_ACT_RELU) t = fmaxf(t, 0.f);
                else if (a.act2 == VP_ACT_HARDTANH20) t = fminf(fmaxf(t, 0.f), 20.f);
                else if (a.act2 == VP_ACT_SILU) t = vp_silu_for<TO>(t);
                v[r] = t;
            }
            st4(reinterpret_cast<TO*>(slab + (mi * 16 + li) * SLAB_ROW), 0, ni * 16 + g * 4, v);
            if (ok) {
                if (nb < a.ysplit) st4(Y2, (size_t)m * a.ldy2 + a.y2off, nb, v);
                if (AUX) {
                    float ad[4];
                    ld4(ADD, (size_t)m * a.ld_add + a.add_off, nb, ad);
                    float s4[4] = {v[0] + ad[0], v[1] + ad[1], v[2] + ad[2], v[3] + ad[3]};
                    st4(AUX, (size_t)m * a.ld_aux + a.aux_off, nb, s4);
                }
            }
            // keep (y - shift) for the column sums; zero for rows / columns outside the problem
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = ok ? (v[r] - sh4[r]) : 0.f;
        }
    }
    {
        constexpr int EPL = 16 / (int)sizeof(TO);              // elements per 16-byte lane chunk
        constexpr int LPR = WCOLS / EPL;                       // lanes per row
        constexpr int RPI = 64 / LPR;                          // rows per wave-instruction
        const int cl = (lane % LPR) * EPL, rl = lane / LPR;
        const int nc = n0 + wn * WCOLS + cl;
        const bool vec16 = ((a.ldy * (int)sizeof(TO)) & 15) == 0 && ((a.yoff * (int)sizeof(TO)) & 15) == 0 &&
                           (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
#pragma unroll
        for (int j = 0; j < MI * 16 / RPI; ++j) {
            const int r = j * RPI + rl;
            const int m = m0 + wm * (MI * 16) + r;
            const u32x4 o = *reinterpret_cast<const u32x4*>(slab + r * SLAB_ROW + cl * (int)sizeof(TO));
            if (m < a.M) {
                TO* dst = Y + (size_t)m * a.ldy + a.yoff + nc;
                if (vec16 && nc + EPL <= a.N) {
                    *reinterpret_cast<u32x4*>(dst) = o;
                } else if constexpr (sizeof(TO) == 2) {        // 8-byte halves (N % 4 == 0, 8-byte alignment guaranteed)
                    if (nc < a.N) *reinterpret_cast<uint2*>(dst) = make_uint2(o[0], o[1]);
                    if (nc + 4 < a.N) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(o[2], o[3]);
                } else {
                    if (nc < a.N) *reinterpret_cast<u32x4*>(dst) = o;     // f32: 4 elements = 16 bytes, always aligned (ldy, yoff % 4)
                }
            }
        }
    }
    if (a.psum) {
        __syncthreads();                                       // slabs are read out before `red` reuses the memory
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
                        const int col = wn * WCOLS + ni * 16 + g * 4 + r;
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

template <typename TI, typename TO, int BN, int MODE>
int launch_conv(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int NI_ = BN >= 64 ? 4 : BN / 16, WN_ = BN / (NI_ * 16), MI_ = BM / ((4 / WN_) * 16);
    constexpr int slab_ = 4 * MI_ * 16 * (NI_ * 16 * (int)sizeof(TO) + 16);              // output slabs of the epilogue
    constexpr int smem = 2 * (BM + BN) * ROWB > slab_ ? 2 * (BM + BN) * ROWB : slab_;
    static bool attr_set = false;
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<TI, TO, BN, MODE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<TI, TO, BN, MODE>), dim3(a.tiles_m * a.tiles_n), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_gemm");
    return VP_OK;
}

// dispatch over tile width and loader mode for one (TI, TO) pair
template <typename TI, typename TO, bool SMALL_TILES>
int dispatch_conv(vp_ctx* ctx, const ConvArgs& a, int bn, int mode, hipStream_t st) {
    if (bn == 128) {
        if (mode == MODE_TAPS) return launch_conv<TI, TO, 128, MODE_TAPS>(ctx, a, st);
        if (mode == MODE_1X1) return launch_conv<TI, TO, 128, MODE_1X1>(ctx, a, st);
        if constexpr (SMALL_TILES) { if (mode == MODE_2D) return launch_conv<TI, TO, 128, MODE_2D>(ctx, a, st); }
    }
    if constexpr (SMALL_TILES) {
        if (bn == 64) {
            if (mode == MODE_TAPS) return launch_conv<TI, TO, 64, MODE_TAPS>(ctx, a, st);
            if (mode == MODE_1X1) return launch_conv<TI, TO, 64, MODE_1X1>(ctx, a, st);
            if (mode == MODE_2D) return launch_conv<TI, TO, 64, MODE_2D>(ctx, a, st);
            if (mode == MODE_1X1_PRO) return launch_conv<TI, TO, 64, MODE_1X1_PRO>(ctx, a, st);
        }
        if (bn == 32) {
            if (mode == MODE_TAPS) return launch_conv<TI, TO, 32, MODE_TAPS>(ctx, a, st);
            if (mode == MODE_1X1) return launch_conv<TI, TO, 32, MODE_1X1>(ctx, a, st);
            if (mode == MODE_2D) return launch_conv<TI, TO, 32, MODE_2D>(ctx, a, st);
            if (mode == MODE_1X1_PRO) return launch_conv<TI, TO, 32, MODE_1X1_PRO>(ctx, a, st);
        }
    }
    VP_FAIL(ctx, VP_EUNSUP, "conv1d: no kernel for tile %d / mode %d", bn, mode);
}

}  // namespace
