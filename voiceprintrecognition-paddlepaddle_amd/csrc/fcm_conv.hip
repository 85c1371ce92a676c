// 3x3 convolution over a (B, T, F, 32) bf16 map with 32 output channels: the FCM head of CAM++ (campplus.py:211-281: BasicResBlock
// conv1 / conv2 / shortcut, FCM.conv2), stride 1 or 2 on the frequency axis, zero 'same' padding.
//
// These layers are N = 32 GEMMs over 10^6 positions: on the general conv GEMM (tile 128 positions x 32 channels, the 2-D loader
// recomputing (t, f) bounds per tap and per 16-byte piece) they ran 2.5x off their HBM roofline (173 us average at B = 256 where
// reading the input and writing the output once takes 70).  Here a workgroup owns TT time rows x all F of one utterance:
//   - the input slab (TT + 2 rows, F + 2 columns: the zero padding is materialised once, so a tap is a constant LDS offset) is
//     loaded with 16-byte coalesced accesses -- a time row of the map is contiguous in HBM;
//   - the 9 x 2 weight fragments (32 x 288 bf16) live in registers for the whole workgroup;
//   - per 16 output positions: 9 ds_read_b128 + 18 v_mfma_f32_16x16x32_bf16 (weights = A, positions = B: a lane ends with 4
//     consecutive channels of one position);
//   - epilogue: bias, BN, residual, ReLU in f32, transposed through a 1.25 KB LDS tile so that a wave stores 1 KB contiguous;
//   - the stride-2 blocks' 1x1 shortcut conv (BN folded) reads the centre-tap fragment that is already in registers: 2 more MFMAs
//     and a second output instead of another pass over the input.
#include "common.h"

namespace {

constexpr int C32 = 32;
constexpr int C32_THREADS = 256;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
constexpr int C32_SROW = 80;                 // bytes per position of the output staging tile (64 + pad: 16-byte aligned, conflict-light)

struct C32Args {
    const bf16_t* x; bf16_t* y; const bf16_t* res; bf16_t* y2;
    const bf16_t* w; const float* bias; const float* scale; const float* shift;        // [32][9 * 32], k = (kt * 3 + kf) * 32 + c
    const bf16_t* w2; const float* bias2; const float* scale2; const float* shift2;    // shortcut 1x1 [32][32] (y2 != NULL)
    int B, T, F_in, F_out, TT, relu;
    // FUSE: x is not read; the slab is FCM.conv1 (1 -> 32 channels, 3x3, BN, ReLU; campplus.py:254-255,274) of feats (B, T, F_in)
    const bf16_t* feats; const float* c1_w; const float* c1_b; const float* c1_scale; const float* c1_shift;   // c1_w [32][9], tap = kt * 3 + kf
};

template <int SF, bool FUSE>
__global__ __launch_bounds__(C32_THREADS) void conv3x3_c32_kernel(const C32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * a.TT;
    const int Fp = a.F_in + 2;                              // padded row: column 0 and F_in + 1 are zeros
    const int rows = a.TT + 2;
    char* slab = smem;
    char* stage = smem + (size_t)rows * Fp * 64 + wv * (2 * 16 * C32_SROW);      // per wave: main tile | shortcut tile

    if constexpr (FUSE) {
        // ---- the slab IS conv1's output: feats rows t0 - 2 .. t0 + TT + 1 (zero outside the utterance, one zero column each side)
        //      go to LDS as f32; a thread then owns 8 channels (weights in registers) of every fourth-of-a-position it visits
        float* fs = reinterpret_cast<float*>(smem + (size_t)rows * Fp * 64 + (C32_THREADS / 64) * (2 * 16 * C32_SROW));
        const int frows = a.TT + 4;
        const __amdgpu_buffer_rsrc_t fsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.feats + (size_t)b * a.T * a.F_in), 0,
                                                                             (unsigned)(a.T * a.F_in * 2), 0x00020000);
        for (int i0 = tid; i0 < frows * Fp; i0 += 4 * C32_THREADS) {       // branch-free: out-of-range offsets read zero
            unsigned short raw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * C32_THREADS;
                const int r = i / Fp, c = i - r * Fp;
                const int t = t0 - 2 + r, f = c - 1;
                const bool ok = i < frows * Fp && t >= 0 && t < a.T && f >= 0 && f < a.F_in;
                raw[k] = __builtin_amdgcn_raw_buffer_load_b16(fsrd, ok ? (unsigned)((t * a.F_in + f) * 2) : 0xfffffff0u, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * C32_THREADS;
                if (i < frows * Fp) fs[i] = __uint_as_float((unsigned)raw[k] << 16);
            }
        }
        const int cg = (tid & 3) * 8;
        float w1[8][9], b1[8], s1[8], h1[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int k = 0; k < 9; ++k) w1[c][k] = a.c1_w[(cg + c) * 9 + k];
            b1[c] = a.c1_b[cg + c]; s1[c] = a.c1_scale[cg + c]; h1[c] = a.c1_shift[cg + c];
        }
        __syncthreads();
        const float inv_f = 1.f / (float)a.F_in;
        for (int i = tid; i < rows * a.F_in * 4; i += C32_THREADS) {
            const int pos = i >> 2;
            const int r = (int)(((float)pos + 0.5f) * inv_f);
            const int f = pos - r * a.F_in;
            const int t = t0 - 1 + r;
            float in[9];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) in[kt * 3 + kf] = fs[(r + kt) * Fp + f + kf];
            bf16x8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float acc = b1[c];
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += w1[c][k] * in[k];
                o[c] = (bf16_t)((t >= 0 && t < a.T) ? fmaxf(acc * s1[c] + h1[c], 0.f) : 0.f);
            }
            *reinterpret_cast<bf16x8*>(slab + ((size_t)r * Fp + f + 1) * 64 + cg * 2) = o;
        }
        for (int i = tid; i < rows * 8; i += C32_THREADS) {  // pad columns: 2 x 64 B per row
            const int r = i >> 3, q = i & 7;
            const int col = (q >> 2) ? Fp - 1 : 0;
            *reinterpret_cast<uint4*>(slab + ((size_t)r * Fp + col) * 64 + (q & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
    // ---- input slab: rows t0 - 1 .. t0 + TT, 16 B per access.  The rows are one contiguous range of the utterance's map: buffer
    //      loads at src0 + 16 i, where a row before the utterance wraps to a huge unsigned offset and a row past it runs off the
    //      descriptor -- both return zeros, with no branch (a conditional load makes hipcc wait for the load right behind it);
    //      eight loads in flight per thread.  The two pad columns of every row are zeros.
    {
        const int per_row = a.F_in * 4;                     // 16-byte chunks of a time row
        const int total = rows * per_row;
        const float inv_row = 1.f / (float)per_row;
        const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x + (size_t)b * a.T * a.F_in * C32), 0,
                                                                             (unsigned)(a.T * a.F_in * 64), 0x00020000);
        const int src0 = (t0 - 1) * a.F_in * 64;
        for (int i0 = tid; i0 < total; i0 += 8 * C32_THREADS) {
            u32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * C32_THREADS;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, i < total ? (unsigned)(src0 + i * 16) : 0xfffffff0u, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * C32_THREADS;
                const int r = (int)(((float)i + 0.5f) * inv_row);       // i / per_row (exact: i < 2^16, per_row >= 8)
                if (i < total) *reinterpret_cast<u32x4_t*>(slab + (size_t)i * 16 + 64 + 128 * r) = v[k];
            }
        }
        for (int i = tid; i < rows * 8; i += C32_THREADS) {  // pad columns: 2 x 64 B per row
            const int r = i >> 3, q = i & 7;
            const int col = (q >> 2) ? Fp - 1 : 0;
            *reinterpret_cast<uint4*>(slab + ((size_t)r * Fp + col) * 64 + (q & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    }
    // ---- weight fragments (registers, whole workgroup lifetime)
    bf16x8 wf[2][9];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            wf[nt][tap] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(nt * 16 + li) * 288 + tap * 32 + g * 8);
    bf16x8 w2f[2];
    if (SF == 2 && a.y2) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w2f[nt] = *reinterpret_cast<const bf16x8*>(a.w2 + (size_t)(nt * 16 + li) * 32 + g * 8);
    } else {
        w2f[0] = w2f[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }

    __syncthreads();

    // ---- per-channel epilogue terms of this lane's 2 x 4 channels
    float bs[2][4], sc[2][4], sh[2][4], bs2[2][4], sc2[2][4], sh2[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = nt * 16 + g * 4 + r;
            bs[nt][r] = a.bias ? a.bias[c] : 0.f;
            sc[nt][r] = a.scale ? a.scale[c] : 1.f;
            sh[nt][r] = a.shift ? a.shift[c] : 0.f;
            bs2[nt][r] = (a.y2 && a.bias2) ? a.bias2[c] : 0.f;
            sc2[nt][r] = (a.y2 && a.scale2) ? a.scale2[c] : 1.f;
            sh2[nt][r] = (a.y2 && a.shift2) ? a.shift2[c] : 0.f;
        }

    const int npos = a.TT * a.F_out;
    const int ntile = (npos + 15) >> 4;
    const size_t out0 = ((size_t)b * a.T + t0) * a.F_out;  // first output position of the workgroup
    const int tvalid = min(a.TT, a.T - t0) * a.F_out;      // positions of rows inside the utterance
    // residual: the workgroup's output range of `res` (a descriptor of zero records when there is none: every load returns zero)
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.res ? a.res + out0 * C32 : a.x), 0,
                                                                         a.res ? (unsigned)(tvalid * 64) : 0u, 0x00020000);
    // the residual in the accumulator layout (4 channels of position p); no residual, a position past the utterance or a tile past
    // the last: an out-of-range offset, zeros.  Fetched TWO tiles of this wave ahead (two register sets, each refilled right after
    // its use): issued at the top of its own tile the load's HBM latency sat exposed in every one of a wave's ~12 tiles --
    // the residual convs ran 197 us at B = 256, F = 40 where the same conv without a residual took 100.
    auto res_load = [&](int tile, u32x2_t (&r)[2]) {
        const int p = tile * 16 + li;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            r[nt] = __builtin_amdgcn_raw_buffer_load_b64(rsrd, (tile < ntile && p < tvalid) ? (unsigned)(p * 64 + (nt * 16 + g * 4) * 2) : 0xfffffff0u, 0, 0);
    };
    auto do_tile = [&](int tile, const u32x2_t (&rraw)[2]) {
        const int p = tile * 16 + li;
        const int pc = min(p, npos - 1);
        const int tt = pc / a.F_out, fo = pc - tt * a.F_out;
        const char* base = slab + ((size_t)tt * Fp + fo * SF) * 64 + g * 16;     // tap (kt, kf): + (kt * Fp + kf) * 64
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, s0 = acc0, s1 = acc0;
        bf16x8 xf[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) xf[kt * 3 + kf] = *reinterpret_cast<const bf16x8*>(base + (kt * Fp + kf) * 64);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][tap], xf[tap], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][tap], xf[tap], acc1, 0, 0, 0);
        }
        if (SF == 2 && a.y2) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[0], xf[4], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[1], xf[4], s1, 0, 0, 0);
        }
        // epilogue in the accumulator layout (4 channels of position p), then the transpose
        float v[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[0][r] = (acc0[r] + bs[0][r]) * sc[0][r] + sh[0][r];
            v[1][r] = (acc1[r] + bs[1][r]) * sc[1][r] + sh[1][r];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bf16x4 rv = __builtin_bit_cast(bf16x4, rraw[nt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt][r] += (float)rv[r];
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(a.relu ? fmaxf(v[nt][r], 0.f) : v[nt][r]);
            *reinterpret_cast<bf16x4*>(stage + li * C32_SROW + (nt * 16 + g * 4) * 2) = o;
        }
        if (SF == 2 && a.y2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4& s = nt ? s1 : s0;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)((s[r] + bs2[nt][r]) * sc2[nt][r] + sh2[nt][r]);
                *reinterpret_cast<bf16x4*>(stage + 16 * C32_SROW + li * C32_SROW + (nt * 16 + g * 4) * 2) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int pos = lane >> 2, q = lane & 3;       // 16 positions x 4 chunks of 16 B = 1 KB contiguous
            const int pp = tile * 16 + pos;
            if (pp < tvalid) {
                *reinterpret_cast<uint4*>(a.y + (out0 + pp) * C32 + q * 8) = *reinterpret_cast<const uint4*>(stage + pos * C32_SROW + q * 16);
                if (SF == 2 && a.y2)
                    *reinterpret_cast<uint4*>(a.y2 + (out0 + pp) * C32 + q * 8) =
                        *reinterpret_cast<const uint4*>(stage + 16 * C32_SROW + pos * C32_SROW + q * 16);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    constexpr int WPG = C32_THREADS / 64;
    u32x2_t rA[2], rB[2];
    res_load(wv, rA);
    res_load(wv + WPG, rB);
    for (int tile = wv; tile < ntile; tile += 2 * WPG) {
        do_tile(tile, rA);
        res_load(tile + 2 * WPG, rA);
        if (tile + WPG < ntile) {
            do_tile(tile + WPG, rB);
            res_load(tile + 3 * WPG, rB);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// A whole stride-1 BasicResBlock (campplus.py:211-243 with the identity shortcut): out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x).
// As two launches the block moves x in, h out, h in, x in again, out: 975 MB at B = 256 on the F = 40 maps.  Here the
// workgroup's x slab carries a two-row halo, h = relu(bn1(conv1(x))) for TT + 2 rows goes into a second LDS slab (bf16, as the
// two-launch path stores it; rows outside the utterance are ZERO -- conv2 pads h, not x) and conv2 reads it from there; the
// residual is the centre of the x slab.  390 MB instead of 975.
struct RB32Args {
    const bf16_t* x; bf16_t* y;
    const bf16_t* w1; const float* b1; const float* s1; const float* h1;
    const bf16_t* w2; const float* b2; const float* s2; const float* h2;
    int B, T, F, TT;
};

__global__ __launch_bounds__(C32_THREADS) void resblock_c32_kernel(const RB32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * a.TT;
    const int Fp = a.F + 2;
    const int xrows = a.TT + 4, mrows = a.TT + 2;
    char* xslab = smem;                                           // rows t0 - 2 .. t0 + TT + 1
    char* mslab = smem + (size_t)xrows * Fp * 64;                 // rows t0 - 1 .. t0 + TT
    char* stage = mslab + (size_t)mrows * Fp * 64 + wv * (16 * C32_SROW);

    // ---- x slab (branch-free: out-of-range offsets read zeros), pad columns of both slabs
    {
        const int per_row = a.F * 4;
        const int total = xrows * per_row;
        const float inv_row = 1.f / (float)per_row;
        const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x + (size_t)b * a.T * a.F * C32), 0,
                                                                             (unsigned)(a.T * a.F * 64), 0x00020000);
        const int src0 = (t0 - 2) * a.F * 64;
        for (int i0 = tid; i0 < total; i0 += 8 * C32_THREADS) {
            u32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * C32_THREADS;
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, i < total ? (unsigned)(src0 + i * 16) : 0xfffffff0u, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * C32_THREADS;
                const int r = (int)(((float)i + 0.5f) * inv_row);
                if (i < total) *reinterpret_cast<u32x4_t*>(xslab + (size_t)i * 16 + 64 + 128 * r) = v[k];
            }
        }
        for (int i = tid; i < (xrows + mrows) * 8; i += C32_THREADS) {
            const int r = i >> 3, q = i & 7;
            const int col = (q >> 2) ? Fp - 1 : 0;
            char* base = r < xrows ? xslab + (size_t)r * Fp * 64 : mslab + (size_t)(r - xrows) * Fp * 64;
            *reinterpret_cast<uint4*>(base + col * 64 + (q & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    bf16x8 wf[2][9];
    auto load_w = [&](const bf16_t* w) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf[nt][tap] = *reinterpret_cast<const bf16x8*>(w + (size_t)(nt * 16 + li) * 288 + tap * 32 + g * 8);
    };
    auto params = [&](const float* bb, const float* ss, const float* hh, float (&bs)[2][4], float (&sc)[2][4], float (&sh)[2][4]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = nt * 16 + g * 4 + r;
                bs[nt][r] = bb ? bb[c] : 0.f; sc[nt][r] = ss ? ss[c] : 1.f; sh[nt][r] = hh ? hh[c] : 0.f;
            }
    };
    float bs[2][4], sc[2][4], sh[2][4];
    load_w(a.w1);
    params(a.b1, a.s1, a.h1, bs, sc, sh);
    __syncthreads();

    // ---- phase 1: h over rows t0 - 1 .. t0 + TT into the mid slab
    {
        const int npos = mrows * a.F;
        const int ntile = (npos + 15) >> 4;
        for (int tile = wv; tile < ntile; tile += C32_THREADS / 64) {
            const int p = tile * 16 + li;
            const int pc = min(p, npos - 1);
            const int r1 = pc / a.F, f = pc - r1 * a.F;
            const char* base = xslab + ((size_t)r1 * Fp + f) * 64 + g * 16;
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            bf16x8 xf[9];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) xf[kt * 3 + kf] = *reinterpret_cast<const bf16x8*>(base + (kt * Fp + kf) * 64);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][tap], xf[tap], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][tap], xf[tap], acc1, 0, 0, 0);
            }
            const int t = t0 - 1 + r1;
            const bool live = p < npos && t >= 0 && t < a.T;
            if (p < npos) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4& ac = nt ? acc1 : acc0;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(live ? fmaxf((ac[r] + bs[nt][r]) * sc[nt][r] + sh[nt][r], 0.f) : 0.f);
                    *reinterpret_cast<bf16x4*>(mslab + ((size_t)r1 * Fp + f + 1) * 64 + (nt * 16 + g * 4) * 2) = o;
                }
            }
        }
    }
    load_w(a.w2);
    params(a.b2, a.s2, a.h2, bs, sc, sh);
    __syncthreads();

    // ---- phase 2: out = relu(bn2(conv2(h)) + x)
    const int npos = a.TT * a.F;
    const int ntile = (npos + 15) >> 4;
    const size_t out0 = ((size_t)b * a.T + t0) * a.F;
    const int tvalid = min(a.TT, a.T - t0) * a.F;
    for (int tile = wv; tile < ntile; tile += C32_THREADS / 64) {
        const int p = tile * 16 + li;
        const int pc = min(p, npos - 1);
        const int tt = pc / a.F, fo = pc - tt * a.F;
        const char* base = mslab + ((size_t)tt * Fp + fo) * 64 + g * 16;
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        bf16x8 xf[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) xf[kt * 3 + kf] = *reinterpret_cast<const bf16x8*>(base + (kt * Fp + kf) * 64);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][tap], xf[tap], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][tap], xf[tap], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4& ac = nt ? acc1 : acc0;
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(xslab + ((size_t)(tt + 2) * Fp + fo + 1) * 64 + (nt * 16 + g * 4) * 2);
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf((ac[r] + bs[nt][r]) * sc[nt][r] + sh[nt][r] + (float)rv[r], 0.f);
            *reinterpret_cast<bf16x4*>(stage + li * C32_SROW + (nt * 16 + g * 4) * 2) = o;
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int pos = lane >> 2, q = lane & 3;
            const int pp = tile * 16 + pos;
            if (pp < tvalid) *reinterpret_cast<uint4*>(a.y + (out0 + pp) * C32 + q * 8) = *reinterpret_cast<const uint4*>(stage + pos * C32_SROW + q * 16);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// y = [relu]( bn(conv3x3(x) + bias) [+ res] ), optionally y2 = bn2(conv1x1_stride(x) + bias2) (stride_f == 2 only).
// Returns VP_EUNSUP when the shape is not covered (the caller falls back to the general conv GEMM).
int vp_conv3x3_c32_bf16(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv, const void* res, int relu,
                        const vp_tdnn_layer* shortcut, void* y2, int B, int T, int F_in, int stride_f, const void* c1_feats,
                        const float* c1_w, const float* c1_b, const float* c1_scale, const float* c1_shift, hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("VPMI_FCM_GENERAL"); off = e && atoi(e) ? 1 : 0; }
    if (off) return VP_EUNSUP;
    if (!conv || conv->cin != C32 || conv->cout != C32 || conv->kw != 9 || (stride_f != 1 && stride_f != 2) || F_in < 2 || B > 65535)
        return VP_EUNSUP;
    const bool fuse = c1_feats != nullptr;
    if (fuse && (stride_f != 2 || !c1_w || !c1_b || !c1_scale || !c1_shift)) return VP_EUNSUP;
    if (shortcut && (stride_f != 2 || !y2 || shortcut->cin != C32 || shortcut->cout != C32 || shortcut->kw != 1)) return VP_EUNSUP;
    if (!fuse && !x) return VP_EUNSUP;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(y2)) & 15)
        return VP_EUNSUP;
    const int F_out = (F_in - 1) / stride_f + 1;
    const int Fp = F_in + 2;
    int TT = (60 * 1024) / (Fp * 64) - 2;                  // slab of <= 60 KB: two workgroups per CU
    if (TT > T) TT = T;
    if (TT < 1) return VP_EUNSUP;
    if (fuse && TT > 2) TT -= 1;                           // room for the f32 feature rows behind the slab
    const size_t smem = (size_t)(TT + 2) * Fp * 64 + (size_t)(C32_THREADS / 64) * 2 * 16 * C32_SROW + (fuse ? (size_t)(TT + 4) * Fp * 4 : 0);
    C32Args a;
    memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.res = (const bf16_t*)res; a.y2 = shortcut ? (bf16_t*)y2 : nullptr;
    a.w = (const bf16_t*)conv->w; a.bias = conv->bias; a.scale = conv->bn_scale; a.shift = conv->bn_shift;
    if (shortcut) { a.w2 = (const bf16_t*)shortcut->w; a.bias2 = shortcut->bias; a.scale2 = shortcut->bn_scale; a.shift2 = shortcut->bn_shift; }
    a.B = B; a.T = T; a.F_in = F_in; a.F_out = F_out; a.TT = TT; a.relu = relu;
    a.feats = (const bf16_t*)c1_feats; a.c1_w = c1_w; a.c1_b = c1_b; a.c1_scale = c1_scale; a.c1_shift = c1_shift;
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        attr_set = true;
    }
    const dim3 grid((T + TT - 1) / TT, B);
    if (smem > 72 * 1024) return VP_EUNSUP;
    if (fuse) hipLaunchKernelGGL((conv3x3_c32_kernel<2, true>), grid, dim3(C32_THREADS), smem, st, a);
    else if (stride_f == 1) hipLaunchKernelGGL((conv3x3_c32_kernel<1, false>), grid, dim3(C32_THREADS), smem, st, a);
    else hipLaunchKernelGGL((conv3x3_c32_kernel<2, false>), grid, dim3(C32_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv3x3_c32");
    return VP_OK;
}

// out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x): a stride-1 BasicResBlock with the identity shortcut in one launch.
// VP_EUNSUP when the shape is not covered (the caller runs the two convs).
int vp_resblock_c32_bf16(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv1, const vp_tdnn_layer* conv2, int B, int T, int F,
                         hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("VPMI_FCM_UNFUSED"); off = e && atoi(e) ? 1 : 0; }
    if (off) return VP_EUNSUP;
    if (!x || !y || !conv1 || !conv2 || conv1->cin != C32 || conv1->cout != C32 || conv1->kw != 9 || conv2->cin != C32 || conv2->cout != C32 ||
        conv2->kw != 9 || F < 2 || B > 65535 || x == y)
        return VP_EUNSUP;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return VP_EUNSUP;
    const int Fp = F + 2;
    int TT = ((64 * 1024) / (Fp * 64) - 6) / 2;            // x slab (TT + 4 rows) + h slab (TT + 2 rows) <= 64 KB: two workgroups per CU
    if (TT > T) TT = T;
    if (TT < 2) return VP_EUNSUP;
    const size_t smem = (size_t)(2 * TT + 6) * Fp * 64 + (size_t)(C32_THREADS / 64) * 16 * C32_SROW;
    if (smem > 72 * 1024) return VP_EUNSUP;
    RB32Args a;
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y;
    a.w1 = (const bf16_t*)conv1->w; a.b1 = conv1->bias; a.s1 = conv1->bn_scale; a.h1 = conv1->bn_shift;
    a.w2 = (const bf16_t*)conv2->w; a.b2 = conv2->bias; a.s2 = conv2->bn_scale; a.h2 = conv2->bn_shift;
    a.B = B; a.T = T; a.F = F; a.TT = TT;
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_c32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(resblock_c32_kernel, dim3((T + TT - 1) / TT, B), dim3(C32_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "resblock_c32");
    return VP_OK;
}
