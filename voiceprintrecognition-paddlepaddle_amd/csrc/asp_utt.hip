// Attentive statistics pooling as ONE kernel per utterance (bf16 engine) -- AttentiveStatisticsPooling.forward, pooling.py:105-123:
//   h      = tanh(BN(ReLU(W_t x + b + rowbias)))          self.tdnn on cat([x, mean, std]): the [mean; std] columns of its weight are
//                                                          the per-utterance bias `rowbias` (a small dense layer ahead of this launch)
//   logits = W_c h (+ b_c: constant over time, it cancels in the softmax);  attn = softmax over time;
//   pooled = [sum_t attn x | sqrt(sum_t attn x^2 - mean^2)]
// Replaces two launches (the 1536 -> 128 conv GEMM on the 128-wide kernel, 84 us, + asp_fused, 85 us, at 256 x 298 frames).  A
// 512-thread workgroup owns ONE utterance (T <= 304 frames) and streams its x rows (T x C bf16, 0.92 MB) through LDS twice, both
// times as K-major stages of 32 channels fed by LDS-DMA rings with four stages in flight (the kernel is HBM-bound: 2 x 234 MB per
// batch; one workgroup per CU must keep ~100 KB in flight to reach the memory's rate):
//   phase 1  h = GEMM 320 x 128 x C: waves 4 (M) x 2 (N), 5 x 4 MFMA tiles each; stage = x (320 x 64 B) + W_t (128 x 64 B)
//   between  h -> bf16 -> LDS in the A-operand layout; every wave takes ITS frame quarter's fragments (5 tiles x 4 k-steps = 80 VGPRs)
//            into registers for good: the h tensor exists nowhere else
//   phase 2  per 32-channel block: logits tile = h (registers) x W_c block (8 KB of the stage); wave = (16 channels, frame quarter);
//            a lane holds its channel's 20 logits, takes max / exp2 / the three weighted sums over them (no online rescaling inside a
//            block), the four frame groups merge by shuffles, the four quarters through a small LDS area one block later.
//            Nothing in the loop is an ordinary load: beside LDS-DMAs in flight hipcc waits vmcnt(0) for any VGPR load and drains the
//            ring (first version: 161 us; 95 us of it was this phase issuing ~900 latency-exposed instructions per block).
// Roofline: HBM (algorithmic bytes: 2 x T*C*2 B per utterance).
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int AU_ATT = 128;
constexpr int AU_THREADS = 512;
constexpr int AU_MAXT = 304;                 // 19 MFMA row tiles
constexpr float AU_LOG2E = 1.4426950408889634f;
// phase 1 ring
constexpr int AU_KS = 32;                    // K per stage: 64-byte LDS rows
constexpr int AU_XROWS = 320;
constexpr int AU_XST = AU_XROWS * 64;        // 20,480 B
constexpr int AU_WST = AU_ATT * 64;          // 8,192 B
constexpr int AU_STAGE = AU_XST + AU_WST;    // 28,672 B
constexpr int AU_NST = 5;                    // 143,360 B; four stages in flight
// phase 2: h image (transient), then a five-slot ring of x (304 x 64 B) + W_c (32 x 256 B) stages, the centres, the merge area
constexpr int AU_HBYTES = AU_MAXT * 256;     // 77,824 B: h rows of 128 bf16, 16-byte chunks XOR-swizzled by row & 15
constexpr int AU_XST2 = AU_MAXT * 64;        // 19,456 B
constexpr int AU_STAGE2 = AU_XST2 + 32 * 256;              // 27,648 B
constexpr int AU_MRG = AU_NST * AU_STAGE;                  // 143,360 (above BOTH rings): [2 group parity][4 blocks][2 channel tiles][4 quarters][16][4] floats
constexpr int AU_SMEM = AU_MRG + 2 * 4 * 2 * 4 * 16 * 4 * 4;       // 159,744 B
static_assert(AU_NST * AU_STAGE <= AU_SMEM && 3 * AU_STAGE2 >= AU_HBYTES, "phase-1 ring fits; slots 3 and 4 lie above the h image");

typedef __attribute__((address_space(3))) void* au_lds_t;

struct AspUttArgs {
    const bf16_t* x;        // (B*T, ldx)
    const bf16_t* wt;       // [128][C]
    const float* bias; const float* rowbias; const float* bn_scale; const float* bn_shift;     // [128], (B, 128) or null, [128], [128]
    const bf16_t* wc;       // [C][128]
    float* pooled;          // (B, 2C)
    int ldx, T, C; float eps;
};

__device__ __forceinline__ float au_tanh(float v) {
    // tanh(v) = 1 - 2 / (exp(2 v) + 1) on the hardware exp2 / rcp (the conv epilogue's form)
    const float e = __builtin_amdgcn_exp2f(v * (2.f * AU_LOG2E));
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// (m, s0, s1, s2) <- merged with (m2, a0, a1, a2): running maximum (log2 domain) and the three sums relative to it.
// ROBUST: a side that holds no frame (weight sum 0: a lane or a frame quarter entirely past T) contributes nothing, whatever its
// maximum field says (the first version merged such a state arithmetically and produced NaN for every utterance shorter than 241
// frames); the plain form is for sides that always hold frames (the lanes of a quarter without frames past T).
template <bool ROBUST>
__device__ __forceinline__ void au_merge(float& m, float& s0, float& s1, float& s2, float m2, float a0, float a1, float a2) {
    if constexpr (ROBUST) {
        const bool e1 = !(s0 > 0.f), e2 = !(a0 > 0.f);
        const float ma = e1 ? m2 : m, mb = e2 ? ma : m2;
        const float M = fmaxf(ma, mb);
        const float f1 = e1 ? 0.f : __builtin_amdgcn_exp2f(m - M), f2 = e2 ? 0.f : __builtin_amdgcn_exp2f(m2 - M);
        s0 = e1 ? 0.f : s0 * f1; s1 = e1 ? 0.f : s1 * f1; s2 = e1 ? 0.f : s2 * f1;
        s0 += e2 ? 0.f : a0 * f2; s1 += e2 ? 0.f : a1 * f2; s2 += e2 ? 0.f : a2 * f2;
        m = M;
    } else {
        const float M = fmaxf(m, m2);
        const float f1 = __builtin_amdgcn_exp2f(m - M), f2 = __builtin_amdgcn_exp2f(m2 - M);
        s0 = fmaf(s0, f1, a0 * f2); s1 = fmaf(s1, f1, a1 * f2); s2 = fmaf(s2, f1, a2 * f2);
        m = M;
    }
}

__global__ __launch_bounds__(AU_THREADS, 1) void asp_utt_kernel(const AspUttArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int T = a.T, C = a.C;
    const bf16_t* xb = a.x + (size_t)b * T * a.ldx;
    // rows past T read as zeros: the descriptor ends with the utterance
    const unsigned xbytes = (unsigned)(((size_t)(T - 1) * a.ldx + C) * 2);
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb), 0, xbytes, 0x00020000);
    constexpr unsigned PAST = 0xf0000000u;
    const unsigned ldxb = (unsigned)a.ldx * 2u;
    // An x piece = 16 rows x 64 B: lane l lands at row l >> 2, position l & 3 and fetches the 16-byte chunk position ^ f(row).
    //   phase 1 reads MFMA fragments (ds_read_b128; rows li, chunk g):       f(row) = (-(row >> 2)) & 3   (conv_gemm_impl.h: amp_pos)
    //   phase 2 reads single values in the accumulator layout (rows 4 g + r, 32 B per frame group): f(row) = ((row >> 2) & 1) << 1 --
    //           a ds_read_u16 serves lanes 0-31 = frame groups g, g + 1 together, 256 B = one bank row apart: they get different halves
    const int prow = lane >> 2;
    const unsigned xl1 = (unsigned)prow * ldxb + (unsigned)(((lane & 3) ^ ((0 - (prow >> 2)) & 3)) << 4);
    const unsigned xl2 = (unsigned)prow * ldxb + (unsigned)(((lane & 3) ^ (((prow >> 2) & 1) << 1)) << 4);

    // ------------------------------------------------------------------------------------------------ phase 1
    {
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, (unsigned)(AU_ATT * C * 2), 0x00020000);
    const unsigned ldwb = (unsigned)C * 2u;
    const unsigned wl = (unsigned)prow * ldwb + (unsigned)(((lane & 3) ^ ((0 - (prow >> 2)) & 3)) << 4);
    const int NK = C / AU_KS;
    // x pieces wv, wv + 8, (wv + 16 for waves 0-3) of the stage's 20; W_t piece wv of its 8
    const bool four = wv < 4;
    auto issue = [&](int k, int slot) {
        char* st = smem + slot * AU_STAGE;
        const unsigned ko = k < NK ? (unsigned)k * (AU_KS * 2) : PAST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wv + 8 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (au_lds_t)(st + p * 1024), 16, xl1 + (ko + (unsigned)(p * 16) * ldxb), 0, 0, 0);
        }
        if (four) {
            const int p = wv + 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (au_lds_t)(st + p * 1024), 16, xl1 + (ko + (unsigned)(p * 16) * ldxb), 0, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (au_lds_t)(st + AU_XST + wv * 1024), 16, wl + (ko + (unsigned)(wv * 16) * ldwb), 0, 0, 0);
    };
    const int wm = wv >> 1, wn = wv & 1;
    const int fsw = ((g ^ ((0 - (li >> 2)) & 3)) << 4);                   // fragment chunk position (row = tile row li)
    const char* fx = smem + (wm * 80 + li) * 64 + fsw;
    const char* fw = smem + AU_XST + (wn * 64 + li) * 64 + fsw;
    f32x4 acc[5][4];
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    int slot = 0;
    for (int k = 0; k < NK; ++k) {
        // stage k has landed when at most the three younger stages' pieces are outstanding
        if (four) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // ... everyone's, and everyone is done reading stage k - 1
        asm volatile("" ::: "memory");
        issue(k + 4, slot == 0 ? AU_NST - 1 : slot - 1);                 // into the buffer stage k - 1 used
        const int so = slot * AU_STAGE;
        bf16x8 xf[5], wf[4];
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(fx + so + mi * 16 * 64);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(fw + so + ni * 16 * 64);
#pragma unroll
        for (int mi = 0; mi < 5; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[mi][ni], 0, 0, 0);
        slot = slot == AU_NST - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the tail's zero-fill DMAs still target the ring
    __syncthreads();                                                     // the ring is dead
    // h = tanh(bn(relu(acc + bias + rowbias))) -> bf16 rows in LDS [0, 76 KB), 16-byte chunks XOR-swizzled by row & 15 (the A-operand
    // layout of phase 2).  acc[mi][ni][r]: frame wm*80 + mi*16 + li, att wn*64 + ni*16 + g*4 + r.  (Ordinary loads: no DMA in flight here.)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n0 = wn * 64 + ni * 16 + g * 4;
        float bs[4], sc[4], sh[4];
        vp_load4(a.bias + n0, bs);
        if (a.rowbias) { float rb[4]; vp_load4(a.rowbias + (size_t)b * AU_ATT + n0, rb); bs[0] += rb[0]; bs[1] += rb[1]; bs[2] += rb[2]; bs[3] += rb[3]; }
        vp_load4(a.bn_scale + n0, sc);
        vp_load4(a.bn_shift + n0, sh);
        const int chunk = n0 >> 3, sub = (n0 & 7) * 2;
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) {
            const int m = wm * 80 + mi * 16 + li;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)au_tanh(fmaxf(acc[mi][ni][r] + bs[r], 0.f) * sc[r] + sh[r]);
            if (m < AU_MAXT) *reinterpret_cast<bf16x4*>(smem + m * 256 + ((chunk ^ (m & 15)) << 4) + sub) = o;
        }
    }
    }

    // ------------------------------------------------------------------------------------------------ phase 2
    // stage n = 32 channels: x (304 rows x 64 B = 19 pieces) + W_c rows (32 x 256 B = 8 pieces of 4 rows: lane l at row l >> 4, position
    // l & 15, fetching chunk position ^ (row & 15)).  Pieces per wave: x wv, wv + 8, (wv + 16 for waves 0-2), W_c wv.
    const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wc), 0, (unsigned)(C * AU_ATT * 2), 0x00020000);
    const int NC = C / 32;
    const bool four2 = wv < 3;
    auto issue2 = [&](int n, int slot) {
        char* st = smem + slot * AU_STAGE2;
        const bool live = n < NC;
        const unsigned xo = live ? (unsigned)n * 64u : PAST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wv + 8 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (au_lds_t)(st + p * 1024), 16, xl2 + (xo + (unsigned)(p * 16) * ldxb), 0, 0, 0);
        }
        if (four2) {
            const int p = wv + 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (au_lds_t)(st + p * 1024), 16, xl2 + (xo + (unsigned)(p * 16) * ldxb), 0, 0, 0);
        }
        // W_c piece wv: rows 4 wv .. 4 wv + 3 of the block; chunk swizzle by (row & 15) = (4 wv + (l >> 4)) & 15
        const unsigned sw = (unsigned)((((lane & 15) ^ ((4 * wv + (lane >> 4)) & 15))) << 4);
        const unsigned co = live ? (unsigned)(n * 32 + wv * 4 + (lane >> 4)) * 256u + sw : PAST;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(csrd, (au_lds_t)(st + AU_XST2 + wv * 1024), 16, co, 0, 0, 0);
    };
    // stages 0 and 1 go to slots 3 and 4, which lie above the h image; the wave then pulls its h fragments
    issue2(0, 3); issue2(1, 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // every wave's rows of h (and the centres) are in LDS
    asm volatile("" ::: "memory");
    const int ct = wv & 1, fq = wv >> 1;                                // 16 channels of the block; frame tiles [5 fq, 5 fq + 5)
    bf16x8 hf[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int row = (fq * 5 + i) * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + min(row, AU_MAXT - 1) * 256 + (((ks * 4 + g) ^ li) << 4));
            if (row >= AU_MAXT) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};       // tile 19 of the last quarter does not exist
            hf[i][ks] = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // everyone holds its fragments: the h image is dead
    asm volatile("" ::: "memory");
    issue2(2, 0); issue2(3, 1);
    // frames of this lane: t(i, r) = (5 fq + i) * 16 + 4 g + r; whole tiles past T only in the last quarter(s)
    const int tbase = fq * 80 + g * 4;
    const bool tail = fq * 80 + 80 > T;                                  // wave-uniform: this quarter holds frames past T
    const int qlive = T - fq * 80;                                       // live frames of the quarter (may be <= 0 or > 80)
    // x values of channel ct*16 + li: chunk ct*2 + (li >> 3) of the 64-byte row, swizzled by (row >> 2) & 3 = g
    const int xoff = tbase * 64 + ((((ct * 2 + (li >> 3)) ^ ((g & 1) << 1))) << 4) + (li & 7) * 2;
    const int woff = AU_XST2 + (ct * 16 + li) * 256;
    // Per block every quarter's wave leaves (max, sum p, sum p x, sum p x^2) of its 16 channels in the merge area; the four quarters of
    // FOUR blocks are merged in one go (64 lanes = 4 blocks x 16 channels) by the waves of quarter G & 3 after the group's last
    // barrier -- a finish per block cost its two waves ~140 instructions while the other six waited at the barrier.
    float* mg = reinterpret_cast<float*>(smem + AU_MRG);                // [group & 1][block & 3][ct][quarter][16 channels][m, s0, s1, s2]
    auto finish = [&](int G) {                                           // group G = blocks 4 G .. 4 G + 3, published by the last barrier
        if (fq == (G & 3)) {
            const int blk = lane >> 4, n = 4 * G + blk;
            const float* q = mg + ((((G & 1) * 4 + blk) * 2 + ct) * 4) * 64 + li * 4;
            float4 o = *reinterpret_cast<const float4*>(q);
            float m = o.x, s0 = o.y, s1 = o.z, s2 = o.w;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                o = *reinterpret_cast<const float4*>(q + k * 64);
                au_merge<true>(m, s0, s1, s2, o.x, o.y, o.z, o.w);
            }
            const float md = s1 / s0;
            const float var = s2 / s0 - md * md;
            const int c = n * 32 + ct * 16 + li;
            if (n < NC) {
                a.pooled[(size_t)b * 2 * C + c] = md;
                a.pooled[(size_t)b * 2 * C + C + c] = sqrtf(fmaxf(var, a.eps));
            }
        }
    };
    // The block loop is software-pipelined inside every wave: iteration n issues block n's MFMAs and x reads, then takes the statistics
    // of block n - 1 from the registers the previous iteration filled.  Measured with s_memtime stamps per wave (round 4, B = 256,
    // T = 298): a block takes ~4.2 k cycles = top (wait + barrier + DMA issue) ~1.1 k, front ~0.8 k, back ~2.4 k.  The back stretch is
    // VALU issue of the TWO waves that share a SIMD (2 x (170 VALU x 4 + 20 v_exp x 16) cycles): the older wave (0-3) finishes its
    // statistics in ~1.2 k cycles and waits at the barrier, the younger one (4-7) gets the issue slots it leaves and finishes ~1 k
    // later -- also when its quarter is entirely past T and it has nothing to compute.  The per-block barrier keeps all eight waves in
    // the same phase, so the VALU idles through top + front; a ring with two blocks per barrier does not fit the LDS beside the merge area.
    auto front = [&](int slot, f32x4 (&lg)[5], unsigned short (&xr)[5][4]) {       // block in `slot`: logits + its x values
        const char* st = smem + slot * AU_STAGE2;
        bf16x8 wf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(st + woff + (((ks * 4 + g) ^ li) << 4));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            lg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) lg[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hf[i][ks], wf[ks], lg[i], 0, 0, 0);
        }
        // the lane's 20 x values (offset (i, r) = i * 1024 + r * 64 from one address).  Plain ds_read_u16: an inline-asm
        // ds_read_u16_d16_hi (bf16 straight into the high half, no shift) measured 6-7 of 200 launches with one wave's block-0 result
        // off on a box where this form gave 0 of 200 twice (same session, tools/asp_determinism.py) -- not pursued further.
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[i][r] = *reinterpret_cast<const unsigned short*>(st + xoff + i * 1024 + r * 64);
    };
    auto back = [&](int n, const f32x4 (&lg)[5], const unsigned short (&xr)[5][4]) {       // statistics of block n -> its quarter state
        float mraw = -1e30f, s0 = 0.f, s1 = 0.f, s2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
        // A tile of 16 frames is wave-uniformly whole, cut by T, or past T: whole tiles run the plain code, tiles past T are skipped,
        // and the (at most one) cut tile masks its dead frames by ONE select on the softmax weight -- the x values and logits of frames
        // past T are finite (zero fill), so 0 x them is 0.  v_max3 by hand: fmaxf() costs a canonicalising v_max per MFMA output.
        auto max4 = [&](const f32x4& v) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mraw) : "v"(v[0]), "v"(v[1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mraw) : "v"(v[2]), "v"(v[3]));
        };
        if (!tail) {
#pragma unroll
            for (int i = 0; i < 5; ++i) max4(lg[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int lim = qlive - i * 16;                           // live frames of tile i (wave-uniform)
                if (lim >= 16) max4(lg[i]);
                else if (lim > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) mraw = (g * 4 + r < lim) ? fmaxf(mraw, lg[i][r]) : mraw;
                }
            }
        }
        // The channel's four lanes (l, l ^ 16, l ^ 32, l ^ 48) agree on ONE maximum before the exponentials, so their sums add up
        // without rescaling.  v_permlane32_swap a, b exchanges a's lanes 32-63 with b's lanes 0-31: with a = b = v, a (op) b is
        // v[l] (op) v[l ^ 32] in every lane; v_permlane16_swap does the same for the 16-lane rows.
        {
            float u = mraw, w = mraw;
            asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
            asm("v_max_f32 %0, %1, %2" : "=v"(mraw) : "v"(u), "v"(w));
            u = mraw; w = mraw;
            asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
            asm("v_max_f32 %0, %1, %2" : "=v"(mraw) : "v"(u), "v"(w));
        }
        const float moff = -mraw * AU_LOG2E;
        auto tile = [&](int i, auto masked, int lim) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(lg[i][r], AU_LOG2E, moff));
                const float xv = __builtin_bit_cast(float, (unsigned)xr[i][r] << 16);
                if constexpr (decltype(masked)::value) p = (g * 4 + r < lim) ? p : 0.f;
                const float tp = p * xv;
                if (r & 1) { t0 += p; t1 += tp; t2 = fmaf(tp, xv, t2); }             // two accumulator sets: shorter dependency chains
                else { s0 += p; s1 += tp; s2 = fmaf(tp, xv, s2); }
            }
        };
        if (!tail) {
#pragma unroll
            for (int i = 0; i < 5; ++i) tile(i, std::false_type{}, 16);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int lim = qlive - i * 16;
                if (lim >= 16) tile(i, std::false_type{}, 16);
                else if (lim > 0) tile(i, std::true_type{}, lim);
            }
        }
        s0 += t0; s1 += t1; s2 += t2;
        // butterfly: swap32(s0, s1) leaves sum s0 in lanes 0-31 and sum s1 in lanes 32-63; s2 against itself; swap16 of the two then
        // gives row 0 = S0, row 1 = S2, row 2 = S1 (row 3 = S2 again): lane (g, li) stores ONE word of channel li's state
        {
            float w = s2;
            asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s0), "+v"(s1));
            asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s2), "+v"(w));
            float u = s0 + s1;
            w += s2;
            asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
            u += w;
            const float word = g == 3 ? mraw * AU_LOG2E : u;             // log2 domain, like au_merge
            const int comp = g == 0 ? 1 : g == 1 ? 3 : g == 2 ? 2 : 0;
            mg[(((((n >> 2) & 1) * 4 + (n & 3)) * 2 + ct) * 4 + fq) * 64 + li * 4 + comp] = word;
        }
    };
    int slot = 3, nextG = 0;
    // top of iteration n: stage n has landed when at most the three younger stages' pieces are outstanding (the finishing waves' result
    // stores are older); lgkmcnt: the raw s_barrier does not wait for this wave's LDS store of its quarter state.  After the barrier:
    // everyone's pieces of stage n are in; stage n - 1 is read out (its slot is re-filled 4 stages ahead); the states of block n - 2 are
    // published -- a group of four blocks is merged as soon as its last state is.
    auto top = [&](int n) {
        if (four2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (n >= 2 && ((n - 2) & 3) == 3) finish(nextG++);              // block n - 2 was the group's last
        issue2(n + 4, slot == 0 ? 4 : slot - 1);                         // the slot stage n - 1 used
    };
    f32x4 lgA[5], lgB[5];
    unsigned short xrA[5][4], xrB[5][4];
    if (NC > 0) {
        top(0);
        front(slot, lgA, xrA);
        slot = slot == 4 ? 0 : slot + 1;
        int n = 1;
        for (; n + 1 < NC; n += 2) {
            top(n);
            front(slot, lgB, xrB);
            back(n - 1, lgA, xrA);
            slot = slot == 4 ? 0 : slot + 1;
            top(n + 1);
            front(slot, lgA, xrA);
            back(n, lgB, xrB);
            slot = slot == 4 ? 0 : slot + 1;
        }
        if (n < NC) {                                                    // NC even: one block left for the B set
            top(n);
            front(slot, lgB, xrB);
            back(n - 1, lgA, xrA);
            back(n, lgB, xrB);
        } else {
            back(n - 1, lgA, xrA);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    while (nextG * 4 < NC) finish(nextG++);
}

}  // namespace

// VP_EUNSUP when the shape is not covered (the caller runs the conv GEMM + asp_fused pair instead).
int vp_asp_utt_bf16(vp_ctx* ctx, const void* x, int ldx, const vp_tdnn_layer* tdnn, const float* rowbias, const void* conv_w,
                    const float* conv_b, int B, int T, int C, int att, float eps, float* pooled, hipStream_t st) {
    static const bool off = getenv("VPMI_ASP_SPLIT") != nullptr;      // A/B: the two-launch path
    if (off || att != AU_ATT || T < 1 || T > AU_MAXT || C < 64 || C % 64 || (ldx & 7) || !tdnn->bn_scale || !tdnn->bn_shift || !tdnn->bias ||
        tdnn->cin != C || tdnn->cout != att || tdnn->kw != 1 ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(tdnn->w) | reinterpret_cast<uintptr_t>(conv_w)) & 15) ||
        (size_t)T * ldx * 2 >= 0xe0000000ull)
        return VP_EUNSUP;
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(asp_utt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AU_SMEM));
        attr_set = true;
    }
    AspUttArgs a;
    a.x = (const bf16_t*)x; a.wt = (const bf16_t*)tdnn->w; a.bias = tdnn->bias; a.rowbias = rowbias; a.bn_scale = tdnn->bn_scale;
    a.bn_shift = tdnn->bn_shift; a.wc = (const bf16_t*)conv_w; a.pooled = pooled;
    (void)conv_b;                                                        // constant over time: cancels in the softmax
    a.ldx = ldx; a.T = T; a.C = C; a.eps = eps;
    hipLaunchKernelGGL(asp_utt_kernel, dim3(B), dim3(AU_THREADS), AU_SMEM, st, a);
    VP_LAUNCH_CHECK(ctx, "asp_utt");
    return VP_OK;
}
