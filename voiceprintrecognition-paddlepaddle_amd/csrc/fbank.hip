// Kaldi Fbank + CMN on gfx950.
//
// Replaces AudioFeaturizer.forward (ppvector/data_utils/featurizer.py:33-60) for feature_method
// 'Fbank' (KaldiFbank.forward, featurizer.py:88-101 -> paddleaudio.compliance.kaldi.fbank).
//
// Bound: HBM: algorithmic bytes per utterance = 4*L (waveform in) + 4*T*n_mels
// (features out) = 287,360 B for 3 s @ 16 kHz, 80 mels.
//
// Kernel 1 (fbank_frames): one workgroup = 4 waves = a tile of FRAMES_PER_WG consecutive frames of
//   one utterance; each wave owns one frame at a time: coalesced load of the 400-sample window
//   (overlapping windows are L1/L2 hits, HBM sees each sample once), DC removal by a wave
//   reduction, pre-emphasis + Povey window on the fly, a 256-point complex Stockham radix-4 FFT of
//   the even/odd-packed frame in LDS (4 stages, one radix-4 butterfly per lane per stage), real-FFT
//   unpack to the 256 power bins, sparse triangular mel bank (CSR, <= ~18 taps per filter), log.
//   Writes raw log-mel (B,T,F) and per-tile column sums for the CMN.
// Kernel 2 (fbank_cmn): subtracts the per-utterance time mean, applies the length mask, optionally
//   emits the bf16 copy the bf16 network consumes.  The feature tensor (24 MB at B=256) is
//   L2/Infinity-Cache resident between the two kernels.
#include "common.h"

#include <math.h>
#include <algorithm>
#include <vector>

namespace {

constexpr int FB_WAVES = 4;
constexpr int FRAMES_PER_WG = 16;
constexpr int FB_MAX_WIN = 512;
constexpr int FB_NFFT = 512;          // only the 512-point (25 ms @ 16 kHz) geometry is built
constexpr int FB_NC = FB_NFFT / 2;    // complex FFT length
constexpr int FB_MAX_MEL = 128;
constexpr int FB_MAX_NNZ = 1024;

struct FbankArgs {
    const float* wav;
    float* out;
    float* psum;       // [B][tiles][n_mels]
    const float* window;
    const float2* tw;  // [512] e^{-2 pi i k/512}
    const int* mel_start;
    const int* mel_bin0;
    const float* mel_w;
    int B, L, T, tiles, win, shift, n_mels, nnz;
    float preemph, log_floor;
    int remove_dc;
};

// FFT work-buffer index with one padding slot every 16 complex points (the Stockham scatter strides
// 4 / 16 / 64 points would otherwise hit the same LDS banks 8-16 ways)
__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(FB_WAVES * 64) void fbank_frames_kernel(FbankArgs a) {
    __shared__ float s_win[FB_MAX_WIN];
    __shared__ float s_melw[FB_MAX_NNZ];
    __shared__ int s_mstart[FB_MAX_MEL + 1];
    __shared__ int s_mbin0[FB_MAX_MEL];
    __shared__ float2 s_buf[FB_WAVES][2][FB_NC + FB_NC / 16];   // one pad slot per 16: breaks the power-of-2 strides
    __shared__ float s_pow[FB_WAVES][FB_NC];
    __shared__ float s_red[FB_WAVES][FB_MAX_MEL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tile = blockIdx.x;
    const int b = blockIdx.y;

    for (int i = tid; i < a.win; i += FB_WAVES * 64) s_win[i] = a.window[i];
    for (int i = tid; i < a.nnz; i += FB_WAVES * 64) s_melw[i] = a.mel_w[i];
    for (int i = tid; i <= a.n_mels; i += FB_WAVES * 64) s_mstart[i] = a.mel_start[i];
    for (int i = tid; i < a.n_mels; i += FB_WAVES * 64) s_mbin0[i] = a.mel_bin0[i];
    // The kernel is bound by LDS instruction issue (~160 per lane per frame in its first version), so everything that is
    // constant per lane lives in registers: the three twiddles of FFT stages 1..3 (stage 0's are 1) and the four
    // real-FFT unpack twiddles.  The frame itself never goes through LDS: a lane loads the two samples of each of its
    // packed points straight from global memory (fully coalesced) and gets the pre-emphasis neighbour by a lane shift.
    float2 twr[3][3], twu[4];
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        const int jm = lane & ((1 << (2 * s)) - 1);
        const int twstep = (FB_NFFT / 4) >> (2 * s);
#pragma unroll
        for (int q = 0; q < 3; ++q) twr[s - 1][q] = a.tw[((q + 1) * jm * twstep) & (FB_NFFT - 1)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) twu[r] = a.tw[lane + 64 * r];
    __syncthreads();

    const float* wav = a.wav + (size_t)b * a.L;
    float acc0 = 0.f, acc1 = 0.f;   // column sums for mel bins lane and lane+64

    // packed point n = lane + 64 r holds samples 2n, 2n+1; the NEXT frame's samples are fetched while this one is transformed
    float nx0[4], nx1[4];
    auto fetch = [&](int t) {
        const float* src = wav + (size_t)min(t, a.T - 1) * a.shift;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 2 * (lane + 64 * r);
            nx0[r] = src[min(i0, a.win - 1)];
            nx1[r] = src[min(i0 + 1, a.win - 1)];
        }
    };
    fetch(tile * FRAMES_PER_WG + wv);

    for (int fi = wv; fi < FRAMES_PER_WG; fi += FB_WAVES) {
        const int t = tile * FRAMES_PER_WG + fi;
        if (t >= a.T) break;                         // wave-uniform
        float x0[4], x1[4];
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 2 * (lane + 64 * r);
            x0[r] = nx0[r]; x1[r] = nx1[r];
            part += (i0 < a.win ? x0[r] : 0.f) + (i0 + 1 < a.win ? x1[r] : 0.f);
        }
        fetch(t + FB_WAVES);                         // clamped; unused past the tile / utterance end
        const float mean = a.remove_dc ? vp_wave_sum(part) / (float)a.win : 0.f;
        // even/odd pack: z[n] = y[2n] + i y[2n+1], y[i] = ((x[i] - mean) - preemph (x[max(i-1,0)] - mean)) window[i], zero past the window
        float2* d0 = s_buf[wv][0];
        float2* d1 = s_buf[wv][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            const int i0 = 2 * n, i1 = 2 * n + 1;
            float prev = __shfl_up(x1[r], 1);                                   // x[2n - 1] lives in the lane below
            const float wrap = r > 0 ? __shfl(x1[r > 0 ? r - 1 : 0], 63) : x0[0];   // lane 0: last sample of the previous 64-point chunk
            if (lane == 0) prev = r > 0 ? wrap : x0[0];                         // frame start: replicate sample 0
            float re = 0.f, im = 0.f;
            if (i0 < a.win) re = ((x0[r] - mean) - a.preemph * (prev - mean)) * s_win[i0];
            if (i1 < a.win) im = ((x1[r] - mean) - a.preemph * (x0[r] - mean)) * s_win[i1];
            d0[pidx(n)] = make_float2(re, im);
        }
        __builtin_amdgcn_wave_barrier();
        // 256-point complex FFT: Stockham radix-4, Ns = 1, 4, 16, 64; thread j = lane
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int Ns = 1 << (2 * s);
            const int jm = lane & (Ns - 1);
            float2 v0 = d0[pidx(lane)];
            float2 v1 = d0[pidx(lane + 64)];
            float2 v2 = d0[pidx(lane + 128)];
            float2 v3 = d0[pidx(lane + 192)];
            if (s > 0) {
                v1 = cmul(v1, twr[s > 0 ? s - 1 : 0][0]);
                v2 = cmul(v2, twr[s > 0 ? s - 1 : 0][1]);
                v3 = cmul(v3, twr[s > 0 ? s - 1 : 0][2]);
            }
            // radix-4 butterfly, forward (W4 = -i)
            float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y);
            float2 d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
            float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y);
            float2 d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
            float2 o0 = make_float2(s02.x + s13.x, s02.y + s13.y);
            float2 o2 = make_float2(s02.x - s13.x, s02.y - s13.y);
            float2 o1 = make_float2(d02.x + d13.y, d02.y - d13.x);   // d02 - i*d13
            float2 o3 = make_float2(d02.x - d13.y, d02.y + d13.x);   // d02 + i*d13
            const int idx = ((lane >> (2 * s)) << (2 * s + 2)) + jm;  // (j/Ns)*Ns*4 + j%Ns
            d1[pidx(idx)] = o0;
            d1[pidx(idx + Ns)] = o1;
            d1[pidx(idx + 2 * Ns)] = o2;
            d1[pidx(idx + 3 * Ns)] = o3;
            __builtin_amdgcn_wave_barrier();
            float2* tmp = d0; d0 = d1; d1 = tmp;
        }
        // real-FFT unpack -> power bins 0..255 (Nyquist bin has zero mel weight in Kaldi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            float2 zk = d0[pidx(k)];
            float2 zn = d0[pidx((FB_NC - k) & (FB_NC - 1))];
            float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));  // (zk - conj(zn)) / (2i)
            float2 wo = cmul(o, twu[r]);
            float xr = e.x + wo.x, xi = e.y + wo.y;
            s_pow[wv][k] = xr * xr + xi * xi;
        }
        __builtin_amdgcn_wave_barrier();
        float* orow = a.out + ((size_t)b * a.T + t) * a.n_mels;
        // mel bins 0..63: one lane per filter.  Bins 64.. are the few widest filters (up to ~20 taps at 80 mels): 4 (or 2) lanes
        // share one of them, a quarter of the taps each, and add up in fixed order -- instead of 16 lanes walking 20 taps
        // while 48 idle.
        auto mel_dot = [&](int m, int sub, int nsub) {
            const int s0 = s_mstart[m], n = s_mstart[m + 1] - s0, k0 = s_mbin0[m];
            const int per = (n + nsub - 1) / nsub;
            const int q0 = sub * per, q1 = min(n, q0 + per);
            float e = 0.f;
            // 4 taps per trip, loads independent of the accumulator: one LDS latency per 4 taps
            for (int q = q0; q < q1; q += 4) {
                float wq[4], pq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = q + u < q1;
                    const float wv_ = s_melw[min(s0 + q + u, a.nnz - 1)];
                    wq[u] = in ? wv_ : 0.f;
                    pq[u] = s_pow[wv][min(k0 + q + u, FB_NC - 1)];
                }
                e += wq[0] * pq[0];
                e += wq[1] * pq[1];
                e += wq[2] * pq[2];
                e += wq[3] * pq[3];
            }
            return e;
        };
        if (lane < a.n_mels) {
            const float v = logf(fmaxf(mel_dot(lane, 0, 1), a.log_floor));
            orow[lane] = v;
            acc0 += v;
        }
        if (a.n_mels > 64) {
            const int nb2 = a.n_mels - 64;
            const int lpb = nb2 <= 16 ? 4 : (nb2 <= 32 ? 2 : 1);         // lanes per bin (wave-uniform)
            const int m = 64 + lane / lpb, sub = lane % lpb;
            float e = m < a.n_mels ? mel_dot(m, sub, lpb) : 0.f;
            if (lpb >= 2) e += __shfl_xor(e, 1);
            if (lpb >= 4) e += __shfl_xor(e, 2);
            const float v = logf(fmaxf(e, a.log_floor));
            if (m < a.n_mels && sub == 0) orow[m] = v;
            // column sum slot lane + 64 belongs to bin 64 + lane: hand the value to that lane
            const float vv = __shfl(v, (lane * lpb) & 63);
            if (lane < nb2) acc1 += vv;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // per-tile column sums (deterministic: fixed wave order)
    s_red[wv][lane] = acc0;
    s_red[wv][lane + 64] = acc1;
    __syncthreads();
    if (tid < a.n_mels) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FB_WAVES; ++w) s += s_red[w][tid];
        a.psum[((size_t)b * a.tiles + tile) * a.n_mels + tid] = s;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// fbank_frames16_kernel: four frames per wave, the 256-point complex FFT as 16 x 16 (two register-resident radix-16
// passes, ONE exchange through LDS) instead of one frame per wave and four radix-4 passes through LDS.
//   lane = (frame f = lane >> 4 of the wave's four, j = lane & 15).  With packed point n = 16 n1 + n2 and bin k = k1 + 16 k2:
//     X[k1 + 16 k2] = sum_n2 W16^(n2 k2) * [ W256^(n2 k1) * sum_n1 z[16 n1 + n2] W16^(n1 k1) ]
//   pass 1: lane j = n2 holds z[16 n1 + j], n1 = 0..15, straight from global memory (16 lanes read 128 contiguous bytes per
//           n1; points past the window are literal zeros the compiler prunes from the butterflies), DFT-16 over n1, twiddle;
//   exchange: element (k1, n2) of the frame goes to LDS slot k1 * 17 + n2 (rows padded by one slot: conflict-free both ways, and
//           every address is a lane base plus an immediate);
//   pass 2: lane j = k1 reads its 16 n2, DFT-16 over n2 -> X[j + 16 k2]; Z goes back to LDS in bin order;
//   unpack: lane j takes the bin pairs (k, 256 - k), k = j + 16 r, r = 0..7 (they share e and w o); no other lane reads those two
//           slots, so the (4x-scaled: the mel weights carry the 0.25) powers overwrite them in place;
//   mel: rounds of 16 filters, lane j = filter 16 r + j of frame f, taps padded to the round's longest filter (filters of a
//        round have similar widths), weights from LDS in [tap][16] order; log; 64 contiguous bytes per frame and round.
// The old kernel ran ~600 VALU + LDS instructions per frame-wave; this one ~190.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int F16_MAX_TAPS = 32;             // longest filter (taps) a round may have: the zeroed tail behind bin 255
constexpr int F16_ZSTRIDE = 2048 + F16_MAX_TAPS * 8 + 128;   // bytes per frame in the per-wave exchange buffer: 256 slots of 8 B, the
                                             // zero tail, and 128 B that put odd frames on the other 32 banks (stride = 128 mod 256)
constexpr int F16_MAX_WPAD = 1536;           // padded mel taps (floats) the kernel keeps in LDS
constexpr int F16_MAX_ROUNDS = FB_MAX_MEL / 16;

struct Fbank16Args {
    const float* wav; float* out; float* psum; const float* window; const float2* tw;
    const float* wpad; const int* mel_bin0;
    const short* wav16; float pcm_scale;       // PCM16 variant: 16-bit samples widened (x pcm_scale) as they are loaded; wav unused
    int B, L, T, tiles, win, shift, n_mels, n_rounds, wpad_len;
    int round_off[F16_MAX_ROUNDS], round_max[F16_MAX_ROUNDS];
    float preemph, log_floor;
    int remove_dc;
};

__device__ __forceinline__ v2f c_mul(v2f a, v2f b) { return v2f{a.x, a.x} * b + v2f{a.y, a.y} * v2f{-b.y, b.x}; }

__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {          // forward, W4 = -i
    const v2f s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    const v2f r13 = v2f{d13.y, -d13.x};                                             // -i * d13
    a0 = s02 + s13; a1 = d02 + r13; a2 = s02 - s13; a3 = d02 - r13;
}

// in-place 16-point forward DFT as 4 x 4; output k lives in v[4 (k & 3) + (k >> 2)]
__device__ __forceinline__ void dft16(v2f (&v)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);          // v[4 c + b] = t[b][c]
    // t[b][c] *= W16^(b c)
    v[4 * 1 + 1] = c_mul(v[4 * 1 + 1], v2f{c1, -s1});
    v[4 * 1 + 2] = c_mul(v[4 * 1 + 2], v2f{h, -h});
    v[4 * 1 + 3] = c_mul(v[4 * 1 + 3], v2f{s1, -c1});
    v[4 * 2 + 1] = c_mul(v[4 * 2 + 1], v2f{h, -h});
    v[4 * 2 + 2] = v2f{v[4 * 2 + 2].y, -v[4 * 2 + 2].x};
    v[4 * 2 + 3] = c_mul(v[4 * 2 + 3], v2f{-h, -h});
    v[4 * 3 + 1] = c_mul(v[4 * 3 + 1], v2f{s1, -c1});
    v[4 * 3 + 2] = c_mul(v[4 * 3 + 2], v2f{-h, -h});
    v[4 * 3 + 3] = c_mul(v[4 * 3 + 3], v2f{-c1, s1});
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);   // v[4 c + d] = Y[c + 4 d]
}
#define DFT16_OUT(v, k) v[4 * ((k) & 3) + ((k) >> 2)]

// NP1 = packed-point rows (of 16) that can be non-zero: ceil(win / 32); 13 for the 25 ms window at 16 kHz
template <int NP1, bool PCM16>
__global__ __launch_bounds__(FB_WAVES * 64, 3) void fbank_frames16_kernel(Fbank16Args a) {
    __shared__ __attribute__((aligned(16))) float s_win[FB_NFFT];
    __shared__ __attribute__((aligned(16))) float2 s_tw1[256];                   // W256^(n2 k1) at [n2 * 16 + k1]
    __shared__ __attribute__((aligned(16))) float s_wpad[F16_MAX_WPAD];
    __shared__ int s_mbin0[FB_MAX_MEL];
    __shared__ __attribute__((aligned(16))) char s_x[FB_WAVES][4 * F16_ZSTRIDE];
    __shared__ float s_red[2][FB_WAVES][FB_MAX_MEL];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int f = lane >> 4, j = lane & 15;

    for (int i = tid; i < FB_NFFT; i += FB_WAVES * 64) s_win[i] = i < a.win ? a.window[i] : 0.f;
    { const int k1 = tid >> 4, n2 = tid & 15; s_tw1[tid] = a.tw[(2 * n2 * k1) & (FB_NFFT - 1)]; }   // [k1][n2]: a wave's 16 lanes (n2) read 128 contiguous bytes
    for (int i = tid; i < a.wpad_len; i += FB_WAVES * 64) s_wpad[i] = a.wpad[i];
    for (int i = tid; i < FB_MAX_MEL; i += FB_WAVES * 64) s_mbin0[i] = i < a.n_mels ? a.mel_bin0[i] : 0;
    v2f twu[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { const float2 w = a.tw[j + 16 * r]; twu[r] = v2f{w.x, w.y}; }

    // Resident workgroups walk the (utterance, tile) list: the tables above are staged once, and the samples of the NEXT tile are
    // fetched while this one is transformed.  samples 2n, 2n+1 of packed point n = 16 n1 + j; the wave's frame is clamped into
    // the utterance (frames past T are computed on frame T-1 and never stored); reads past the utterance return zero.
    const int total = a.tiles * a.B;
    v2u raw[NP1];
    auto fetch = [&](int id) {
        const int bb = id / a.tiles, tt = id - bb * a.tiles;
        const int tf = min(tt * FRAMES_PER_WG + wv * 4 + f, a.T - 1);
        if constexpr (PCM16) {                             // two 16-bit samples per lane and row: 64 contiguous bytes per 16 lanes
            const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(a.wav16 + (size_t)bb * a.L), 0,
                                                                                 (unsigned)a.L * 2u, 0x00020000);
            const unsigned base = (unsigned)(tf * a.shift + 2 * j) * 2u;
#pragma unroll
            for (int n1 = 0; n1 < NP1; ++n1) raw[n1].x = __builtin_amdgcn_raw_buffer_load_b32(wsrd, base + (unsigned)n1 * 64u, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wav + (size_t)bb * a.L), 0,
                                                                                 (unsigned)a.L * 4u, 0x00020000);
            const unsigned base = (unsigned)(tf * a.shift + 2 * j) * 4u;
#pragma unroll
            for (int n1 = 0; n1 < NP1; ++n1) raw[n1] = __builtin_amdgcn_raw_buffer_load_b64(wsrd, base + (unsigned)n1 * 128u, 0, 0);
        }
    };
    if ((int)blockIdx.x < total) fetch(blockIdx.x);
    __syncthreads();
    int par = 0;
    for (int id = blockIdx.x; id < total; id += gridDim.x, par ^= 1) {
    const int b = id / a.tiles, tile = id - b * a.tiles;
    const int t = tile * FRAMES_PER_WG + wv * 4 + f;
    const int tl = min(t, a.T - 1);

    // ---- DC removal, pre-emphasis, window, even/odd pack
    v2f x[NP1];
    float part = 0.f;
#pragma unroll
    for (int n1 = 0; n1 < NP1; ++n1) {
        if constexpr (PCM16) {
            const unsigned pr = raw[n1].x;                  // low half = sample 2n, high half = sample 2n + 1 (sign-extended)
            x[n1] = v2f{(float)(short)(pr & 0xffffu) * a.pcm_scale, (float)((int)pr >> 16) * a.pcm_scale};
        } else {
            x[n1] = __builtin_bit_cast(v2f, raw[n1]);       // whole-vector cast (an element-wise bit_cast of an ext-vector lvalue reads element 0)
        }
        const int i0 = 32 * n1 + 2 * j;
        if (n1 < NP1 - 1 && NP1 < 16) part += x[n1].x + x[n1].y;                 // NP1 = ceil(win / 32): only the last row can cross the window end
        else part += (i0 < a.win ? x[n1].x : 0.f) + (i0 + 1 < a.win ? x[n1].y : 0.f);
    }
    if (id + (int)gridDim.x < total) fetch(id + gridDim.x);
    if (a.remove_dc) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) part += __shfl_xor(part, o);
        const float mean = part / (float)a.win;
#pragma unroll
        for (int n1 = 0; n1 < NP1; ++n1) x[n1] -= v2f{mean, mean};
    }
    v2f v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = v2f{0.f, 0.f};
#pragma unroll
    for (int n1 = 0; n1 < NP1; ++n1) {
        // x[2n - 1]: the odd sample of lane j - 1 (DPP row_shr:1), or for lane 0 of lane 15 one row up (DPP row_ror:1 of the
        // previous register, kept where row_shr has no source lane); the very first sample replicates itself
        const float odd = x[n1].y, odd_up = x[n1 > 0 ? n1 - 1 : 0].y, first = x[0].x;      // scalars: see the bit_cast note above
        const int wrap = n1 > 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(odd_up), 0x121, 0xf, 0xf, false) : __float_as_int(first);
        const float prev = __int_as_float(__builtin_amdgcn_update_dpp(wrap, __float_as_int(odd), 0x111, 0xf, 0xf, false));
        const float2 w = *reinterpret_cast<const float2*>(&s_win[32 * n1 + 2 * j]);
        v[n1] = v2f{(x[n1].x - a.preemph * prev) * w.x, (x[n1].y - a.preemph * x[n1].x) * w.y};
    }
    // ---- pass 1: DFT-16 over n1, twiddle by W256^(j k1), exchange
    dft16(v);
    char* xb = s_x[wv] + f * F16_ZSTRIDE;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        v2f o = DFT16_OUT(v, k1);
        if (k1 > 0) { const float2 w = s_tw1[k1 * 16 + j]; o = c_mul(o, v2f{w.x, w.y}); }      // (was [n2][k1]: 16 lanes 128 B apart = one bank pair, PMC: 64 % of the LDS cycles were conflicts)
        *reinterpret_cast<v2f*>(xb + (k1 * 17 + j) * 8) = o;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = *reinterpret_cast<const v2f*>(xb + (j * 17 + n2) * 8);
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2: DFT-16 over n2 -> Z[j + 16 k2], back to LDS in bin order
    dft16(v);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) *reinterpret_cast<v2f*>(xb + (j + 16 * k2) * 8) = DFT16_OUT(v, k2);
    __builtin_amdgcn_wave_barrier();
    // ---- real-FFT unpack of the pairs (k, 256 - k): 2 X[k] = e + w o, 2 X[256 - k] = conj(e - w o),
    //      e = Z[k] + conj(Z[256 - k]), o = (Z[k] - conj(Z[256 - k])) / i, w = W512^k.  The mel weights carry the 0.25.
    //      All pairs are read first; the powers then go back as a COMPACT float array P[k] at xb + 4 k (over the dead Z): the mel
    //      taps below read consecutive dwords instead of every other one.
    float pw[16];
    float p128 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = j + 16 * r;
        const v2f zk = *reinterpret_cast<const v2f*>(xb + k * 8);
        const v2f zn = *reinterpret_cast<const v2f*>(xb + ((256 - k) & 255) * 8);
        const v2f e = v2f{zk.x + zn.x, zk.y - zn.y};
        const v2f o = v2f{zk.y + zn.y, zn.x - zk.x};
        const v2f wo = c_mul(o, twu[r]);
        const v2f p = e + wo, q = e - wo;
        pw[2 * r] = p.x * p.x + p.y * p.y;
        pw[2 * r + 1] = q.x * q.x + q.y * q.y;
    }
    if (j == 0) {                                       // bin 128 pairs with itself: |X[128]|^2 = |Z[128]|^2 (scaled like the rest)
        const v2f z = *reinterpret_cast<const v2f*>(xb + 128 * 8);
        p128 = 4.f * (z.x * z.x + z.y * z.y);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = j + 16 * r;
        *reinterpret_cast<float*>(xb + k * 4) = pw[2 * r];
        if (k > 0) *reinterpret_cast<float*>(xb + (256 - k) * 4) = pw[2 * r + 1];
    }
    if (j == 0) *reinterpret_cast<float*>(xb + 128 * 4) = p128;
#pragma unroll
    for (int r = 0; r < F16_MAX_TAPS / 16; ++r) *reinterpret_cast<float*>(xb + (256 + j + 16 * r) * 4) = 0.f;   // taps past bin 255 (zero weight) must read finite
    __builtin_amdgcn_wave_barrier();
    // ---- mel rounds: lane j = filter 16 r + j
    float* orow = a.out + ((size_t)b * a.T + tl) * a.n_mels;
    const bool tvalid = t < a.T;
    float csum[F16_MAX_ROUNDS];
#pragma unroll
    for (int r = 0; r < F16_MAX_ROUNDS; ++r) {
        csum[r] = 0.f;
        if (r < a.n_rounds) {                           // uniform
            const int m = 16 * r + j;
            const float* wp = s_wpad + a.round_off[r] + j;
            const char* pp = xb + s_mbin0[min(m, FB_MAX_MEL - 1)] * 4;
            const int nq = a.round_max[r];              // multiple of 4 (host-padded)
            float e0 = 0.f, e1 = 0.f;
            for (int q = 0; q < nq; q += 4) {
                const float w0 = wp[(q + 0) * 16], w1 = wp[(q + 1) * 16], w2 = wp[(q + 2) * 16], w3 = wp[(q + 3) * 16];
                const float p0 = *reinterpret_cast<const float*>(pp + q * 4), p1 = *reinterpret_cast<const float*>(pp + q * 4 + 4);
                const float p2 = *reinterpret_cast<const float*>(pp + q * 4 + 8), p3 = *reinterpret_cast<const float*>(pp + q * 4 + 12);
                e0 += w0 * p0; e1 += w1 * p1; e0 += w2 * p2; e1 += w3 * p3;
            }
            const float val = __logf(fmaxf(e0 + e1, a.log_floor));
            if (m < a.n_mels && tvalid) { orow[m] = val; csum[r] = val; }
        }
    }
    // ---- per-tile column sums (fixed order: frames of a wave by shuffles, then the four waves); two buffers, one barrier per tile
#pragma unroll
    for (int r = 0; r < F16_MAX_ROUNDS; ++r) {
        if (r < a.n_rounds) {
            float s = csum[r];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (f == 0) s_red[par][wv][16 * r + j] = s;
        }
    }
    __syncthreads();
    if (tid < a.n_mels) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FB_WAVES; ++w) s += s_red[par][w][tid];
        a.psum[((size_t)b * a.tiles + tile) * a.n_mels + tid] = s;
    }
    }
}

struct CmnArgs {
    float* out;
    bf16_t* out_bf16;
    const float* psum;
    const float* lens_ratio;
    int B, T, tiles, n_mels;
};

__global__ __launch_bounds__(256) void fbank_cmn_kernel(CmnArgs a) {
    __shared__ float s_mean[FB_MAX_MEL];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < a.n_mels) {
        float s = 0.f;
        for (int i = 0; i < a.tiles; ++i) s += a.psum[((size_t)b * a.tiles + i) * a.n_mels + tid];
        s_mean[tid] = s / (float)a.T;
    }
    __syncthreads();
    int valid = a.T;
    if (a.lens_ratio) valid = (int)(a.lens_ratio[b] * (float)a.T);      // astype(int32): truncation
    const int per_blk = (((a.T * a.n_mels + gridDim.x - 1) / gridDim.x) + 3) & ~3;
    const int e0 = blockIdx.x * per_blk;
    const int e1 = min(e0 + per_blk, a.T * a.n_mels);
    float* o = a.out + (size_t)b * a.T * a.n_mels;
    bf16_t* ob = a.out_bf16 ? a.out_bf16 + (size_t)b * a.T * a.n_mels : nullptr;
    if ((a.n_mels & 3) == 0) {               // 16-byte accesses: 4 consecutive mel bins of one frame
        for (int e = e0 + 4 * tid; e < e1; e += 1024) {
            const int t = e / a.n_mels;
            const int m = e - t * a.n_mels;
            float4 v = *reinterpret_cast<const float4*>(o + e);
            const bool keep = t < valid;
            v.x = keep ? v.x - s_mean[m] : 0.f; v.y = keep ? v.y - s_mean[m + 1] : 0.f;
            v.z = keep ? v.z - s_mean[m + 2] : 0.f; v.w = keep ? v.w - s_mean[m + 3] : 0.f;
            *reinterpret_cast<float4*>(o + e) = v;
            if (ob) {
                bf16_t q[4] = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
                *reinterpret_cast<uint2*>(ob + e) = *reinterpret_cast<const uint2*>(q);
            }
        }
        return;
    }
    for (int e = e0 + tid; e < e1; e += 256) {
        const int t = e / a.n_mels;
        const int m = e - t * a.n_mels;
        float v = (t < valid) ? (o[e] - s_mean[m]) : 0.f;
        o[e] = v;
        if (ob) ob[e] = (bf16_t)v;
    }
}

// Ragged batch: utterance b owns frames [0, n_frames[b]); its time mean is taken over those frames only and the rest of
// its rows are zero -- per-utterance featurisation followed by collate_fn's zero padding (reader.py:102-103 +
// collate_fn.py:5-23).  One workgroup per utterance: column means (each thread strides the rows of its mel bin group), then
// the subtraction.  Fixed-order sums.
struct CmnRaggedArgs { float* out; bf16_t* out_bf16; const int* n_frames; int T, n_mels; };

__global__ __launch_bounds__(256) void fbank_cmn_ragged_kernel(CmnRaggedArgs a) {
    __shared__ float s_part[4][FB_MAX_MEL];
    __shared__ float s_mean[FB_MAX_MEL];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int valid = min(max(a.n_frames[b], 0), a.T);
    float* o = a.out + (size_t)b * a.T * a.n_mels;
    bf16_t* ob = a.out_bf16 ? a.out_bf16 + (size_t)b * a.T * a.n_mels : nullptr;
    const int m = tid & 63, rg = tid >> 6;                         // 64 mel bins per pass x 4 row groups
    for (int m0 = 0; m0 < a.n_mels; m0 += 64) {
        float s = 0.f;
        if (m0 + m < a.n_mels)
            for (int t = rg; t < valid; t += 4) s += o[(size_t)t * a.n_mels + m0 + m];
        if (m0 + m < a.n_mels) s_part[rg][m0 + m] = s;
    }
    __syncthreads();
    if (tid < a.n_mels) s_mean[tid] = (s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid]) / (float)(valid > 0 ? valid : 1);
    __syncthreads();
    const int n = a.T * a.n_mels;
    for (int e = tid; e < n; e += 256) {
        const int t = e / a.n_mels;
        const float v = (t < valid) ? (o[e] - s_mean[e - t * a.n_mels]) : 0.f;
        o[e] = v;
        if (ob) ob[e] = (bf16_t)v;
    }
}

__global__ void frames_of_samples_kernel(const int* n_samples, int B, int win, int shift, int T, int* n_frames) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int n = n_samples[b];
    const int f = n < win ? 0 : 1 + (n - win) / shift;
    n_frames[b] = f < T ? f : T;
}

double mel_of(double f) { return 1127.0 * log(1.0 + f / 700.0); }

bool same_opts(const vp_fbank_opts& x, const vp_fbank_opts& y) { return memcmp(&x, &y, sizeof(x)) == 0; }

int build_tables(vp_ctx* ctx, const vp_fbank_opts* o) {
    if (ctx->fb_valid && same_opts(ctx->fb_opts, *o)) return VP_OK;
    vp_fbank_release_tables(ctx);
    const int win = (int)(o->sample_rate * o->frame_length_ms * 0.001f);
    const int shift = (int)(o->sample_rate * o->frame_shift_ms * 0.001f);
    int nfft = 1;
    while (nfft < win) nfft <<= 1;
    if (nfft != FB_NFFT || win > FB_MAX_WIN || win < 2)
        VP_FAIL(ctx, VP_EUNSUP, "fbank: only a 512-point FFT geometry is built (window %d -> nfft %d)", win, nfft);
    if (o->n_mels < 1 || o->n_mels > FB_MAX_MEL) VP_FAIL(ctx, VP_EUNSUP, "fbank: n_mels %d out of range", o->n_mels);
    std::vector<float> window(win);
    for (int i = 0; i < win; ++i)
        window[i] = (float)pow(0.5 - 0.5 * cos(2.0 * M_PI * i / (win - 1)), 0.85);   // povey
    std::vector<float2> tw(nfft);
    for (int k = 0; k < nfft; ++k) {
        double ang = -2.0 * M_PI * k / nfft;
        tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    // Kaldi mel banks over bins [0, nfft/2); Nyquist column is zero
    const int nbins = nfft / 2;
    const double nyq = 0.5 * o->sample_rate;
    double hi = o->high_freq;
    if (hi <= 0.0) hi += nyq;
    const double bin_w = (double)o->sample_rate / nfft;
    const double mlo = mel_of(o->low_freq), mhi = mel_of(hi);
    const double delta = (mhi - mlo) / (o->n_mels + 1);
    std::vector<int> start(o->n_mels + 1), bin0(o->n_mels);
    std::vector<float> wts;
    for (int m = 0; m < o->n_mels; ++m) {
        const double l = mlo + m * delta, c = l + delta, r = l + 2.0 * delta;
        start[m] = (int)wts.size();
        int first = -1, last = -2;
        std::vector<float> row(nbins, 0.f);
        for (int k = 0; k < nbins; ++k) {
            const double x = mel_of(bin_w * k);
            const double up = (x - l) / (c - l), down = (r - x) / (r - c);
            const double w = fmax(0.0, fmin(up, down));
            row[k] = (float)w;
            if (w > 0.0) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        bin0[m] = first;
        for (int k = first; k <= last; ++k) wts.push_back(row[k]);
    }
    start[o->n_mels] = (int)wts.size();
    if (wts.size() > FB_MAX_NNZ) VP_FAIL(ctx, VP_EUNSUP, "fbank: mel bank too dense (%zu taps)", wts.size());
    if (wts.empty()) wts.push_back(0.f);
    // padded round layout of the four-frames-per-wave kernel: round r = filters 16 r .. 16 r + 15, taps padded to the round's longest
    // filter (rounded up to 4) in [tap][16] order, scaled by 0.25 (its power spectrum is 4 |X|^2)
    {
        const int nr = (o->n_mels + 15) / 16;
        std::vector<float> wpad;
        bool ok = nr <= F16_MAX_ROUNDS;
        for (int r = 0; r < nr && ok; ++r) {
            int mx = 0;
            for (int m = 16 * r; m < std::min(16 * r + 16, o->n_mels); ++m) mx = std::max(mx, start[m + 1] - start[m]);
            mx = std::max(4, (mx + 3) / 4 * 4);
            if (mx > F16_MAX_TAPS) ok = false;
            ctx->fb_round_off[r] = (int)wpad.size();
            ctx->fb_round_max[r] = mx;
            for (int q = 0; q < mx; ++q)
                for (int jj = 0; jj < 16; ++jj) {
                    const int m = 16 * r + jj;
                    const bool in = m < o->n_mels && q < start[m + 1] - start[m];
                    wpad.push_back(in ? 0.25f * wts[start[m] + q] : 0.f);
                }
        }
        ctx->fb_f16 = ok && (int)wpad.size() <= F16_MAX_WPAD;
        ctx->fb_rounds = nr;
        ctx->fb_wpad_len = (int)wpad.size();
        if (wpad.empty()) wpad.push_back(0.f);
        VP_HIP(ctx, hipMalloc(&ctx->fb_wpad, wpad.size() * sizeof(float)));
        VP_HIP(ctx, hipMemcpy(ctx->fb_wpad, wpad.data(), wpad.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    VP_HIP(ctx, hipMalloc(&ctx->fb_window, win * sizeof(float)));
    VP_HIP(ctx, hipMalloc(&ctx->fb_twiddle, nfft * sizeof(float2)));
    VP_HIP(ctx, hipMalloc(&ctx->fb_mel_start, (o->n_mels + 1) * sizeof(int)));
    VP_HIP(ctx, hipMalloc(&ctx->fb_mel_bin0, o->n_mels * sizeof(int)));
    VP_HIP(ctx, hipMalloc(&ctx->fb_mel_w, wts.size() * sizeof(float)));
    VP_HIP(ctx, hipMemcpy(ctx->fb_window, window.data(), win * sizeof(float), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->fb_twiddle, tw.data(), nfft * sizeof(float2), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->fb_mel_start, start.data(), (o->n_mels + 1) * sizeof(int), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->fb_mel_bin0, bin0.data(), o->n_mels * sizeof(int), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->fb_mel_w, wts.data(), wts.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->fb_opts = *o;
    ctx->fb_win = win; ctx->fb_shift = shift; ctx->fb_nfft = nfft; ctx->fb_nmel = o->n_mels;
    ctx->fb_nnz = (int)wts.size();
    ctx->fb_valid = 1;
    return VP_OK;
}

}  // namespace

// CMN pass shared by the Fbank and MelSpectrogram paths: subtract the per-utterance time mean (from the
// per-tile column sums), apply the length mask, optionally emit the bf16 twin.
int vp_feat_cmn(vp_ctx* ctx, float* out, void* out_bf16, const float* psum, const float* lens_ratio, int B, int T,
                int tiles, int n_mels, hipStream_t st) {
    CmnArgs c;
    c.out = out; c.out_bf16 = (bf16_t*)out_bf16; c.psum = psum; c.lens_ratio = lens_ratio;
    c.B = B; c.T = T; c.tiles = tiles; c.n_mels = n_mels;
    int gx = (T * n_mels + 4095) / 4096;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(fbank_cmn_kernel, dim3(gx, B), dim3(256), 0, st, c);
    VP_LAUNCH_CHECK(ctx, "feat_cmn");
    return VP_OK;
}

int vp_fbank_release_tables(vp_ctx* ctx) {
    if (!ctx) return VP_OK;
    if (ctx->fb_window) (void)hipFree(ctx->fb_window);
    if (ctx->fb_twiddle) (void)hipFree(ctx->fb_twiddle);
    if (ctx->fb_mel_start) (void)hipFree(ctx->fb_mel_start);
    if (ctx->fb_mel_bin0) (void)hipFree(ctx->fb_mel_bin0);
    if (ctx->fb_mel_w) (void)hipFree(ctx->fb_mel_w);
    if (ctx->fb_wpad) (void)hipFree(ctx->fb_wpad);
    ctx->fb_wpad = nullptr;
    ctx->fb_window = nullptr; ctx->fb_twiddle = nullptr; ctx->fb_mel_start = nullptr;
    ctx->fb_mel_bin0 = nullptr; ctx->fb_mel_w = nullptr; ctx->fb_valid = 0;
    return VP_OK;
}

// Frame kernel launch shared by the rectangular and the ragged entry points.
static int launch_frames(vp_ctx* ctx, const float* wav, const short* wav16, float pcm_scale, float* out, float* psum, int B, int L, int T,
                         int tiles, const vp_fbank_opts* o, hipStream_t st) {
    static int force_old = -1;
    if (force_old < 0) { const char* e = getenv("VPMI_FBANK_OLD"); force_old = e && atoi(e) ? 1 : 0; }
    if (ctx->fb_f16 && (!force_old || wav16)) {
        Fbank16Args a;
        memset(&a, 0, sizeof(a));
        a.wav = wav; a.wav16 = wav16; a.pcm_scale = pcm_scale; a.out = out; a.psum = psum; a.window = ctx->fb_window; a.tw = ctx->fb_twiddle;
        a.wpad = ctx->fb_wpad; a.mel_bin0 = ctx->fb_mel_bin0;
        a.B = B; a.L = L; a.T = T; a.tiles = tiles; a.win = ctx->fb_win; a.shift = ctx->fb_shift;
        a.n_mels = o->n_mels; a.n_rounds = ctx->fb_rounds; a.wpad_len = ctx->fb_wpad_len;
        for (int r = 0; r < F16_MAX_ROUNDS; ++r) { a.round_off[r] = ctx->fb_round_off[r]; a.round_max[r] = ctx->fb_round_max[r]; }
        a.preemph = o->preemph; a.log_floor = o->log_floor; a.remove_dc = o->remove_dc;
        static int slots = 0;                      // three resident workgroups per CU (LDS 52 KB each)
        if (slots == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || v <= 0) v = 256;
            slots = 3 * v;
        }
        const long long total = (long long)tiles * B;
        const int grid = (int)(total < slots ? total : slots);
        const bool np13 = (ctx->fb_win + 31) / 32 == 13;
        if (wav16) {
            if (np13) hipLaunchKernelGGL((fbank_frames16_kernel<13, true>), dim3(grid), dim3(FB_WAVES * 64), 0, st, a);
            else hipLaunchKernelGGL((fbank_frames16_kernel<16, true>), dim3(grid), dim3(FB_WAVES * 64), 0, st, a);
        } else {
            if (np13) hipLaunchKernelGGL((fbank_frames16_kernel<13, false>), dim3(grid), dim3(FB_WAVES * 64), 0, st, a);
            else hipLaunchKernelGGL((fbank_frames16_kernel<16, false>), dim3(grid), dim3(FB_WAVES * 64), 0, st, a);
        }
        VP_LAUNCH_CHECK(ctx, "fbank_frames16");
        return VP_OK;
    }
    if (wav16) VP_FAIL(ctx, VP_EUNSUP, "fbank: 16-bit PCM input needs the four-frames-per-wave kernel (mel bank not covered)");
    FbankArgs a;
    a.wav = wav; a.out = out; a.psum = psum; a.window = ctx->fb_window; a.tw = ctx->fb_twiddle;
    a.mel_start = ctx->fb_mel_start; a.mel_bin0 = ctx->fb_mel_bin0; a.mel_w = ctx->fb_mel_w;
    a.B = B; a.L = L; a.T = T; a.tiles = tiles; a.win = ctx->fb_win; a.shift = ctx->fb_shift;
    a.n_mels = o->n_mels; a.nnz = ctx->fb_nnz; a.preemph = o->preemph; a.log_floor = o->log_floor;
    a.remove_dc = o->remove_dc;
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(tiles, B), dim3(FB_WAVES * 64), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "fbank_frames");
    return VP_OK;
}

extern "C" {

void vp_fbank_default_opts(vp_fbank_opts* o) {
    o->sample_rate = 16000; o->n_mels = 23; o->frame_length_ms = 25.f; o->frame_shift_ms = 10.f;
    o->preemph = 0.97f; o->remove_dc = 1; o->low_freq = 20.f; o->high_freq = 0.f; o->log_floor = 1e-7f;
}

int vp_fbank_num_frames(const vp_fbank_opts* o, int n_samples) {
    const int win = (int)(o->sample_rate * o->frame_length_ms * 0.001f);
    const int shift = (int)(o->sample_rate * o->frame_shift_ms * 0.001f);
    if (n_samples < win || shift <= 0) return 0;
    return 1 + (n_samples - win) / shift;
}

size_t vp_fbank_workspace_bytes(const vp_fbank_opts* o, int B, int L) {
    const int T = vp_fbank_num_frames(o, L);
    const int tiles = (T + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
    return vp_align_up((size_t)B * (tiles > 0 ? tiles : 1) * o->n_mels * sizeof(float), 256);
}

int vp_fbank_cmn_f32(vp_ctx* ctx, const float* wav, const float* lens_ratio, int B, int L,
                     const vp_fbank_opts* o, float* out, void* out_bf16, void* ws, size_t ws_bytes,
                     vp_stream stream) {
    if (!ctx || !wav || !o || !out || B <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank: bad arguments");
    const int T = vp_fbank_num_frames(o, L);
    if (T <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank: %d samples give no frame", L);
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "fbank: batch %d > 65535", B);
    int rc = build_tables(ctx, o);
    if (rc != VP_OK) return rc;
    if (!ws || ws_bytes < vp_fbank_workspace_bytes(o, B, L)) VP_FAIL(ctx, VP_EWORKSPACE, "fbank: workspace too small");
    const int tiles = (T + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = launch_frames(ctx, wav, nullptr, 0.f, out, (float*)ws, B, L, T, tiles, o, st))) return rc;
    return vp_feat_cmn(ctx, out, out_bf16, (const float*)ws, lens_ratio, B, T, tiles, o->n_mels, st);
}

int vp_fbank_cmn_pcm16(vp_ctx* ctx, const int16_t* wav, float pcm_scale, const float* lens_ratio, int B, int L,
                       const vp_fbank_opts* o, float* out, void* out_bf16, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !wav || !o || !out || B <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank: bad arguments");
    const int T = vp_fbank_num_frames(o, L);
    if (T <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank: %d samples give no frame", L);
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "fbank: batch %d > 65535", B);
    if (L & 1) VP_FAIL(ctx, VP_EUNSUP, "fbank: 16-bit PCM rows must hold an even number of samples (4-byte aligned pairs)");
    int rc = build_tables(ctx, o);
    if (rc != VP_OK) return rc;
    if (ctx->fb_shift & 1) VP_FAIL(ctx, VP_EUNSUP, "fbank: 16-bit PCM input needs an even frame shift (4-byte aligned sample pairs)");
    if (!ws || ws_bytes < vp_fbank_workspace_bytes(o, B, L)) VP_FAIL(ctx, VP_EWORKSPACE, "fbank: workspace too small");
    const int tiles = (T + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = launch_frames(ctx, nullptr, wav, pcm_scale, out, (float*)ws, B, L, T, tiles, o, st))) return rc;
    return vp_feat_cmn(ctx, out, out_bf16, (const float*)ws, lens_ratio, B, T, tiles, o->n_mels, st);
}

int vp_fbank_cmn_ragged_f32(vp_ctx* ctx, const float* wav, const int32_t* n_samples, int B, int L, const vp_fbank_opts* o, float* out,
                            void* out_bf16, int32_t* n_frames, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !wav || !n_samples || !n_frames || !o || !out || B <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank_ragged: bad arguments");
    const int T = vp_fbank_num_frames(o, L);
    if (T <= 0) VP_FAIL(ctx, VP_EINVAL, "fbank: %d samples give no frame", L);
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "fbank: batch %d > 65535", B);
    int rc = build_tables(ctx, o);
    if (rc != VP_OK) return rc;
    if (!ws || ws_bytes < vp_fbank_workspace_bytes(o, B, L)) VP_FAIL(ctx, VP_EWORKSPACE, "fbank: workspace too small");
    const int tiles = (T + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = launch_frames(ctx, wav, nullptr, 0.f, out, (float*)ws, B, L, T, tiles, o, st))) return rc;
    hipLaunchKernelGGL(frames_of_samples_kernel, dim3((B + 255) / 256), dim3(256), 0, st, n_samples, B, ctx->fb_win, ctx->fb_shift, T, n_frames);
    VP_LAUNCH_CHECK(ctx, "frames_of_samples");
    CmnRaggedArgs c{out, (bf16_t*)out_bf16, n_frames, T, o->n_mels};
    hipLaunchKernelGGL(fbank_cmn_ragged_kernel, dim3(B), dim3(256), 0, st, c);
    VP_LAUNCH_CHECK(ctx, "fbank_cmn_ragged");
    return VP_OK;
}

}  // extern "C"
