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
    ctx->fb_window = nullptr; ctx->fb_twiddle = nullptr; ctx->fb_mel_start = nullptr;
    ctx->fb_mel_bin0 = nullptr; ctx->fb_mel_w = nullptr; ctx->fb_valid = 0;
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
    FbankArgs a;
    a.wav = wav; a.out = out; a.psum = (float*)ws; a.window = ctx->fb_window; a.tw = ctx->fb_twiddle;
    a.mel_start = ctx->fb_mel_start; a.mel_bin0 = ctx->fb_mel_bin0; a.mel_w = ctx->fb_mel_w;
    a.B = B; a.L = L; a.T = T; a.tiles = tiles; a.win = ctx->fb_win; a.shift = ctx->fb_shift;
    a.n_mels = o->n_mels; a.nnz = ctx->fb_nnz; a.preemph = o->preemph; a.log_floor = o->log_floor;
    a.remove_dc = o->remove_dc;
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(tiles, B), dim3(FB_WAVES * 64), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "fbank_frames");
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
    FbankArgs a;
    a.wav = wav; a.out = out; a.psum = (float*)ws; a.window = ctx->fb_window; a.tw = ctx->fb_twiddle;
    a.mel_start = ctx->fb_mel_start; a.mel_bin0 = ctx->fb_mel_bin0; a.mel_w = ctx->fb_mel_w;
    a.B = B; a.L = L; a.T = T; a.tiles = tiles; a.win = ctx->fb_win; a.shift = ctx->fb_shift;
    a.n_mels = o->n_mels; a.nnz = ctx->fb_nnz; a.preemph = o->preemph; a.log_floor = o->log_floor;
    a.remove_dc = o->remove_dc;
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(tiles, B), dim3(FB_WAVES * 64), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "fbank_frames");
    hipLaunchKernelGGL(frames_of_samples_kernel, dim3((B + 255) / 256), dim3(256), 0, st, n_samples, B, ctx->fb_win, ctx->fb_shift, T, n_frames);
    VP_LAUNCH_CHECK(ctx, "frames_of_samples");
    CmnRaggedArgs c{out, (bf16_t*)out_bf16, n_frames, T, o->n_mels};
    hipLaunchKernelGGL(fbank_cmn_ragged_kernel, dim3(B), dim3(256), 0, st, c);
    VP_LAUNCH_CHECK(ctx, "fbank_cmn_ragged");
    return VP_OK;
}

}  // extern "C"
