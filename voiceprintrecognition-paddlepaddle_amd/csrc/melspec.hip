// MelSpectrogram + CMN on gfx950.
//
// Replaces AudioFeaturizer.forward (ppvector/data_utils/featurizer.py:33-60) for feature_method
// 'MelSpectrogram' -> paddle.audio.features.MelSpectrogram(**method_args) (featurizer.py:22-23; librosa
// semantics): centred STFT with reflect padding, periodic Hann window (win_length <= n_fft, centred in
// the FFT frame), |X|^power, Slaney-scale / Slaney-normalised mel bank, LINEAR power (no log), then the
// featurizer's transpose + time-mean subtraction + length mask.
// Bound: HBM -- algorithmic bytes per utterance = 4*L + 4*T*n_mels.
// One wave transforms one frame: n_fft real points as an n_fft/2-point complex Stockham FFT in LDS
// (radix-4 stages, one radix-2 stage when n_fft/2 is not a power of 4), real-FFT unpack to the
// n_fft/2+1 power bins, sparse mel bank (CSR), per-tile column sums for the CMN pass (shared with
// the Fbank path).  The next frame's samples are fetched while the current one is transformed.
#include "common.h"

#include <math.h>
#include <vector>

int vp_feat_cmn(vp_ctx* ctx, float* out, void* out_bf16, const float* psum, const float* lens_ratio, int B, int T,
                int tiles, int n_mels, hipStream_t st);

namespace {

constexpr int MS_FRAMES_PER_WG = 16;
constexpr int MS_MAX_MEL = 128;
constexpr int MS_MAX_NNZ = 4096;

struct MelArgs {
    const float* wav; float* out; float* psum;
    const float* window;      // [n_fft], zero outside the centred win_length
    const float2* tw;         // [n_fft] e^{-2 pi i k / n_fft}
    const int* mel_start; const int* mel_bin0; const float* mel_w;
    int B, L, T, tiles, hop, n_mels, nnz;
    float power;
    int log_db; float amin, db_off;      // LogMelSpectrogram: 10 log10(max(e, amin)) - db_off
};

__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ int reflect_idx(int s, int L) { s = s < 0 ? -s : s; return s >= L ? 2 * (L - 1) - s : s; }

template <int NC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void melspec_frames_kernel(MelArgs a) {
    constexpr int NFFT = 2 * NC;
    constexpr int PPL = NC / 64;                 // complex points per lane
    constexpr int NB4 = NC / 256;                // radix-4 butterflies per lane per stage
    constexpr int LOG2 = NC == 256 ? 8 : (NC == 512 ? 9 : 10);
    constexpr int S4 = LOG2 / 2;                 // radix-4 stages
    constexpr bool R2 = (LOG2 & 1) != 0;         // one trailing radix-2 stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_tw = reinterpret_cast<float2*>(smem);                        // [NFFT]
    float* s_win = reinterpret_cast<float*>(s_tw + NFFT);                  // [NFFT]
    float* s_melw = s_win + NFFT;                                          // [MS_MAX_NNZ]
    int* s_mstart = reinterpret_cast<int*>(s_melw + MS_MAX_NNZ);           // [MS_MAX_MEL + 1]
    int* s_mbin0 = s_mstart + MS_MAX_MEL + 1;                              // [MS_MAX_MEL]
    float* s_red = reinterpret_cast<float*>(s_mbin0 + MS_MAX_MEL);         // [WAVES][MS_MAX_MEL]
    float2* s_buf = reinterpret_cast<float2*>(s_red + WAVES * MS_MAX_MEL + 1);   // [WAVES][2][NC + NC/16]
    float* s_pow = reinterpret_cast<float*>(s_buf + WAVES * 2 * (NC + NC / 16)); // [WAVES][NC + 1]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    for (int i = tid; i < NFFT; i += WAVES * 64) { s_tw[i] = a.tw[i]; s_win[i] = a.window[i]; }
    for (int i = tid; i < a.nnz; i += WAVES * 64) s_melw[i] = a.mel_w[i];
    for (int i = tid; i <= a.n_mels; i += WAVES * 64) s_mstart[i] = a.mel_start[i];
    for (int i = tid; i < a.n_mels; i += WAVES * 64) s_mbin0[i] = a.mel_bin0[i];
    __syncthreads();

    const float* wav = a.wav + (size_t)b * a.L;
    float2* d0 = s_buf + (size_t)wv * 2 * (NC + NC / 16);
    float2* d1 = d0 + (NC + NC / 16);
    float* pw = s_pow + (size_t)wv * (NC + 1);
    float acc0 = 0.f, acc1 = 0.f;

    float2 nxt[PPL];
    auto fetch = [&](int t) {                     // z[n] = x[2n] + i x[2n+1] of the centred, reflect-padded frame
        const int base = min(t, a.T - 1) * a.hop - NC;
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const int n = lane + 64 * u;
            nxt[u].x = wav[reflect_idx(base + 2 * n, a.L)];
            nxt[u].y = wav[reflect_idx(base + 2 * n + 1, a.L)];
        }
    };
    fetch(tile * MS_FRAMES_PER_WG + wv);

    for (int fi = wv; fi < MS_FRAMES_PER_WG; fi += WAVES) {
        const int t = tile * MS_FRAMES_PER_WG + fi;
        if (t >= a.T) break;                      // wave-uniform
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const int n = lane + 64 * u;
            d0[pidx(n)] = make_float2(nxt[u].x * s_win[2 * n], nxt[u].y * s_win[2 * n + 1]);
        }
        fetch(t + WAVES);
        __builtin_amdgcn_wave_barrier();
        float2* src = d0;
        float2* dst = d1;
#pragma unroll
        for (int s = 0; s < S4; ++s) {
            const int Ns = 1 << (2 * s);
            const int twstep = NFFT >> (2 * s + 2);                       // n_fft / (4 Ns)
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const int j = lane + 64 * u;
                const int jm = j & (Ns - 1);
                float2 v0 = src[pidx(j)];
                float2 v1 = cmul(src[pidx(j + NC / 4)], s_tw[(jm * twstep) & (NFFT - 1)]);
                float2 v2 = cmul(src[pidx(j + NC / 2)], s_tw[(2 * jm * twstep) & (NFFT - 1)]);
                float2 v3 = cmul(src[pidx(j + 3 * NC / 4)], s_tw[(3 * jm * twstep) & (NFFT - 1)]);
                const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
                const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
                const int idx = ((j >> (2 * s)) << (2 * s + 2)) + jm;
                dst[pidx(idx)] = make_float2(s02.x + s13.x, s02.y + s13.y);
                dst[pidx(idx + Ns)] = make_float2(d02.x + d13.y, d02.y - d13.x);
                dst[pidx(idx + 2 * Ns)] = make_float2(s02.x - s13.x, s02.y - s13.y);
                dst[pidx(idx + 3 * Ns)] = make_float2(d02.x - d13.y, d02.y + d13.x);
            }
            __builtin_amdgcn_wave_barrier();
            float2* tmp = src; src = dst; dst = tmp;
        }
        if constexpr (R2) {                       // Ns = NC/2: one radix-2 stage finishes the transform
#pragma unroll
            for (int u = 0; u < NC / 128; ++u) {
                const int j = lane + 64 * u;
                const float2 v0 = src[pidx(j)];
                const float2 v1 = cmul(src[pidx(j + NC / 2)], s_tw[(j * 2) & (NFFT - 1)]);
                dst[pidx(j)] = make_float2(v0.x + v1.x, v0.y + v1.y);
                dst[pidx(j + NC / 2)] = make_float2(v0.x - v1.x, v0.y - v1.y);
            }
            __builtin_amdgcn_wave_barrier();
            float2* tmp = src; src = dst; dst = tmp;
        }
        // real-FFT unpack -> |X|^power for bins 0 .. NC (Nyquist included)
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const int k = lane + 64 * u;
            const float2 zk = src[pidx(k)];
            const float2 zn = src[pidx((NC - k) & (NC - 1))];
            const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
            const float2 wo = cmul(o, s_tw[k]);
            const float xr = e.x + wo.x, xi = e.y + wo.y;
            float p = xr * xr + xi * xi;
            if (a.power != 2.f) p = a.power == 1.f ? sqrtf(p) : powf(p, 0.5f * a.power);
            pw[k] = p;
        }
        if (lane == 0) {
            const float2 z0 = src[pidx(0)];
            float p = (z0.x - z0.y) * (z0.x - z0.y);
            if (a.power != 2.f) p = a.power == 1.f ? sqrtf(p) : powf(p, 0.5f * a.power);
            pw[NC] = p;
        }
        __builtin_amdgcn_wave_barrier();
        float* orow = a.out + ((size_t)b * a.T + t) * a.n_mels;
        for (int m = lane, it = 0; m < a.n_mels; m += 64, ++it) {
            const int s0 = s_mstart[m], n = s_mstart[m + 1] - s0, k0 = s_mbin0[m];
            float e = 0.f;
            for (int q = 0; q < n; q += 4) {
                float wq[4], pq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = q + u < n;
                    const float w_ = s_melw[min(s0 + q + u, a.nnz - 1)];
                    wq[u] = in ? w_ : 0.f;
                    pq[u] = pw[min(k0 + q + u, NC)];
                }
                e += wq[0] * pq[0]; e += wq[1] * pq[1]; e += wq[2] * pq[2]; e += wq[3] * pq[3];
            }
            if (a.log_db) e = 10.f * log10f(fmaxf(e, a.amin)) - a.db_off;
            orow[m] = e;
            if (it == 0) acc0 += e; else acc1 += e;
        }
        __builtin_amdgcn_wave_barrier();
    }
    s_red[wv * MS_MAX_MEL + lane] = acc0;
    s_red[wv * MS_MAX_MEL + lane + 64] = acc1;
    __syncthreads();
    if (tid < a.n_mels) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += s_red[w * MS_MAX_MEL + tid];
        a.psum[((size_t)b * a.tiles + tile) * a.n_mels + tid] = s;
    }
}

template <int NC, int WAVES>
size_t mel_smem() {
    return (size_t)2 * NC * 8 + (size_t)2 * NC * 4 + MS_MAX_NNZ * 4 + (2 * MS_MAX_MEL + 1) * 4 + (WAVES * MS_MAX_MEL + 1) * 4 +
           (size_t)WAVES * 2 * (NC + NC / 16) * 8 + (size_t)WAVES * (NC + 1) * 4 + 64;
}

template <int NC, int WAVES>
int launch_mel(vp_ctx* ctx, const MelArgs& a, hipStream_t st) {
    const size_t smem = mel_smem<NC, WAVES>();
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(melspec_frames_kernel<NC, WAVES>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((melspec_frames_kernel<NC, WAVES>), dim3(a.tiles, a.B), dim3(WAVES * 64), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "melspec_frames");
    return VP_OK;
}

double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int build_mel_tables(vp_ctx* ctx, const vp_mel_opts* o) {
    if (ctx->ms_valid && memcmp(&ctx->ms_opts, o, sizeof(*o)) == 0) return VP_OK;
    vp_mel_release_tables(ctx);
    const int n_fft = o->n_fft, win_length = o->win_length > 0 ? o->win_length : n_fft;
    if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) VP_FAIL(ctx, VP_EUNSUP, "melspec: n_fft %d not built (512, 1024, 2048)", n_fft);
    if (win_length > n_fft || win_length < 2 || o->hop_length < 1) VP_FAIL(ctx, VP_EINVAL, "melspec: bad window / hop");
    if (o->n_mels < 1 || o->n_mels > MS_MAX_MEL) VP_FAIL(ctx, VP_EUNSUP, "melspec: n_mels %d out of range", o->n_mels);
    std::vector<float> window(n_fft, 0.f);
    const int lpad = (n_fft - win_length) / 2;
    for (int i = 0; i < win_length; ++i) window[lpad + i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / win_length));
    std::vector<float2> tw(n_fft);
    for (int k = 0; k < n_fft; ++k) tw[k] = make_float2((float)cos(-2.0 * M_PI * k / n_fft), (float)sin(-2.0 * M_PI * k / n_fft));
    const int nbins = n_fft / 2 + 1;
    const double f_hi = o->f_max > 0.f ? o->f_max : 0.5 * o->sample_rate;
    std::vector<double> mel_f(o->n_mels + 2);
    const double mlo = hz_to_mel(o->f_min), mhi = hz_to_mel(f_hi);
    for (int i = 0; i < o->n_mels + 2; ++i) mel_f[i] = mel_to_hz(mlo + (mhi - mlo) * i / (o->n_mels + 1));
    std::vector<int> start(o->n_mels + 1), bin0(o->n_mels);
    std::vector<float> wts;
    for (int m = 0; m < o->n_mels; ++m) {
        start[m] = (int)wts.size();
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        int first = -1, last = -2;
        std::vector<float> row(nbins, 0.f);
        for (int k = 0; k < nbins; ++k) {
            const double f = 0.5 * o->sample_rate * k / (nbins - 1);
            const double lower = (f - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
            const double upper = (mel_f[m + 2] - f) / (mel_f[m + 2] - mel_f[m + 1]);
            double w = lower < upper ? lower : upper;
            if (w < 0.0) w = 0.0;
            row[k] = (float)(w * enorm);
            if (w > 0.0) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        bin0[m] = first;
        for (int k = first; k <= last; ++k) wts.push_back(row[k]);
    }
    start[o->n_mels] = (int)wts.size();
    if (wts.size() > MS_MAX_NNZ) VP_FAIL(ctx, VP_EUNSUP, "melspec: mel bank too dense (%zu taps)", wts.size());
    if (wts.empty()) wts.push_back(0.f);
    VP_HIP(ctx, hipMalloc(&ctx->ms_window, n_fft * sizeof(float)));
    VP_HIP(ctx, hipMalloc(&ctx->ms_twiddle, n_fft * sizeof(float2)));
    VP_HIP(ctx, hipMalloc(&ctx->ms_mel_start, (o->n_mels + 1) * sizeof(int)));
    VP_HIP(ctx, hipMalloc(&ctx->ms_mel_bin0, o->n_mels * sizeof(int)));
    VP_HIP(ctx, hipMalloc(&ctx->ms_mel_w, wts.size() * sizeof(float)));
    VP_HIP(ctx, hipMemcpy(ctx->ms_window, window.data(), n_fft * sizeof(float), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->ms_twiddle, tw.data(), n_fft * sizeof(float2), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->ms_mel_start, start.data(), (o->n_mels + 1) * sizeof(int), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->ms_mel_bin0, bin0.data(), o->n_mels * sizeof(int), hipMemcpyHostToDevice));
    VP_HIP(ctx, hipMemcpy(ctx->ms_mel_w, wts.data(), wts.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->ms_opts = *o;
    ctx->ms_nnz = (int)wts.size();
    ctx->ms_valid = 1;
    return VP_OK;
}

}  // namespace

int vp_mel_release_tables(vp_ctx* ctx) {
    if (!ctx) return VP_OK;
    if (ctx->ms_window) (void)hipFree(ctx->ms_window);
    if (ctx->ms_twiddle) (void)hipFree(ctx->ms_twiddle);
    if (ctx->ms_mel_start) (void)hipFree(ctx->ms_mel_start);
    if (ctx->ms_mel_bin0) (void)hipFree(ctx->ms_mel_bin0);
    if (ctx->ms_mel_w) (void)hipFree(ctx->ms_mel_w);
    ctx->ms_window = nullptr; ctx->ms_twiddle = nullptr; ctx->ms_mel_start = nullptr; ctx->ms_mel_bin0 = nullptr;
    ctx->ms_mel_w = nullptr; ctx->ms_valid = 0;
    return VP_OK;
}

extern "C" {

void vp_mel_default_opts(vp_mel_opts* o) {
    o->sample_rate = 22050; o->n_fft = 2048; o->hop_length = 512; o->win_length = 0; o->n_mels = 64;
    o->f_min = 50.f; o->f_max = 0.f; o->power = 2.f; o->log_db = 0; o->amin = 1e-10f; o->ref_value = 1.f;
}

int vp_mel_num_frames(const vp_mel_opts* o, int n_samples) { return o->hop_length > 0 ? 1 + n_samples / o->hop_length : 0; }

size_t vp_mel_workspace_bytes(const vp_mel_opts* o, int B, int L) {
    const int T = vp_mel_num_frames(o, L);
    const int tiles = (T + MS_FRAMES_PER_WG - 1) / MS_FRAMES_PER_WG;
    return vp_align_up((size_t)B * (tiles > 0 ? tiles : 1) * o->n_mels * sizeof(float), 256);
}

int vp_melspec_cmn_f32(vp_ctx* ctx, const float* wav, const float* lens_ratio, int B, int L, const vp_mel_opts* o,
                       float* out, void* out_bf16, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !wav || !o || !out || B <= 0) VP_FAIL(ctx, VP_EINVAL, "melspec: bad arguments");
    if (L <= o->n_fft / 2) VP_FAIL(ctx, VP_EINVAL, "melspec: %d samples are too few for reflect padding of %d", L, o->n_fft / 2);
    if (B > 65535) VP_FAIL(ctx, VP_EINVAL, "melspec: batch %d > 65535", B);
    int rc = build_mel_tables(ctx, o);
    if (rc) return rc;
    if (!ws || ws_bytes < vp_mel_workspace_bytes(o, B, L)) VP_FAIL(ctx, VP_EWORKSPACE, "melspec: workspace too small");
    const int T = vp_mel_num_frames(o, L);
    const int tiles = (T + MS_FRAMES_PER_WG - 1) / MS_FRAMES_PER_WG;
    hipStream_t st = (hipStream_t)stream;
    MelArgs a;
    a.wav = wav; a.out = out; a.psum = (float*)ws; a.window = ctx->ms_window; a.tw = ctx->ms_twiddle;
    a.mel_start = ctx->ms_mel_start; a.mel_bin0 = ctx->ms_mel_bin0; a.mel_w = ctx->ms_mel_w;
    a.B = B; a.L = L; a.T = T; a.tiles = tiles; a.hop = o->hop_length; a.n_mels = o->n_mels; a.nnz = ctx->ms_nnz;
    a.power = o->power;
    a.log_db = o->log_db; a.amin = o->amin; a.db_off = 10.f * log10f(fmaxf(o->ref_value, o->amin));
    if (o->n_fft == 512) rc = launch_mel<256, 4>(ctx, a, st);
    else if (o->n_fft == 1024) rc = launch_mel<512, 4>(ctx, a, st);
    else rc = launch_mel<1024, 2>(ctx, a, st);
    if (rc) return rc;
    return vp_feat_cmn(ctx, out, out_bf16, (const float*)ws, lens_ratio, B, T, tiles, o->n_mels, st);
}

}  // extern "C"
