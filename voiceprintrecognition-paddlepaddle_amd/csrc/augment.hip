// Batch assembly either side of the featurizer: SpecAugment masks and zero-padded collation, on the GPU.
//
// SpecAugment: the reference calls yeaudio.SpecAugmentor(**aug_conf.spec_aug) on each (T, F) feature
// (ppvector/data_utils/reader.py:105-107,150-151; parameters configs/augmentation.yml:36-48).  yeaudio is not
// vendored; its published algorithm (frequency masks, then time masks, each filled with the CURRENT mean of the
// feature -- or zero) is restated in oracle/augment.py.  The random draws stay on the host (same Python `random`
// call sequence as the restatement); this kernel applies the drawn masks to the whole batch in place.
// Collation: collate_fn (ppvector/data_utils/collate_fn.py:5-23) = zero-pad every (T_i, F) feature to T_max.
#include "common.h"

namespace {

template <typename T>
struct SaArgs { T* x; const int* fmask; const int* tmask; int Tn, F, nf, nt, zero; };

// one workgroup per utterance; masks are applied in order, each with its own mean of the current tensor
template <typename T>
__global__ __launch_bounds__(256) void spec_augment_kernel(SaArgs<T> a) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    T* x = a.x + (size_t)b * a.Tn * a.F;
    const int n = a.Tn * a.F;
    for (int k = 0; k < a.nf + a.nt; ++k) {
        const bool is_f = k < a.nf;
        const int* mk = is_f ? a.fmask + ((size_t)b * a.nf + k) * 2 : a.tmask + ((size_t)b * a.nt + (k - a.nf)) * 2;
        const int s0 = mk[0], wd = mk[1];
        if (wd <= 0) continue;                                   // uniform per workgroup
        float fill = 0.f;
        if (!a.zero) {
            float s = 0.f;
            for (int i = tid; i < n; i += 256) s += vp_to_f32(x[i]);
            s = vp_wave_sum(s);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            fill = (red[0] + red[1] + red[2] + red[3]) / (float)n;
            __syncthreads();
        }
        const T fv = vp_from_f32<T>(fill);
        if (is_f) {
            for (int i = tid; i < a.Tn * wd; i += 256) x[(size_t)(i / wd) * a.F + s0 + i % wd] = fv;
        } else {
            for (int i = tid; i < wd * a.F; i += 256) x[(size_t)s0 * a.F + i] = fv;
        }
        __threadfence_block();
        __syncthreads();
    }
}

template <typename T>
struct PadArgs { const T* const* src; const int* lens; T* out; int Tmax, F; };

template <typename T>
__global__ __launch_bounds__(256) void pad_batch_kernel(PadArgs<T> a) {
    const int b = blockIdx.y;
    const T* s = a.src[b];
    const long long nv = (long long)a.lens[b] * a.F, n = (long long)a.Tmax * a.F;
    T* o = a.out + (size_t)b * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        o[i] = i < nv ? s[i] : vp_from_f32<T>(0.f);
}

// One workgroup per utterance: mean square over the WHOLE source (the reference normalises before it crops), then the
// gained crop, zero-padded to L.  Fixed-order reduction (256 strided partials, wave shuffles, 4 waves).
struct WaveArgs {
    const float* const* src; const int* lens; const int* starts; const float* gain_db; float* out; int* n_valid;
    int L, normalize; float target_db;
};

__global__ __launch_bounds__(256) void wave_batch_kernel(WaveArgs a) {
    __shared__ float sm[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = a.src[b];
    const int n = a.lens[b];
    float gain = 1.f;
    if (a.normalize) {
        float ss = 0.f;
        for (int i = tid; i < n; i += 256) { const float v = s[i]; ss += v * v; }
        ss = vp_wave_sum(ss);
        if ((tid & 63) == 0) sm[tid >> 6] = ss;
        __syncthreads();
        const float ms = (sm[0] + sm[1] + sm[2] + sm[3]) / (float)(n > 0 ? n : 1);
        if (ms > 0.f) gain = exp10f((a.target_db - 10.f * log10f(ms)) * 0.05f);
    } else if (a.gain_db) {
        gain = exp10f(a.gain_db[b] * 0.05f);
    }
    int st = a.starts ? a.starts[b] : 0;
    st = st < 0 ? 0 : (st > n ? n : st);
    const int nv = n - st < a.L ? n - st : a.L;
    float* o = a.out + (size_t)b * a.L;
    for (int i = tid; i < a.L; i += 256) o[i] = i < nv ? s[st + i] * gain : 0.f;
    if (tid == 0 && a.n_valid) a.n_valid[b] = nv;
}

// Speed perturbation = resampling by linear interpolation onto new_len points spread over [0, len] (yeaudio
// AudioSegment.change_speed: np.interp(np.linspace(0, len, new_len), arange(len), samples)); positions in f64 as numpy has them,
// the last points clamp to the final sample.
struct SpeedArgs { const float* const* src; const int* lens; const int* new_lens; float* const* dst; };

__global__ __launch_bounds__(256) void speed_perturb_kernel(SpeedArgs a) {
    const int b = blockIdx.y;
    const int n = a.lens[b], m = a.new_lens[b];
    const float* s = a.src[b];
    float* o = a.dst[b];
    const double step = m > 1 ? (double)n / (double)(m - 1) : 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const double p = (double)i * step;
        int j = (int)p;
        float v;
        if (j >= n - 1) {
            v = s[n - 1];
        } else {
            const double y0 = (double)s[j], y1 = (double)s[j + 1];
            v = (float)((y1 - y0) * (p - (double)j) + y0);
        }
        o[i] = v;
    }
}

}  // namespace

extern "C" {

int vp_spec_augment(vp_ctx* ctx, int dtype, void* feats, int B, int T, int F, const int32_t* fmask, int n_freq_masks,
                    const int32_t* tmask, int n_time_masks, int replace_with_zero, vp_stream stream) {
    if (!ctx || !feats || B <= 0 || T <= 0 || F <= 0 || n_freq_masks < 0 || n_time_masks < 0 ||
        (n_freq_masks && !fmask) || (n_time_masks && !tmask))
        VP_FAIL(ctx, VP_EINVAL, "spec_augment: bad arguments");
    if (n_freq_masks + n_time_masks == 0) return VP_OK;
    if (dtype == VP_BF16) {
        SaArgs<bf16_t> a{(bf16_t*)feats, fmask, tmask, T, F, n_freq_masks, n_time_masks, replace_with_zero};
        hipLaunchKernelGGL(spec_augment_kernel<bf16_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    } else if (dtype == VP_F32) {
        SaArgs<float> a{(float*)feats, fmask, tmask, T, F, n_freq_masks, n_time_masks, replace_with_zero};
        hipLaunchKernelGGL(spec_augment_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        VP_FAIL(ctx, VP_EINVAL, "spec_augment: bad dtype");
    }
    VP_LAUNCH_CHECK(ctx, "spec_augment");
    return VP_OK;
}

int vp_pad_batch(vp_ctx* ctx, int dtype, const void* const* srcs, const int32_t* lens, int B, int Tmax, int F, void* out,
                 vp_stream stream) {
    if (!ctx || !srcs || !lens || !out || B <= 0 || Tmax <= 0 || F <= 0 || B > 65535) VP_FAIL(ctx, VP_EINVAL, "pad_batch: bad arguments");
    const long long n = (long long)Tmax * F;
    unsigned gx = (unsigned)((n + 255) / 256);
    if (gx > 1024) gx = 1024;
    if (dtype == VP_BF16) {
        PadArgs<bf16_t> a{(const bf16_t* const*)srcs, lens, (bf16_t*)out, Tmax, F};
        hipLaunchKernelGGL(pad_batch_kernel<bf16_t>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, a);
    } else if (dtype == VP_F32) {
        PadArgs<float> a{(const float* const*)srcs, lens, (float*)out, Tmax, F};
        hipLaunchKernelGGL(pad_batch_kernel<float>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        VP_FAIL(ctx, VP_EINVAL, "pad_batch: bad dtype");
    }
    VP_LAUNCH_CHECK(ctx, "pad_batch");
    return VP_OK;
}

int vp_wave_batch_f32(vp_ctx* ctx, const float* const* srcs, const int32_t* lens, const int32_t* starts, int B, int L, int normalize,
                      float target_db, const float* gain_db, float* out, int32_t* n_valid, vp_stream stream) {
    if (!ctx || !srcs || !lens || !out || B <= 0 || L <= 0) VP_FAIL(ctx, VP_EINVAL, "wave_batch: bad arguments");
    WaveArgs a{srcs, lens, starts, gain_db, out, n_valid, L, normalize, target_db};
    hipLaunchKernelGGL(wave_batch_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "wave_batch");
    return VP_OK;
}

int vp_speed_perturb_f32(vp_ctx* ctx, const float* const* srcs, const int32_t* lens, const int32_t* new_lens, float* const* dsts,
                         int B, int max_new_len, vp_stream stream) {
    if (!ctx || !srcs || !lens || !new_lens || !dsts || B <= 0 || B > 65535 || max_new_len <= 0)
        VP_FAIL(ctx, VP_EINVAL, "speed_perturb: bad arguments");
    unsigned gx = (unsigned)((max_new_len + 1023) / 1024);
    if (gx > 256) gx = 256;
    SpeedArgs a{srcs, lens, new_lens, dsts};
    hipLaunchKernelGGL(speed_perturb_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "speed_perturb");
    return VP_OK;
}

}  // extern "C"
