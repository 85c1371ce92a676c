// Shared host/device helpers for libvpmi (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vpmi.h"

struct vp_ctx {
    int device;
    char err[512];
    // Fbank tables (built lazily per option set; tiny, read-only)
    vp_fbank_opts fb_opts;
    int fb_valid;
    int fb_win, fb_shift, fb_nfft, fb_nmel, fb_nnz;
    float* fb_window;     // [win]
    float2* fb_twiddle;   // [nfft/2] e^{-2 pi i k / nfft}
    int* fb_mel_start;    // [nmel + 1] CSR offsets into fb_mel_w
    int* fb_mel_bin0;     // [nmel] first FFT bin of each filter
    float* fb_mel_w;      // [nnz]
    float* fb_wpad;       // padded [round][tap][16] mel weights of the four-frames-per-wave kernel (x 0.25)
    int fb_f16, fb_rounds, fb_wpad_len, fb_round_off[8], fb_round_max[8];
    // MelSpectrogram tables
    vp_mel_opts ms_opts;
    int ms_valid, ms_nnz;
    float* ms_window;     // [n_fft]
    float2* ms_twiddle;   // [n_fft]
    int* ms_mel_start;
    int* ms_mel_bin0;
    float* ms_mel_w;
    // device margin table of the margin-softmax losses (vp_set_margin_table): [margin, cos m, sin m, cos(pi - m), 1 + cos(pi - m)]
    const float* margin_table;
    // counters of the in-kernel grid barrier (res2_train.hip): eight arrival counters 32 words apart, [256] departures, [257] bail-out
    // flag (VP_FAULT_WORD); zero between launches
    unsigned* grid_bar;
    unsigned* grid_bar_own;   // the context's own allocation of them (vp_set_grid_barrier_words may point grid_bar at caller-owned words)
    // CUs the fused grid-barrier kernels leave to kernels of other queues (a collective running beside the step): a launch of more than
    // (#CUs - reserve) workgroups takes the per-chunk path instead (vp_set_grid_reserve_cus)
    int grid_reserve_cus;
};
constexpr int VP_FAULT_WORD = 8 * 32 + 1;
// the bail-out word the optimiser kernels test before they update anything (nullptr: no barrier words on this context)
static inline const unsigned* vp_fault_word(const vp_ctx* ctx) { return ctx && ctx->grid_bar ? ctx->grid_bar + VP_FAULT_WORD : nullptr; }

#define VP_FAIL(ctx, code, ...)                                      \
    do {                                                             \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                               \
    } while (0)

#define VP_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) VP_FAIL(ctx, VP_EHIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define VP_LAUNCH_CHECK(ctx, name)                                                            \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) VP_FAIL(ctx, VP_EHIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline size_t vp_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename T> struct vp_elem;
template <> struct vp_elem<float> { static constexpr int id = VP_F32; };
template <> struct vp_elem<bf16_t> { static constexpr int id = VP_BF16; };

__device__ __forceinline__ float vp_to_f32(float v) { return v; }
__device__ __forceinline__ float vp_to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T vp_from_f32(float v);
template <> __device__ __forceinline__ float vp_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t vp_from_f32<bf16_t>(float v) { return (bf16_t)v; }
// 4 consecutive elements (16 B f32 / 8 B bf16; the pointer must be aligned accordingly)
__device__ __forceinline__ void vp_load4(const float* p, float v[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void vp_load4(const bf16_t* p, float v[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void vp_store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void vp_store4(bf16_t* p, const float v[4]) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

__device__ __forceinline__ float vp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float vp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

static inline size_t vp_dtype_size(int dt) { return dt == VP_BF16 ? 2 : 4; }
// a backbone's `dtype` -> the element type of its tensors (VP_F32X3 = f32 tensors, split-precision contractions)
static inline int vp_storage_dtype(int dt) { return dt == VP_F32X3 ? VP_F32 : dt; }
static inline bool vp_backbone_dtype_ok(int dt) { return dt == VP_F32 || dt == VP_BF16 || dt == VP_F32X3; }
// dtype fields of a conv descriptor from a backbone's `dtype`
static inline void vp_desc_dtype(vp_conv1d_desc& d, int dt) {
    d.dtype_in = d.dtype_out = vp_storage_dtype(dt);
    d.mfma_bf16 = dt == VP_F32X3 ? 2 : 0;
}
// weights of a layer for a conv descriptor whose dtype fields are set: a split-precision backbone (VP_F32X3) hands over the layer's
// pre-split weights when it has them (only the activations are then split while staging)
static inline void vp_desc_weights(vp_conv1d_desc& d, const vp_tdnn_layer& L) {
    d.w = L.w;
    if (d.mfma_bf16 == 2 && d.dtype_in == VP_F32 && d.dtype_out == VP_F32 && L.w_hl) { d.w = L.w_hl; d.mfma_bf16 = 3; }
}

// M-tile of the conv GEMM: the partial time-sum arrays are indexed by it
#define VP_CONV_BM 128

// internal launchers shared between translation units
int vp_dense_f32_ex(vp_ctx* ctx, const float* a, int lda, const float* w, int w_is_kn, const float* bias,
                    const float* rowscale, const float* colscale, int M, int N, int K, int act, float* out,
                    int ldo, hipStream_t st);
int vp_asp_softmax_stats_ex(vp_ctx* ctx, int dtype, const float* logits, const void* x, int ldx, int xoff,
                            const float* center, int ldc, int B, int T, int C, float eps, float* pooled, hipStream_t st);
struct VpAspBufs { void* h; float* e; float* psum; float* psumsq; float* stats; float* rowbias; float* pooled; };
int vp_run_asp(vp_ctx* ctx, const vp_asp_weights& A, int dtype, const void* x, int ldx, const float* shift,
               int B, int T, const VpAspBufs& w, hipStream_t st);
int vp_asp_utt_bf16(vp_ctx* ctx, const void* x, int ldx, const vp_tdnn_layer* tdnn, const float* rowbias, const void* conv_w,
                    const float* conv_b, int B, int T, int C, int att, float eps, float* pooled, hipStream_t st);
int vp_wgrad_tr256_splits(long long M, int N, int K);
int vp_wgrad_tr256_bf16(vp_ctx* ctx, const void* x, int ldx, int xoff, const void* dz, int lddz, long long M, int N, int K, float* part,
                        int* splits_out, hipStream_t st);
int vp_conv3x3_c1(vp_ctx* ctx, int dtype, const void* feats, void* out, const float* w, const float* bias,
                  const float* scale, const float* shift, int B, int T, int F, int C, hipStream_t st);
int vp_se_scale_residual_ex(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, const float* s,
                            const void* res, int ldr, int roff, void* out, int ldo, int ooff, int B, int T, int C,
                            int relu, hipStream_t st);
int vp_im2col_hl32(vp_ctx* ctx, const float* x, int B, int T, int F, int KW, int dil, int pad, void* out, int Kp, hipStream_t st);
int vp_time_moments(vp_ctx* ctx, int dtype, const void* x, int ldx, int B, int T, int C, float eps, int unbiased,
                    float* stats, hipStream_t st);
int vp_se_gate(vp_ctx* ctx, const float* psum, const float* shift, int B, int T, int C, int H, const float* w1, const float* b1,
               const float* w2, const float* b2, float* out, hipStream_t st);
int vp_copy_cols(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, void* y, int ldy, int yoff, long long rows, int C,
                 hipStream_t st);
int vp_aff_combine(vp_ctx* ctx, int dtype, const void* t, int ldt, const void* x, int ldx, int xoff, const void* y, int ldy,
                   int yoff, void* o, int ldo, int ooff, long long rows, int C, hipStream_t st);
int vp_res2_chain_bf16(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T,
                       int C, int width, hipStream_t st);
int vp_res2_chain_x3(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C, int width,
                     hipStream_t st);
bool vp_res2_chain_x3_ok(const vp_tdnn_layer* layers, int nconv, int B, int T, int C, int width);
int vp_asp_fused_x3(vp_ctx* ctx, const void* h, const float* w, const float* bias, const void* x, int ldx, const float* center, int ldc,
                    int B, int T, int C, int att, float eps, float* pooled, hipStream_t st);
int vp_asp_fused_bf16(vp_ctx* ctx, const void* h, const void* w, const float* bias, const void* x, int ldx,
                      const float* center, int ldc, int B, int T, int C, int att, float eps, float* pooled,
                      hipStream_t st);
int vp_row_inv_norm(vp_ctx* ctx, const float* x, int rows, int D, int ld, float eps, float* inv, hipStream_t st);
int vp_conv3x3_c32_bf16(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv, const void* res, int relu,
                        const vp_tdnn_layer* shortcut, void* y2, int B, int T, int F_in, int stride_f, const void* c1_feats,
                        const float* c1_w, const float* c1_b, const float* c1_scale, const float* c1_shift, hipStream_t st);
int vp_resblock_c32_bf16(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv1, const vp_tdnn_layer* conv2, int B, int T, int F,
                         hipStream_t st);
int vp_pointwise_bf16(vp_ctx* ctx, const vp_conv1d_desc* d, int only_where_it_wins, hipStream_t st);
int vp_cam_block_bf16(vp_ctx* ctx, const vp_cam_layer* layers, int n_layers, void* cat, int ld, int ch0, int B, int Tn, int seg_len,
                      int bn_channels, int growth, hipStream_t st);

// kernels' host launchers (defined in the .hip files)
int vp_fbank_release_tables(vp_ctx* ctx);
int vp_mel_release_tables(vp_ctx* ctx);

// tanh for an epilogue whose result is stored as TO: bf16 outputs (8 mantissa bits) take 1 - 2 / (2^(2 x log2 e) + 1) on the
// hardware exp2 / rcp (5 instructions, |error| ~1e-7); f32 outputs keep tanhf (~40 instructions: the parity instrument).
template <typename TO>
__device__ __forceinline__ float vp_tanh_for(float x) {
    if constexpr (sizeof(TO) == 2) {
        const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
        return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    } else {
        return tanhf(x);
    }
}

// x * sigmoid(x) likewise: hardware exp2 / rcp for bf16 outputs, the exact division for f32 outputs
template <typename TO>
__device__ __forceinline__ float vp_silu_for(float x) {
    if constexpr (sizeof(TO) == 2) return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    else return x / (1.f + __expf(-x));
}
