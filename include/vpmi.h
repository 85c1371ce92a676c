/*
 * vpmi.h -- C ABI of libvpmi.so, the MI355X (gfx950) engine for the ppvector hot path.
 *
 * The reference (yeyupiaoling/VoiceprintRecognition-PaddlePaddle, ppvector 1.1.1) is pure Python and
 * has NO FFI / operator-plugin interface: its seams are Python call signatures into PaddlePaddle
 * (SURVEY.md section 8(b)).  Each entry point below therefore names the reference Python call site whose
 * arithmetic it replaces; INTEGRATION.md shows the ctypes binding a ppvector maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _h; the caller (PyTorch) owns all
 *     buffers and the stream; calls are asynchronous on `stream` and hold no mutable global state
 *     beyond the opaque vp_ctx (which owns small read-only device tables: twiddles, window, mel bank);
 *   - activations are "frame-major" (B, T, C): one frame = one contiguous channel vector -- the
 *     reference's (B, T, F) feature layout (featurizer.py:46) kept end to end, so its
 *     transpose([0,2,1]) (ecapa_tdnn.py:256) never materialises;
 *   - return 0 on success, <0 = VP_E*; never throws; vp_last_error(ctx) gives the text;
 *   - no torch types anywhere in this header.
 */
#ifndef VPMI_H
#define VPMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPMI_VERSION 100

typedef struct vp_ctx vp_ctx;
typedef void* vp_stream; /* hipStream_t */

enum { VP_OK = 0, VP_EINVAL = -1, VP_ENOMEM = -2, VP_EHIP = -3, VP_EUNSUP = -4, VP_EWORKSPACE = -5 };
enum { VP_F32 = 0, VP_BF16 = 1,                    /* element type of activations / GEMM weights     */
       VP_F32X3 = 2,                               /* `dtype` of a whole-backbone weights struct only: f32 tensors (as VP_F32), every
                                                      conv / GEMM in split precision (vp_conv1d_desc.mfma_bf16 = 2)                  */
       VP_HL32 = 3 };                              /* split bf16 planes, the storage form of split precision: a row of C channels
                                                      (C % 32 == 0) is C / 32 groups of 128 bytes = [32 bf16 hi | 32 bf16 lo], value =
                                                      hi + lo, hi = bf16(v), lo = bf16(v - hi).  4 bytes per element: leading dimensions
                                                      and offsets count elements exactly as for f32 (multiples of 32).  The consumer's
                                                      LDS K-stage row IS a group, so wide layers stream it by LDS-DMA               */
enum { VP_PAD_NONE = 0, VP_PAD_REFLECT = 1, VP_PAD_ZERO = 2 };
enum { VP_ACT_NONE = 0, VP_ACT_RELU = 1, VP_ACT_SIGMOID = 2, VP_ACT_TANH = 3,
       VP_ACT_HARDTANH20 = 4,   /* clamp to [0, 20]: ERes2Net's "ReLU" (models/eres2net.py:12-20) */
       VP_ACT_SILU = 5 };

int vp_version(void);
vp_ctx* vp_create(int device);                      /* NULL on failure                                */
void vp_destroy(vp_ctx* ctx);
const char* vp_last_error(vp_ctx* ctx);             /* host string, valid until the next failing call */
/* Margin of the margin-softmax losses as DEVICE data: table = 5 device floats [m, cos m, sin m, cos(pi - m), 1 + cos(pi - m)]
 * (or NULL: back to the `margin` launch scalars).  While set, every loss entry point below reads the margin from the table
 * when its kernel RUNS, so a launch sequence captured in a HIP graph follows MarginScheduler.step (optimizer/scheduler.py:69,76:
 * criterion.update(margin=...) every step of the ramp) without being re-captured.  Replaces nothing in the reference: there the
 * margin is a Python attribute read by eager ops (loss/aamloss.py:49-53). */
int vp_set_margin_table(vp_ctx* ctx, const float* table);

/* ------------------------------------------------------------------------------------------------
 * Fbank + CMN  -- replaces AudioFeaturizer.forward (ppvector/data_utils/featurizer.py:33-60) with
 * feature_method 'Fbank' -> KaldiFbank.forward (featurizer.py:88-101) ->
 * paddleaudio.compliance.kaldi.fbank(waveform, sr=, n_mels=) (call site featurizer.py:97).
 * wav (B, L) f32 in [-1,1]  ->  out (B, T, n_mels) f32, time-mean subtracted over the padded T,
 * rows t >= int32(lens_ratio[b] * T) zeroed when lens_ratio != NULL (featurizer.py:51-59).
 * out_bf16 (optional) receives the same tensor rounded to bf16 for the bf16 network path.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int sample_rate;        /* 16000 */
    int n_mels;             /* paddleaudio default 23; configs use 80 */
    float frame_length_ms;  /* 25 */
    float frame_shift_ms;   /* 10 */
    float preemph;          /* 0.97 */
    int remove_dc;          /* 1 */
    float low_freq;         /* 20 */
    float high_freq;        /* 0 -> Nyquist */
    float log_floor;        /* 1e-7 */
} vp_fbank_opts;

void vp_fbank_default_opts(vp_fbank_opts* o);
int vp_fbank_num_frames(const vp_fbank_opts* o, int n_samples);       /* snip_edges frame count   */
size_t vp_fbank_workspace_bytes(const vp_fbank_opts* o, int B, int L);
int vp_fbank_cmn_f32(vp_ctx* ctx, const float* wav, const float* lens_ratio, int B, int L,
                     const vp_fbank_opts* o, float* out, void* out_bf16, void* ws, size_t ws_bytes,
                     vp_stream stream);
/* vp_fbank_cmn_pcm16 -- the same featurizer over 16-bit PCM as the decoder delivers it (the reference's readers convert to float32 on
 * the host: yeaudio AudioSegment, x / 32768): samples are widened as the frame kernel loads them (x pcm_scale), so the host -> device
 * copy in front of featurizer.py:33-60 carries half the bytes.  L and the frame shift must be even.  Same outputs, bit for bit, as
 * vp_fbank_cmn_f32 over the widened samples. */
int vp_fbank_cmn_pcm16(vp_ctx* ctx, const int16_t* wav, float pcm_scale, const float* lens_ratio, int B, int L,
                       const vp_fbank_opts* o, float* out, void* out_bf16, void* ws, size_t ws_bytes, vp_stream stream);

/* Ragged batch, the training loader's semantics: the reference featurises every utterance on its own (AudioFeaturizer on
 * one waveform, time mean over ITS frames, data_utils/reader.py:102-103) and collate_fn zero-pads the features
 * (collate_fn.py:5-23).  wav (B, L) holds utterance b in its first n_samples[b] samples; frames [0, n_frames[b]),
 * n_frames[b] = snip-edges count of n_samples[b], are mean-normalised over that range and the remaining rows are zero.
 * n_frames (B) is written for the caller (collate_fn's input_lens). */
int vp_fbank_cmn_ragged_f32(vp_ctx* ctx, const float* wav, const int32_t* n_samples, int B, int L, const vp_fbank_opts* o,
                            float* out, void* out_bf16, int32_t* n_frames, void* ws, size_t ws_bytes, vp_stream stream);

/* MelSpectrogram + CMN -- replaces AudioFeaturizer.forward with feature_method 'MelSpectrogram' ->
 * paddle.audio.features.MelSpectrogram(**method_args) (featurizer.py:22-23): centred STFT (reflect pad,
 * periodic Hann, win_length <= n_fft), |X|^power, Slaney mel bank (htk=False, norm='slaney'), linear
 * power; then transpose, time-mean subtraction, length mask as above.  n_fft in {512, 1024, 2048}. */
typedef struct {
    int sample_rate;        /* paddle default 22050 */
    int n_fft;              /* 2048 */
    int hop_length;         /* 512 */
    int win_length;         /* 0 -> n_fft */
    int n_mels;             /* 64 */
    float f_min;            /* 50 */
    float f_max;            /* 0 -> sample_rate / 2 */
    float power;            /* 2 */
    int log_db;             /* 0: linear power (MelSpectrogram); 1: LogMelSpectrogram (featurizer.py:20-21) =
                             * power_to_db: 10 log10(max(x, amin)) - 10 log10(max(ref_value, amin)), top_db None */
    float amin;             /* 1e-10 */
    float ref_value;        /* 1.0 */
} vp_mel_opts;

void vp_mel_default_opts(vp_mel_opts* o);
int vp_mel_num_frames(const vp_mel_opts* o, int n_samples);           /* 1 + L / hop              */
size_t vp_mel_workspace_bytes(const vp_mel_opts* o, int B, int L);
int vp_melspec_cmn_f32(vp_ctx* ctx, const float* wav, const float* lens_ratio, int B, int L,
                       const vp_mel_opts* o, float* out, void* out_bf16, void* ws, size_t ws_bytes,
                       vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * conv1d as implicit GEMM with fused epilogue -- replaces, per launch, the reference chain
 *   Conv1d.forward (models/utils.py:65-93: reflect "same" pad + nn.Conv1D)  /  nn.Conv1D (tdnn.py:13-21)
 *   -> activation -> BatchNorm1d (models/utils.py:96-119, eval mode)   i.e. TDNNBlock (utils.py:147-148),
 * plus the Res2Net "x_i + y_{i-1}" hand-off (ecapa_tdnn.py:36-47) and the time-sums that SEBlock
 * (ecapa_tdnn.py:69-78) and AttentiveStatisticsPooling's global context (pooling.py:97-104) need.
 *
 *   y[b,t,n] = act2( bn( act( bias[n] + rowbias[b,n] + sum_{j<KW} sum_{c<Cin}
 *                    w[n][j*Cin+c] * pro(x[b, src(t,j), c]) ) ) * gate[b,seg(t),n] + res[b,t,n] )
 *   src(t,j) = t*stride - pad_left + j*dilation, reflected / zero-filled per pad_mode.
 * x, y, add_in, aux are (B*T, ld) row-major with a channel offset (so slices of a concat buffer
 * are addressed in place); output columns [0, ysplit) can additionally be stored to a second
 * tensor y2 (Res2Net's pass-through chunk y_0 = x_0 lands in the concat buffer for free).
 * aux = y + add_in (same dtype as y).  psum/psumsq: per (M-tile, utterance-segment) partial sums
 * of (y - bn_shift) and its square over the tile's rows, layout [tiles_m][nseg][Cout] f32.
 * Alignment: Cin, ldx, xoff multiples of 16 B / sizeof(elem); Cout, ldy, yoff, ysplit multiples of 4.
 * Each operand tensor must be smaller than 4 GiB (32-bit buffer offsets).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int dtype_in, dtype_out;
    int B, T_in, T_out;
    int Cin, Cout, KW, dilation, stride, pad_left, pad_mode;   /* KW = total taps (KT*KF for 2-D) */
    const void* x;   int ldx, xoff;
    const void* w;                       /* [Cout][KW*Cin], dtype_in; 2-D: tap = kt*KF + kf */
    const float* bias;                   /* [Cout] or NULL */
    const float* rowbias;                /* [B][Cout] or NULL */
    int act;                             /* VP_ACT_NONE | VP_ACT_RELU, applied before BN */
    const float* bn_scale;               /* [Cout] or NULL: gamma / sqrt(var + eps) */
    const float* bn_shift;               /* [Cout] or NULL: beta - mean * scale */
    int act2;                            /* VP_ACT_NONE | VP_ACT_TANH | VP_ACT_RELU | VP_ACT_HARDTANH20 | VP_ACT_SILU, applied last */
    void* y;         int ldy, yoff;
    void* y2;        int ldy2, y2off, ysplit;   /* columns [0, ysplit) are ALSO stored to y2 */
    const void* add_in; int ld_add, add_off;
    void* aux;       int ld_aux, aux_off;
    float* psum;
    float* psumsq;
    /* --- extensions used by CAM++ (campplus.py); zero / NULL = off --------------------------------- */
    int F_in, F_out, KF, stride_f, pad_f;        /* 2-D conv over (time, freq): rows are (b, t, f)      */
    const float* pro_scale;              /* [Cin] BN-affine + ReLU on the INPUT channels (KW == 1)      */
    const float* pro_shift;
    const void* res; int ld_res, res_off;        /* residual added after BN, before act2 (dtype_out)     */
    const float* gate; int gate_len, gate_nseg;  /* [B*gate_nseg][Cout]: y *= gate[b, t / gate_len, :]   */
    /* --- mixed-precision training (trainer.py:209-229, auto_cast O1) --------------------------------------------------- */
    int mfma_bf16;                       /* f32 tensors only: 0 = exact f32 matrix cores; 1 = round x and w to bf16 while staging
                                            and run the bf16 matrix cores (f32 accumulate, f32 out); 2 = split precision: x and w
                                            each become bf16 hi + lo while staging, hi*hi + hi*lo + lo*hi on the bf16 matrix cores
                                            (~2^-16 relative per product: the reference's 1e-4 score tolerance at 1/3 of the
                                            bf16 rate = 5.3x the exact-f32 rate); 3 = as 2, with `w` ALREADY split: split bf16
                                            planes (VP_HL32) [Cout][Kw], Kw = KW*Cin rounded up to a multiple of 32 with zero
                                            columns (vp_tdnn_layer.w_hl) -- only x is split while staging                 */
} vp_conv1d_desc;

int vp_conv1d_tiles_m(int B, int T_out);            /* rows of the psum arrays                     */
int vp_conv1d_nseg(int T_out);                      /* utterance segments per M-tile               */
int vp_conv1d_fwd(vp_ctx* ctx, const vp_conv1d_desc* d, vp_stream stream);
/* Tuning knob (not part of the reference surface): K-loop schedule of the 256-wide bf16 kernel vp_conv1d_fwd dispatches the
 * wide layers to -- -1 = default, 0 = never (128-wide kernel), 1..3 = the two-stage schedules, 4 = half-tile
 * ring, 5 = half-tile ring with resident workgroups (csrc/conv_gemm.hip).  Process-wide; returns the previous value; out-of-range values only query.  Results
 * are identical for every schedule. */
int vp_conv256_select(int schedule);

/* mean / std over time from the conv1d partial sums: stats[b][0:C] = mean, stats[b][C:2C] = std,
 * std = sqrt(max(E[(x-mean)^2], eps)) -- pooling.py:90-93 with the all-ones mask of pooling.py:94-101
 * (lengths=None, the only branch the shipped entry points reach, trainer.py:210). */
int vp_moments_finalize(vp_ctx* ctx, const float* psum, const float* psumsq, const float* shift,
                        int B, int T, int C, float eps, int want_std, float* stats, vp_stream stream);
/* The same from the fused sums of a TRAINING conv -- sums of z = ReLU(conv + bias), the layer's output being y = scale z + shift
 * (batch-statistics BatchNorm): SEBlock's squeeze mean (ecapa_tdnn.py:66-71) and ASP's context statistics (pooling.py:97-104) without a
 * pass over y. */
int vp_moments_finalize_affine(vp_ctx* ctx, const float* psum, const float* psumsq, const float* scale, const float* shift,
                               int B, int T, int C, float eps, int want_std, float* stats, vp_stream stream);

/* small dense layer in exact f32 (f32 MFMA): out[M][N] = act(a[M][K] @ W + bias).
 * w_is_kn = 0: W given as [N][K];  1: W given as [K][N] (Paddle Linear / fc.py weight layout). */
int vp_dense_f32(vp_ctx* ctx, const float* a, int lda, const float* w, int w_is_kn, const float* bias,
                 int M, int N, int K, int act, float* out, int ldo, vp_stream stream);

/* f32 -> bf16 (round to nearest even) of a contiguous tensor; x 16-B aligned, y 8-B aligned. */
int vp_cast_f32_bf16(vp_ctx* ctx, const float* x, void* y, long long n, vp_stream stream);


/* SE gate applied + residual: out = x * s[b, c] + res  (ecapa_tdnn.py:82 and :142). */
int vp_se_scale_residual(vp_ctx* ctx, int dtype, const void* x, int ldx, int xoff, const float* s,
                         const void* res, int ldr, int roff, void* out, int ldo, int ooff,
                         int B, int T, int C, vp_stream stream);
/* The same on f32 tensors, plus the result once more as bf16 at shadow[m * ld_shadow + shadow_off + c]: under enable_amp the block
 * output (ecapa_tdnn.py:139-142) is the GEMM operand of the next block's tdnn1 and of the MFA concatenation (:262-263). */
int vp_se_scale_residual_shadow(vp_ctx* ctx, const float* x, int ldx, int xoff, const float* s, const float* res, int ldr, int roff,
                                float* out, int ldo, int ooff, void* shadow, int ld_shadow, int shadow_off, int B, int T, int C,
                                vp_stream stream);

/* attention softmax over time + weighted mean/std (pooling.py:114-123):
 * logits (B*T, C) f32, x (B*T, ldx) -> pooled (B, 2C) f32 = [mean | std]. */
int vp_asp_softmax_stats(vp_ctx* ctx, int dtype, const float* logits, const void* x, int ldx, int xoff,
                         int B, int T, int C, float eps, float* pooled, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-backbone forward, eval mode -- replaces EcapaTdnn.forward (models/ecapa_tdnn.py:245-276)
 * and TDNN.forward (models/tdnn.py:46-68).  Weights are packed by the host (BN folded to
 * scale/shift, conv weights as [Cout][KW*Cin] in `dtype`).
 * ---------------------------------------------------------------------------------------------- */
#define VP_MAX_SE_BLOCKS 8
#define VP_MAX_RES2 15

typedef struct {
    const void* w;            /* [cout][kw*cin] in the network dtype */
    const float* bias;        /* [cout] */
    const float* bn_scale;    /* [cout] or NULL */
    const float* bn_shift;    /* [cout] or NULL */
    int cin, cout, kw, dil;
    const void* w_hl;         /* optional (NULL = absent): the same weights as split bf16 planes (VP_HL32), [cout][Kp], Kp = kw*cin
                                 rounded up to a multiple of 32 with zero columns -- what the split-precision fast path of a
                                 VP_F32X3 backbone reads (LDS-DMA-able)                                                  */
} vp_tdnn_layer;

typedef struct {
    vp_tdnn_layer tdnn1;
    vp_tdnn_layer res2[VP_MAX_RES2];
    vp_tdnn_layer tdnn2;
    const float* se_w1;       /* [C][se_ch]  f32, input-major (Conv1D weight transposed) */
    const float* se_b1;       /* [se_ch] */
    const float* se_w2;       /* [se_ch][C]  f32, input-major */
    const float* se_b2;       /* [C] */
} vp_se_res2_block;

typedef struct {
    vp_tdnn_layer tdnn;       /* x-part of the attention TDNN: w [att][C]; bias, BN as usual */
    const float* w_ctx;       /* [att][2C] f32: columns acting on the tiled mean | std, or NULL */
    const void* conv_w;       /* [C][att] network dtype */
    const float* conv_b;      /* [C] */
    int C, att;
} vp_asp_weights;

typedef struct {
    int dtype;                /* VP_F32 | VP_F32X3 (f32 tensors, split-precision contractions) | VP_BF16 */
    int feat_dim, embd_dim, n_blocks, res2_scale, se_ch;
    vp_tdnn_layer block0;
    vp_se_res2_block blk[VP_MAX_SE_BLOCKS];
    vp_tdnn_layer mfa;
    vp_asp_weights asp;
    const float* fc_w;        /* [embd][2*C_mfa] f32, asp_bn folded in */
    const float* fc_b;        /* [embd] */
} vp_ecapa_weights;

/* The three fused bf16 kernels of the ECAPA forward, one door each (vp_ecapa_fwd runs them in sequence).
 * vp_res2_chain_fwd: Res2NetBlock.forward (ecapa_tdnn.py:36-47) for slices 1 .. nconv of t1 (B*T, C) bf16 -- y_1 = f_1(x_1),
 *   y_j = f_j(x_j + y_{j-1}), f_j = BN(ReLU(conv k3, dilation, reflect pad)) -- written to columns [j*width, (j+1)*width) of r2;
 *   slice 0 of r2 is left alone (the producing conv writes it).  One workgroup per utterance, activations ping-pong in LDS.
 *   VP_EUNSUP unless width == 64, equal dilations, 2 <= T <= 384 (callers fall back to one vp_conv1d_fwd per conv).
 * vp_asp_fused_fwd: AttentiveStatisticsPooling.forward (pooling.py:105-123) from the attention hidden layer on: logits =
 *   w (C x att) h + bias, softmax over time, weighted mean and std of x; h (B*T, att) bf16, x (B*T, ldx) bf16, center (B, ldc)
 *   f32 = per-utterance channel means (numerical centre of the variance), pooled (B, 2C) f32 = [mean | std].  att == 128.
 * vp_se_gate_fwd: SEBlock.forward (ecapa_tdnn.py:69-82) without the final product: s = sigmoid(w2' relu(w1' mean + b1) + b2),
 *   mean = shift + (sum of the producing conv's partial time sums) / T; psum as vp_conv1d_fwd writes it; w1 [C][H], w2 [H][C]. */
int vp_res2_chain_fwd(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C,
                      int width, vp_stream stream);
int vp_asp_fused_fwd(vp_ctx* ctx, const void* h, const void* w, const float* bias, const void* x, int ldx, const float* center,
                     int ldc, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream);
int vp_se_gate_fwd(vp_ctx* ctx, const float* psum, const float* shift, int B, int T, int C, int H, const float* w1, const float* b1,
                   const float* w2, const float* b2, float* out, vp_stream stream);
/* vp_asp_utt_fwd: the WHOLE AttentiveStatisticsPooling.forward (pooling.py:105-123) of the bf16 engine as one kernel per utterance:
 *   h = tanh(BN(ReLU(tdnn.w x + tdnn.bias + rowbias[b]))) (rowbias = the [mean; std] columns of the attention TDNN applied to the global
 *   context, (B, att) f32 or NULL), logits = conv_w h (+ conv_b: constant over time, cancels), softmax over time, pooled (B, 2C) f32 =
 *   [sum_t a x | sqrt(max(sum_t a x^2 - mean^2, eps))].  The reference computes the CENTRED form sqrt(clip(sum_t a (x - mean)^2, eps))
 *   (pooling.py:44-46 `_compute_statistics`); the uncentred form needs one pass over x and equals it up to the cancellation in
 *   E[x^2] - mean^2: with f32 sums over bf16 x the absolute error of the variance is <= ~2^-22 (mean^2 + var) per channel, i.e. the std
 *   loses accuracy only where |mean| >> std (|mean| / std > ~100 for 1e-3 relative); BatchNorm in front of ASP keeps the MFA output's
 *   per-channel |mean| / std of order one, and tests/test_gpu_kernels.py::test_asp_utt_kernel_vs_float64 bounds the result against the
 *   centred float64 form (3e-4 abs).  x (B*T, ldx) bf16, tdnn.w [att][C] bf16,
 *   conv_w [C][att] bf16.  VP_EUNSUP unless att == 128, C % 64 == 0, T <= 304, tdnn = 1x1 with BN (vp_ecapa_fwd then runs the
 *   conv GEMM + vp_asp_fused_fwd pair). */
int vp_asp_utt_fwd(vp_ctx* ctx, const void* x, int ldx, const vp_tdnn_layer* tdnn, const float* rowbias, const void* conv_w,
                   const float* conv_b, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream);

/* The two fused kernels of the SPLIT-PRECISION ECAPA forward (w->dtype = VP_F32X3 with split weights present: csrc/ecapa.hip), tensors
 * stored as split bf16 planes (VP_HL32), every product hi*hi + hi*lo + lo*hi on the bf16 matrix cores, f32 accumulate:
 * vp_res2_chain_x3_fwd: Res2NetBlock.forward (ecapa_tdnn.py:36-47) as vp_res2_chain_fwd, t1 / r2 hl32 (B*T, C), weights layers[j].w_hl
 *   (hl32 [64][192]).  The utterance is cut into time segments with recomputed halos so that a segment's ping-pong buffers and one
 *   conv's weights fit the LDS (csrc/res2_x3.hip).  VP_EUNSUP unless width == 64, equal dilations, w_hl present, the segments fit.
 * vp_asp_fused_x3_fwd: AttentiveStatisticsPooling.forward (pooling.py:112-123) from the attention hidden layer on, as vp_asp_fused_fwd:
 *   h (B*T, att) hl32, w [C][att] F32 (split in registers), x (B*T, ldx) hl32.  att == 128, C and ldx multiples of 32. */
int vp_res2_chain_x3_fwd(vp_ctx* ctx, const vp_tdnn_layer* layers, int nconv, const void* t1, void* r2, int B, int T, int C,
                         int width, vp_stream stream);
int vp_asp_fused_x3_fwd(vp_ctx* ctx, const void* h, const float* w, const float* bias, const void* x, int ldx, const float* center,
                        int ldc, int B, int T, int C, int att, float eps, float* pooled, vp_stream stream);

size_t vp_ecapa_workspace_bytes(const vp_ecapa_weights* w, int B, int T);
/* feats: (B, T, feat_dim) in w->dtype (f32 for VP_F32X3); emb: (B, embd_dim) f32. */
int vp_ecapa_fwd(vp_ctx* ctx, const vp_ecapa_weights* w, const void* feats, int B, int T, float* emb,
                 void* ws, size_t ws_bytes, vp_stream stream);

typedef struct {
    int dtype;                /* VP_F32 | VP_F32X3 | VP_BF16 (as vp_ecapa_weights) */
    int feat_dim, embd_dim, channels;
    vp_tdnn_layer td[5];      /* un-padded convs, ReLU then BN (td[4]: no BN) */
    vp_asp_weights asp;
    const float* lin_w;       /* [embd][2*channels] f32 with bn5 and bn6 folded in */
    const float* lin_b;       /* [embd] */
} vp_tdnn_weights;

size_t vp_tdnn_workspace_bytes(const vp_tdnn_weights* w, int B, int T);
int vp_tdnn_fwd(vp_ctx* ctx, const vp_tdnn_weights* w, const void* feats, int B, int T, float* emb,
                void* ws, size_t ws_bytes, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * CAM++ backbone forward, eval mode -- replaces CAMPPlus.forward (models/campplus.py:331-335):
 * FCM head (:246-281), TDNNLayer (:38-64), CAMDenseTDNNBlock x3 (:145-173) with CAMLayer context
 * gating (:67-106), TransitLayer (:176-189), BN-ReLU + statistics_pooling (:24-30), DenseLayer (:192-208).
 * 2-D conv weights: [Cout][tap*Cin + c] with tap = kt*3 + kf (time-major taps); vp_tdnn_layer.kw = 9
 * for 3x3, 1 for the 1x1 shortcut.  tdnn.w is permuted so that its input channel index is f*32 + c.
 * ---------------------------------------------------------------------------------------------- */
#define VP_MAX_CAM_LAYERS 64
#define VP_MAX_CAM_BLOCKS 4

typedef struct {
    vp_tdnn_layer conv1, conv2, shortcut;   /* BN folded to bn_scale/bn_shift of each */
    int stride, has_shortcut;
} vp_resblock;

typedef struct {
    const float* bn1_scale;   /* [cin]: nonlinear1 (BN) of the layer input, ReLU follows */
    const float* bn1_shift;
    vp_tdnn_layer linear1;    /* 1x1 cin -> bn_channels; bn_scale/shift = nonlinear2, ReLU follows */
    vp_tdnn_layer local;      /* cam_layer.linear_local: k3, dilation, zero 'same' pad, bias only */
    const float* ctx_w1;      /* [bn_channels][bn_channels/2] f32, input-major (Conv1D weight transposed) */
    const float* ctx_b1;
    const float* ctx_w2;      /* [bn_channels/2][growth] f32, input-major */
    const float* ctx_b2;
} vp_cam_layer;

typedef struct {
    const float* bn_scale;    /* [cin] */
    const float* bn_shift;
    vp_tdnn_layer linear;     /* 1x1 cin -> cin/2, bias */
} vp_transit;

typedef struct {
    int dtype;                /* VP_F32 | VP_F32X3 | VP_BF16 (as vp_ecapa_weights) */
    int feat_dim, embd_dim, m_channels, init_channels, growth, bn_channels, seg_len;
    int n_blocks;
    int block_layers[VP_MAX_CAM_BLOCKS];
    const float* fcm1_w;      /* [32][9] f32, tap = kt*3 + kf */
    const float* fcm1_b;
    const float* fcm1_scale;
    const float* fcm1_shift;
    vp_resblock res[4];
    vp_tdnn_layer fcm_conv2;
    vp_tdnn_layer tdnn;
    vp_cam_layer layers[VP_MAX_CAM_LAYERS];
    vp_transit transit[VP_MAX_CAM_BLOCKS];
    const float* out_bn_scale;
    const float* out_bn_shift;
    const float* dense_w;     /* [embd][2*C_final] f32 with the trailing BN folded in */
    const float* dense_b;
} vp_campplus_weights;

/* Kernel doors of the CAM++ forward's two fused bf16 kernels (kernel-level parity tests; vp_campplus_fwd uses them internally):
 * vp_cam_block_fwd: CAMDenseTDNNBlock.forward (models/campplus.py:145-173) over n_layers CAMDenseTDNNLayers (:109-142, CAMLayer
 *   :67-106): cat (B*Tn, ld) bf16 holds the block input in columns [0, ch0); layer l reads columns [0, ch0 + 32 l) and appends
 *   its 32 output channels in place.  VP_EUNSUP outside bottleneck 128 / growth 32 / k3 / T' <= 160 / <= 24 layers.
 * vp_conv3x3_c32_fwd: one 3x3 conv of the FCM head over a (B, T, F_in, 32) bf16 map (BasicResBlock conv1 / conv2, FCM.conv2:
 *   models/campplus.py:211-281), frequency stride 1 or 2, zero padding: y = [relu](bn(conv(x) + bias) [+ res]); with `shortcut`
 *   (stride 2) also y2 = bn_s(conv1x1_s2(x) + bias_s) (:232-239); with c1_feats the input map is FCM.conv1 (:254-255, 274) of the
 *   (B, T, F_in) bf16 features, produced inside the kernel (x is not read). */
int vp_cam_block_fwd(vp_ctx* ctx, const vp_cam_layer* layers, int n_layers, void* cat, int ld, int ch0, int B, int Tn, int seg_len,
                     int bn_channels, int growth, vp_stream stream);
/* vp_resblock_c32_fwd: a whole stride-1 BasicResBlock with the identity shortcut (models/campplus.py:211-243) over a (B, T, F, 32) bf16
 * map in one launch: y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x); h never leaves the chip.  x and y must not alias. */
int vp_resblock_c32_fwd(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv1, const vp_tdnn_layer* conv2, int B, int T, int F,
                        vp_stream stream);
int vp_conv3x3_c32_fwd(vp_ctx* ctx, const void* x, void* y, const vp_tdnn_layer* conv, const void* res, int relu,
                       const vp_tdnn_layer* shortcut, void* y2, int B, int T, int F_in, int stride_f, const void* c1_feats,
                       const float* c1_w, const float* c1_b, const float* c1_scale, const float* c1_shift, vp_stream stream);

size_t vp_campplus_workspace_bytes(const vp_campplus_weights* w, int B, int T);
int vp_campplus_fwd(vp_ctx* ctx, const vp_campplus_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Batch assembly around the featurizer.
 * vp_spec_augment -- applies, in place, the masks the host drew for each utterance of a (B, T, F) batch; replaces the
 *   per-sample yeaudio SpecAugmentor call at data_utils/reader.py:105-107 (parameters configs/augmentation.yml:36-48).
 *   fmask [B][n_freq_masks][2] = (first bin, width), tmask [B][n_time_masks][2] = (first frame, width); width 0 = no
 *   mask.  Frequency masks first, then time masks; each is filled with the mean of the utterance's feature as it is
 *   at that moment, or with zero.
 * vp_pad_batch -- collate_fn (data_utils/collate_fn.py:5-23): out[b] = srcs[b] (lens[b] x F) zero-padded to Tmax.
 *   srcs is a DEVICE array of B device pointers.
 * vp_wave_batch_f32 -- the waveform side of the same assembly, the step in front of the featurizer: decibel normalisation
 *   over the WHOLE utterance (data_utils/reader.py:97-98, yeaudio AudioSegment.normalize(target_db): gain =
 *   10^((target_db - 10 log10(mean x^2)) / 20); a silent source keeps gain 1), then the crop to max_duration
 *   (reader.py:100-101: starts[b] = the host's random start in training, 0 otherwise; NULL = 0) and the zero padding to
 *   L that predict_batch applies to ragged waveforms (predict.py:246-254).  normalize = 0 applies gain_db[b] instead
 *   (VolumePerturbAugmentor's draw, reader.py:155-156; NULL = unity).  n_valid[b] (optional) = samples kept, the
 *   numerator of predict_batch's input_lens_ratio.  srcs: DEVICE array of B device pointers, lens: samples per source.
 * ---------------------------------------------------------------------------------------------- */
int vp_spec_augment(vp_ctx* ctx, int dtype, void* feats, int B, int T, int F, const int32_t* fmask, int n_freq_masks,
                    const int32_t* tmask, int n_time_masks, int replace_with_zero, vp_stream stream);
int vp_pad_batch(vp_ctx* ctx, int dtype, const void* const* srcs, const int32_t* lens, int B, int Tmax, int F, void* out,
                 vp_stream stream);
int vp_wave_batch_f32(vp_ctx* ctx, const float* const* srcs, const int32_t* lens, const int32_t* starts, int B, int L,
                      int normalize, float target_db, const float* gain_db, float* out, int32_t* n_valid, vp_stream stream);
/* Speed perturbation (SpeedPerturbAugmentor -> AudioSegment.change_speed, call site data_utils/reader.py:155-156;
 * configs/augmentation.yml:1-6; yeaudio is third party: resampling by linear interpolation, dst[b][i] = interp of srcs[b] at
 * i * lens[b] / (new_lens[b] - 1), the tail clamped to the last sample; the host draws the rate from {1.0, 0.9, 1.1} and passes
 * new_lens[b] = int(lens[b] / rate)).  srcs / dsts: DEVICE arrays of B device pointers. */
int vp_speed_perturb_f32(vp_ctx* ctx, const float* const* srcs, const int32_t* lens, const int32_t* new_lens,
                         float* const* dsts, int B, int max_new_len, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * ResNetSE backbone forward, eval mode -- replaces ResNetSE.forward (models/resnet_se.py:121-139) with
 * SEBottleneck (:8-45) / SELayer (:48-63) blocks and ASP pooling.  2-D conv weights as for CAM++
 * ([Cout][tap*Cin + c], tap = kt*3 + kf).  SE Linear weights stay in Paddle's [in, out] layout.  The ASP /
 * Linear weights are permuted by the host to the engine's channel order f*C + c (reference: c*F + f).
 * ---------------------------------------------------------------------------------------------- */
#define VP_MAX_RSE_BLOCKS 32

typedef struct {
    vp_tdnn_layer conv1, conv2, conv3, down;  /* 1x1, 3x3 (kw = 9, stride), 1x1, optional strided 1x1; BN folded */
    const float* se_w1;       /* [C][C/8] f32 (Paddle Linear layout) */
    const float* se_b1;
    const float* se_w2;       /* [C/8][C] f32 */
    const float* se_b2;
    int stride, has_down;
} vp_rse_block;

typedef struct {
    int dtype;                /* VP_F32 | VP_F32X3 | VP_BF16 (as vp_ecapa_weights) */
    int feat_dim, embd_dim, n_blocks, c1_channels;
    const float* c1_w;        /* [32][9] f32, tap = kt*3 + kf */
    const float* c1_b;
    const float* c1_scale;
    const float* c1_shift;
    vp_rse_block blk[VP_MAX_RSE_BLOCKS];
    vp_asp_weights asp;       /* C = F/8 * C4 channels */
    const float* lin_w;       /* [embd][2*C] f32, bn2 and bn3 folded */
    const float* lin_b;
} vp_resnetse_weights;

/* vp_pointwise_fwd -- kernel door (tests): the streaming 1x1 conv of the few-channel full-resolution stages (SEBottleneck conv1 /
 * conv3 / downsample, models/resnet_se.py:8-45): same vp_conv1d_desc semantics as vp_conv1d_fwd for KW = 1, stride 1, bf16, bias /
 * ReLU / BN / ReLU epilogue and the fused per-utterance column sums (psum, T_out = positions per utterance). */
int vp_pointwise_fwd(vp_ctx* ctx, const vp_conv1d_desc* d, vp_stream stream);
size_t vp_resnetse_workspace_bytes(const vp_resnetse_weights* w, int B, int T);
int vp_resnetse_fwd(vp_ctx* ctx, const vp_resnetse_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * ERes2Net backbone forward, eval mode -- replaces ERes2Net.forward (models/eres2net.py:239-263) with
 * BasicBlockERes2Net (:56-108), BasicBlockERes2Net_diff_AFF (:111-169), AFF (:33-53), the stride-2 stage
 * fusion convs and TemporalStatsPool (models/pooling.py:128-146); ERes2NetV2 (:266-462) is the same graph with one fusion and
 * chunk widths 13 / 26 / 52 / 104, which the host zero-pads to multiples of 8.  2-D conv weights [Cout][tap*Cin + c],
 * tap = kt*3 + kf; BatchNorm folded to scale/shift; seg_1 permuted to the engine's (f*C + c) order.
 * ---------------------------------------------------------------------------------------------- */
#define VP_MAX_ERE_BLOCKS 40
#define VP_MAX_ERE_SCALE 4

typedef struct { vp_tdnn_layer c1, c2; } vp_aff_weights;   /* 1x1 (2C -> C/r, BN, SiLU), 1x1 (C/r -> C, BN) */

typedef struct {
    vp_tdnn_layer conv1;                      /* 1x1, stride on both axes, BN folded (Hardtanh(0,20) follows) */
    vp_tdnn_layer convs[VP_MAX_ERE_SCALE];    /* 3x3 on `width` channels, BN folded */
    vp_tdnn_layer conv3;                      /* 1x1 width*scale -> planes*expansion, BN folded */
    vp_tdnn_layer shortcut;                   /* optional strided 1x1 + BN */
    vp_aff_weights fuse[VP_MAX_ERE_SCALE - 1];
    int stride, has_shortcut, use_aff, width, scale;
} vp_ere_block;

typedef struct {
    int dtype;                /* VP_F32 | VP_F32X3 | VP_BF16 (as vp_ecapa_weights) */
    int feat_dim, embd_dim, n_blocks, m_channels;
    int stage_blocks[4];
    int first_fuse;           /* 0: ERes2Net (three bottom-up fusions, down / fuse [0..2]); 2: ERes2NetV2 (layer3_ds + fuse34 only, slot 2) */
    const float* c1_w;        /* [m][9] f32, tap = kt*3 + kf */
    const float* c1_b;
    const float* c1_scale;
    const float* c1_shift;
    vp_ere_block blk[VP_MAX_ERE_BLOCKS];
    vp_tdnn_layer down[3];    /* layer{1,2,3}_downsample: 3x3 stride 2, bias only */
    vp_aff_weights fuse[3];   /* fuse_mode12 / 123 / 1234 */
    const float* seg_w;       /* [embd][2 * F/8 * C] f32 */
    const float* seg_b;
} vp_eres2net_weights;

size_t vp_eres2net_workspace_bytes(const vp_eres2net_weights* w, int B, int T);
int vp_eres2net_fwd(vp_ctx* ctx, const vp_eres2net_weights* w, const void* feats, int B, int T, float* emb,
                    void* ws, size_t ws_bytes, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Cosine classifier + AAM-softmax loss -- replaces SpeakerIdentification.forward 'Cosine' branch
 * (models/fc.py:41-53) and AAMLoss.forward (loss/aamloss.py:28-47) incl. CrossEntropyLoss
 * (label_smoothing, mean reduction).  emb (B, D) f32; W (D, C) f32 (fc.py:31 layout); labels int64.
 * logits (B, C) f32 = cosines (optional output); row_loss (B) f32; loss (1) f32 = mean.
 * ---------------------------------------------------------------------------------------------- */
size_t vp_cosine_logits_workspace_bytes(int B, int D, int C);
int vp_cosine_logits_f32(vp_ctx* ctx, const float* emb, const float* W, int B, int D, int C, float* logits,
                         void* ws, size_t ws_bytes, vp_stream stream);                 /* fc.py:49 alone   */
int vp_aam_ce_fwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, float margin, float scale,
                  float label_smoothing, int easy_margin, float* loss, float* row_loss,
                  vp_stream stream);                                                    /* aamloss.py:28-47 */
size_t vp_cosine_aam_workspace_bytes(int B, int D, int C);
int vp_cosine_aam_ce_fwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D,
                         int C, float margin, float scale, float label_smoothing, int easy_margin,
                         float* loss, float* logits, float* row_loss, void* ws, size_t ws_bytes,
                         vp_stream stream);

/* Head + loss class-tiled (csrc/head_tiled.hip): the same value as vp_cosine_aam_ce_fwd without the (B, C) logits -- a workgroup
 * per 64 classes streams its slice of W once (column norms from the same bytes), forms the cosines of every utterance on the exact-f32
 * matrix cores and keeps only per-row online-softmax partials; a merge kernel finishes log-sum-exp / label smoothing per row.
 * Replaces SpeakerIdentification.forward 'Cosine' (fc.py:41-53) + AAMLoss.forward (aamloss.py:28-47) when only the loss is wanted
 * (evaluation of the training objective, bench.py's step; BASELINE configs[4]: 200 000 classes).  D % 4 == 0, D <= 256.
 * row_loss (B), lse (B, optional: log-sum-exp per row), cinv (C, optional: column inverse norms) and pred (B int32, optional: argmax of
 * the un-margined cosine = the prediction trainer.py:233-236 takes from outputs["logits"], first index on ties) are by-products. */
size_t vp_cosine_aam_tiled_workspace_bytes(int B, int D, int C);
int vp_cosine_aam_tiled_fwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                            float scale, float label_smoothing, int easy_margin, float* loss, float* row_loss, float* lse,
                            float* cinv, int* pred, void* ws, size_t ws_bytes, vp_stream stream);

/* The training step's head: forward value + d emb + d W class-tiled (no (B, C) cosine / gradient tensors; W streamed twice instead of
 * four times): per 64-class tile the cosines are recomputed and three f32-MFMA products run out of LDS / registers.  B <= 128 (the per-GPU
 * batch of BASELINE configs[4]) and D == 192, else VP_EUNSUP -- callers then take vp_cosine_aam_ce_bwd.  Same results as that entry point
 * to f32 rounding.  Replaces the autograd of fc.py:41-53 + aamloss.py:28-47 (trainer.py:213-219). */
size_t vp_cosine_aam_tiled_bwd_workspace_bytes(int B, int D, int C);
int vp_cosine_aam_tiled_bwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                            float scale, float label_smoothing, int easy_margin, float grad_scale, float* demb, float* dW, float* loss,
                            int* pred, void* ws, size_t ws_bytes, vp_stream stream);

/* Backward of head + loss (the autograd the reference gets from paddle for fc.py:41-53 + aamloss.py:28-47; called per
 * step by PPVectorTrainer.__train_epoch, trainer.py:213-219): demb (B, D) = d loss / d emb, dW (D, C) = d loss / d W,
 * both f32 and both scaled by grad_scale (1.0 for plain backward); loss (1) optional (the forward value, recomputed). */
size_t vp_cosine_aam_ce_bwd_workspace_bytes(int B, int D, int C);
int vp_cosine_aam_ce_bwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                         float scale, float label_smoothing, int easy_margin, float grad_scale, float* demb, float* dW,
                         float* loss, void* ws, size_t ws_bytes, vp_stream stream);

/* The same backward split at the reference's module boundary (classifier and loss are separate objects there:
 * models/fc.py and loss/aamloss.py): dlogits from the loss, then demb / dW from dlogits. */
int vp_aam_ce_bwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, float margin, float scale,
                  float label_smoothing, int easy_margin, float grad_scale, float* dlogits, float* loss, float* row_loss,
                  vp_stream stream);
size_t vp_cosine_logits_bwd_workspace_bytes(int B, int D, int C);
int vp_cosine_logits_bwd(vp_ctx* ctx, const float* emb, const float* W, const float* dcos, int B, int D, int C, float* demb,
                         float* dW, void* ws, size_t ws_bytes, vp_stream stream);

/* The rest of the reference's loss package over the same (B, C) head logits (build_loss, loss/__init__.py:16-22;
 * trainer.py:180,213), forward and backward, one pass per row with an online log-sum-exp -- no one-hot / margin tensors:
 *   VP_LOSS_AAM        loss/aamloss.py:28-47        (same arithmetic as vp_aam_ce_*)
 *   VP_LOSS_AM         loss/amloss.py:14-25         out = scale * (cos - margin * onehot)
 *   VP_LOSS_ARM        loss/armloss.py:14-31        out = where(z - z[label] < 0, 0, z), z as AM
 *   VP_LOSS_CE         loss/celoss.py:11-19         out = logits (margin / scale ignored)
 *   VP_LOSS_SUBCENTER  loss/subcenterloss.py:32-54  logits (B, C*K), class value = max over its K columns c*K .. c*K+K-1,
 *                                                   then the AAM margin; the gradient goes to the winning sub-centre
 * then CrossEntropyLoss(label_smoothing), mean over the batch.  K must be 1 except for VP_LOSS_SUBCENTER.
 * _bwd: dlogits (B, C*K) = grad_scale * d loss / d logits; loss (1) + row_loss (B) optional. */
enum { VP_LOSS_AAM = 0, VP_LOSS_AM = 1, VP_LOSS_ARM = 2, VP_LOSS_CE = 3, VP_LOSS_SUBCENTER = 4 };
int vp_margin_ce_fwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, int K, int kind, float margin,
                     float scale, float label_smoothing, int easy_margin, float* loss, float* row_loss, vp_stream stream);
int vp_margin_ce_bwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, int K, int kind, float margin,
                     float scale, float label_smoothing, int easy_margin, float grad_scale, float* dlogits, float* loss,
                     float* row_loss, vp_stream stream);
/* SphereFace2.forward (loss/sphereface2.py:47-69): loss = mean_b sum_c [ onehot * lanbuda * log(1 + exp(-P)) +
 * (1 - onehot) * (1 - lanbuda) * log(1 + exp(N)) ], P / N = scale * (g(cos) -/+ margin) + bias for margin type 'C'
 * (margin_type_a = 0) or the arc forms for 'A'; g(z) = 2 ((z + 1) / 2)^t - 1.  bias: device pointer to the module's (1, 1)
 * parameter (NULL = 0).  dlogits (B, C) / dbias (1) + row_dbias (B) optional, both scaled by grad_scale. */
int vp_sphereface2(vp_ctx* ctx, const float* logits, const int64_t* labels, const float* bias, int B, int C, float margin,
                   float scale, float lanbuda, int t, int margin_type_a, float grad_scale, float* loss, float* row_loss,
                   float* dlogits, float* dbias, float* row_dbias, vp_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Training-side building blocks (f32 engine) -- what paddle's autograd and optimiser run under
 * PPVectorTrainer.__train_epoch (trainer.py:202-274) for the layers of models/utils.py:22-148.
 *
 * vp_conv1d_wgrad_f32: dW[n][tap*Cin + c] = sum over (b, t) of dz[b,t,n] * x[b, src(t, tap), c], src = the forward's tap
 *   index rule (reflect / zero / none).  `d` is the FORWARD conv's descriptor (x, ldx, xoff, geometry).  The data gradient
 *   needs no new kernel: it is vp_conv1d_fwd over dz with the taps reversed and the channel roles swapped
 *   (W'[c][(KW-1-j)*Cout + n] = W[n][j*Cin + c], zero padding dil*(KW-1) - pad_left), see ppvector/train/functions.py.
 * vp_col_sums_f32: sums[0][c] = sum_m a[m][c]; sums[1][c] = sum_m a[m][c] * (b[m][c] - bmean[c]) * bscale[c] (when b):
 *   the two reductions of BatchNorm backward (d beta, d gamma), and the conv bias gradient.
 * vp_bn_train_finalize: batch mean / biased variance from a conv's fused column sums (all nparts = tiles * nseg partial
 *   rows), folded scale/shift for the apply pass, saved mean / invstd, running statistics updated in place
 *   (running = momentum * running + (1 - momentum) * batch; utils.py:108-115, momentum 0.9).
 * vp_affine_rows_f32: y = z * scale + shift (then ReLU when relu: the conv -> BN -> ReLU order of the 2-D models).   vp_bn_relu_bwd_f32: dz through BN (batch statistics) and the ReLU before it.
 * vp_adam_step_f32: Adam with coupled L2 (optimizer/__init__.py:12-18), bias-corrected, on a flat f32 buffer.
 * ---------------------------------------------------------------------------------------------- */
size_t vp_conv1d_wgrad_workspace_bytes(const vp_conv1d_desc* d);
int vp_conv1d_wgrad_f32(vp_ctx* ctx, const vp_conv1d_desc* d, const float* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                        vp_stream stream);
/* vp_conv1d_wgrad_oik_f32: the same gradient reduced straight into the model's (Cout, Cin, KW) layout (utils.py:22-93 stores
 * nn.Conv1D weights that way).  vp_conv_weight_layouts_f32: from that tensor, the forward weight panel wp (Cout, KW * Cin) and the
 * data-gradient panel w2 (Cin, KW * Cout; taps reversed, channel roles swapped) in one launch; either output may be NULL. */
int vp_conv1d_wgrad_oik_f32(vp_ctx* ctx, const vp_conv1d_desc* d, const float* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                            vp_stream stream);
int vp_conv_weight_layouts_f32(vp_ctx* ctx, const float* w, int Cout, int Cin, int KW, float* wp, float* w2, vp_stream stream);
/* The same pair of panels for a 2-D conv (nn.Conv2D weights are (Cout, Cin, KF, KT): resnet_se.py, eres2net.py, campplus.py FCM): the kernels'
 * tap order is kt * KF + kf, so wp is (Cout, KT, KF, Cin) and w2 (Cin, KT, KF, Cout) with both tap axes reversed.  With a 2-D descriptor
 * (KF > 1) vp_conv1d_wgrad_oik_f32 likewise reduces straight into (Cout, Cin, KF, KT). */
int vp_conv2d_weight_layouts_f32(vp_ctx* ctx, const float* w, int Cout, int Cin, int KF, int KT, float* wp, float* w2, vp_stream stream);
/* vp_prep_weights_bf16: for n f32 matrices w[i] (rows[i] x cols[i], row pitch ld[i] >= cols[i] -- HOST arrays of n entries): the bf16 copy
 * w16[i] [rows][cols] and / or the bf16 transpose wt16[i] [cols][rows] (either may be NULL) in one launch per 24 matrices -- the forward and
 * data-gradient weight panels of every wide 1x1 layer of a mixed-precision step (paddle.amp.auto_cast casts each conv's weight at each
 * call, trainer.py:209-213; here: once per step, all layers together). */
int vp_prep_weights_bf16(vp_ctx* ctx, const void* const* w, void* const* w16, void* const* wt16, const int* rows, const int* cols,
                         const int* ld, int n, vp_stream stream);
size_t vp_col_sums_workspace_bytes(long long M, int C);
int vp_col_sums_f32(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                    long long M, int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream);
int vp_bn_train_finalize(vp_ctx* ctx, const float* psum, const float* psumsq, int nparts, long long M, int C, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                         float* invstd, float* scale, float* shift, vp_stream stream);
int vp_affine_rows_f32(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y,
                       int ldy, int relu, vp_stream stream);
int vp_bn_relu_bwd_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                       const float* gamma, const float* sums, long long M, int C, int relu_mask, float* dz, int lddz,
                       vp_stream stream);
/* vp_bn_relu_bwd_dbias_f32: the same dz plus dbias[c] = sum_m dz[m][c] from the pass that writes dz -- the bias gradient of the
 * conv in front of the ReLU (utils.py:122-148: conv(+bias) -> ReLU -> BN) without a second read of dz. */
size_t vp_bn_relu_bwd_dbias_workspace_bytes(long long M, int C);
int vp_bn_relu_bwd_dbias_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, float* dz, int lddz,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream);
/* Mixed precision, wide layers: operands that are already bf16 in memory.  vp_bn_relu_bwd_dbias_bf16out writes dz as bf16 (lddz in
 * elements; dbias from the unrounded values); vp_conv1d_wgrad_bf16_oik takes x (d->x with d->dtype_in = VP_BF16) and dz both bf16
 * and reduces dW (f32) into the model's (Cout, Cin, KW) layout.  The forward and data-gradient convs of such a layer are
 * vp_conv1d_fwd with dtype_in = VP_BF16, dtype_out = VP_F32.  Same roundings as mfma_bf16 over f32 tensors (round-to-nearest-even of
 * x, W and dz), half the operand bytes. */
int vp_bn_relu_bwd_dbias_bf16out(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                                 const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz,
                                 float* dbias, void* ws, size_t ws_bytes, vp_stream stream);
int vp_conv1d_wgrad_bf16_oik(vp_ctx* ctx, const vp_conv1d_desc* d, const void* dz, int lddz, float* dW, void* ws, size_t ws_bytes,
                             vp_stream stream);
/* ... and the pre-BatchNorm activation z of such a layer kept as bf16 (the forward conv is then vp_conv1d_fwd bf16 -> bf16, i.e. the
 * 256-wide LDS-DMA kernel, its fused column sums taken from the f32 accumulators): the BatchNorm apply, the two backward
 * reductions and the backward itself with z (and dz) bf16 in memory, everything else f32 as in the functions they mirror. */
int vp_affine_rows_b16_f32(vp_ctx* ctx, const void* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y,
                           int ldy, int relu, vp_stream stream);
int vp_col_sums_f32_b16(vp_ctx* ctx, const float* a, int lda, const void* b, int ldb, const float* bmean, const float* bscale, long long M,
                        int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream);
int vp_bn_relu_bwd_dbias_b16(vp_ctx* ctx, const float* dy, int lddy, const void* z, int ldz, const float* mean, const float* invstd,
                             const float* gamma, const float* sums, long long M, int C, int relu_mask, void* dz, int lddz,
                             float* dbias, void* ws, size_t ws_bytes, vp_stream stream);
/* vp_pack_segments_f32: dst[offs[i] .. offs[i] + sizes[i]) = srcs[i] (zeros where srcs[i] is NULL), HOST arrays of n entries -- the
 * parameters' gradient tensors into the optimiser's flat buffer in one or two launches (what fleet's fused gradient buffers do for
 * the reference's DataParallel; trainer.py:213-229 only sees loss.backward() / optimizer.step()). */
int vp_pack_segments_f32(vp_ctx* ctx, const void* const* srcs, const long long* offs, const long long* sizes, int n, float* dst,
                         vp_stream stream);
int vp_adam_step_f32(vp_ctx* ctx, float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, vp_stream stream);
/* The other optimisers build_optimizer can name (optimizer/__init__.py:12-18 resolves any paddle.optimizer class; [3P-memory]
 * update rules): vp_adamw_step_f32 = Adam with DECOUPLED decay p *= 1 - lr * coeff (paddle.optimizer.AdamW, coeff = weight_decay,
 * default 0.01); vp_momentum_step_f32: g += wd * p; vel = mu * vel + g; p -= lr * vel, or p -= lr * (g + mu * vel) with
 * use_nesterov (paddle.optimizer.Momentum; momentum = 0 is paddle.optimizer.SGD). */
int vp_adamw_step_f32(vp_ctx* ctx, float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1, float beta2,
                      float eps, float coeff, int step, float grad_scale, vp_stream stream);
int vp_momentum_step_f32(vp_ctx* ctx, float* param, const float* grad, float* velocity, long long n, float lr, float momentum,
                         float weight_decay, int use_nesterov, float grad_scale, vp_stream stream);

/* ASP in training (models/pooling.py:69-125): per-utterance sums (gradient of the context bias), the global-context
 * statistics [mean | sqrt(max(var, eps))] and their backward, the backward of softmax-over-time + weighted mean/std
 * (forward = vp_asp_softmax_stats), tanh / sigmoid and their backward.  e (B*T, C) f32 logits, x (B*T, ldx) f32, pooled (B, 2C). */
int vp_utt_sums_f32(vp_ctx* ctx, const float* a, int lda, int B, int T, int C, float* out, vp_stream stream);
int vp_time_stats_f32(vp_ctx* ctx, const float* x, int ldx, int B, int T, int C, float eps, int unbiased /* 1: TSTP, sqrt(var_unbiased + eps) */,
                      float* stats, vp_stream stream);
/* vp_time_stats_bwd_add_f32: the same gradient plus `add` (the other consumers' gradients of x; may alias dx), one pass. */
int vp_time_stats_bwd_add_f32(vp_ctx* ctx, const float* x, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                              int unbiased, const float* add, int ldadd, float* dx, int lddx, vp_stream stream);
int vp_time_stats_bwd_f32(vp_ctx* ctx, const float* x, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                          int unbiased, float* dx, int lddx, vp_stream stream);
int vp_attn_stats_bwd_f32(vp_ctx* ctx, const float* e, const float* x, int ldx, const float* pooled, const float* dpooled, int B, int T,
                          int C, float eps, float* de, float* dx, int lddx, vp_stream stream);
/* vp_attn_stats_bwd_de16: the same with d e written as bf16 (mixed precision: the logits conv's data- and weight-gradient GEMMs read it
 * through vp_conv1d_fwd bf16 -> f32 and vp_conv1d_wgrad_bf16_oik). */
int vp_attn_stats_bwd_de16(vp_ctx* ctx, const float* e, const float* x, int ldx, const float* pooled, const float* dpooled, int B, int T,
                           int C, float eps, void* de, float* dx, int lddx, vp_stream stream);
/* The same pair with the LOGITS stored as bf16 (enable_amp: the logits conv writes bf16, as Paddle's O1 conv does; pooling.py:114-123 then
 * takes the softmax in f32).  T <= 320, else VP_EUNSUP and the caller keeps f32 logits. */
int vp_asp_softmax_stats_l16(vp_ctx* ctx, const void* logits_bf16, const void* x, int x_dtype, int ldx, int xoff, int B, int T, int C, float eps,
                             float* pooled, vp_stream stream);
int vp_attn_stats_bwd_e16(vp_ctx* ctx, const void* e_bf16, const void* x, int x_dtype, int ldx, const float* pooled, const float* dpooled, int B,
                          int T, int C, float eps, void* de_bf16, float* dx, int lddx, vp_stream stream);
/* x (the pooled tensor) stored as bf16 by its producer (x_dtype = VP_BF16 above); the pieces around it: the BatchNorm apply pass that writes
 * it (z bf16 -> y bf16), the context statistics' backward reading it, per-utterance sums of a bf16 gradient (d rowbias). */
int vp_affine_rows_b16_b16(vp_ctx* ctx, const void* z, int ldz, const float* scale, const float* shift, long long M, int C, void* y,
                           int ldy, int relu, vp_stream stream);
int vp_affine_rows_f32_b16(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, void* y,
                           int ldy, int relu, vp_stream stream);
int vp_time_stats_bwd_add_x16(vp_ctx* ctx, const void* x_bf16, int ldx, const float* stats, const float* dstats, int B, int T, int C, float eps,
                              int unbiased, const float* add, int ldadd, float* dx, int lddx, vp_stream stream);
int vp_utt_sums_b16(vp_ctx* ctx, const void* a_bf16, int lda, int B, int T, int C, float* out, vp_stream stream);
int vp_utt_dot_x16(vp_ctx* ctx, const float* dy, const void* x_bf16, int B, int T, int C, float* ds, vp_stream stream);   /* vp_utt_dot_f32 over a bf16 x */
int vp_act_f32(vp_ctx* ctx, int act /* VP_ACT_RELU .. VP_ACT_SILU; the backward takes the OUTPUT y, except SiLU: the input */, const float* x, long long n, float* y, vp_stream stream);
int vp_act_bwd_f32(vp_ctx* ctx, int act, const float* dy, const float* y, long long n, float* dz, vp_stream stream);
/* vp_reflect_fold_f32: adjoint of the reflect padding of Conv1d (models/utils.py:89-91): dxp (B, T + 2 pad, C) -> dx (B, T, C).
 * vp_scale_rows_bwd_f32: backward of the SE gate x * s (ecapa_tdnn.py:82): dx = dy * s, ds[b] = sum_t dy * x. */
/* vp_zero_insert_2d_f32: dz (B, T_out, F_out, C) -> up (B, T_in, F_in, C), zeros between the samples: the data gradient of a
 * stride-s 2-D conv is then a stride-1 vp_conv1d_fwd over `up`.  vp_relu_bwd_f32: dz = [y > 0] dy. */
int vp_zero_insert_2d_f32(vp_ctx* ctx, const float* dz, int B, int T_out, int F_out, int C, int T_in, int F_in, int stride_t, int stride_f,
                          float* up, vp_stream stream);
int vp_relu_bwd_f32(vp_ctx* ctx, const float* dy, const float* y, long long n, float* dz, vp_stream stream);
/* CAM++ context gate in training (models/campplus.py:88-106): ctx (B, nseg, C) = mean_t x + per-100-frame-segment mean;
 * the gate y * m[b, seg(t)]; and their backward.  x, y (B*T, C) dense f32; nseg = ceil(T / seg_len). */
int vp_seg_ctx_f32(vp_ctx* ctx, const float* x, int B, int T, int C, int seg_len, float* out, vp_stream stream);
int vp_seg_ctx_bwd_f32(vp_ctx* ctx, const float* dctx, int B, int T, int C, int seg_len, float* dx, vp_stream stream);
int vp_seg_scale_f32(vp_ctx* ctx, const float* y, const float* m, int B, int T, int C, int seg_len, float* out, vp_stream stream);
int vp_seg_scale_bwd_f32(vp_ctx* ctx, const float* g, const float* y, const float* m, int B, int T, int C, int seg_len, float* dy, float* dm,
                         vp_stream stream);
/* The whole gate of a CAMLayer behind its local conv as ONE launch forward and TWO backward, one workgroup per utterance
 * (CAMLayer.forward, models/campplus.py:88-95: context = x.mean(-1) + seg_pooling(x); m = sigmoid(linear2(relu(linear1(context))));
 * return linear_local(x) * m).  h (B*T, C) = the layer's input, y (B*T, O) = linear_local(h) from the conv GEMM, W1 (H, C), W2 (O, H):
 * vp_cam_gate_fwd_f32  : out = y * m[b, seg(t)];  ctx (B*nseg, C), hid (B*nseg, H), m (B*nseg, O) kept for backward.
 * vp_cam_gate_bwd_f32  : from g = d out: dy = g * m, dpre2 / dpre1 = the gradients at the two dense layers' pre-activations, dh = the part
 *                        of d h that arrives through the context (the caller adds it in the local conv's data-gradient epilogue),
 *                        dyb (B, O) = per-utterance column sums of dy (the local conv's bias gradient, summed by the next call).
 * vp_cam_gate_wgrad_f32: dW1 = dpre1^T ctx, db1, dW2 = dpre2^T hid, db2, dbl[o] = sum_b dyb[b][o] (dbl / dyb may be NULL), rows summed in order.
 * C, O % 4 == 0, C <= 1024, O <= 128, 16-byte aligned rows, one utterance's context within 128 KB of LDS, else VP_EUNSUP (the caller keeps the
 * per-op path: vp_seg_ctx_f32 -> two vp_conv1d_fwd -> vp_seg_scale_f32). */
int vp_cam_gate_fwd_f32(vp_ctx* ctx, const float* h, int ldh, const float* y, int ldy, const float* w1, const float* b1, const float* w2,
                        const float* b2, int B, int T, int C, int H, int O, int seg_len, float* ctx_out, float* hid_out, float* m_out,
                        float* out, int ldo, vp_stream stream);
int vp_cam_gate_bwd_f32(vp_ctx* ctx, const float* g, int ldg, const float* y, int ldy, const float* hid, const float* m, const float* w1,
                        const float* w2, int B, int T, int C, int H, int O, int seg_len, float* dy, int lddy, float* dpre1, float* dpre2,
                        float* dh, int lddh, float* dyb, vp_stream stream);
int vp_cam_gate_wgrad_f32(vp_ctx* ctx, const float* dpre1, const float* dpre2, const float* ctx_in, const float* hid, const float* dyb, int B,
                          int nseg, int C, int H, int O, float* dw1, float* db1, float* dw2, float* db2, float* dbl, vp_stream stream);
/* AFF output (eres2net.py:48-51) and its backward: o = x (1 + t) + y (1 - t) on dense (rows, C) f32 tensors. */
int vp_aff_combine_f32(vp_ctx* ctx, const float* t, const float* x, const float* y, long long rows, int C, float* out, vp_stream stream);
int vp_aff_combine_bwd_f32(vp_ctx* ctx, const float* g, const float* t, const float* x, const float* y, long long n, float* dx, float* dy,
                           float* dt, vp_stream stream);
int vp_reflect_fold_f32(vp_ctx* ctx, const float* dxp, int B, int T, int pad, int C, float* dx, vp_stream stream);
/* Res2NetBlock as one tape entry (ecapa_tdnn.py:11-47; train/functions.py Res2Fn):
 * vp_affine_rows_aux_f32: y = z * scale + shift into a channel slice (ldy) of the concatenated output, and (when aux) aux = y + add
 *   -- the next chunk's input y_i + x_{i+1} -- from the same pass.
 * vp_reflect_fold_into_f32: vp_reflect_fold_f32 written into a channel slice of d x (rows lddx apart) and, when add is given,
 *   sum = folded + add (dense): the previous chunk's output gradient. */
int vp_affine_rows_aux_f32(vp_ctx* ctx, const float* z, int ldz, const float* scale, const float* shift, long long M, int C, float* y, int ldy,
                           const float* add, int ld_add, float* aux, int ld_aux, vp_stream stream);
int vp_reflect_fold_into_f32(vp_ctx* ctx, const float* dxp, int B, int T, int pad, int C, float* dx, int lddx, const float* add, int ld_add,
                             float* sum, vp_stream stream);
/* Res2NetBlock in training mode under enable_amp, ONE launch per direction (csrc/res2_train.hip; ecapa_tdnn.py:11-47 forward, the
 * autograd paddle derives from it backward; trainer.py:209-244): a workgroup per utterance walks the scale - 1 chunk convs
 * (k = 3, dilation `dil`, reflect 'same' padding, 64 -> 64 channels) with the batch statistics of every chunk's BatchNorm exchanged
 * through an in-kernel grid barrier -- needs B <= the device's CU count, T <= 368, 2 dil + 2 <= T, width == 64, C == scale * 64;
 * otherwise VP_EUNSUP and callers run the per-chunk entry points (vp_conv1d_fwd / vp_bn_train_finalize / ...).
 * Forward:  x (B*T, C) -> out (B*T, C); saves z (scale-1, B*T, 64) f32 = ReLU(conv + bias), inb (scale-1, B*T, 64) bf16 = each conv's input
 *           as the matrix cores read it, stats (scale-1, 2, 64) = batch mean and 1 / sqrt(var + eps); run_mean / run_var updated in place.
 * Backward: x = d out, out = d x; reads z, stats, w, gamma; writes dzb (scale-1, B*T, 64) bf16 = d(conv output) -- the operand of the
 *           weight gradients, which the caller takes with vp_conv1d_wgrad_bf16_oik(x = inb[j], dz = dzb[j]) -- and
 *           dvec (scale-1, 3, 64) = d bias, d gamma, d beta of every chunk. */
typedef struct {
    int B, T, C, scale, width, dil;
    float momentum, eps;
    const float* x;
    float* out;
    const float* w[7];            /* (64, 64, 3) f32 each: the model's Conv1D weights */
    const float* bias[7];
    const float* gamma[7];
    const float* beta[7];
    float* run_mean[7];           /* may be NULL */
    float* run_var[7];
    float* z;
    void* inb;
    void* dzb;
    float* stats;
    float* dvec;
    void* out_bf16;               /* forward, optional: out once more as bf16 (B*T, C), the operand of the conv behind the block;
                                     with it, `out` may be NULL (no f32 copy is written) */
    int x_is_bf16;                /* forward: x points to bf16 (the producer wrote the block input as bf16 only) */
} vp_res2_train_desc;
/* nbatch convs of identical geometry in one launch (the chunk convs of a Res2Net block after vp_res2_train_bwd): conv c reads
 * x + c * x_bstride and dz + c * dz_bstride (bf16 elements), writes dW + c * Cout * Cin * KW; ws = nbatch x vp_conv1d_wgrad_workspace_bytes(d). */
int vp_conv1d_wgrad_bf16_oik_batched(vp_ctx* ctx, const vp_conv1d_desc* d, const void* dz, int lddz, float* dW, int nbatch, long long x_bstride,
                                     long long dz_bstride, void* ws, size_t ws_bytes, vp_stream stream);
/* SEBlock's two dense layers in training (ecapa_tdnn.py:50-82 with lengths=None; csrc/se_train.hip): mean (B, C) -> a = ReLU(W1 mean + b1) (B, H)
 * -> s = sigmoid(W2 a + b2) (B, C), W1 (H, C) and W2 (C, H) as the model's Conv1D weights; one launch forward, two backward (d mean and the
 * four parameter gradients from d s).  round_bf16 = enable_amp (operands rounded where the matrix cores round them).  C, H <= 1024. */
int vp_se_dense_train_fwd(vp_ctx* ctx, const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, int B, int C,
                          int H, int round_bf16, float* a, float* s, vp_stream stream);
size_t vp_se_dense_train_bwd_workspace_bytes(int B, int C, int H);
int vp_se_dense_train_bwd(vp_ctx* ctx, const float* ds, const float* mean, const float* a, const float* s, const float* w1, const float* w2,
                          int B, int C, int H, int round_bf16, float* dmean, float* dw1, float* db1, float* dw2, float* db2, void* ws,
                          size_t ws_bytes, vp_stream stream);
size_t vp_res2_train_workspace_bytes(int B, int scale);
int vp_res2_train_fwd(vp_ctx* ctx, const vp_res2_train_desc* d, void* ws, size_t ws_bytes, vp_stream stream);
int vp_res2_train_bwd(vp_ctx* ctx, const vp_res2_train_desc* d, void* ws, size_t ws_bytes, vp_stream stream);
/* 0 unless a grid barrier of this context gave up waiting since the last reset (it never does on a healthy launch; tests assert it).
   Host-synchronising 4-byte read.  While the word is set the optimiser entry points (vp_adam_step_f32 ...) leave the parameters
   untouched and vp_res2_train_fwd / vp_bn_train_finalize leave the BatchNorm running statistics untouched: a step computed from incomplete statistics
   never reaches the weights (reference: trainer.py:206-274 has no such failure mode -- this guards OUR fused kernels). */
int vp_grid_barrier_status(vp_ctx* ctx);
/* Clears the barrier words (after a bail-out, once the caller has switched to the per-chunk kernels). */
int vp_grid_barrier_reset(vp_ctx* ctx, vp_stream stream);
/* The 512 barrier words (2048 bytes, 256-byte aligned, zeroed) may live in CALLER-owned device memory: word 257 is the bail-out flag
 * every persistent-state writer tests (the optimiser entry points, vp_bn_train_finalize's running statistics, vp_res2_train_fwd's).
 * The data-parallel step (ppvector/train/step.py; reference: fleet.distributed_model keeps replicas identical, trainer.py:316-320)
 * owns them as a tensor so that it can MAX-all-reduce the flag with the gradients -- every rank drops the same steps -- and read it
 * back every step with an asynchronous 4-byte copy.  NULL returns to the context's own words.  Call before any launch is captured
 * into a graph (captured launches keep the pointer they saw). */
int vp_set_grid_barrier_words(vp_ctx* ctx, void* words);
/* CUs the grid-barrier kernels must leave free for kernels of OTHER queues (the collective running beside a data-parallel step,
 * trainer.py:316-320): a launch of more than (#CUs - n_cus) workgroups returns VP_EUNSUP (callers run the per-chunk entry points).
 * Independently of the reserve, a launch is refused when the runtime's occupancy query says the grid cannot be resident at once. */
int vp_set_grid_reserve_cus(vp_ctx* ctx, int n_cus);
/* Diagnostic: n_workgroups workgroups that each hold a CU slot and lds_bytes of LDS for ~usec microseconds on `stream` -- the
 * stand-in for a persistent kernel of another queue in the co-residency tests of the grid-barrier kernels (no reference counterpart). */
int vp_occupy_cus(vp_ctx* ctx, int n_workgroups, int lds_bytes, int usec, vp_stream stream);
/* The SE block's backward as two passes (SEBlock + residual, ecapa_tdnn.py:50-82, 139-141):
 * vp_utt_dot_f32: ds[b][c] = sum_t dy[b,t,c] * x[b,t,c];  vp_scale_shift_rows_f32: dx[b,t,c] = dy[b,t,c] * s[b][c] + dm[b][c] / T
 * (dm = the gradient that reached the squeeze mean through the two dense layers).  C % 4 == 0, contiguous (B*T, C) tensors. */
int vp_utt_dot_f32(vp_ctx* ctx, const float* dy, const float* x, int B, int T, int C, float* ds, vp_stream stream);
/* The conv -> ReLU -> BatchNorm -> SEBlock -> + residual tail of an SE-Res2 block (ecapa_tdnn.py:125-142) under enable_amp WITHOUT the two
 * tensors between the conv and the SE gate: h = BN(z) is never stored forward (vp_se_scale_residual_z16: out = bf16(z * bn_scale + bn_shift)
 * * s[b] + res over the conv's bf16 pre-BatchNorm output z), and the SE block's input gradient dh = dout * s[b] + dmean[b] / T is never stored
 * backward: vp_utt_dot_z16 (ds[b][c] = sum_t dout * h), then the BatchNorm-backward reductions and the BatchNorm + ReLU backward read dout
 * and form dh on the fly (vp_col_sums_f32_b16_utt / vp_bn_relu_bwd_dbias_b16_utt: d y = dy * utt_scale[b] + utt_shift[b] / T, b = row / T).
 * Bit-identical to the passes they replace (vp_affine_rows_b16_b16, vp_utt_dot_x16, vp_scale_shift_rows_f32 + the plain reductions). */
int vp_se_scale_residual_z16(vp_ctx* ctx, const void* z, int ldz, const float* bn_scale, const float* bn_shift, const float* s, const void* res,
                             int ldr, int roff, void* out, int ldo, int ooff, int B, int T, int C, vp_stream stream);
int vp_utt_dot_z16(vp_ctx* ctx, const float* dy, const void* z_bf16, const float* bn_scale, const float* bn_shift, int B, int T, int C,
                   float* ds, vp_stream stream);
int vp_col_sums_f32_b16_utt(vp_ctx* ctx, const float* a, int lda, const float* utt_scale, const float* utt_shift, int T, const void* b, int ldb,
                            const float* bmean, const float* bscale, long long M, int C, float* sums, void* ws, size_t ws_bytes, vp_stream stream);
/* The context-statistics gradient of a pooling layer folded into the BatchNorm-backward passes of the TDNNBlock that feeds it (MFA -> ASP,
 * ecapa_tdnn.py:262-267 + pooling.py:97-104): the gradient reaching y = BN(z) through [mean_t y | std_t y] is alpha[b][c] + beta[b][c] * y
 * (vp_time_stats_bwd_coeffs: ab = [alpha | beta], each (B, C)); vp_col_sums_f32_b16_ctx / vp_bn_relu_bwd_dbias_b16_ctx add it to their d y
 * on the fly with y = bf16(z * bn_scale + bn_shift) re-formed from the bf16 z they read anyway -- the vp_time_stats_bwd_add_x16 pass
 * (read x, read + write d x: 1.17 GB at 256 x 298 x 1536) never runs. */
int vp_time_stats_bwd_coeffs(vp_ctx* ctx, const float* stats, const float* dstats, int B, int T, int C, float eps, float* ab, vp_stream stream);
int vp_col_sums_f32_b16_ctx(vp_ctx* ctx, const float* a, int lda, const float* alpha, const float* beta, int T, const float* bn_scale,
                            const float* bn_shift, const void* b, int ldb, const float* bmean, const float* bscale, long long M, int C, float* sums,
                            void* ws, size_t ws_bytes, vp_stream stream);
int vp_bn_relu_bwd_dbias_b16_ctx(vp_ctx* ctx, const float* dy, int lddy, const float* alpha, const float* beta, int T, const float* bn_scale,
                                 const float* bn_shift, const void* z, int ldz, const float* mean, const float* invstd, const float* gamma,
                                 const float* sums, long long M, int C, int relu_mask, void* dz, int lddz, float* dbias, void* ws,
                                 size_t ws_bytes, vp_stream stream);
int vp_bn_relu_bwd_dbias_b16_utt(vp_ctx* ctx, const float* dy, int lddy, const float* utt_scale, const float* utt_shift, int T, const void* z,
                                 int ldz, const float* mean, const float* invstd, const float* gamma, const float* sums, long long M, int C,
                                 int relu_mask, void* dz, int lddz, float* dbias, void* ws, size_t ws_bytes, vp_stream stream);
int vp_scale_shift_rows_f32(vp_ctx* ctx, const float* dy, const float* s, const float* dm, int B, int T, int C, float* dx, vp_stream stream);
/* [mean | std] over the frames of f32 (B, T, C) for MANY frames per utterance (the (B, T*F', C) feature maps of ResNetSE's SE squeeze,
 * resnet_se.py:60-66): the frames spread over ~2048 workgroups, chunk partials reduced in order.  Else VP_EUNSUP -> vp_time_stats_f32. */
size_t vp_time_stats_workspace_bytes(int B, int T, int C);
int vp_time_stats_ws_f32(vp_ctx* ctx, const float* x, int ldx, int B, int T, int C, float eps, int unbiased, float* stats, void* ws,
                         size_t ws_bytes, vp_stream stream);
/* BatchNorm -> ReLU units (Conv2D -> BatchNorm2D -> ReLU, resnet_se.py:72-74; eres2net.py; campplus.py FCM): the ReLU's backward folded into
 * the two BatchNorm-backward passes -- d y counts only where the unit's output z * mask_scale + mask_shift (mask_scale = gamma * invstd,
 * mask_shift = beta - mean * mask_scale: what vp_bn_train_finalize returned) was positive; no d(activation) tensor.  vp_col_sums_masked_f32:
 * sums[0] = sum m dy, sums[1] = sum m dy zhat;  vp_bn_relu_bwd_masked_f32: dz from them.  C % 4 == 0, 16-byte aligned (else VP_EUNSUP).
 * mask_hi > 0: the unit's activation is Hardtanh(0, mask_hi) (ERes2Net's ReLU = nn.Hardtanh(0, 20), eres2net.py:14-22): d y also counts only
 * where the output was BELOW mask_hi; mask_hi = 0: plain ReLU.  The forward's apply pass: vp_affine_rows_f32 with relu = 2 (clamp to [0, 20]). */
int vp_col_sums_masked_f32(vp_ctx* ctx, const float* a, int lda, const float* b, int ldb, const float* bmean, const float* bscale,
                           const float* mask_scale, const float* mask_shift, float mask_hi, long long M, int C, float* sums, void* ws,
                           size_t ws_bytes, vp_stream stream);
int vp_bn_relu_bwd_masked_f32(vp_ctx* ctx, const float* dy, int lddy, const float* z, int ldz, const float* mean, const float* invstd,
                              const float* gamma, const float* sums, const float* mask_scale, const float* mask_shift, float mask_hi,
                              long long M, int C, float* dz, int lddz, int accumulate /* dz += : a DenseNet block's shared gradient buffer */,
                              vp_stream stream);
int vp_scale_rows_bwd_f32(vp_ctx* ctx, const float* dy, const float* x, const float* s, int B, int T, int C, float* dx, float* ds,
                          vp_stream stream);
/* The same for utterances of many positions (the (B, T*F', C) feature maps of ResNetSE / ERes2Net, resnet_se.py:60-75): positions spread over
 * ~2048 workgroups, per-chunk partial sums reduced in fixed order.  C % 4 == 0, C <= 1024, 16-byte aligned tensors; else VP_EUNSUP. */
size_t vp_scale_rows_bwd_workspace_bytes(int B, int T, int C);
int vp_scale_rows_bwd_ws_f32(vp_ctx* ctx, const float* dy, const float* x, const float* s, int B, int T, int C, float* dx, float* ds, void* ws,
                             size_t ws_bytes, vp_stream stream);

/* Trial scoring -- replaces the per-trial sklearn cosine_similarity loop of
 * PPVectorTrainer.evaluate (trainer.py:416-423) and PPVectorPredictor.contrast (predict.py:282):
 * scores[i][j] = <a_i, b_j> / (|a_i| |b_j|). */
size_t vp_cosine_scores_workspace_bytes(int Na, int Nb, int D);
int vp_cosine_scores_f32(vp_ctx* ctx, const float* a, const float* b, int Na, int Nb, int D, float* scores,
                         void* ws, size_t ws_bytes, vp_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VPMI_H */
