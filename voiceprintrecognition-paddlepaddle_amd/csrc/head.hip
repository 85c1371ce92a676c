// Cosine classifier + AAM-softmax cross-entropy, and cosine trial scoring.
//
// Replaces SpeakerIdentification.forward 'Cosine' (ppvector/models/fc.py:41-53:
// logits = normalize(x, axis=1) @ normalize(W, axis=0), W [D, C]) and AAMLoss.forward
// (ppvector/loss/aamloss.py:28-47: sine = sqrt(1 - cos^2) -- no clamp, as the reference;
// phi = cos*cos_m - sine*sin_m; hard/easy margin select; one-hot mix; * scale;
// CrossEntropyLoss(label_smoothing), mean), and the scoring loops trainer.py:416-423 / predict.py:282.
// The (B, C) logits are produced in exact f32 on the f32 matrix cores and read once by the loss
// kernel (online log-sum-exp per row, fixed-order reductions).
#include "common.h"

#include <math.h>

namespace {

// 64 columns x 4 row-groups per workgroup: coalesced 256-B row reads, fixed-order 4-way reduction
__global__ __launch_bounds__(256) void col_inv_norm_kernel(const float* w, int D, int C, float eps, float* inv) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lc;
    float s = 0.f;
    if (c < C)
        for (int d = dg; d < D; d += 4) { const float v = w[(size_t)d * C + c]; s += v * v; }
    sm[dg][lc] = s;
    __syncthreads();
    if (dg == 0 && c < C) inv[c] = 1.f / fmaxf(sqrtf(sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc]), eps);
}

struct AamArgs {
    const float* logits; const long long* labels; float* row_loss;
    int B, C; float cos_m, sin_m, th, mmm, scale, ls; int easy;
};

__global__ __launch_bounds__(256) void aam_ce_rows_kernel(AamArgs a) {
    __shared__ float sm[3][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = a.logits + (size_t)b * a.C;
    const int y = (int)a.labels[b];
    float mx = -INFINITY, se = 0.f, so = 0.f;       // running max, sum exp(out - mx), sum out
    for (int c = tid; c < a.C; c += 256) {
        const float cs = row[c];
        float o = cs;
        if (c == y) {
            const float sine = sqrtf(1.f - cs * cs);
            const float phi = cs * a.cos_m - sine * a.sin_m;
            o = a.easy ? (cs > 0.f ? phi : cs) : (cs > a.th ? phi : cs - a.mmm);
        }
        o *= a.scale;
        so += o;
        if (o > mx) { se = se * expf(mx - o) + 1.f; mx = o; }
        else se += expf(o - mx);
    }
    // wave merge, then the 4 waves (fixed order)
    const float wmx = vp_wave_max(mx);
    se = vp_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
    so = vp_wave_sum(so);
    if (lane == 0) { sm[0][wv] = wmx; sm[1][wv] = se; sm[2][wv] = so; }
    __syncthreads();
    if (tid == 0) {
        float M = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
        float S = 0.f, O = 0.f;
        for (int w = 0; w < 4; ++w) {
            S += (sm[0][w] == -INFINITY) ? 0.f : sm[1][w] * expf(sm[0][w] - M);
            O += sm[2][w];
        }
        const float lse = M + logf(S);
        // target logit, recomputed by one thread (cheap, keeps the reduction single-pass)
        const float cs = row[y];
        const float sine = sqrtf(1.f - cs * cs);
        const float phi = cs * a.cos_m - sine * a.sin_m;
        float oy = a.easy ? (cs > 0.f ? phi : cs) : (cs > a.th ? phi : cs - a.mmm);
        oy *= a.scale;
        // -sum_c q_c log p_c,  q = (1 - ls) onehot + ls / C
        const float nll = lse - oy;
        const float smooth = lse - O / (float)a.C;
        a.row_loss[b] = (1.f - a.ls) * nll + a.ls * smooth;
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* v, int n, float* out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = vp_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sm[0] + sm[1] + sm[2] + sm[3]) / (float)n;
}

}  // namespace

extern "C" {

size_t vp_cosine_logits_workspace_bytes(int B, int D, int C) {
    (void)D;
    return vp_align_up((size_t)B * 4, 256) + vp_align_up((size_t)C * 4, 256);
}

int vp_cosine_logits_f32(vp_ctx* ctx, const float* emb, const float* W, int B, int D, int C, float* logits,
                         void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !logits || B <= 0 || D <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_logits: bad arguments");
    if (!ws || ws_bytes < vp_cosine_logits_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_logits: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* rinv = (float*)ws;
    float* cinv = (float*)((char*)ws + vp_align_up((size_t)B * 4, 256));
    int rc = vp_row_inv_norm(ctx, emb, B, D, D, 1e-12f, rinv, st);
    if (rc) return rc;
    hipLaunchKernelGGL(col_inv_norm_kernel, dim3((C + 63) / 64), dim3(256), 0, st, W, D, C, 1e-12f, cinv);
    VP_LAUNCH_CHECK(ctx, "col_inv_norm");
    return vp_dense_f32_ex(ctx, emb, D, W, /*w_is_kn=*/1, nullptr, rinv, cinv, B, C, D, VP_ACT_NONE, logits, C, st);
}

int vp_aam_ce_fwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, float margin, float scale,
                  float label_smoothing, int easy_margin, float* loss, float* row_loss, vp_stream stream) {
    if (!ctx || !logits || !labels || !loss || !row_loss || B <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "aam_ce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    AamArgs a;
    a.logits = logits; a.labels = (const long long*)labels; a.row_loss = row_loss; a.B = B; a.C = C;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.ls = label_smoothing; a.easy = easy_margin;
    hipLaunchKernelGGL(aam_ce_rows_kernel, dim3(B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "aam_ce_rows");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
    VP_LAUNCH_CHECK(ctx, "mean");
    return VP_OK;
}

size_t vp_cosine_aam_workspace_bytes(int B, int D, int C) {
    return vp_cosine_logits_workspace_bytes(B, D, C) + vp_align_up((size_t)B * C * 4, 256);
}

int vp_cosine_aam_ce_fwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D,
                         int C, float margin, float scale, float label_smoothing, int easy_margin,
                         float* loss, float* logits, float* row_loss, void* ws, size_t ws_bytes,
                         vp_stream stream) {
    if (!ctx || !ws || ws_bytes < vp_cosine_aam_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_aam: workspace too small");
    const size_t lw = vp_cosine_logits_workspace_bytes(B, D, C);
    float* lg = logits ? logits : (float*)((char*)ws + lw);
    int rc = vp_cosine_logits_f32(ctx, emb, W, B, D, C, lg, ws, lw, stream);
    if (rc) return rc;
    return vp_aam_ce_fwd(ctx, lg, labels, B, C, margin, scale, label_smoothing, easy_margin, loss, row_loss, stream);
}

size_t vp_cosine_scores_workspace_bytes(int Na, int Nb, int D) {
    (void)D;
    return vp_align_up((size_t)Na * 4, 256) + vp_align_up((size_t)Nb * 4, 256);
}

int vp_cosine_scores_f32(vp_ctx* ctx, const float* a, const float* b, int Na, int Nb, int D, float* scores,
                         void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !a || !b || !scores || Na <= 0 || Nb <= 0 || D <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_scores: bad arguments");
    if (!ws || ws_bytes < vp_cosine_scores_workspace_bytes(Na, Nb, D)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_scores: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ainv = (float*)ws;
    float* binv = (float*)((char*)ws + vp_align_up((size_t)Na * 4, 256));
    int rc = vp_row_inv_norm(ctx, a, Na, D, D, 0.f, ainv, st);
    if (rc) return rc;
    rc = vp_row_inv_norm(ctx, b, Nb, D, D, 0.f, binv, st);
    if (rc) return rc;
    return vp_dense_f32_ex(ctx, a, D, b, /*w_is_kn=*/0, nullptr, ainv, binv, Na, Nb, D, VP_ACT_NONE, scores, Nb, st);
}

}  // extern "C"
