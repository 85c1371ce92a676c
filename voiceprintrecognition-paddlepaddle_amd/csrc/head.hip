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

// 32 columns x 8 row-groups per workgroup: coalesced 128-B row reads, eight independent loads in flight per thread (the
// 64 x 4 shape with a plain loop was a 48-deep dependent-latency chain on 44 workgroups: 21 us for a 2 MB matrix),
// fixed-order 8-way reduction
__global__ __launch_bounds__(256) void col_inv_norm_kernel(const float* w, int D, int C, float eps, float* inv) {
    __shared__ float sm[8][32];
    const int lc = threadIdx.x & 31, dg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + lc;
    const int cc = c < C ? c : C - 1;                              // clamped: loads stay unconditional
    float s = 0.f;
    for (int d0 = dg; d0 < D; d0 += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int d = d0 + 8 * u;
            v[u] = w[(size_t)(d < D ? d : 0) * C + cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (d0 + 8 * u < D) ? v[u] * v[u] : 0.f;
    }
    sm[dg][lc] = s;
    __syncthreads();
    if (dg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sm[r][lc];
        inv[c] = 1.f / fmaxf(sqrtf(t), eps);
    }
}

struct AamArgs {
    const float* logits; const long long* labels; float* row_loss;
    int B, C; float cos_m, sin_m, th, mmm, scale, ls; int easy;
    const float* mt;                 // device margin table (vp_set_margin_table) or NULL: the launch scalars above
};

__global__ __launch_bounds__(256) void aam_ce_rows_kernel(AamArgs a) {
    __shared__ float sm[3][4];
    if (a.mt) { a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
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

// ------------------------------------------------------------------------------------------------ backward
// d loss / d cos for one row, pre-multiplied by the column inverse norms (what both backward GEMMs consume):
//   out_c = scale * (c == y ? margin(cos_c) : cos_c);  p = softmax(out);  q = (1 - ls) onehot + ls / C
//   G[b][c] = gscale / B * (p_c - q_c) * scale * d margin / d cos * cinv[c]
struct AamBwdArgs {
    const float* logits; const long long* labels; const float* cinv; float* G; float* row_loss;
    int B, C; float cos_m, sin_m, th, mmm, scale, ls, gscale; int easy;
    const float* mt;
};

__global__ __launch_bounds__(256) void aam_ce_bwd_rows_kernel(AamBwdArgs a) {
    __shared__ float sm[3][4];
    __shared__ float s_lse;
    if (a.mt) { a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = a.logits + (size_t)b * a.C;
    const int y = (int)a.labels[b];
    auto out_of = [&](int c, float cs, float& dm) {
        float o = cs;
        dm = 1.f;
        if (c == y) {
            const float sine = sqrtf(1.f - cs * cs);
            const float phi = cs * a.cos_m - sine * a.sin_m;
            const bool use_phi = a.easy ? (cs > 0.f) : (cs > a.th);
            o = use_phi ? phi : (a.easy ? cs : cs - a.mmm);
            if (use_phi) dm = a.cos_m + cs * a.sin_m / sine;
        }
        return o * a.scale;
    };
    float mx = -INFINITY, se = 0.f, so = 0.f;
    for (int c = tid; c < a.C; c += 256) {
        float dm;
        const float o = out_of(c, row[c], dm);
        so += o;
        if (o > mx) { se = se * expf(mx - o) + 1.f; mx = o; }
        else se += expf(o - mx);
    }
    const float wmx = vp_wave_max(mx);
    se = vp_wave_sum(mx == -INFINITY ? 0.f : se * expf(mx - wmx));
    so = vp_wave_sum(so);
    if (lane == 0) { sm[0][wv] = wmx; sm[1][wv] = se; sm[2][wv] = so; }
    __syncthreads();
    if (tid == 0) {
        const float M = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
        float S = 0.f, O = 0.f;
        for (int w = 0; w < 4; ++w) {
            S += (sm[0][w] == -INFINITY) ? 0.f : sm[1][w] * expf(sm[0][w] - M);
            O += sm[2][w];
        }
        const float lse = M + logf(S);
        s_lse = lse;
        if (a.row_loss) {
            float dm;
            const float oy = out_of(y, row[y], dm);
            a.row_loss[b] = (1.f - a.ls) * (lse - oy) + a.ls * (lse - O / (float)a.C);
        }
    }
    __syncthreads();
    const float lse = s_lse;
    const float k = a.gscale / (float)a.B * a.scale;
    const float qoff = a.ls / (float)a.C;
    float* g = a.G + (size_t)b * a.C;
    for (int c = tid; c < a.C; c += 256) {
        float dm;
        const float o = out_of(c, row[c], dm);
        const float q = qoff + (c == y ? 1.f - a.ls : 0.f);
        g[c] = k * (expf(o - lse) - q) * dm * (a.cinv ? a.cinv[c] : 1.f);
    }
}

// g[b][c] = d[b][c] * cinv[c]
__global__ __launch_bounds__(256) void scale_cols_kernel(const float* d, const float* cinv, long long n, int C, float* g) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) g[i] = d[i] * cinv[i % C];
}

// xnT[d][b] = x[b][d] * rinv[b]
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float* x, const float* rinv, int B, int D, float* xnT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    const int d = i / B, b = i - d * B;
    xnT[i] = x[(size_t)b * D + d] * rinv[b];
}

// dx[b][:] = rinv * (dxn - x * rinv^2 * <x, dxn>)   (normalize backward, F.normalize(x, axis=1))
__global__ __launch_bounds__(256) void row_normalize_bwd_kernel(const float* x, const float* dxn, const float* rinv, int B, int D,
                                                                 float* dx) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* xr = x + (size_t)b * D;
    const float* gr = dxn + (size_t)b * D;
    float t = 0.f;
    for (int d = lane; d < D; d += 64) t += xr[d] * gr[d];
    t = vp_wave_sum(t);
    const float r = rinv[b];
    for (int d = lane; d < D; d += 64) dx[(size_t)b * D + d] = r * (gr[d] - xr[d] * r * r * t);
}

// dW[d][c] = dwn'[d][c] - cinv[c]^2 * W[d][c] * sum_d' W[d'][c] dwn'[d'][c]   (dwn' = xn^T (G cinv): see above)
__global__ __launch_bounds__(256) void col_normalize_bwd_kernel(const float* W, const float* dwn, const float* cinv, int D, int C,
                                                                 float* dW) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lc;
    float s = 0.f;
    if (c < C)
        for (int d = dg; d < D; d += 4) s += W[(size_t)d * C + c] * dwn[(size_t)d * C + c];
    sm[dg][lc] = s;
    __syncthreads();
    if (c < C) {
        const float t = (sm[0][lc] + sm[1][lc] + sm[2][lc] + sm[3][lc]) * cinv[c] * cinv[c];
        for (int d = dg; d < D; d += 4) dW[(size_t)d * C + c] = dwn[(size_t)d * C + c] - W[(size_t)d * C + c] * t;
    }
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
    hipLaunchKernelGGL(col_inv_norm_kernel, dim3((C + 31) / 32), dim3(256), 0, st, W, D, C, 1e-12f, cinv);
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
    a.scale = scale; a.ls = label_smoothing; a.easy = easy_margin; a.mt = ctx->margin_table;
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

size_t vp_cosine_aam_ce_bwd_workspace_bytes(int B, int D, int C) {
    return vp_cosine_logits_workspace_bytes(B, D, C) + 2 * vp_align_up((size_t)B * C * 4, 256) + vp_align_up((size_t)D * C * 4, 256) +
           2 * vp_align_up((size_t)B * D * 4, 256) + vp_align_up((size_t)B * 4, 256);
}

int vp_cosine_aam_ce_bwd(vp_ctx* ctx, const float* emb, const float* W, const int64_t* labels, int B, int D, int C, float margin,
                         float scale, float label_smoothing, int easy_margin, float grad_scale, float* demb, float* dW,
                         float* loss, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !labels || !demb || !dW || B <= 0 || D <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_aam_bwd: bad arguments");
    if (!ws || ws_bytes < vp_cosine_aam_ce_bwd_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_aam_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)ws;
    const size_t lw = vp_cosine_logits_workspace_bytes(B, D, C);
    float* rinv = (float*)p;
    float* cinv = (float*)(p + vp_align_up((size_t)B * 4, 256));
    p += lw;
    float* cosv = (float*)p; p += vp_align_up((size_t)B * C * 4, 256);
    float* G = (float*)p; p += vp_align_up((size_t)B * C * 4, 256);
    float* dwn = (float*)p; p += vp_align_up((size_t)D * C * 4, 256);
    float* xnT = (float*)p; p += vp_align_up((size_t)B * D * 4, 256);
    float* dxn = (float*)p; p += vp_align_up((size_t)B * D * 4, 256);
    float* row_loss = (float*)p;
    int rc = vp_cosine_logits_f32(ctx, emb, W, B, D, C, cosv, ws, lw, stream);
    if (rc) return rc;
    AamBwdArgs a;
    a.logits = cosv; a.labels = (const long long*)labels; a.cinv = cinv; a.G = G; a.row_loss = loss ? row_loss : nullptr;
    a.B = B; a.C = C;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.ls = label_smoothing; a.gscale = grad_scale; a.easy = easy_margin; a.mt = ctx->margin_table;
    hipLaunchKernelGGL(aam_ce_bwd_rows_kernel, dim3(B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "aam_ce_bwd_rows");
    if (loss) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
        VP_LAUNCH_CHECK(ctx, "mean");
    }
    // d xn = (G cinv) W^T;  d wn' = xn^T (G cinv)
    if ((rc = vp_dense_f32_ex(ctx, G, C, W, /*w_is_kn=*/0, nullptr, nullptr, nullptr, B, D, C, VP_ACT_NONE, dxn, D, st))) return rc;
    hipLaunchKernelGGL(transpose_scale_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, emb, rinv, B, D, xnT);
    VP_LAUNCH_CHECK(ctx, "transpose_scale");
    if ((rc = vp_dense_f32_ex(ctx, xnT, B, G, /*w_is_kn=*/1, nullptr, nullptr, nullptr, D, C, B, VP_ACT_NONE, dwn, C, st))) return rc;
    hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, emb, dxn, rinv, B, D, demb);
    VP_LAUNCH_CHECK(ctx, "row_normalize_bwd");
    hipLaunchKernelGGL(col_normalize_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, st, W, dwn, cinv, D, C, dW);
    VP_LAUNCH_CHECK(ctx, "col_normalize_bwd");
    return VP_OK;
}

// Backward of AAMLoss alone (aamloss.py:28-47): dlogits (B, C) = grad_scale * d loss / d cos.  loss (1) optional.
int vp_aam_ce_bwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, float margin, float scale,
                  float label_smoothing, int easy_margin, float grad_scale, float* dlogits, float* loss, float* row_loss,
                  vp_stream stream) {
    if (!ctx || !logits || !labels || !dlogits || B <= 0 || C <= 0 || (loss && !row_loss)) VP_FAIL(ctx, VP_EINVAL, "aam_ce_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    AamBwdArgs a;
    a.logits = logits; a.labels = (const long long*)labels; a.cinv = nullptr; a.G = dlogits; a.row_loss = loss ? row_loss : nullptr;
    a.B = B; a.C = C;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.scale = scale; a.ls = label_smoothing; a.gscale = grad_scale; a.easy = easy_margin; a.mt = ctx->margin_table;
    hipLaunchKernelGGL(aam_ce_bwd_rows_kernel, dim3(B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "aam_ce_bwd_rows");
    if (loss) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, B, loss);
        VP_LAUNCH_CHECK(ctx, "mean");
    }
    return VP_OK;
}

size_t vp_cosine_logits_bwd_workspace_bytes(int B, int D, int C) {
    return vp_cosine_logits_workspace_bytes(B, D, C) + vp_align_up((size_t)B * C * 4, 256) + vp_align_up((size_t)D * C * 4, 256) +
           2 * vp_align_up((size_t)B * D * 4, 256);
}

// Backward of the cosine classifier alone (fc.py:41-53): demb (B, D), dW (D, C) from dcos (B, C).
int vp_cosine_logits_bwd(vp_ctx* ctx, const float* emb, const float* W, const float* dcos, int B, int D, int C, float* demb,
                         float* dW, void* ws, size_t ws_bytes, vp_stream stream) {
    if (!ctx || !emb || !W || !dcos || !demb || !dW || B <= 0 || D <= 0 || C <= 0) VP_FAIL(ctx, VP_EINVAL, "cosine_logits_bwd: bad arguments");
    if (!ws || ws_bytes < vp_cosine_logits_bwd_workspace_bytes(B, D, C)) VP_FAIL(ctx, VP_EWORKSPACE, "cosine_logits_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)ws;
    float* rinv = (float*)p;
    float* cinv = (float*)(p + vp_align_up((size_t)B * 4, 256));
    p += vp_cosine_logits_workspace_bytes(B, D, C);
    float* G = (float*)p; p += vp_align_up((size_t)B * C * 4, 256);
    float* dwn = (float*)p; p += vp_align_up((size_t)D * C * 4, 256);
    float* xnT = (float*)p; p += vp_align_up((size_t)B * D * 4, 256);
    float* dxn = (float*)p;
    int rc = vp_row_inv_norm(ctx, emb, B, D, D, 1e-12f, rinv, st);
    if (rc) return rc;
    hipLaunchKernelGGL(col_inv_norm_kernel, dim3((C + 31) / 32), dim3(256), 0, st, W, D, C, 1e-12f, cinv);
    VP_LAUNCH_CHECK(ctx, "col_inv_norm");
    const long long n = (long long)B * C;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dcos, cinv, n, C, G);
    VP_LAUNCH_CHECK(ctx, "scale_cols");
    if ((rc = vp_dense_f32_ex(ctx, G, C, W, /*w_is_kn=*/0, nullptr, nullptr, nullptr, B, D, C, VP_ACT_NONE, dxn, D, st))) return rc;
    hipLaunchKernelGGL(transpose_scale_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, emb, rinv, B, D, xnT);
    VP_LAUNCH_CHECK(ctx, "transpose_scale");
    if ((rc = vp_dense_f32_ex(ctx, xnT, B, G, /*w_is_kn=*/1, nullptr, nullptr, nullptr, D, C, B, VP_ACT_NONE, dwn, C, st))) return rc;
    hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, emb, dxn, rinv, B, D, demb);
    VP_LAUNCH_CHECK(ctx, "row_normalize_bwd");
    hipLaunchKernelGGL(col_normalize_bwd_kernel, dim3((C + 63) / 64), dim3(256), 0, st, W, dwn, cinv, D, C, dW);
    VP_LAUNCH_CHECK(ctx, "col_normalize_bwd");
    return VP_OK;
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
