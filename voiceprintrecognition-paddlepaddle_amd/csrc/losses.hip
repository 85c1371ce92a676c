// The margin-softmax family next to AAMLoss, and SphereFace2, forward and backward over the (B, C) cosine logits.
//
// Replaces (the reference's loss package, every one consumed through build_loss -> criterion(outputs, labels),
// trainer.py:180,213):
//   AMLoss.forward        ppvector/loss/amloss.py:14-25         out = scale * (cos - m * onehot)
//   ARMLoss.forward       ppvector/loss/armloss.py:14-31        out = where(z - z[y] < 0, 0, z), z as AMLoss
//   CELoss.forward        ppvector/loss/celoss.py:11-19         out = logits
//   SubCenterLoss.forward ppvector/loss/subcenterloss.py:32-54  cos = max_k logits[c*K + k], then the AAM margin
//   AAMLoss.forward       ppvector/loss/aamloss.py:28-47        (kind VP_LOSS_AAM: same arithmetic as csrc/head.hip)
//   SphereFace2.forward   ppvector/loss/sphereface2.py:47-69    per-class binary losses, margin types 'A' / 'C'
// followed by CrossEntropyLoss(label_smoothing) with mean (or sum / B) reduction.  One workgroup per utterance walks its
// logits row twice: online log-sum-exp, then the gradient.  No one-hot / margin / prediction tensors exist (the reference
// builds three to six (B, C) temporaries and fills the margin with a Python loop over the batch).  Reductions are fixed-order.
#include "common.h"

#include <math.h>

namespace {

struct MarginArgs {
    const float* logits; const long long* labels; float* G; float* row_loss;
    int B, C, K, kind, easy;
    float cos_m, sin_m, th, mmm, margin, scale, ls, gscale;
    const float* mt;                 // device margin table (vp_set_margin_table) or NULL
};

// value of class c: its logit, or the max over its K sub-centres (first max wins; arg = winning sub-centre)
__device__ __forceinline__ float class_value(const float* row, int c, int K, int& arg) {
    float best = row[(size_t)c * K];
    arg = 0;
    for (int k = 1; k < K; ++k) {
        const float v = row[(size_t)c * K + k];
        if (v > best) { best = v; arg = k; }
    }
    return best;
}

// the logit that enters the softmax and its derivative w.r.t. the class value; zy = scaled target logit (ARM only)
__device__ __forceinline__ float margin_out(const MarginArgs& a, bool target, float v, float zy, float& dm) {
    switch (a.kind) {
    case VP_LOSS_AM: dm = a.scale; return a.scale * (target ? v - a.margin : v);
    case VP_LOSS_ARM: {
        const float z = a.scale * (target ? v - a.margin : v);
        if (z - zy < 0.f) { dm = 0.f; return 0.f; }
        dm = a.scale;
        return z;
    }
    case VP_LOSS_CE: dm = 1.f; return v;
    default: {                                                       // VP_LOSS_AAM, VP_LOSS_SUBCENTER
        float o = v;
        dm = a.scale;
        if (target) {
            const float sine = sqrtf(1.f - v * v);
            const float phi = v * a.cos_m - sine * a.sin_m;
            const bool use_phi = a.easy ? (v > 0.f) : (v > a.th);
            o = use_phi ? phi : (a.easy ? v : v - a.mmm);
            if (use_phi) dm = a.scale * (a.cos_m + v * a.sin_m / sine);
        }
        return o * a.scale;
    }
    }
}

__global__ __launch_bounds__(256) void margin_ce_rows_kernel(MarginArgs a) {
    __shared__ float sm[3][4];
    __shared__ float s_lse;
    if (a.mt) {
        a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4];
        if (a.kind == VP_LOSS_AM || a.kind == VP_LOSS_ARM) a.margin = a.mt[0];
    }
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = a.logits + (size_t)b * a.C * a.K;
    const int y = (int)a.labels[b];
    int arg;
    const float vy = class_value(row, y, a.K, arg);
    const float zy = a.scale * (vy - a.margin);
    float mx = -INFINITY, se = 0.f, so = 0.f;
    for (int c = tid; c < a.C; c += 256) {
        float dm;
        const float o = margin_out(a, c == y, class_value(row, c, a.K, arg), zy, dm);
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
            const float oy = margin_out(a, true, vy, zy, dm);
            a.row_loss[b] = (1.f - a.ls) * (lse - oy) + a.ls * (lse - O / (float)a.C);
        }
    }
    if (!a.G) return;
    __syncthreads();
    const float lse = s_lse;
    const float k = a.gscale / (float)a.B;
    const float qoff = a.ls / (float)a.C;
    float* g = a.G + (size_t)b * a.C * a.K;
    for (int c = tid; c < a.C; c += 256) {
        float dm;
        const float o = margin_out(a, c == y, class_value(row, c, a.K, arg), zy, dm);
        const float q = qoff + (c == y ? 1.f - a.ls : 0.f);
        const float gc = k * (expf(o - lse) - q) * dm;
        for (int kk = 0; kk < a.K; ++kk) g[(size_t)c * a.K + kk] = kk == arg ? gc : 0.f;
    }
}

struct SphereArgs {
    const float* logits; const long long* labels; const float* bias; float* G; float* row_loss; float* row_dbias;
    int B, C, t, type_a;
    float cos_m, sin_m, th, mmm, margin, scale, lam, gscale;
    const float* mt;
};

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// g(z) = 2 ((z + 1) / 2)^t - 1 and its derivative t ((z + 1) / 2)^(t - 1)
__device__ __forceinline__ float fun_g(float z, int t, float& dg) {
    const float h = (z + 1.f) * 0.5f;
    float p = 1.f;
    for (int i = 1; i < t; ++i) p *= h;
    dg = (float)t * p;
    return 2.f * p * h - 1.f;
}

// per-class loss term and d term / d cos, d term / d bias
__device__ __forceinline__ float sphere_term(const SphereArgs& a, bool target, float cs, float bias, float& dcos, float& dbias) {
    float z = cs, dz = 1.f;
    if (a.type_a) {
        const float sn = sqrtf(1.f - cs * cs);
        if (target) {
            if (cs > a.th) { z = cs * a.cos_m - sn * a.sin_m; dz = a.cos_m + cs * a.sin_m / sn; }
            else z = cs - a.mmm;
        } else { z = cs * a.cos_m + sn * a.sin_m; dz = a.cos_m - cs * a.sin_m / sn; }
    }
    float dg;
    float gz = fun_g(z, a.t, dg);
    if (!a.type_a) gz += target ? -a.margin : a.margin;
    const float x = a.scale * gz + bias;
    if (target) {
        dbias = -a.lam * sigmoidf(-x);
        dcos = dbias * a.scale * dg * dz;
        return a.lam * softplus(-x);
    }
    dbias = (1.f - a.lam) * sigmoidf(x);
    dcos = dbias * a.scale * dg * dz;
    return (1.f - a.lam) * softplus(x);
}

__global__ __launch_bounds__(256) void sphereface2_rows_kernel(SphereArgs a) {
    __shared__ float sm[2][4];
    if (a.mt) { a.margin = a.mt[0]; a.cos_m = a.mt[1]; a.sin_m = a.mt[2]; a.th = a.mt[3]; a.mmm = a.mt[4]; }
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* row = a.logits + (size_t)b * a.C;
    const int y = (int)a.labels[b];
    const float bias = a.bias ? a.bias[0] : 0.f;
    const float k = a.gscale / (float)a.B;
    float sl = 0.f, sb = 0.f;
    for (int c = tid; c < a.C; c += 256) {
        float dcos, dbias;
        sl += sphere_term(a, c == y, row[c], bias, dcos, dbias);
        sb += dbias;
        if (a.G) a.G[(size_t)b * a.C + c] = k * dcos;
    }
    sl = vp_wave_sum(sl);
    sb = vp_wave_sum(sb);
    if (lane == 0) { sm[0][wv] = sl; sm[1][wv] = sb; }
    __syncthreads();
    if (tid == 0) {
        if (a.row_loss) a.row_loss[b] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        if (a.row_dbias) a.row_dbias[b] = k * (sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]);
    }
}

// out[0] = (mean ? 1 / n : 1) * sum v, fixed order
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* v, int n, int mean, float* out) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = vp_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sm[0] + sm[1] + sm[2] + sm[3]) * (mean ? 1.f / (float)n : 1.f);
}

int launch_margin(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, int K, int kind, float margin, float scale,
                  float ls, int easy, float gscale, float* G, float* loss, float* row_loss, hipStream_t st) {
    MarginArgs a;
    a.logits = logits; a.labels = (const long long*)labels; a.G = G; a.row_loss = row_loss;
    a.B = B; a.C = C; a.K = K; a.kind = kind; a.easy = easy;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.margin = (kind == VP_LOSS_AM || kind == VP_LOSS_ARM) ? margin : 0.f;
    a.scale = kind == VP_LOSS_CE ? 1.f : scale;
    a.ls = ls; a.gscale = gscale; a.mt = ctx->margin_table;
    hipLaunchKernelGGL(margin_ce_rows_kernel, dim3(B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "margin_ce_rows");
    if (loss) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, B, 1, loss);
        VP_LAUNCH_CHECK(ctx, "reduce_rows");
    }
    return VP_OK;
}

bool margin_args_ok(const float* logits, const int64_t* labels, int B, int C, int K, int kind) {
    if (!logits || !labels || B <= 0 || C <= 0 || K <= 0) return false;
    if (kind < VP_LOSS_AAM || kind > VP_LOSS_SUBCENTER) return false;
    return kind == VP_LOSS_SUBCENTER || K == 1;
}

}  // namespace

extern "C" {

int vp_margin_ce_fwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, int K, int kind, float margin, float scale,
                     float label_smoothing, int easy_margin, float* loss, float* row_loss, vp_stream stream) {
    if (!ctx || !margin_args_ok(logits, labels, B, C, K, kind) || !loss || !row_loss) VP_FAIL(ctx, VP_EINVAL, "margin_ce_fwd: bad arguments");
    return launch_margin(ctx, logits, labels, B, C, K, kind, margin, scale, label_smoothing, easy_margin, 0.f, nullptr, loss, row_loss,
                         (hipStream_t)stream);
}

int vp_margin_ce_bwd(vp_ctx* ctx, const float* logits, const int64_t* labels, int B, int C, int K, int kind, float margin, float scale,
                     float label_smoothing, int easy_margin, float grad_scale, float* dlogits, float* loss, float* row_loss,
                     vp_stream stream) {
    if (!ctx || !margin_args_ok(logits, labels, B, C, K, kind) || !dlogits || (loss && !row_loss))
        VP_FAIL(ctx, VP_EINVAL, "margin_ce_bwd: bad arguments");
    return launch_margin(ctx, logits, labels, B, C, K, kind, margin, scale, label_smoothing, easy_margin, grad_scale, dlogits, loss,
                         loss ? row_loss : nullptr, (hipStream_t)stream);
}

int vp_sphereface2(vp_ctx* ctx, const float* logits, const int64_t* labels, const float* bias, int B, int C, float margin, float scale,
                   float lanbuda, int t, int margin_type_a, float grad_scale, float* loss, float* row_loss, float* dlogits, float* dbias,
                   float* row_dbias, vp_stream stream) {
    if (!ctx || !logits || !labels || B <= 0 || C <= 0 || t < 1 || !loss || !row_loss || (dbias && !row_dbias))
        VP_FAIL(ctx, VP_EINVAL, "sphereface2: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    SphereArgs a;
    a.logits = logits; a.labels = (const long long*)labels; a.bias = bias; a.G = dlogits; a.row_loss = row_loss;
    a.row_dbias = dbias ? row_dbias : nullptr;
    a.B = B; a.C = C; a.t = t; a.type_a = margin_type_a;
    a.cos_m = (float)cos((double)margin); a.sin_m = (float)sin((double)margin);
    a.th = (float)cos(M_PI - (double)margin); a.mmm = (float)(1.0 + cos(M_PI - (double)margin));
    a.margin = margin; a.scale = scale; a.lam = lanbuda; a.gscale = grad_scale; a.mt = ctx->margin_table;
    hipLaunchKernelGGL(sphereface2_rows_kernel, dim3(B), dim3(256), 0, st, a);
    VP_LAUNCH_CHECK(ctx, "sphereface2_rows");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, B, 1, loss);
    VP_LAUNCH_CHECK(ctx, "reduce_rows");
    if (dbias) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, st, row_dbias, B, 0, dbias);
        VP_LAUNCH_CHECK(ctx, "reduce_rows");
    }
    return VP_OK;
}

}  // extern "C"
