// SEBlock's two dense layers in training (ecapa_tdnn.py:50-82, lengths=None; the autograd paddle derives for them, trainer.py:226-244):
//   a = ReLU(W1 mean + b1),  s = sigmoid(W2 a + b2)       mean (B, C), W1 (H, C), W2 (C, H): the Conv1D weights as the model stores them
// As 1x1 "convs" over B rows these were 12 launches per block and step (two GEMM-kernel launches forward; per layer an activation
// backward, a column sum, a weight-gradient GEMM + its partial-sum reduce and a data-gradient GEMM backward), each a single
// 128-wide tile walking K in dependent steps: ~20 us apiece for 0.1 MFLOP -- a quarter of the B = 32 step.  Here: one launch
// forward (a workgroup per utterance, a wave per output row, lanes over the reduction index), two backward (per-utterance chain
// d s -> d z2 -> d a -> d z1 -> d mean; then the four parameter gradients, a thread per weight, the batch as the reduction).
// round_bf16 = the enable_amp flavour: operands rounded to bf16 where the matrix-core path rounds them (products of two bf16 are
// exact in f32), f32 accumulation; 0 = plain f32 FMAs (the f32 engine: same values as the f32 MFMA chain up to summation order).
#include "common.h"

namespace {

constexpr int ST_THREADS = 1024;

__device__ __forceinline__ float st_rnd(float v, int round_bf16) { return round_bf16 ? (float)(bf16_t)v : v; }

struct SeTrainArgs {
    const float* mean; const float* w1; const float* b1; const float* w2; const float* b2;
    float* a; float* s;                 // forward outputs (backward inputs)
    const float* ds; float* dmean; float* dz1; float* dz2;
    float* dw1; float* db1; float* dw2; float* db2;
    int B, C, H, rnd;
};

// out[o] = sum_k W[o][k] x[k] for o < N, W output-major (rows of K floats), x in LDS.  LPO lanes share an output (each takes float4 columns
// j, j + LPO, ...), 1024 / LPO outputs per pass, every lane's loads of a pass issued before the first is used: the first version gave a
// wave one output at a time -- 32 dependent global round trips per wave for the 512 outputs of the second layer, 66 us per launch.
template <int LPO>
__device__ __forceinline__ void st_matvec(const float* __restrict__ W, int K, int N, const float* x, int rnd, float* out_s) {
    const int tid = threadIdx.x, j = tid % LPO, og = tid / LPO;
    constexpr int OPP = ST_THREADS / LPO;                      // outputs per pass
    const int kv = K >> 2;                                      // float4 columns (K % 4 == 0: host-checked)
    for (int o0 = 0; o0 < N; o0 += OPP) {
        const int o = o0 + og;
        float acc = 0.f;
        if (o < N) {
            const float4* w = reinterpret_cast<const float4*>(W + (size_t)o * K);
            for (int c0 = j; c0 < kv; c0 += 8 * LPO) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = c0 + u * LPO < kv ? w[c0 + u * LPO] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = 4 * min(c0 + u * LPO, kv - 1);
                    acc += st_rnd(v[u].x, rnd) * x[k] + st_rnd(v[u].y, rnd) * x[k + 1] + st_rnd(v[u].z, rnd) * x[k + 2] + st_rnd(v[u].w, rnd) * x[k + 3];
                }
            }
        }
#pragma unroll
        for (int m = 1; m < LPO; m <<= 1) acc += __shfl_xor(acc, m);
        if (j == 0 && o < N) out_s[o] = acc;
    }
}

__global__ __launch_bounds__(ST_THREADS) void se_dense_fwd_kernel(SeTrainArgs p) {
    extern __shared__ float sm[];        // mean[C] | a[H] | z[max(C, H)]
    float* mean_s = sm;
    float* a_s = sm + p.C;
    float* z_s = a_s + p.H;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < p.C; c += ST_THREADS) mean_s[c] = st_rnd(p.mean[(size_t)b * p.C + c], p.rnd);
    __syncthreads();
    st_matvec<16>(p.w1, p.C, p.H, mean_s, p.rnd, z_s);
    __syncthreads();
    for (int o = tid; o < p.H; o += ST_THREADS) {
        const float v = fmaxf(z_s[o] + p.b1[o], 0.f);
        p.a[(size_t)b * p.H + o] = v;
        a_s[o] = st_rnd(v, p.rnd);
    }
    __syncthreads();
    st_matvec<4>(p.w2, p.H, p.C, a_s, p.rnd, z_s);
    __syncthreads();
    for (int o = tid; o < p.C; o += ST_THREADS) p.s[(size_t)b * p.C + o] = 1.f / (1.f + __expf(-(z_s[o] + p.b2[o])));
}

// per utterance: d z2 = d s * s (1 - s);  d a = W2^T d z2;  d z1 = d a [a > 0];  d mean = W1^T d z1
__global__ __launch_bounds__(ST_THREADS) void se_dense_bwd_kernel(SeTrainArgs p) {
    extern __shared__ float sm[];        // dz2[C] | dz1[H] | part[1024] (groups x outputs <= the thread count)
    float* dz2_s = sm;
    float* dz1_s = sm + p.C;
    float* part = dz1_s + p.H;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < p.C; c += ST_THREADS) {
        const float s = p.s[(size_t)b * p.C + c];
        const float v = p.ds[(size_t)b * p.C + c] * s * (1.f - s);
        p.dz2[(size_t)b * p.C + c] = v;
        dz2_s[c] = st_rnd(v, p.rnd);
    }
    __syncthreads();
    {   // d a[h] = sum_c W2[c][h] dz2[c]: thread -> h (contiguous in W2's rows), the c range split over the thread groups
        const int groups = ST_THREADS / p.H;                    // H <= 1024 (host-checked), H | 1024 not required
        const int h = tid % p.H, q = tid / p.H;
        if (q < groups) {
            const int c0 = (int)((long long)p.C * q / groups), c1 = (int)((long long)p.C * (q + 1) / groups);
            float acc = 0.f;
#pragma unroll 8
            for (int c = c0; c < c1; ++c) acc += st_rnd(p.w2[(size_t)c * p.H + h], p.rnd) * dz2_s[c];
            part[q * p.H + h] = acc;
        }
        __syncthreads();
        if (tid < p.H) {
            float acc = 0.f;
            for (int g = 0; g < groups; ++g) acc += part[g * p.H + tid];
            const float v = p.a[(size_t)b * p.H + tid] > 0.f ? acc : 0.f;
            p.dz1[(size_t)b * p.H + tid] = v;
            dz1_s[tid] = st_rnd(v, p.rnd);
        }
        __syncthreads();
    }
    {   // d mean[c] = sum_h W1[h][c] dz1[h]
        const int groups = max(1, ST_THREADS / p.C);
        const int c = tid % p.C, q = tid / p.C;
        for (int c2 = c; c2 < p.C; c2 += ST_THREADS) {          // C > 1024 is host-rejected; this loop runs once
            if (q < groups) {
                const int h0 = (int)((long long)p.H * q / groups), h1 = (int)((long long)p.H * (q + 1) / groups);
                float acc = 0.f;
#pragma unroll 8
                for (int h = h0; h < h1; ++h) acc += st_rnd(p.w1[(size_t)h * p.C + c2], p.rnd) * dz1_s[h];
                part[q * p.C + c2] = acc;
            }
        }
        __syncthreads();
        if (tid < p.C) {
            float acc = 0.f;
            for (int g = 0; g < groups; ++g) acc += part[g * p.C + tid];
            p.dmean[(size_t)b * p.C + tid] = acc;
        }
    }
}

// parameter gradients: blockIdx.y = 0: d W2[c][h] = sum_b dz2[b][c] a[b][h], d b2[c];  1: d W1[h][c] = sum_b dz1[b][h] mean[b][c], d b1[h]
// A workgroup owns 256 / ncol rows... simply: thread -> one weight (row r, column j contiguous across lanes), the batch in order.
__global__ __launch_bounds__(256) void se_dense_wgrad_kernel(SeTrainArgs p) {
    const bool second = blockIdx.y == 0;
    const int nrow = second ? p.C : p.H, ncol = second ? p.H : p.C;
    const float* dz = second ? p.dz2 : p.dz1;                   // (B, nrow)
    const float* x = second ? p.a : p.mean;                     // (B, ncol)
    float* dw = second ? p.dw2 : p.dw1;
    float* db = second ? p.db2 : p.db1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)nrow * ncol) return;
    const int r = (int)(idx / ncol), j = (int)(idx - (long long)r * ncol);
    float acc = 0.f, bsum = 0.f;
#pragma unroll 4
    for (int b = 0; b < p.B; ++b) {
        const float g = dz[(size_t)b * nrow + r];
        acc += st_rnd(g, p.rnd) * st_rnd(x[(size_t)b * ncol + j], p.rnd);
        bsum += g;
    }
    dw[idx] = acc;
    if (j == 0) db[r] = bsum;
}

}  // namespace

extern "C" {

int vp_se_dense_train_fwd(vp_ctx* ctx, const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, int B, int C,
                          int H, int round_bf16, float* a, float* s, vp_stream stream) {
    if (!ctx || !mean || !w1 || !b1 || !w2 || !b2 || !a || !s || B <= 0) VP_FAIL(ctx, VP_EINVAL, "se_dense_train_fwd: bad arguments");
    if (C < 4 || H < 4 || C > 1024 || H > 1024 || (C | H) & 3) return VP_EUNSUP;
    SeTrainArgs p;
    memset(&p, 0, sizeof(p));
    p.mean = mean; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.a = a; p.s = s; p.B = B; p.C = C; p.H = H; p.rnd = round_bf16;
    hipLaunchKernelGGL(se_dense_fwd_kernel, dim3(B), dim3(ST_THREADS), (size_t)(C + H + (C > H ? C : H)) * sizeof(float), (hipStream_t)stream, p);
    VP_LAUNCH_CHECK(ctx, "se_dense_fwd");
    return VP_OK;
}

size_t vp_se_dense_train_bwd_workspace_bytes(int B, int C, int H) { return (size_t)B * (C + H) * sizeof(float) + 256; }

int vp_se_dense_train_bwd(vp_ctx* ctx, const float* ds, const float* mean, const float* a, const float* s, const float* w1, const float* w2,
                          int B, int C, int H, int round_bf16, float* dmean, float* dw1, float* db1, float* dw2, float* db2, void* ws,
                          size_t ws_bytes, vp_stream stream) {
    if (!ctx || !ds || !mean || !a || !s || !w1 || !w2 || !dmean || !dw1 || !db1 || !dw2 || !db2 || B <= 0)
        VP_FAIL(ctx, VP_EINVAL, "se_dense_train_bwd: bad arguments");
    if (C < 1 || H < 1 || C > 1024 || H > 1024) return VP_EUNSUP;
    if (!ws || ws_bytes < vp_se_dense_train_bwd_workspace_bytes(B, C, H)) VP_FAIL(ctx, VP_EWORKSPACE, "se_dense_train_bwd: workspace too small");
    SeTrainArgs p;
    memset(&p, 0, sizeof(p));
    p.mean = mean; p.w1 = w1; p.w2 = w2; p.a = const_cast<float*>(a); p.s = const_cast<float*>(s); p.ds = ds; p.dmean = dmean;
    p.dz2 = (float*)ws; p.dz1 = p.dz2 + (size_t)B * C;
    p.dw1 = dw1; p.db1 = db1; p.dw2 = dw2; p.db2 = db2; p.B = B; p.C = C; p.H = H; p.rnd = round_bf16;
    const size_t smem = (size_t)(C + H + ST_THREADS) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_dense_bwd_kernel, dim3(B), dim3(ST_THREADS), smem, st, p);
    VP_LAUNCH_CHECK(ctx, "se_dense_bwd");
    hipLaunchKernelGGL(se_dense_wgrad_kernel, dim3((unsigned)(((long long)C * H + 255) / 256), 2), dim3(256), 0, st, p);
    VP_LAUNCH_CHECK(ctx, "se_dense_wgrad");
    return VP_OK;
}

}  // extern "C"
