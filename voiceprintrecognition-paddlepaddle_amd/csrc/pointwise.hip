// Pointwise (1x1) convolution with few channels over a very long position axis (bf16): y = act2(bn(act(W x + bias))) with
// K = Cin <= 128 and N = Cout <= 128 -- the 1x1 convs of the full-resolution stages of the 2-D backbones (SEBottleneck.conv1 / conv3 /
// downsample of ResNetSE stage 1, resnet_se.py:8-45: 1.5 M positions x 32..128 channels at 64 utterances).
//
// These are HBM-streaming problems (conv3 32 -> 128: 98 MB in, 390 MB out, 6 GFLOP), and the tiled conv GEMM spends its time around
// a single K-step: register-staged loads -> LDS -> fragments -> MFMA -> slab epilogue, the input read once per 64-wide N tile
// (228 us where the traffic takes 90).  Here the whole weight matrix lives in registers (N / 16 x K / 32 fragments), a wave
// streams 16 positions at a time -- the activation fragment IS a 16-byte global load (positions are the MFMA B operand: lane
// (g, n) needs 8 consecutive channels of position n) -- and the epilogue goes through a per-wave LDS tile so that every store is
// 16 bytes of a contiguous run.  One workgroup = one 128-row tile of the conv GEMM's partial-sum layout, so the fused
// per-utterance column sums (the SE squeeze) drop into the same psum array the SE gate kernel reads.
#include "common.h"

namespace {

constexpr int PW_THREADS = 256;
constexpr int PW_ROWS = 128;               // rows per workgroup = VP_CONV_BM

struct PwArgs {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const float* bias; const float* scale; const float* shift; float* psum;
    const bf16_t* res;                         // optional residual, added before act2 (rows as y)
    int ldx, xoff, ldy, yoff, ld_res, res_off, M, T, nseg, act, act2, Ntot;   // Ntot: all output channels (blockIdx.y picks a chunk of N)
    unsigned x_bytes;
};

template <int KS, int NT>                  // K = 32 KS, N = 16 NT
__global__ __launch_bounds__(PW_THREADS) void pointwise_kernel(const PwArgs a) {
    constexpr int K = 32 * KS, N = 16 * NT;
    constexpr int SROW = N * 2 + 16;                        // bytes per position of the staging tile
    __shared__ __attribute__((aligned(16))) char stage[4][16 * SROW];
    __shared__ float red[4][2][N];
    __shared__ __attribute__((aligned(16))) float par[3][N];       // bias | scale | shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * PW_ROWS;
    const int n_off = blockIdx.y * N;                        // this workgroup's output-channel chunk (the input is read once per chunk)
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, a.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // activation fragments of the wave's two 16-position tiles first (the only HBM latency of the kernel), then the weights
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t xr[2][KS];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int m = m0 + wv * 32 + tl * 16 + li;
        const unsigned base = m < a.M ? (unsigned)(((size_t)m * a.ldx + a.xoff) * 2) + (unsigned)g * 16u : OOB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xr[tl][ks] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, base != OOB ? base + ks * 64u : OOB, 0, 0);
    }
    for (int i = tid; i < N; i += PW_THREADS) {
        par[0][i] = a.bias ? a.bias[n_off + i] : 0.f; par[1][i] = a.scale ? a.scale[n_off + i] : 1.f; par[2][i] = a.shift ? a.shift[n_off + i] : 0.f;
    }
    bf16x8 wf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(n_off + nt * 16 + li) * K + ks * 32 + g * 8);

    __syncthreads();
    // utterance boundary inside the workgroup's 128 rows (T >= 128, host-checked: at most one)
    const int bfirst = m0 / a.T;
    const int rb = min(PW_ROWS, (bfirst + 1) * a.T - m0);   // rows [0, rb) belong to utterance bfirst
    float s_lo[NT][4], s_hi[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_lo[nt][r] = 0.f; s_hi[nt][r] = 0.f; }

#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xf = __builtin_bit_cast(bf16x8, xr[tl][ks]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xf, acc[nt], 0, 0, 0);
        }
        const int row = wv * 32 + tl * 16 + li;             // of the workgroup's 128
        const bool live = m0 + row < a.M;
        char* st = stage[wv];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c0 = nt * 16 + g * 4;
            const float4 bb = *reinterpret_cast<const float4*>(&par[0][c0]);
            const float4 ss = *reinterpret_cast<const float4*>(&par[1][c0]);
            const float4 hh = *reinterpret_cast<const float4*>(&par[2][c0]);
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w}, sv[4] = {ss.x, ss.y, ss.z, ss.w}, hv[4] = {hh.x, hh.y, hh.z, hh.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.res && live) {
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(a.res + (size_t)(m0 + row) * a.ld_res + a.res_off + n_off + c0);
                rv[0] = (float)rr[0]; rv[1] = (float)rr[1]; rv[2] = (float)rr[2]; rv[3] = (float)rr[3];
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[nt][r] + bv[r];
                if (a.act == VP_ACT_RELU) t = fmaxf(t, 0.f);
                const float pre = t * sv[r];                                       // y - shift: what the column sums carry
                t = pre + hv[r] + rv[r];
                if (a.act2 == VP_ACT_RELU) t = fmaxf(t, 0.f);
                else if (a.act2 == VP_ACT_HARDTANH20) t = fminf(fmaxf(t, 0.f), 20.f);
                v[r] = t;
                if (a.psum) {
                    const float d = live ? t - hv[r] : 0.f;
                    if (row < rb) s_lo[nt][r] += d; else s_hi[nt][r] += d;
                }
            }
            bf16x4 o;
            o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
            *reinterpret_cast<bf16x4*>(st + li * SROW + c0 * 2) = o;
        }
        __builtin_amdgcn_wave_barrier();
        // 16 positions x N channels: N / 8 chunks of 16 B per position, 64 lanes per pass
        constexpr int CPP = N / 8;                          // chunks per position
#pragma unroll
        for (int i = lane; i < 16 * CPP; i += 64) {
            const int pos = i / CPP, q = i - pos * CPP;
            const int m = m0 + wv * 32 + tl * 16 + pos;
            if (m < a.M)
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldy + a.yoff + n_off + q * 8) = *reinterpret_cast<const uint4*>(st + pos * SROW + q * 16);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.psum) {
        // lanes of equal g hold disjoint positions of the same channels: reduce over li, then over the four waves in fixed order
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float lo = s_lo[nt][r], hi = s_hi[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { lo += __shfl_xor(lo, o); hi += __shfl_xor(hi, o); }
                if (li == 0) { red[wv][0][nt * 16 + g * 4 + r] = lo; red[wv][1][nt * 16 + g * 4 + r] = hi; }
            }
        __syncthreads();
        for (int i = tid; i < 2 * N; i += PW_THREADS) {
            const int sgi = i / N, c = i - sgi * N;
            // segment sgi exists iff some row of the tile belongs to utterance bfirst + sgi
            const bool has = sgi == 0 ? true : (rb < PW_ROWS && m0 + rb < a.M);
            if (has) a.psum[((size_t)blockIdx.x * a.nseg + sgi) * a.Ntot + n_off + c] = red[0][sgi][c] + red[1][sgi][c] + red[2][sgi][c] + red[3][sgi][c];
        }
    }
}

template <int KS, int NT>
void launch_pw(const PwArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((pointwise_kernel<KS, NT>), dim3((a.M + PW_ROWS - 1) / PW_ROWS, a.Ntot / (16 * NT)), dim3(PW_THREADS), 0, st, a);
}

}  // namespace

// Returns VP_EUNSUP when the shape is not covered (the caller falls back to vp_conv1d_fwd).  d: a 1x1, stride-1 conv descriptor
// (bf16 in / out); T_out = positions per utterance (psum segmentation).  only_where_it_wins: the backbones' dispatch rule;
// the test door passes 0 and reaches every shape the kernel covers.
int vp_pointwise_bf16(vp_ctx* ctx, const vp_conv1d_desc* d, int only_where_it_wins, hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("VPMI_PW_GENERAL"); off = e && atoi(e) ? 1 : 0; }
    if (off) return VP_EUNSUP;
    if (d->dtype_in != VP_BF16 || d->dtype_out != VP_BF16 || d->KW != 1 || d->stride != 1 || d->KF > 1 || d->F_in > 1 || d->F_out > 1)
        return VP_EUNSUP;
    if (d->rowbias || d->aux || d->add_in || d->gate || d->pro_scale || d->ysplit || d->psumsq) return VP_EUNSUP;
    if ((d->act != VP_ACT_NONE && d->act != VP_ACT_RELU) ||
        (d->act2 != VP_ACT_NONE && d->act2 != VP_ACT_RELU && d->act2 != VP_ACT_HARDTANH20)) return VP_EUNSUP;
    if (d->res && (d->ld_res % 4 || d->res_off % 4 || (reinterpret_cast<uintptr_t>(d->res) & 7))) return VP_EUNSUP;
    const int K = d->Cin, Nn = d->Cout;
    if (K != 32 && K != 64 && K != 128 && K != 256) return VP_EUNSUP;
    // measured (ResNetSE / ERes2Net, B = 64): the streaming kernel beats the tiled conv GEMM for K, N <= 128 with K N <= 8192 (the
    // full-resolution stages); with more channels the GEMM's operand reuse wins (4.23 -> 4.53 ms when everything was routed here)
    if (only_where_it_wins && (K > 128 || Nn > 128 || K * Nn > 8192)) return VP_EUNSUP;
    // output channels in chunks (grid.y) such that a chunk's weight fragments stay within 128 registers (64 beside the column sums)
    int cap = (d->psum ? 8192 : 16384) / K;
    if (cap > 128) cap = 128;
    int Nc = cap < Nn ? cap : Nn;
    Nc = Nc >= 128 ? 128 : (Nc >= 64 ? 64 : (Nc >= 32 ? 32 : 0));
    if (Nc == 0 || Nn % Nc || Nn > 2048) return VP_EUNSUP;
    if (d->ldx % 8 || d->xoff % 8 || d->ldy % 8 || d->yoff % 8) return VP_EUNSUP;
    if ((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->w)) & 15) return VP_EUNSUP;
    const long long M = (long long)d->B * d->T_out;
    if (M < 128 * 256) return VP_EUNSUP;                                    // small problems: the tiled kernel is fine
    if (d->psum && d->T_out < PW_ROWS) return VP_EUNSUP;                    // at most one utterance boundary per 128-row tile
    const unsigned long long xb = ((unsigned long long)M - 1) * d->ldx * 2 + (unsigned long long)(d->xoff + K) * 2;
    if (xb >= 0xffffff00ull || M > 0x7fffffffLL) return VP_EUNSUP;
    PwArgs a;
    a.x = (const bf16_t*)d->x; a.w = (const bf16_t*)d->w; a.y = (bf16_t*)d->y; a.bias = d->bias; a.scale = d->bn_scale; a.shift = d->bn_shift;
    a.res = (const bf16_t*)d->res; a.ld_res = d->ld_res; a.res_off = d->res_off;
    a.psum = d->psum; a.ldx = d->ldx; a.xoff = d->xoff; a.ldy = d->ldy; a.yoff = d->yoff; a.M = (int)M; a.T = d->T_out;
    a.nseg = vp_conv1d_nseg(d->T_out); a.act = d->act; a.act2 = d->act2; a.x_bytes = (unsigned)xb; a.Ntot = Nn;
    const int ks = K / 32, nt = Nc / 16;
    if (ks == 1 && nt == 2) launch_pw<1, 2>(a, st);
    else if (ks == 1 && nt == 4) launch_pw<1, 4>(a, st);
    else if (ks == 1 && nt == 8) launch_pw<1, 8>(a, st);
    else if (ks == 2 && nt == 2) launch_pw<2, 2>(a, st);
    else if (ks == 2 && nt == 4) launch_pw<2, 4>(a, st);
    else if (ks == 2 && nt == 8) launch_pw<2, 8>(a, st);
    else if (ks == 4 && nt == 2) launch_pw<4, 2>(a, st);
    else if (ks == 4 && nt == 4) launch_pw<4, 4>(a, st);
    else if (ks == 4 && nt == 8) launch_pw<4, 8>(a, st);
    else if (ks == 8 && nt == 2) launch_pw<8, 2>(a, st);
    else if (ks == 8 && nt == 4) launch_pw<8, 4>(a, st);
    else return VP_EUNSUP;
    VP_LAUNCH_CHECK(ctx, "pointwise");
    return VP_OK;
}
