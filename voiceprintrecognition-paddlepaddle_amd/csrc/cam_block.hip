// One CAMDenseTDNNBlock of CAM++ as ONE kernel: a resident workgroup per utterance walks every layer of the block.
//
// Replaces CAMDenseTDNNBlock.forward (ppvector/models/campplus.py:145-173): for each CAMDenseTDNNLayer (:109-142)
//   h = relu(bn2(linear1(relu(bn1(x[:, :ch])))));  m = sigmoid(W2 relu(W1 (mean_t h + segmean h) + b1) + b2)  (CAMLayer :67-106)
//   x[:, ch : ch + 32] = linear_local(h) * m          (k3, dilation d, zero 'same' padding)
// As separate launches a layer is three kernels (1x1 GEMM with the BN-ReLU prologue, context gate, local conv) over
// (B*T', C) tensors: 156 launches per forward, each a full HBM round trip of h, 3.3 ms of a 5.4 ms forward at B = 256 and
// launch-bound at the 64 utterances per GPU of the data-parallel configuration.  An utterance's layer is small
// (T' = 149 frames x 128 bottleneck channels), so here the whole block runs out of one CU:
//   phase 1  1x1 GEMM (T' x 128 x ch): the 8 waves split the 128 bottleneck channels; the input streams through LDS in
//            64-channel chunks -- loaded once (16 B per lane, L1-bypassing: the columns were written by this workgroup a layer
//            ago), BN1 + ReLU applied in registers, stored bf16 with the XOR swizzle the fragment reads undo -- double-buffered,
//            one barrier per chunk; weight fragments go L2 -> registers, a chunk ahead.  Epilogue: bias, BN2, ReLU -> h in LDS
//            (bf16, 272-byte rows) + per-segment column sums for the context.
//   phase 2  context gate (two small matvecs per segment, f32, weights from L2).
//   phase 3  local conv as a GEMM (T' x 32 x 384) over h in LDS (taps = row shifts, a zero row for the padding), weights staged
//            in LDS; epilogue: bias, gate, 32 new bf16 channels appended to the utterance's rows of the concat buffer.
// h, the context and the gate never touch HBM; the concat buffer is read once per layer (that IS the DenseNet) and extended in place.
#include "common.h"

#include <type_traits>

namespace {

constexpr int CB_THREADS = 512;
constexpr int CB_WAVES = 8;
constexpr int CB_BNC = 128;                // bottleneck channels (= 8 waves x 16)
constexpr int CB_GR = 32;                  // growth
constexpr int CB_H = CB_BNC / 2;           // hidden width of the context gate
constexpr int CB_MT = 10;                  // frame tiles of 16: T' <= 160
constexpr int CB_TP = CB_MT * 16;
constexpr int CB_HROW = 2 * CB_BNC + 16;   // bytes per h row (272: conflict-free 16-byte fragment reads down a column of rows)
constexpr int CB_WLROW = 2 * 3 * CB_BNC + 16;   // bytes per local-conv weight row (784)
constexpr int CB_MAX_LAYERS = 24;
constexpr int CB_MAX_SEG = 4;
constexpr int CB_MAX_CH = 1024;

struct CamLayerP {
    const float* bn1_scale; const float* bn1_shift;
    const bf16_t* w1; const float* b1; const float* bn2_scale; const float* bn2_shift;
    const bf16_t* wl; const float* bl;
    const float* ctx_w1; const float* ctx_b1; const float* ctx_w2; const float* ctx_b2;
    int dil, pad_;
};

struct CamBlockArgs {
    bf16_t* cat;                 // (B*Tn, ld): columns [0, ch0) hold the block input; layer l appends columns [ch0 + 32 l, +32)
    int ld, ch0, Tn, seg_len, nseg, n_layers;
    unsigned cat_bytes;
#ifdef VP_TIMING
    unsigned long long* dbg;
#endif
    CamLayerP L[CB_MAX_LAYERS];
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(CB_THREADS, 1) void cam_block_kernel(const CamBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: xs[2][TP][128 B] | h[(TP + 1)][272 B] (last row = zeros) | wl[32][784 B] | bn1[2][CB_MAX_CH] f32 |
    //         csum[nseg + 1][128] f32 | hid[CB_MAX_SEG][64] f32 | gate[CB_MAX_SEG][32] f32 | part[512] f32
    char* xs = smem;
    char* hb = xs + 2 * CB_TP * 128;
    char* wls = hb + (CB_TP + 1) * CB_HROW;
    float* bn1 = reinterpret_cast<float*>(wls + CB_GR * CB_WLROW);
    float* csum = bn1 + 2 * CB_MAX_CH;
    float* hid = csum + (CB_MAX_SEG + 1) * CB_BNC;
    float* gate = hid + CB_MAX_SEG * CB_H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int Tn = a.Tn;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(a.cat, 0, a.cat_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    constexpr int SC1 = 16;                        // cache policy bit 4: agent scope -- the load is served by L2, not by this CU's L1

    // zero row of h (the local conv's padding) -- written once
    for (int i = tid; i < CB_HROW / 4; i += CB_THREADS) reinterpret_cast<float*>(hb + CB_TP * CB_HROW)[i] = 0.f;

    // this thread's items of a chunk: (frame t = i >> 3, 16-byte column group c8 = i & 7), i = tid + 512 q; c8 is the same for all q
    const int c8 = tid & 7;
    unsigned rowoff[3];
    int trow[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int t = (tid + CB_THREADS * q) >> 3;
        trow[q] = t;
        rowoff[q] = (t < Tn && t < CB_TP) ? (unsigned)(((size_t)b * Tn + t) * a.ld * 2) + (unsigned)c8 * 16u : OOB;
    }

#ifdef VP_TIMING
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tq = wall_clock64();
#define CB_STAMP(i) { const unsigned long long n_ = wall_clock64(); tacc[i] += n_ - tq; tq = n_; }
#else
#define CB_STAMP(i)
#endif
    int ch = a.ch0;
    for (int l = 0; l < a.n_layers; ++l, ch += CB_GR) {
        const CamLayerP& P = a.L[l];
        // ---- stage the layer's BN1 affine and local-conv weights (read after the first barriers below)
        for (int i = tid; i < ch; i += CB_THREADS) { bn1[i] = P.bn1_scale[i]; bn1[CB_MAX_CH + i] = P.bn1_shift[i]; }
        for (int i = tid; i < CB_GR * 48; i += CB_THREADS) {                        // 32 rows x 48 chunks of 16 B
            const int r = i / 48, c = i - r * 48;
            *reinterpret_cast<uint4*>(wls + r * CB_WLROW + c * 16) = *reinterpret_cast<const uint4*>(P.wl + (size_t)r * 384 + c * 8);
        }
        const int nchunk = (ch + 63) >> 6;
        // ---- phase 1: h = relu(bn2(W1 relu(bn1(x)) + b1))
        f32x4 acc[CB_MT];
#pragma unroll
        for (int mt = 0; mt < CB_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4_t xr[4][3];                                  // ring of four chunk slots: a chunk is fetched three iterations before it is staged
        bf16x8 wfn[2][2];                                  // [slot = chunk parity][k-step]: fetched two chunks ahead, BEFORE the x fetch of
                                                           // the same iteration (waiting for the newest load would wait for all of them)
        auto fetch_x = [&](int kc, auto slot) {
            constexpr int SL = decltype(slot)::value;
            const unsigned cb = (unsigned)kc * 128u;
            const bool cok = kc * 64 + c8 * 8 < ch;                                 // ch % 8 == 0 (host-checked): a group is all in or all out
#pragma unroll
            for (int q = 0; q < 3; ++q)
#ifdef CB_EXP_NOX
                xr[SL][q] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, OOB + (cok ? 0u : 1u) + cb * 0u, 0, SC1);
#else
                xr[SL][q] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, (cok && rowoff[q] != OOB) ? rowoff[q] + cb : OOB, 0, SC1);
#endif
        };
        // weight fragments straight from L2: rows 16 wv + li of W1, 8 k per lane; k past ch is an out-of-range offset (zeros) -- a
        // select on the loaded value would make hipcc wait for EVERY load in flight right behind the load
        const __amdgpu_buffer_rsrc_t w1srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P.w1), 0, (unsigned)(CB_BNC * ch * 2), 0x00020000);
        const unsigned w1row = (unsigned)((wv * 16 + li) * ch + g * 8) * 2u;
        auto fetch_w = [&](int kc, auto slot) {
            constexpr int SL = decltype(slot)::value & 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k = kc * 64 + ks * 32 + g * 8;
#ifdef CB_EXP_NOW
                wfn[SL][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1srd, OOB + (k < ch ? 0u : 1u), 0, 0));
#else
                wfn[SL][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w1srd, k < ch ? w1row + (unsigned)(kc * 64 + ks * 32) * 2u : OOB, 0, 0));
#endif
            }
        };
        // BN1 + ReLU on the fetched 16-byte groups, into chunk buffer `buf`
        auto stage_x = [&](int kc, int buf, auto slot) {
            constexpr int SL = decltype(slot)::value;
            const int cbase = kc * 64 + c8 * 8;
            float sc[8], sh[8];
            if (cbase < ch) {
                *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(bn1 + cbase);
                *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(bn1 + cbase + 4);
                *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(bn1 + CB_MAX_CH + cbase);
                *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(bn1 + CB_MAX_CH + cbase + 4);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int t = trow[q];
                if (t < CB_TP) {
                    const bf16x8 v = __builtin_bit_cast(bf16x8, xr[SL][q]);
                    bf16x8 o;                                  // rows past the utterance carry relu(shift): they only feed h rows the epilogue zeroes
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
                    *reinterpret_cast<bf16x8*>(xs + buf * (CB_TP * 128) + t * 128 + ((c8 ^ (t & 7)) << 4)) = o;
                }
            }
        };
        __syncthreads();                                   // bn1 / wl staged; the previous layer's stores have been waited for below
        CB_STAMP(0)
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
        fetch_w(0, S0{});
        fetch_x(0, S0{});
        fetch_w(1, S1{});
        fetch_x(1, S1{});
        fetch_x(2, S2{});
        stage_x(0, 0, S0{});
        __syncthreads();
        // iteration kc: fetch W chunk kc + 2 and x chunk kc + 3 (into the slots chunk kc leaves), MFMAs of chunk kc, stage chunk kc + 1, barrier
        auto step = [&](int kc, auto sfetch, auto sstage, auto scur) {
            constexpr int CUR = decltype(scur)::value & 1;
            const bool more = kc + 1 < nchunk;
            const bf16x8 wf[2] = {wfn[CUR][0], wfn[CUR][1]};
            fetch_w(kc + 2, scur);                         // past the last chunk: out-of-range offsets, zeros nobody uses
            fetch_x(kc + 3, sfetch);
            const char* xb = xs + (kc & 1) * (CB_TP * 128);
            // ten fragment reads, then their ten MFMAs (left to itself hipcc alternates read -> wait -> MFMA: one LDS latency each)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xf[CB_MT];
#pragma unroll
                for (int mt = 0; mt < CB_MT; ++mt) {
                    const int t = mt * 16 + li;
                    xf[mt] = *reinterpret_cast<const bf16x8*>(xb + t * 128 + (((ks * 4 + g) ^ (t & 7)) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < CB_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf[mt], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) stage_x(kc + 1, (kc + 1) & 1, sstage);
            __syncthreads();
        };
        for (int kc = 0; kc < nchunk; kc += 4) {
            step(kc, S3{}, S1{}, S0{});
            if (kc + 1 < nchunk) step(kc + 1, S0{}, S2{}, S1{});
            if (kc + 2 < nchunk) step(kc + 2, S1{}, S3{}, S0{});
            if (kc + 3 < nchunk) step(kc + 3, S2{}, S0{}, S1{});
        }
        CB_STAMP(1)
        // the context gate's weights (128 x 64 + 64 x 32 f32 = 40 KB = the two chunk buffers, idle from here on): one round of
        // independent 16-byte loads, in flight under the epilogue
        float4 gw[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) gw[q] = reinterpret_cast<const float4*>(P.ctx_w1)[tid + CB_THREADS * q];
        gw[4] = reinterpret_cast<const float4*>(P.ctx_w2)[tid];
        // epilogue: lane = 4 channels 16 wv + 4 g + r of frame 16 mt + li
        {
            const int c0 = wv * 16 + g * 4;
            const float4 bb = *reinterpret_cast<const float4*>(P.b1 + c0);
            const float4 ss = *reinterpret_cast<const float4*>(P.bn2_scale + c0);
            const float4 hh = *reinterpret_cast<const float4*>(P.bn2_shift + c0);
            float sums[CB_MAX_SEG][4];
#pragma unroll
            for (int s = 0; s < CB_MAX_SEG; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) sums[s][r] = 0.f;
#pragma unroll
            for (int mt = 0; mt < CB_MT; ++mt) {
                const int t = mt * 16 + li;
                const bool live = t < Tn;
                float v[4];
                v[0] = live ? fmaxf((acc[mt][0] + bb.x) * ss.x + hh.x, 0.f) : 0.f;
                v[1] = live ? fmaxf((acc[mt][1] + bb.y) * ss.y + hh.y, 0.f) : 0.f;
                v[2] = live ? fmaxf((acc[mt][2] + bb.z) * ss.z + hh.z, 0.f) : 0.f;
                v[3] = live ? fmaxf((acc[mt][3] + bb.w) * ss.w + hh.w, 0.f) : 0.f;
                bf16x4 o;
                o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                *reinterpret_cast<bf16x4*>(hb + t * CB_HROW + c0 * 2) = o;
                // the context is taken over the bf16 h the next kernels of the unfused path read
                const int sg = t / a.seg_len;
#pragma unroll
                for (int s = 0; s < CB_MAX_SEG; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sums[s][r] += (sg == s) ? (float)o[r] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < CB_MAX_SEG; ++s) {
                if (s < a.nseg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = sums[s][r];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                        if (li == 0) csum[s * CB_BNC + c0 + r] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(xs)[tid + CB_THREADS * q] = gw[q];
        reinterpret_cast<float4*>(xs)[4 * CB_THREADS + tid] = gw[4];
        __syncthreads();
        CB_STAMP(2)
        // ---- phase 2: context gate.  ctx[s] = mean_t h + mean_{t in s} h; hid = relu(W1^T ctx + b1); gate = sigmoid(W2^T hid + b2)
        // (the gate's weights were fetched before the epilogue and now sit in the idle chunk buffers)
        if (tid < CB_BNC) {
            float tot = 0.f;
            for (int s = 0; s < a.nseg; ++s) tot += csum[s * CB_BNC + tid];
            const float mean = tot / (float)Tn;
            for (int s = 0; s < a.nseg; ++s) {
                const int len = min(a.seg_len, Tn - s * a.seg_len);
                csum[s * CB_BNC + tid] = mean + csum[s * CB_BNC + tid] / (float)len;
            }
        }
        __syncthreads();
        {
            const float* w1s = reinterpret_cast<const float*>(xs);                 // [128][64]
            const int nout = a.nseg * CB_H;                       // <= 256: (s, j) outputs, the 128-long sum split over 4 threads x 32
            const int kq = tid & 3;
            for (int base = 0; base < nout; base += CB_THREADS / 4) {
                const int o = base + (tid >> 2);
                float p = 0.f;
                if (o < nout) {
                    const int s = o / CB_H, j = o - s * CB_H;
                    const float* cx = csum + s * CB_BNC + kq * 32;
                    const float* w = w1s + (kq * 32) * CB_H + j;
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) p += cx[k] * w[k * CB_H];
                }
                p += __shfl_xor(p, 1);
                p += __shfl_xor(p, 2);
                if (o < nout && kq == 0) {
                    const int s = o / CB_H, j = o - s * CB_H;
                    hid[s * CB_H + j] = fmaxf(p + P.ctx_b1[j], 0.f);
                }
            }
        }
        __syncthreads();
        {
            const float* w2s = reinterpret_cast<const float*>(xs) + CB_BNC * CB_H;  // [64][32]
            const int nout = a.nseg * CB_GR;                      // <= 128
            const int o = tid >> 2, kq = tid & 3;
            float p = 0.f;
            if (o < nout) {
                const int s = o / CB_GR, j = o - s * CB_GR;
                const float* hx = hid + s * CB_H + kq * 16;
                const float* w = w2s + (kq * 16) * CB_GR + j;
#pragma unroll
                for (int k = 0; k < 16; ++k) p += hx[k] * w[k * CB_GR];
            }
            p += __shfl_xor(p, 1);
            p += __shfl_xor(p, 2);
            if (o < nout && kq == 0) {
                const int s = o / CB_GR, j = o - s * CB_GR;
                gate[s * CB_GR + j] = 1.f / (1.f + __expf(-(p + P.ctx_b2[j])));
            }
        }
        __syncthreads();
        CB_STAMP(3)
        // ---- phase 3: y = (Wl * h + bl) * gate, appended as columns [ch, ch + 32).  20 output tiles (10 frame tiles x 2 channel tiles)
        for (int it = wv; it < CB_MT * 2; it += CB_WAVES) {
            const int mt = it >> 1, nt = it & 1;
            const int t = mt * 16 + li;
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int ts = t + (tap - 1) * P.dil;
                const char* hrow = hb + ((ts >= 0 && ts < Tn) ? ts : CB_TP) * CB_HROW;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wls + (nt * 16 + li) * CB_WLROW + (tap * 128 + ks * 32 + g * 8) * 2);
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(hrow + (ks * 32 + g * 8) * 2);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, c, 0, 0, 0);
                }
            }
            if (t < Tn) {
                const int o0 = nt * 16 + g * 4;
                const float4 bl = *reinterpret_cast<const float4*>(P.bl + o0);
                const float* gt = gate + (t / a.seg_len) * CB_GR + o0;
                bf16x4 o;
                o[0] = (bf16_t)((c[0] + bl.x) * gt[0]); o[1] = (bf16_t)((c[1] + bl.y) * gt[1]);
                o[2] = (bf16_t)((c[2] + bl.z) * gt[2]); o[3] = (bf16_t)((c[3] + bl.w) * gt[3]);
                *reinterpret_cast<bf16x4*>(a.cat + ((size_t)b * Tn + t) * a.ld + ch + o0) = o;
            }
        }
        // the new columns are in L2 before any wave of this workgroup reads them as part of the next layer's input, and every wave
        // is done with this layer's LDS tables before the next layer's are staged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CB_STAMP(4)
    }
#ifdef VP_TIMING
    if (a.dbg && tid == 0)
        for (int i = 0; i < 5; ++i) a.dbg[(size_t)b * 5 + i] = tacc[i];
#endif
}

}  // namespace

#ifdef VP_TIMING
static unsigned long long* g_cam_dbg = nullptr;
extern "C" void vp_dbg_cam_buffer(void* p) { g_cam_dbg = (unsigned long long*)p; }
#endif

// One dense block of CAM++ (bf16 engine).  Returns VP_EUNSUP when the shape is not covered (the caller falls back to the per-layer
// launches): bottleneck 128, growth 32, k3 local convs, T' <= 160, <= 4 context segments, <= 24 layers, <= 1024 channels.
int vp_cam_block_bf16(vp_ctx* ctx, const vp_cam_layer* layers, int n_layers, void* cat, int ld, int ch0, int B, int Tn, int seg_len,
                      int bn_channels, int growth, hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("VPMI_CAM_UNFUSED"); off = e && atoi(e) ? 1 : 0; }
    if (off) return VP_EUNSUP;
    const int nseg = (Tn + seg_len - 1) / seg_len;
    if (bn_channels != CB_BNC || growth != CB_GR || Tn > CB_TP || Tn < 2 || nseg > CB_MAX_SEG || n_layers < 1 || n_layers > CB_MAX_LAYERS ||
        ch0 % 8 || ld % 8 || ch0 + n_layers * CB_GR > CB_MAX_CH || ch0 + n_layers * CB_GR > ld || B > 65535)
        return VP_EUNSUP;
    const unsigned long long bytes = (unsigned long long)B * Tn * ld * 2;
    if (bytes >= 0xffffff00ull || (reinterpret_cast<uintptr_t>(cat) & 15)) return VP_EUNSUP;
    CamBlockArgs a;
    memset(&a, 0, sizeof(a));
    a.cat = (bf16_t*)cat; a.ld = ld; a.ch0 = ch0; a.Tn = Tn; a.seg_len = seg_len; a.nseg = nseg; a.n_layers = n_layers;
    a.cat_bytes = (unsigned)bytes;
#ifdef VP_TIMING
    a.dbg = g_cam_dbg;
#endif
    for (int l = 0; l < n_layers; ++l) {
        const vp_cam_layer& L = layers[l];
        if (L.linear1.kw != 1 || L.linear1.cin != ch0 + l * CB_GR || L.linear1.cout != CB_BNC || L.local.kw != 3 || L.local.cin != CB_BNC ||
            L.local.cout != CB_GR || !L.linear1.bias || !L.linear1.bn_scale || !L.linear1.bn_shift || !L.local.bias || L.local.bn_scale ||
            L.local.dil < 1)
            return VP_EUNSUP;
        CamLayerP& P = a.L[l];
        P.bn1_scale = L.bn1_scale; P.bn1_shift = L.bn1_shift;
        P.w1 = (const bf16_t*)L.linear1.w; P.b1 = L.linear1.bias; P.bn2_scale = L.linear1.bn_scale; P.bn2_shift = L.linear1.bn_shift;
        P.wl = (const bf16_t*)L.local.w; P.bl = L.local.bias;
        P.ctx_w1 = L.ctx_w1; P.ctx_b1 = L.ctx_b1; P.ctx_w2 = L.ctx_w2; P.ctx_b2 = L.ctx_b2;
        P.dil = L.local.dil;
    }
    constexpr size_t smem = (size_t)2 * CB_TP * 128 + (size_t)(CB_TP + 1) * CB_HROW + (size_t)CB_GR * CB_WLROW +
                            (size_t)(2 * CB_MAX_CH + (CB_MAX_SEG + 1) * CB_BNC + CB_MAX_SEG * CB_H + CB_MAX_SEG * CB_GR + CB_THREADS) * 4;
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cam_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(cam_block_kernel, dim3(B), dim3(CB_THREADS), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "cam_block");
    return VP_OK;
}
