// Res2Net chain of one SE-Res2 block in TRAINING mode (batch-statistics BatchNorm), mixed precision: one launch per direction.
//
// Replaces, for enable_amp steps, the per-chunk launch sequence of Res2NetBlock.forward / backward (ppvector/models/ecapa_tdnn.py:
// 36-47; trainer.py:226-244 drives it): y_0 = x_0, y_i = BN_i(ReLU(conv_i(x_i + y_{i-1}))), concat.  As separate launches every
// chunk is conv -> BN finalise -> BN apply (+ hand-off add) forward and column sums -> BN/ReLU backward -> data-gradient conv ->
// reflect fold backward: ~16 launches of 10-25 us over 19 MB tensors per chunk, 21 chunks per step (3.2 of the 12.9 ms step at
// B = 256, most of the 4.5 ms at B = 32).
//
// Here ONE workgroup owns one utterance for the whole chain (its T x 64 activations stay in LDS / registers between the chunks,
// as in the inference kernel res2_chain.hip) and the batch statistics -- the only cross-utterance dependency -- go through a
// grid barrier per chunk: every workgroup writes its per-channel partial sums, all workgroups meet (an agent-scope counter;
// B <= #CUs workgroups at one per CU are co-resident by construction), and every workgroup reduces the B partials in the same
// fixed order, so all of them hold bit-identical statistics.  Same roundings as the per-chunk path: conv operands (input, weights,
// dz) rounded to bf16 on their way into LDS, f32 accumulation, everything else f32.
//
// Register state between phases is in ROW layout (lane -> rows rl + 4k of the wave's 16-row tile, 4 consecutive channels): every
// global access is 256 contiguous bytes per row (f32) or 128 (bf16); MFMA accumulators pass through a wave-private LDS slab.
//
// Forward saves for backward: z_i = ReLU(conv_i + bias) (f32), the conv inputs as the matrix cores read them (bf16), mean / invstd.
// Backward: BatchNorm + ReLU backward (one grid barrier for sum dy, sum dy zhat), dz -> LDS + global (bf16: the weight-gradient
// GEMM's operand, run by the caller through vp_conv1d_wgrad_bf16_oik), the data gradient as G[u] = sum_tap W_tap^T dz[u - s_tap]
// over the utterance plus ONE extra 16-row tile holding the 2 d rows outside [0, T) whose mirror images the reflect padding
// folds back (ecapa_tdnn.py: Conv1d padding_mode='reflect'), hand-off d y_{i-1} = d in_i + d out_{i-1}.
#include "common.h"

namespace {

constexpr int RT_W = 64;
constexpr int RT_K = 3 * RT_W;
constexpr int RT_THREADS = 512;
constexpr int RT_WAVES = 8;
constexpr int RT_ROUNDS = 3;
constexpr int RT_MAXC = 7;
constexpr int RT_SLD = 68;                  // floats per slab row (64 + 4: the accumulator-layout writes spread over banks)
constexpr int RT_WT_BYTES = RT_W * RT_K * 2;

struct Res2TrainArgs {
    const float* x;                         // forward: x (M, C);  backward: d out (M, C)
    float* out;                             // forward: out (M, C); backward: d x (M, C)
    const float* w[RT_MAXC];                // (64, 64, 3) f32, the model's layout
    const float* bias[RT_MAXC];
    const float* gamma[RT_MAXC];
    const float* beta[RT_MAXC];
    float* rmean[RT_MAXC];
    float* rvar[RT_MAXC];
    float* z;                               // [nconv][M][64] f32
    bf16_t* inb;                            // [nconv][M][64] bf16   (forward only)
    bf16_t* dzb;                            // [nconv][M][64] bf16   (backward only)
    bf16_t* outb;                           // forward, optional: out once more as bf16 (M, C) -- the operand of the conv that follows
    float* stats;                           // [nconv][2][64]: mean, invstd
    float* dvec;                            // [nconv][3][64]: d bias, d gamma, d beta  (backward only)
    int x16;                                // forward: x is bf16 (M, C)
    float* part;                            // workspace [nconv][B][128] (+ backward: [nconv][B][64] behind it)
    unsigned* bar;                          // grid-barrier words (rt_grid_barrier)
    int B, T, C, nconv, dil, TP;
    float momentum, eps, invM;
};

__device__ __forceinline__ int rt_reflect(int t, int T) {
    t = t < 0 ? -t : t;
    return t >= T ? 2 * (T - 1) - t : t;
}

// Grid barrier + the exchange of per-workgroup partial sums WITHOUT cache maintenance.  The only data that crosses workgroups inside
// these kernels are the partial-sum arrays, so they are written and read with agent-scope relaxed atomics (sc1: they bypass the
// XCD-local L2 state) and ordered by hand: every writer drains its stores (s_waitcnt vmcnt(0)) before its workgroup's arrival is
// counted, every reader issues its loads after it has seen the count.  The first version used __threadfence() + acquire polling:
// each of those is an L2 write-back / invalidate on a multi-XCD part, 2048 wave-level cache flushes per barrier plus one per poll --
// 128 us per chunk.  Arrivals are spread over 8 counters (64 B apart... one cache line each: 256 arrivals on one address serialise).
constexpr int RT_NBAR = 8;                  // arrival counters, 32 words apart; [RT_NBAR * 32] departures, [RT_NBAR * 32 + 1] bail-out flag
__device__ __forceinline__ void rt_put(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rt_get(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rt_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// round: 1, 2, ... within the launch.  Workgroup b arrives on counter b % 8; counter c is complete at round * (number of b with b % 8 == c).
// arrive = this workgroup's partial sums are stored; wait = everybody's are.  Work that does not need the totals goes between the two.
__device__ __forceinline__ void rt_arrive(unsigned* bar, int b) {
    rt_drain();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar + (b % RT_NBAR) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rt_wait(unsigned* bar, unsigned round, int nwg) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned members = lane < RT_NBAR ? (unsigned)((nwg - lane + RT_NBAR - 1) / RT_NBAR) : 0u;
        const unsigned target = round * members;
        unsigned spins = 0;
        for (;;) {
            const unsigned v = lane < RT_NBAR ? __hip_atomic_load(bar + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(v >= target)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) {       // ~3 s; never on a healthy launch (B <= #CUs, waits are < 1 ms): leave instead of hanging the device
                if (lane == 0) __hip_atomic_store(bar + RT_NBAR * 32 + 1, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void rt_grid_barrier(unsigned* bar, unsigned round, int b, int nwg) {
    rt_arrive(bar, b);
    rt_wait(bar, round, nwg);
}

// the launch's last workgroup to leave re-arms the counters for the next launch on the stream
__device__ __forceinline__ void rt_grid_leave(unsigned* bar, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = __hip_atomic_fetch_add(bar + RT_NBAR * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == nwg - 1) {
            for (int c = 0; c < RT_NBAR; ++c) __hip_atomic_store(bar + c * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar + RT_NBAR * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// byte position of channel c (multiple of 4) of row t in a [rows][128 B] bf16 buffer with the 16-byte chunks XOR-swizzled by row
__device__ __forceinline__ int rt_pos(int t, int c) { return t * 128 + ((((c >> 3) ^ (t & 7))) << 4) + (c & 7) * 2; }

// weights of one conv, f32 (o, c, tap) in global -> bf16 [tap][row][128 B] in LDS; row = o, k = c (forward: W as the A operand)
// or row = c, k = o (data gradient: W^T)
template <bool TRANSPOSED>
__device__ __forceinline__ void rt_load_w(const float* __restrict__ w, char* dst, int tid) {
    constexpr int N = RT_W * RT_K / RT_THREADS / 2;          // 24 per thread in two batches of 12 loads in flight
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
    float v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = w[tid + (h * N + u) * RT_THREADS];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int i = tid + (h * N + u) * RT_THREADS;
        const int oc = i / 3, tap = i - oc * 3;
        const int o = oc >> 6, c = oc & 63;
        const int row = TRANSPOSED ? c : o, k = TRANSPOSED ? o : c;
        *reinterpret_cast<bf16_t*>(dst + tap * 8192 + row * 128 + (((k >> 3) ^ (row & 7)) << 4) + (k & 7) * 2) = (bf16_t)v[u];
    }
    }
}

// sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same channels)
__device__ __forceinline__ float rt_sum_rows(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Row-layout global accesses go through buffer instructions: descriptor + per-lane offset (one VGPR per pitch, or RT_PAST for a lane
// whose row is past the utterance: its loads return zeros, its stores are dropped) + a SCALAR offset for (utterance, tile, row group,
// channel slice).  With flat 64-bit addresses the compiler hoisted the 12 row addresses of every tensor out of the chunk loop and kept
// ~100 VGPRs alive (scratch spills).
typedef __attribute__((ext_vector_type(4))) unsigned int rt_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int rt_u32x2;
constexpr unsigned RT_PAST = 0xf0000000u;      // + any scalar offset < 0x10000000 stays out of range without wrapping

__device__ __forceinline__ float4 rt_ld(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
    const rt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
    return make_float4(__builtin_bit_cast(float, (unsigned)v[0]), __builtin_bit_cast(float, (unsigned)v[1]),
                       __builtin_bit_cast(float, (unsigned)v[2]), __builtin_bit_cast(float, (unsigned)v[3]));
}
// x of the forward, f32 or bf16 (wave-uniform): the byte offsets are written for f32 and halved for bf16 (RT_PAST / 2 stays out of range)
__device__ __forceinline__ float4 rt_ldx(__amdgpu_buffer_rsrc_t r, bool x16, unsigned vo, unsigned so) {
    if (!x16) return rt_ld(r, vo, so);
    const rt_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo >> 1, so >> 1, 0);
    return make_float4(__builtin_bit_cast(float, (unsigned)v[0] << 16), __builtin_bit_cast(float, (unsigned)v[0] & 0xffff0000u),
                       __builtin_bit_cast(float, (unsigned)v[1] << 16), __builtin_bit_cast(float, (unsigned)v[1] & 0xffff0000u));
}
__device__ __forceinline__ void rt_st(__amdgpu_buffer_rsrc_t r, float4 v, unsigned vo, unsigned so) {
    rt_u32x4 u;
    u[0] = __builtin_bit_cast(unsigned, v.x); u[1] = __builtin_bit_cast(unsigned, v.y);
    u[2] = __builtin_bit_cast(unsigned, v.z); u[3] = __builtin_bit_cast(unsigned, v.w);
    // offset entirely in the VGPR, soffset = 0: hipcc pads the 'store data > 64 bits, then a VALU write of the data registers' hazard
    // only for stores WITHOUT a register soffset (LLVM createsVALUHazard), but gfx950 shows it with one too -- with so in an SGPR the
    // y rows of a chunk were stored and overwritten by the next v_pk_add in the same cycle pair (intermittently stale rows in `out`)
    __builtin_amdgcn_raw_buffer_store_b128(u, r, vo + so, 0, 0);
}
__device__ __forceinline__ void rt_st8(__amdgpu_buffer_rsrc_t r, bf16x4 v, unsigned vo, unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rt_u32x2, v), r, vo, so, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rt_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
}
// between a wave's accumulator-layout writes of its slab and its row-layout reads (and back): the LDS queue drained, and a compiler
// barrier (per THREAD the two address patterns never overlap, so nothing else keeps the accesses in program order)
// every workgroup sums the B partial rows (128 floats each, from byte offset `base` of the workspace) in the same order:
// thread -> 4 channels (tid & 31), rows q, q + 16, ... (q = tid >> 5), 16-byte loads past the caches (sc0 sc1), then the 16 row groups
// in LDS order.  tot: [16][128].
__device__ __forceinline__ void rt_totals(__amdgpu_buffer_rsrc_t rp, unsigned base, int B, int tid, float* tot) {
    const unsigned vo = (unsigned)(tid & 31) * 16;
    const int q = tid >> 5;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int bb = q; bb < B; bb += 16) {
        const rt_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, vo, base + (unsigned)bb * 512, 17);
        s.x += __builtin_bit_cast(float, (unsigned)v[0]); s.y += __builtin_bit_cast(float, (unsigned)v[1]);
        s.z += __builtin_bit_cast(float, (unsigned)v[2]); s.w += __builtin_bit_cast(float, (unsigned)v[3]);
    }
    *reinterpret_cast<float4*>(tot + q * 128 + (tid & 31) * 4) = s;
}
__device__ __forceinline__ float rt_total16(const float* tot, int c) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += tot[q * 128 + c];
    return s;
}

__device__ __forceinline__ void rt_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(RT_THREADS) void res2_train_fwd_kernel(Res2TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* act = smem;                                                    // [TP][128 B]
    char* wts = act + a.TP * 128;                                        // [3][64][128 B]
    float* slab_all = reinterpret_cast<float*>(wts + RT_WT_BYTES);       // [8][16][RT_SLD]
    float* red = slab_all + RT_WAVES * 16 * RT_SLD;                      // [8][128]
    float* tot = red + RT_WAVES * 128;                                   // [16][128]
    float* prm = tot + 16 * 128;                                         // [2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int rl = lane >> 4, cq = (lane & 15) * 4;                      // row layout: rows rl + 4 k, channels cq .. cq + 3
    const int b = blockIdx.x;
    const unsigned row0 = (unsigned)b * a.T;
    const unsigned Mrows = (unsigned)a.B * a.T;
    const int ntile = a.TP / 16;
    float* slab = slab_all + wv * 16 * RT_SLD;
    const bool x16 = a.x16 != 0;
    const __amdgpu_buffer_rsrc_t rx = rt_rsrc(a.x, (size_t)Mrows * a.C * (x16 ? 2 : 4));
    const __amdgpu_buffer_rsrc_t ro = rt_rsrc(a.out ? (const void*)a.out : (const void*)a.x, a.out ? (size_t)Mrows * a.C * 4 : 0);   // (absent: every store out of range)
    const __amdgpu_buffer_rsrc_t rz = rt_rsrc(a.z, (size_t)a.nconv * Mrows * RT_W * 4), ri = rt_rsrc(a.inb, (size_t)a.nconv * Mrows * RT_W * 2);
    const unsigned lx = (unsigned)(rl * a.C + cq) * 4, lz = (unsigned)(rl * RT_W + cq) * 4, lb = (unsigned)(rl * RT_W + cq) * 2;
    const unsigned pitch = (unsigned)a.C * 4;
    const __amdgpu_buffer_rsrc_t rp = rt_rsrc(a.part, (size_t)a.nconv * a.B * (128 + 64) * 4);
    const __amdgpu_buffer_rsrc_t rob = rt_rsrc(a.outb ? (const void*)a.outb : (const void*)a.out, a.outb ? (size_t)Mrows * a.C * 2 : 0);   // (absent: every store out of range)
    const unsigned lxb = (unsigned)(rl * a.C + cq) * 2;

    // ---- prologue: out[:, 0:64] = x[:, 0:64]; in_1 = x[:, 64:128] -> LDS (bf16) and the saved operand; weights of conv 0
    rt_load_w<false>(a.w[0], wts, tid);
#pragma unroll
    for (int r = 0; r < RT_ROUNDS; ++r) {
        const int mt = wv + r * RT_WAVES;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = mt * 16 + rl + 4 * k;
            const bool ok = t < a.T;
            const unsigned so = (row0 + mt * 16 + 4 * k) * pitch;
            const float4 v0 = rt_ldx(rx, x16, ok ? lx : RT_PAST, so);
            const float4 v1 = rt_ldx(rx, x16, ok ? lx : RT_PAST, so + RT_W * 4);
            rt_st(ro, v0, ok ? lx : RT_PAST, so);
            {
                bf16x4 ob;
                ob[0] = (bf16_t)v0.x; ob[1] = (bf16_t)v0.y; ob[2] = (bf16_t)v0.z; ob[3] = (bf16_t)v0.w;
                rt_st8(rob, ob, ok ? lxb : RT_PAST, so >> 1);
            }
            bf16x4 o;
            o[0] = (bf16_t)v1.x; o[1] = (bf16_t)v1.y; o[2] = (bf16_t)v1.z; o[3] = (bf16_t)v1.w;
            if (ok) *reinterpret_cast<bf16x4*>(act + rt_pos(t, cq)) = o;
            rt_st8(ri, o, ok ? lb : RT_PAST, (row0 + mt * 16 + 4 * k) * (RT_W * 2));
        }
    }
    __syncthreads();

    for (int j = 0; j < a.nconv; ++j) {
        const bool has_next = j + 1 < a.nconv;
        f32x4 acc[RT_ROUNDS][4];
        {   // ---- conv j: weight fragments in registers, 3 rounds of 24 MFMAs (as res2_chain.hip)
            bf16x8 wf[4][6];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ni * 16 + li;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const int c = (s & 1) * 4 + g;
                    wf[ni][s] = *reinterpret_cast<const bf16x8*>(wts + (s >> 1) * 8192 + n * 128 + ((c ^ (n & 7)) << 4));
                }
            }
#pragma unroll
            for (int r = 0; r < RT_ROUNDS; ++r) {
                const int mt = wv + r * RT_WAVES;
                const int t = min(mt * 16 + li, a.TP - 1);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[r][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (mt >= ntile) continue;                             // wave-uniform: an idle tile slot
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    int ts = rt_reflect(t + (tap - 1) * a.dil, a.T);
                    ts = min(max(ts, 0), a.T - 1);                     // rows >= T only feed discarded outputs; keep the reads on written rows
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int c = ks * 4 + g;
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(act + ts * 128 + ((c ^ (ts & 7)) << 4));
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc[r][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][tap * 2 + ks], xf, acc[r][ni], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                               // every wave is done with act and wts
        // ---- z = ReLU(acc + bias): accumulator layout -> slab -> row layout; this utterance's share of the statistics
        float4 rr[RT_ROUNDS][4], xn[RT_ROUNDS][4];
        {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            const float* bj = a.bias[j];
#pragma unroll
            for (int r = 0; r < RT_ROUNDS; ++r) {
                const int mt = wv + r * RT_WAVES;
#pragma unroll
                for (int k = 0; k < 4; ++k) rr[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mt < ntile) {                                      // wave-uniform
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const int nb = ni * 16 + g * 4;
                        const float4 bb = *reinterpret_cast<const float4*>(bj + nb);
                        float4 v;
                        v.x = fmaxf(acc[r][ni][0] + bb.x, 0.f); v.y = fmaxf(acc[r][ni][1] + bb.y, 0.f);
                        v.z = fmaxf(acc[r][ni][2] + bb.z, 0.f); v.w = fmaxf(acc[r][ni][3] + bb.w, 0.f);
                        *reinterpret_cast<float4*>(slab + li * RT_SLD + nb) = v;
                    }
                    rt_wave_sync();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float4 v = *reinterpret_cast<const float4*>(slab + (rl + 4 * k) * RT_SLD + cq);
                        if (!(mt * 16 + rl + 4 * k < a.T)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        rr[r][k] = v;
                        s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
                        s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
                    }
                    rt_wave_sync();
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] = rt_sum_rows(s1[e]); s2[e] = rt_sum_rows(s2[e]); }
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[wv * 128 + cq + e] = s1[e]; red[wv * 128 + 64 + cq + e] = s2[e]; }
            }
        }
        __syncthreads();
        if (tid < 128) {
            float p = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < RT_WAVES; ++w8) p += red[w8 * 128 + tid];
            rt_put(a.part + ((size_t)j * a.B + b) * 128 + tid, p);
        }
        rt_arrive(a.bar, b);
        // ---- under the barrier's latency: z out, the next chunk's x in, the next conv's weights
#pragma unroll
        for (int r = 0; r < RT_ROUNDS; ++r) {
            const int mt = wv + r * RT_WAVES;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                              // (past the utterance: out-of-range offsets -- zeros in, nothing out)
                const bool ok = mt * 16 + rl + 4 * k < a.T;
                xn[r][k] = rt_ldx(rx, x16, (has_next && ok) ? lx : RT_PAST, (row0 + mt * 16 + 4 * k) * pitch + (j + 2) * (RT_W * 4));
                rt_st(rz, rr[r][k], ok ? lz : RT_PAST, ((unsigned)j * Mrows + row0 + mt * 16 + 4 * k) * (RT_W * 4));
            }
        }
        if (has_next) rt_load_w<false>(a.w[j + 1], wts, tid);
        rt_wait(a.bar, (unsigned)(j + 1), a.B);
        rt_totals(rp, (unsigned)j * a.B * 512, a.B, tid, tot);
        __syncthreads();
        if (tid < 64) {
            const float t1 = rt_total16(tot, tid), t2 = rt_total16(tot, 64 + tid);
            const float mu = t1 * a.invM;
            const float var = fmaxf(t2 * a.invM - mu * mu, 0.f);
            const float is = rsqrtf(var + a.eps);
            const float sc = a.gamma[j][tid] * is;
            prm[tid] = sc;
            prm[64 + tid] = a.beta[j][tid] - mu * sc;
            if (b == 0) {
                a.stats[(j * 2 + 0) * 64 + tid] = mu;
                a.stats[(j * 2 + 1) * 64 + tid] = is;
                // a barrier that gave up has produced these from incomplete sums: the running statistics keep their values (the
                // optimiser kernel drops the step's update on the same word, train_ops.hip)
                const bool sound = __hip_atomic_load(a.bar + RT_NBAR * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
                if (sound && a.rmean[j]) a.rmean[j][tid] = a.momentum * a.rmean[j][tid] + (1.f - a.momentum) * mu;
                if (sound && a.rvar[j]) a.rvar[j][tid] = a.momentum * a.rvar[j][tid] + (1.f - a.momentum) * var;
            }
        }
        __syncthreads();
        // ---- y_i = BN(z) -> out slice; in_{i+1} = y_i + x_{i+1} -> LDS (bf16) + saved operand
        const float4 sc = *reinterpret_cast<const float4*>(prm + cq);
        const float4 sh = *reinterpret_cast<const float4*>(prm + 64 + cq);
#pragma unroll
        for (int r = 0; r < RT_ROUNDS; ++r) {
            const int mt = wv + r * RT_WAVES;
            if (mt < ntile) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = mt * 16 + rl + 4 * k;
                    const bool ok = t < a.T;
                    float4 y;
                    y.x = rr[r][k].x * sc.x + sh.x; y.y = rr[r][k].y * sc.y + sh.y;
                    y.z = rr[r][k].z * sc.z + sh.z; y.w = rr[r][k].w * sc.w + sh.w;
                    rt_st(ro, y, ok ? lx : RT_PAST, (row0 + mt * 16 + 4 * k) * pitch + (j + 1) * (RT_W * 4));
                    {
                        bf16x4 ob;
                        ob[0] = (bf16_t)y.x; ob[1] = (bf16_t)y.y; ob[2] = (bf16_t)y.z; ob[3] = (bf16_t)y.w;
                        rt_st8(rob, ob, ok ? lxb : RT_PAST, ((row0 + mt * 16 + 4 * k) * pitch + (j + 1) * (RT_W * 4)) >> 1);
                    }
                    if (has_next) {
                        bf16x4 o;
                        o[0] = (bf16_t)(y.x + xn[r][k].x); o[1] = (bf16_t)(y.y + xn[r][k].y);
                        o[2] = (bf16_t)(y.z + xn[r][k].z); o[3] = (bf16_t)(y.w + xn[r][k].w);
                        if (ok) *reinterpret_cast<bf16x4*>(act + rt_pos(t, cq)) = o;
                        rt_st8(ri, o, ok ? lb : RT_PAST, ((unsigned)(j + 1) * Mrows + row0 + mt * 16 + 4 * k) * (RT_W * 2));
                    }
                }
            }
        }
        __syncthreads();                                               // act and wts hold the next conv's operands
    }
    rt_grid_leave(a.bar, a.B);
}

__global__ __launch_bounds__(RT_THREADS) void res2_train_bwd_kernel(Res2TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dzl = smem;                                                    // [TP + 1][128 B]; row TP = zeros
    char* wts0 = dzl + (a.TP + 1) * 128;                                 // [2][3][64 c][128 B over o]: conv j's W^T in buffer j & 1
    float* slab_all = reinterpret_cast<float*>(wts0 + 2 * RT_WT_BYTES);  // [8][16][RT_SLD]
    float* gx = slab_all + RT_WAVES * 16 * RT_SLD;                       // [16][64]: G rows outside [0, T)
    float* red = gx + 16 * 64;                                           // [8][128]
    float* tot = red + RT_WAVES * 128;                                   // [16][128]
    float* prm = tot + 16 * 128;                                         // [3][64]: gamma * invstd, sum dy / M, sum dy zhat / M
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int rl = lane >> 4, cq = (lane & 15) * 4;
    const int b = blockIdx.x;
    const unsigned row0 = (unsigned)b * a.T;
    const unsigned Mrows = (unsigned)a.B * a.T;
    const int ntile = a.TP / 16;
    const int d = a.dil, T = a.T;
    float* slab = slab_all + wv * 16 * RT_SLD;
    float* pbias = a.part + (size_t)a.nconv * a.B * 128;                 // [nconv][B][64]
    const __amdgpu_buffer_rsrc_t rx = rt_rsrc(a.x, (size_t)Mrows * a.C * 4), ro = rt_rsrc(a.out, (size_t)Mrows * a.C * 4);
    const __amdgpu_buffer_rsrc_t rz = rt_rsrc(a.z, (size_t)a.nconv * Mrows * RT_W * 4), ri = rt_rsrc(a.dzb, (size_t)a.nconv * Mrows * RT_W * 2);
    const unsigned lx = (unsigned)(rl * a.C + cq) * 4, lz = (unsigned)(rl * RT_W + cq) * 4, lb = (unsigned)(rl * RT_W + cq) * 2;
    const unsigned pitch = (unsigned)a.C * 4;
    const __amdgpu_buffer_rsrc_t rp = rt_rsrc(a.part, (size_t)a.nconv * a.B * (128 + 64) * 4);

    // ---- prologue: d x[:, 0:64] = d out[:, 0:64]; d y of the last chunk; W^T of the last conv; the zero row
    rt_load_w<true>(a.w[a.nconv - 1], wts0 + ((a.nconv - 1) & 1) * RT_WT_BYTES, tid);
    if (tid < 32) *reinterpret_cast<float*>(dzl + a.TP * 128 + tid * 4) = 0.f;
    float4 dy[RT_ROUNDS][4];
#pragma unroll
    for (int r = 0; r < RT_ROUNDS; ++r) {
        const int mt = wv + r * RT_WAVES;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = mt * 16 + rl + 4 * k < T;
            const unsigned so = (row0 + mt * 16 + 4 * k) * pitch;
            rt_st(ro, rt_ld(rx, ok ? lx : RT_PAST, so), ok ? lx : RT_PAST, so);
            dy[r][k] = rt_ld(rx, ok ? lx : RT_PAST, so + a.nconv * (RT_W * 4));
        }
    }

    const unsigned lx0 = lx, lz0 = lz, lb0 = lb;
    for (int j = a.nconv - 1; j >= 0; --j) {
        // (opaque per-iteration copies: otherwise the 36 validity-selected offsets are hoisted out of the loop and spilled)
        unsigned lx = lx0, lz = lz0, lb = lb0;
        asm volatile("" : "+v"(lx), "+v"(lz), "+v"(lb));
        const float4 mu = *reinterpret_cast<const float4*>(a.stats + (j * 2 + 0) * 64 + cq);
        const float4 is = *reinterpret_cast<const float4*>(a.stats + (j * 2 + 1) * 64 + cq);
        // ---- sums of dy and dy * zhat over this utterance (rows past it were loaded as zeros)
        float4 zz[RT_ROUNDS][4];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RT_ROUNDS; ++r) {
            const int mt = wv + r * RT_WAVES;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                zz[r][k] = rt_ld(rz, (mt * 16 + rl + 4 * k < T) ? lz : RT_PAST, ((unsigned)j * Mrows + row0 + mt * 16 + 4 * k) * (RT_W * 4));
        }
#pragma unroll
        for (int r = 0; r < RT_ROUNDS; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = zz[r][k], q = dy[r][k];                // dy = 0 on the rows past the utterance
                s1[0] += q.x; s1[1] += q.y; s1[2] += q.z; s1[3] += q.w;
                s2[0] += q.x * ((v.x - mu.x) * is.x); s2[1] += q.y * ((v.y - mu.y) * is.y);
                s2[2] += q.z * ((v.z - mu.z) * is.z); s2[3] += q.w * ((v.w - mu.w) * is.w);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] = rt_sum_rows(s1[e]); s2[e] = rt_sum_rows(s2[e]); }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[wv * 128 + cq + e] = s1[e]; red[wv * 128 + 64 + cq + e] = s2[e]; }
        }
        __syncthreads();
        if (tid < 128) {
            float p = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < RT_WAVES; ++w8) p += red[w8 * 128 + tid];
            rt_put(a.part + ((size_t)j * a.B + b) * 128 + tid, p);
        }
        rt_arrive(a.bar, b);
        if (j > 0) rt_load_w<true>(a.w[j - 1], wts0 + ((j - 1) & 1) * RT_WT_BYTES, tid);      // under the barrier's latency
        rt_wait(a.bar, (unsigned)(a.nconv - j), a.B);
        rt_totals(rp, (unsigned)j * a.B * 512, a.B, tid, tot);
        __syncthreads();
        if (tid < 64) {
            const float t1 = rt_total16(tot, tid), t2 = rt_total16(tot, 64 + tid);
            prm[tid] = a.gamma[j][tid] * a.stats[(j * 2 + 1) * 64 + tid];
            prm[64 + tid] = t1 * a.invM;
            prm[128 + tid] = t2 * a.invM;
            if (b == 0) { a.dvec[(j * 3 + 2) * 64 + tid] = t1; a.dvec[(j * 3 + 1) * 64 + tid] = t2; }
        }
        __syncthreads();
        // ---- dz = [z > 0] gamma invstd (dy - mean dy - zhat mean(dy zhat)) -> LDS + global (bf16); bias-gradient partial
        {
            const float4 k1 = *reinterpret_cast<const float4*>(prm + cq);
            const float4 m1 = *reinterpret_cast<const float4*>(prm + 64 + cq);
            const float4 m2 = *reinterpret_cast<const float4*>(prm + 128 + cq);
            float sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RT_ROUNDS; ++r) {
                const int mt = wv + r * RT_WAVES;
                if (mt < ntile) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int t = mt * 16 + rl + 4 * k;
                        const bool ok = t < T;
                        const float4 z4 = zz[r][k], q = dy[r][k];       // rows past the utterance: z = 0 -> dz = 0
                        float4 v;
                        v.x = z4.x > 0.f ? k1.x * (q.x - m1.x - (z4.x - mu.x) * is.x * m2.x) : 0.f;
                        v.y = z4.y > 0.f ? k1.y * (q.y - m1.y - (z4.y - mu.y) * is.y * m2.y) : 0.f;
                        v.z = z4.z > 0.f ? k1.z * (q.z - m1.z - (z4.z - mu.z) * is.z * m2.z) : 0.f;
                        v.w = z4.w > 0.f ? k1.w * (q.w - m1.w - (z4.w - mu.w) * is.w * m2.w) : 0.f;
                        sb[0] += v.x; sb[1] += v.y; sb[2] += v.z; sb[3] += v.w;
                        bf16x4 o;
                        o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
                        *reinterpret_cast<bf16x4*>(dzl + rt_pos(t, cq)) = o;            // rows T .. TP - 1: zeros
                        rt_st8(ri, o, ok ? lb : RT_PAST, ((unsigned)j * Mrows + row0 + mt * 16 + 4 * k) * (RT_W * 2));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sb[e] = rt_sum_rows(sb[e]);
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wv * 128 + cq + e] = sb[e];
            }
        }
        __syncthreads();                                               // dz rows, W^T and the bias partials are in LDS
        if (tid < 64) {
            float p = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < RT_WAVES; ++w8) p += red[w8 * 128 + tid];
            rt_put(pbias + ((size_t)j * a.B + b) * 64 + tid, p);
        }
        // ---- data gradient in padded coordinates: G[u][c] = sum_tap sum_o W[o][c][tap] dz[u - (tap - 1) d][o]
        float4 gg[RT_ROUNDS][4];
        {
            const char* wl = wts0 + (j & 1) * RT_WT_BYTES;
            bf16x8 wf[4][6];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = ni * 16 + li;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const int c = (s & 1) * 4 + g;
                    wf[ni][s] = *reinterpret_cast<const bf16x8*>(wl + (s >> 1) * 8192 + n * 128 + ((c ^ (n & 7)) << 4));
                }
            }
#pragma unroll
            for (int r = 0; r < RT_ROUNDS; ++r) {
                const int mt = wv + r * RT_WAVES;
                const bool extra = mt == RT_WAVES * RT_ROUNDS - 1;     // the last tile slot (never an utterance tile: ntile <= 23)
#pragma unroll
                for (int k = 0; k < 4; ++k) gg[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mt < ntile || extra) {                             // wave-uniform
                    int u;
                    if (extra) u = li < d ? -1 - li : (li < 2 * d ? T + li - d : -0x10000);
                    else u = mt * 16 + li;
                    f32x4 acc[4];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int tap = 0; tap < 3; ++tap) {
                        const int tsr = u - (tap - 1) * d;
                        const int ts = (tsr >= 0 && tsr < T) ? tsr : a.TP;             // the zero row
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const int c = ks * 4 + g;
                            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(dzl + ts * 128 + ((c ^ (ts & 7)) << 4));
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
                                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][tap * 2 + ks], xf, acc[ni], 0, 0, 0);
                        }
                    }
                    if (extra) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<float4*>(gx + li * 64 + ni * 16 + g * 4) = make_float4(acc[ni][0], acc[ni][1], acc[ni][2], acc[ni][3]);
                    } else {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            *reinterpret_cast<float4*>(slab + li * RT_SLD + ni * 16 + g * 4) = make_float4(acc[ni][0], acc[ni][1], acc[ni][2], acc[ni][3]);
                        rt_wave_sync();
#pragma unroll
                        for (int k = 0; k < 4; ++k) gg[r][k] = *reinterpret_cast<const float4*>(slab + (rl + 4 * k) * RT_SLD + cq);
                        rt_wave_sync();
                    }
                }
            }
        }
        __syncthreads();                                               // gx complete; dzl is dead
        // ---- fold the mirrored rows, d x slice, hand-off to the previous chunk
#pragma unroll
        for (int r = 0; r < RT_ROUNDS; ++r) {
            const int mt = wv + r * RT_WAVES;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = mt * 16 + rl + 4 * k;
                const bool ok = t < T;
                const unsigned so = (row0 + mt * 16 + 4 * k) * pitch;
                float4 v = gg[r][k];
                if (t >= 1 && t <= d) {                                 // G[-t]
                    const float4 e = *reinterpret_cast<const float4*>(gx + (t - 1) * 64 + cq);
                    v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
                }
                if (t >= T - 1 - d && t <= T - 2) {                     // G[2 (T - 1) - t]
                    const float4 e = *reinterpret_cast<const float4*>(gx + (d + (T - 2 - t)) * 64 + cq);
                    v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
                }
                rt_st(ro, v, ok ? lx : RT_PAST, so + (j + 1) * (RT_W * 4));
                const float4 o = rt_ld(rx, (ok && j > 0) ? lx : RT_PAST, so + j * (RT_W * 4));
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the utterance stay out of the next chunk's sums
                dy[r][k] = v;
            }
        }
        __syncthreads();                                               // gx, red reused by the next chunk
    }
    // ---- bias gradients: conv j's B partials summed by workgroup j (mod B), fixed order
    rt_grid_barrier(a.bar, (unsigned)(a.nconv + 1), b, a.B);
    for (int j = b; j < a.nconv; j += a.B) {
        const int c = tid & 63, q = tid >> 6;
        const float* pp = pbias + (size_t)j * a.B * 64 + c;
        float s = 0.f;
#pragma unroll 8
        for (int bb = q; bb < a.B; bb += RT_WAVES) s += rt_get(pp + (size_t)bb * 64);
        red[q * 128 + c] = s;
        __syncthreads();
        if (tid < 64) {
            float p = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < RT_WAVES; ++w8) p += red[w8 * 128 + tid];
            a.dvec[(j * 3 + 0) * 64 + tid] = p;
        }
        __syncthreads();
    }
    rt_grid_leave(a.bar, a.B);
}

}  // namespace

static int rt_num_cus(vp_ctx* ctx) {
    static int cus[16] = {0};
    const int dev = ctx->device & 15;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n < 1) n = 1;
        cus[dev] = n;
    }
    return cus[dev];
}

// Does a grid of nwg workgroups of `kernel` fit the device at once?  hipOccupancyMaxActiveBlocksPerMultiprocessor is the runtime's own
// answer for the kernel's registers and THIS launch's LDS (smem grows with T): asked on every call -- a cheap host query -- and never
// cached (a per-(device, kernel) cache kept the first T's answer, and a first failure disabled the kernel for good); the
// reserve is subtracted from the CU count.
static bool rt_fits(vp_ctx* ctx, const void* kernel, size_t smem, int nwg) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, RT_THREADS, smem) != hipSuccess || nb < 1) return false;
    const long long slots = (long long)nb * (rt_num_cus(ctx) - ctx->grid_reserve_cus);
    return nwg <= slots && nwg <= rt_num_cus(ctx) - ctx->grid_reserve_cus;      // (the dispatcher places one workgroup per CU first)
}

static int rt_fill(vp_ctx* ctx, const vp_res2_train_desc* d, void* ws, size_t ws_bytes, bool bwd, Res2TrainArgs& a) {
    if (!ctx || !d) return VP_EINVAL;
    const int nconv = d->scale - 1;
    if (d->width != RT_W || nconv < 1 || nconv > RT_MAXC || d->C != d->scale * RT_W) return VP_EUNSUP;
    const int TP = (d->T + 15) / 16 * 16;
    if (d->dil < 1 || d->dil > 8 || d->T < 2 * d->dil + 2 || TP / 16 > RT_WAVES * RT_ROUNDS - 1 || d->B < 1) return VP_EUNSUP;
    // the grid barrier needs every workgroup resident: one per CU (100-140 KB of LDS each; checked against the occupancy the runtime
    // computes for the kernel in rt_fits), minus the CUs the caller asked to leave to other queues (vp_set_grid_reserve_cus)
    if (d->B > rt_num_cus(ctx) - ctx->grid_reserve_cus) return VP_EUNSUP;
    if (!ctx->grid_bar) return VP_EUNSUP;
    if ((size_t)d->B * d->T * d->C * 4 >= 0x0ff00000ull) return VP_EUNSUP;    // 32-bit buffer offsets with room for the out-of-range marker
    // forward: `out` (f32) may be absent when the bf16 copy is asked for -- a caller whose only consumer reads the bf16 copy (tdnn2's GEMM operand)
    // saves the 156 MB f32 store per block
    const bool out_ok = d->out || (!bwd && d->out_bf16);
    if (!d->x || !out_ok || !d->z || !d->stats || !ws || (bwd ? (!d->dzb || !d->dvec) : !d->inb)) VP_FAIL(ctx, VP_EINVAL, "res2_train: null argument");
    if (ws_bytes < vp_res2_train_workspace_bytes(d->B, d->scale)) VP_FAIL(ctx, VP_EWORKSPACE, "res2_train: workspace too small");
    if ((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->out) | reinterpret_cast<uintptr_t>(d->z) |
         reinterpret_cast<uintptr_t>(d->inb) | reinterpret_cast<uintptr_t>(d->dzb)) & 15)
        VP_FAIL(ctx, VP_EINVAL, "res2_train: tensors must be 16-byte aligned");
    memset(&a, 0, sizeof(a));
    a.x = d->x; a.out = d->out; a.z = d->z; a.inb = (bf16_t*)d->inb; a.dzb = (bf16_t*)d->dzb; a.outb = bwd ? nullptr : (bf16_t*)d->out_bf16; a.stats = d->stats; a.dvec = d->dvec;
    a.x16 = bwd ? 0 : d->x_is_bf16;
    a.part = (float*)ws; a.bar = ctx->grid_bar;
    for (int j = 0; j < nconv; ++j) {
        if (!d->w[j] || !d->gamma[j] || (!bwd && (!d->bias[j] || !d->beta[j]))) VP_FAIL(ctx, VP_EINVAL, "res2_train: null parameter");
        a.w[j] = d->w[j]; a.bias[j] = d->bias[j]; a.gamma[j] = d->gamma[j]; a.beta[j] = d->beta[j];
        a.rmean[j] = d->run_mean[j]; a.rvar[j] = d->run_var[j];
    }
    a.B = d->B; a.T = d->T; a.C = d->C; a.nconv = nconv; a.dil = d->dil; a.TP = TP;
    a.momentum = d->momentum; a.eps = d->eps; a.invM = 1.f / ((float)d->B * (float)d->T);
    return VP_OK;
}

extern "C" {

size_t vp_res2_train_workspace_bytes(int B, int scale) {
    return (size_t)(scale - 1) * B * (128 + 64) * sizeof(float) + 256;
}

int vp_res2_train_fwd(vp_ctx* ctx, const vp_res2_train_desc* d, void* ws, size_t ws_bytes, vp_stream stream) {
    Res2TrainArgs a;
    const int rc = rt_fill(ctx, d, ws, ws_bytes, false, a);
    if (rc != VP_OK) return rc;
    const size_t smem = (size_t)a.TP * 128 + RT_WT_BYTES + (RT_WAVES * 16 * RT_SLD + RT_WAVES * 128 + 16 * 128 + 2 * 64) * sizeof(float);
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(res2_train_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (!rt_fits(ctx, reinterpret_cast<const void*>(res2_train_fwd_kernel), smem, a.B)) return VP_EUNSUP;
    hipLaunchKernelGGL(res2_train_fwd_kernel, dim3(a.B), dim3(RT_THREADS), smem, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "res2_train_fwd");
    return VP_OK;
}

int vp_res2_train_bwd(vp_ctx* ctx, const vp_res2_train_desc* d, void* ws, size_t ws_bytes, vp_stream stream) {
    Res2TrainArgs a;
    const int rc = rt_fill(ctx, d, ws, ws_bytes, true, a);
    if (rc != VP_OK) return rc;
    const size_t smem = (size_t)(a.TP + 1) * 128 + 2 * RT_WT_BYTES +
                        (RT_WAVES * 16 * RT_SLD + 16 * 64 + RT_WAVES * 128 + 16 * 128 + 3 * 64) * sizeof(float);
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(res2_train_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (!rt_fits(ctx, reinterpret_cast<const void*>(res2_train_bwd_kernel), smem, a.B)) return VP_EUNSUP;
    hipLaunchKernelGGL(res2_train_bwd_kernel, dim3(a.B), dim3(RT_THREADS), smem, (hipStream_t)stream, a);
    VP_LAUNCH_CHECK(ctx, "res2_train_bwd");
    return VP_OK;
}

// Re-arms the barrier words after a bail-out (arrival / departure counters are left unbalanced by it, the flag stays set until
// cleared): the caller switches to the per-chunk kernels first, or the next fused launch meets the same residency problem.
int vp_grid_barrier_reset(vp_ctx* ctx, vp_stream stream) {
    if (!ctx || !ctx->grid_bar) return VP_EINVAL;
    VP_HIP(ctx, hipMemsetAsync(ctx->grid_bar, 0, 2048, (hipStream_t)stream));
    return VP_OK;
}

// 0 while no grid barrier of this context gave up waiting since the last reset; synchronises the host with the device (a 4-byte
// copy): GraphedTrainStep polls it every few steps and before every checkpoint, the optimiser kernels test the word on the device
int vp_grid_barrier_status(vp_ctx* ctx) {
    if (!ctx || !ctx->grid_bar) return -1;
    unsigned v = 0;
    if (hipMemcpy(&v, ctx->grid_bar + RT_NBAR * 32 + 1, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

}  // extern "C"
