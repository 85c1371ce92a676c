// 256 x 256 x 64 conv GEMM for the wide bf16 layers (1x1 / tapped 1-D, Cin % 64 == 0).
//
// Why a second kernel: the 128 x 128 kernel (conv_gemm_impl.h) moves 64 FLOP per byte staged and
// pays a ds_write_b128 pass (~79 B/clk/CU) for every byte, so its LDS pipe is as busy as its MFMA
// pipe.  Here one workgroup of 8 waves owns a 256-position x 256-channel tile (128 FLOP per staged
// byte), each wave a 64 x 128 sub-tile (128 accumulator VGPRs), and the operands go HBM/L2 -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs and no ds_write pass.  The DMA writes a
// wave's 64 x 16 B linearly, so the XOR swizzle that keeps ds_read_b128 conflict-free is applied to
// the per-lane SOURCE chunk (and again on the read) instead of the destination.  Out-of-range rows /
// padding taps use out-of-range buffer offsets: the DMA writes zeros.
// Two LDS stages of 64 KB; per K-step: wait own DMAs -> barrier -> issue next stage -> 64 MFMAs/wave.
// Epilogue semantics are those of conv_gemm_impl.h (same ConvArgs, same psum layout per 128 rows).
#include "conv_gemm_impl.h"

#include <type_traits>

namespace {

constexpr int T2 = 256;                      // tile edge (positions and channels)
constexpr int MODE_TAPS_GEN = 4;             // tapped 1-D conv with Cin % 64 != 0: a K-step may straddle taps (role-split schedule only)
constexpr int STAGE2 = 2 * T2 * ROWB;        // X panel + W panel
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int SCHED>
__global__ __launch_bounds__(512) void conv_gemm256_kernel(const ConvArgs a) {
    constexpr int MI = 4, NI = 8;            // per wave: 64 positions x 128 channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VP_TIMING
    const unsigned long long tk0 = wall_clock64();
    unsigned long long tk1 = 0, twait = 0, tbar = 0, ck1 = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: scalar branches on the wave's role
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, g = lane >> 4;

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int gsz = a.group_m * a.tiles_n;
    const int grp = swz / gsz, rem = swz - grp * gsz;
    const int gm = min(a.group_m, a.tiles_m - grp * a.group_m);
    const int tn = rem / gm;
    const int tm = grp * a.group_m + (rem - tn * gm);
    const int m0 = tm * T2, n0 = tn * T2;

    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

    // staging: wave wv fills rows [32 wv, 32 wv + 32) of both panels, 8 rows x 128 B per DMA; lane l
    // lands at row (l >> 3), slot (l & 7) and therefore fetches chunk slot ^ row
    const int srow = lane >> 3;
    const unsigned cb = (unsigned)((lane & 7) ^ srow) << 4;
    const bool zero_pad = a.pad_mode == VP_PAD_ZERO;
    const unsigned ldxb = (unsigned)a.ldx * 2u;
    unsigned rowoff[4], rowfix[4], woff[4];
    int tpos[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wv * 32 + i * 8 + srow;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / a.T_out;
        const int t = mm - b * a.T_out;
        tpos[i] = t * a.stride - a.pad_left;
        rowoff[i] = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.ldx + (unsigned)a.xoff) * 2u : OOB;
        const int traw = tpos[i];
        int ts = traw < 0 ? -traw : traw;
        ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
        const bool inr = traw >= 0 && traw < a.T_in;
        rowfix[i] = (ok && (inr || !zero_pad)) ? rowoff[i] + (unsigned)ts * ldxb + cb : OOB;
        const int n = n0 + wv * 32 + i * 8 + srow;
        woff[i] = n < a.N ? (unsigned)n * (unsigned)a.K * 2u + cb : OOB;
    }

    // SCHED 2 staging roles: waves 0..3 stage the X panel (rows 64 (wv & 3) + 8 i), waves 4..7 the W panel, and
    // each role issues its 8 pieces inside a different half of the K-step (see the loop below)
    const bool w_role = wv >= 4;
    unsigned qoff[8];
    int qpos[8];
    if constexpr (SCHED == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (wv & 3) * 64 + i * 8 + srow;
            if (w_role) {
                const int n = n0 + r;
                qoff[i] = n < a.N ? (unsigned)n * (unsigned)a.K * 2u + cb : OOB;
                qpos[i] = 0;
            } else {
                const int m = m0 + r;
                const bool ok = m < a.M;
                const int mm = ok ? m : 0;
                const int b = mm / a.T_out;
                const int t = mm - b * a.T_out;
                qpos[i] = t * a.stride - a.pad_left;
                const unsigned ro = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.ldx + (unsigned)a.xoff) * 2u : OOB;
                if constexpr (MODE == MODE_1X1) {
                    const int traw = qpos[i];
                    int ts = traw < 0 ? -traw : traw;
                    ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                    const bool inr = traw >= 0 && traw < a.T_in;
                    qoff[i] = (ok && (inr || !zero_pad)) ? ro + (unsigned)ts * ldxb + cb : OOB;
                } else {
                    qoff[i] = ro;
                }
            }
        }
    }
    auto role_piece = [&](int kt, int s, int i) {
        char* dst = smem + s * STAGE2 + (w_role ? T2 * ROWB : 0) + ((wv & 3) * 64 + i * 8) * ROWB;
        const unsigned kb = (unsigned)kt * (unsigned)ROWB;
        if constexpr (MODE == MODE_TAPS_GEN) {
            // this lane's 16-B chunk q of the K axis: tap j = q / (Cin / 8), channel chunk q - j * cpt; q >= KC reads zero
            const int q = kt * 8 + (int)(cb >> 4);
            const bool kv = q < a.KC;
            if (w_role) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)dst, 16, (kv && qoff[i] != OOB) ? qoff[i] + kb : OOB, 0, 0, 0);
            } else {
                const int j = q / a.cpt;
                const unsigned cbyte = (unsigned)(q - j * a.cpt) << 4;
                const int traw = qpos[i] + j * a.dilation;
                int ts = traw < 0 ? -traw : traw;
                ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                const bool inr = traw >= 0 && traw < a.T_in;
                const bool ok = kv && qoff[i] != OOB && (inr || !zero_pad);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, ok ? qoff[i] + (unsigned)ts * ldxb + cbyte : OOB, 0, 0, 0);
            }
        } else if (w_role) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)dst, 16, qoff[i] != OOB ? qoff[i] + kb : OOB, 0, 0, 0);
        } else if constexpr (MODE == MODE_1X1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, qoff[i] != OOB ? qoff[i] + kb : OOB, 0, 0, 0);
        } else {
            const int k0 = kt * 64;
            const int j = k0 / a.Cin;
            const unsigned cbase = (unsigned)(k0 - j * a.Cin) * 2u + cb;
            const int traw = qpos[i] + j * a.dilation;
            int ts = traw < 0 ? -traw : traw;
            ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
            const bool inr = traw >= 0 && traw < a.T_in;
            const bool ok = qoff[i] != OOB && (inr || !zero_pad);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, ok ? qoff[i] + (unsigned)ts * ldxb + cbase : OOB, 0, 0, 0);
        }
    };

    // one DMA piece (8 rows x 128 B of one panel): pieces 0..3 = X rows 8i.., pieces 4..7 = W rows
    auto stage_piece = [&](int kt, int s, int i) {
        char* Xs = smem + s * STAGE2 + wv * (32 * ROWB);
        char* Ws = Xs + T2 * ROWB;
        const unsigned kb = (unsigned)kt * (unsigned)ROWB;
        if (i >= 4) {
            const int w = i - 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)(Ws + w * (8 * ROWB)), 16,
                                                     woff[w] != OOB ? woff[w] + kb : OOB, 0, 0, 0);
        } else if constexpr (MODE == MODE_1X1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(Xs + i * (8 * ROWB)), 16,
                                                     rowfix[i] != OOB ? rowfix[i] + kb : OOB, 0, 0, 0);
        } else {
            const int k0 = kt * 64;                     // Cin % 64 == 0: one tap per K-step
            const int j = k0 / a.Cin;
            const unsigned cbase = (unsigned)(k0 - j * a.Cin) * 2u + cb;
            const int traw = tpos[i] + j * a.dilation;
            int ts = traw < 0 ? -traw : traw;
            ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
            const bool inr = traw >= 0 && traw < a.T_in;
            const bool ok = rowoff[i] != OOB && (inr || !zero_pad);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(Xs + i * (8 * ROWB)), 16,
                                                     ok ? rowoff[i] + (unsigned)ts * ldxb + cbase : OOB, 0, 0, 0);
        }
    };
    auto stage = [&](int kt, int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_piece(kt, s, i);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    [[maybe_unused]] auto compute = [&](int s) {
        const char* Xs = smem + s * STAGE2;
        const char* Ws = Xs + T2 * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#ifdef VP_EXP_NOLDS
            Frag<bf16_t> xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi].v = __builtin_bit_cast(bf16x8, u32x4{(unsigned)s, (unsigned)mi, 3u, 4u});
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni].v = __builtin_bit_cast(bf16x8, u32x4{(unsigned)s, (unsigned)ni, 5u, (unsigned)lane});
#else
            Frag<bf16_t> xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) load_frag(Xs, wm * 64 + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) load_frag(Ws, wn * 128 + ni * 16 + li, ks, g, wf[ni]);
#endif
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(wf[ni], xf[mi], acc[mi][ni]);
        }
    };

    const int KT = a.KT;
    if constexpr (SCHED == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) role_piece(0, 0, i);
        // Role-split schedule: one barrier per K-step as in SCHED 0, but the two waves of a SIMD issue their
        // DMA pieces in DIFFERENT halves of the step (waves 0..3: X panel during the ks = 0 MFMAs; waves 4..7:
        // the L2-resident W panel during the ks = 1 MFMAs).  A wave stalls ~100 cycles per piece at issue (the
        // vector-memory path takes ~64 B/clk/CU); while it does, its SIMD partner -- which has no DMA in this
        // half -- keeps the MFMA pipe fed, instead of both stalling together.
        const int my_half = w_role ? 1 : 0;
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef VP_TIMING
            if (kt == 0) { tk1 = wall_clock64(); ck1 = clock64(); }
#endif
            const bool more = kt + 1 < KT;
            const char* Xs = smem + (kt & 1) * STAGE2;
            const char* Ws = Xs + T2 * ROWB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag<bf16_t> xf[MI], wf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) load_frag(Xs, wm * 64 + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) load_frag(Ws, wn * 128 + ni * 16 + li, ks, g, wf[ni]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (more && ks == my_half) {
                        role_piece(kt + 1, (kt + 1) & 1, 2 * mi);
                        role_piece(kt + 1, (kt + 1) & 1, 2 * mi + 1);
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) mma(wf[ni], xf[mi], acc[mi][ni]);
                }
            }
        }
    } else {
    stage(0, 0);
    if constexpr (SCHED == 0) {
    for (int kt = 0; kt < KT; ++kt) {
        // own DMAs of stage kt have landed; after the barrier so have everyone's, and every wave
        // is done reading the other buffer (stage kt-1), which the next DMAs overwrite
#ifdef VP_TIMING
        const unsigned long long tw0 = wall_clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef VP_TIMING
        const unsigned long long tw1 = wall_clock64();
#endif
        __syncthreads();
#ifdef VP_TIMING
        if (kt == 0) { tk1 = wall_clock64(); ck1 = clock64(); }
        else { twait += tw1 - tw0; tbar += wall_clock64() - tw1; }
#endif
        // the next stage's 8 DMA pieces are spread over this step's MFMA groups: the vector-memory path takes
        // ~64 B/clk/CU, so a burst of all 64 pieces right after the barrier stalls every wave at issue
        const bool more = kt + 1 < KT;
        const char* Xs = smem + (kt & 1) * STAGE2;
        const char* Ws = Xs + T2 * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<bf16_t> xf[MI], wf[NI];
#ifdef VP_EXP_NOLDS
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi].v = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, (unsigned)mi, 3u, 4u});
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni].v = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, (unsigned)ni, 5u, (unsigned)lane});
#else
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) load_frag(Xs, wm * 64 + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) load_frag(Ws, wn * 128 + ni * 16 + li, ks, g, wf[ni]);
#endif
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#ifndef VP_EXP_NODMA
                if (more) stage_piece(kt + 1, (kt + 1) & 1, ks * 4 + mi);
#endif
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma(wf[ni], xf[mi], acc[mi][ni]);
            }
        }
    }
    } else {
    // Ping-pong schedule.  Every K-step is four phases closed by a workgroup barrier; a wave alternates
    // L phases (fetch the 12 fragments of one 32-deep half step from LDS, issue DMAs) and C phases (its 32
    // MFMAs on those fragments).  Waves 4..7 -- the second wave of each SIMD -- run ONE phase behind waves
    // 0..3, so while one wave of a SIMD waits on LDS its partner owns the MFMA pipe:
    //   phase      4kt        4kt+1      4kt+2      4kt+3
    //   waves 0-3  L(kt,0)    C(kt,0)    L(kt,1)    C(kt,1)
    //   waves 4-7  C(kt-1,1)  L(kt,0)    C(kt,0)    L(kt,1)
    // Buffer (kt+1)&1 was last read in phase 4kt-1, so stage kt+1 is issued in phase 4kt (waves 0-3) /
    // 4kt+1 (waves 4-7), waited for (vmcnt) before the barrier that closes phase 4kt+3 and first read in
    // phase 4kt+4.  Raw s_barrier: __syncthreads() would drain the DMAs in flight at every phase.
    // Each role gets its own straight-line loop (one branch at the top): a single loop with per-phase role
    // tests makes hipcc shuffle the 176 live accumulator / fragment registers through scratch.
    Frag<bf16_t> xf[MI], wf[NI];
    auto fetch = [&](int kt, int ks) {
        const char* Xs = smem + (kt & 1) * STAGE2;
        const char* Ws = Xs + T2 * ROWB;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) load_frag(Xs, wm * 64 + mi * 16 + li, ks, g, xf[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) load_frag(Ws, wn * 128 + ni * 16 + li, ks, g, wf[ni]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma(wf[ni], xf[mi], acc[mi][ni]);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef VP_TIMING
    tk1 = wall_clock64();
#endif
    if (wv < 4) {
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
            fetch(kt, 0);
            __builtin_amdgcn_s_barrier();
            mfmas();
            __builtin_amdgcn_s_barrier();
            fetch(kt, 1);
            __builtin_amdgcn_s_barrier();
            mfmas();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                          // phase 4 KT: the late waves' last MFMAs
    } else {
        __builtin_amdgcn_s_barrier();                          // phase 0: the early waves' first fetch
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
            fetch(kt, 0);
            __builtin_amdgcn_s_barrier();
            mfmas();
            __builtin_amdgcn_s_barrier();
            fetch(kt, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mfmas();
            __builtin_amdgcn_s_barrier();
        }
    }
    }
    }
#ifdef VP_TIMING
    const unsigned long long tk2 = wall_clock64();
    const unsigned long long ck2 = clock64();
#endif

    // ------------------------------------------------------------------ epilogue
    // y = act2( bn( act( acc + bias + rowbias ) ) + res );  aux = y + add_in;  psum/psumsq over (y - shift)
    // (conv_gemm_impl.h semantics, no gate).  Written as ROLLED loops over an LDS image of the
    // accumulators: the fully unrolled per-register form of the 128-wide kernel is ~25k instructions for
    // 128 accumulators per lane and ran 15 us per tile on instruction fetch alone.  Per 64-channel half:
    //   A. dump the wave's 64 x 64 f32 accumulators into its own LDS slab (row stride 272 B);
    //   B. 16 iterations: lane = (row j*4 + lane/16, channels 4*(lane%16)..+3): math, 8-B store (16 lanes
    //      cover 128 contiguous bytes of a position), (y - shift) written back to the slab;
    //   C. column sums: lane = one channel, 64 rows, per utterance segment, into red[wm][seg][col].
    constexpr int OROW = 272;
    constexpr int SLABS = 8 * 64 * OROW;
    __syncthreads();                                           // every wave is done reading the K panels
#ifdef VP_TIMING
    const unsigned long long te0 = wall_clock64();
    unsigned long long te1 = 0, te1b = 0;
#endif
    char* slab = smem + wv * (64 * OROW);
    float* red = reinterpret_cast<float*>(smem + SLABS);       // [2 stats][4 wm][2 seg][256 col]
    bf16_t* __restrict__ Y = static_cast<bf16_t*>(a.y);
    bf16_t* __restrict__ Y2 = static_cast<bf16_t*>(a.y2);
    const bf16_t* __restrict__ ADD = static_cast<const bf16_t*>(a.add_in);
    const bf16_t* __restrict__ RES = static_cast<const bf16_t*>(a.res);
    bf16_t* __restrict__ AUX = static_cast<bf16_t*>(a.aux);
    const int mw = m0 + wm * 64;                               // first position of this wave
    const int bfirst = (m0 + (wm >> 1) * 128) / a.T_out;       // first utterance of the 128-row half
    const int q4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4*>(slab + (mi * 16 + li) * OROW + (ni * 16 + g * 4) * 4) = acc[mi][h * 4 + ni];
#ifdef VP_TIMING
        if (h == 0) te1b = wall_clock64();
#endif
        const int nb = n0 + wn * 128 + h * 64 + c4;
        const bool nvalid = nb < a.N;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f}, sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvalid) {
            if (a.bias) load4(a.bias + nb, bias4);
            if (a.bn_scale) load4(a.bn_scale + nb, sc4);
            if (a.bn_shift) load4(a.bn_shift + nb, sh4);
        }
        // FAST rows: the whole 64 x 64 block is inside the problem and only the per-channel terms are
        // active (bias, ReLU, BN affine, ReLU) -- lane = (row 8j + lane/8, channels 8*(lane%8)..+7), one
        // 16-B store per lane, 8 lanes = 128 contiguous bytes of a position.  ~40 VALU per 8 values; the
        // general loop below spends ~25 per VALUE on masks and 64-bit addressing.
        const int nh = n0 + wn * 128 + h * 64;
        const bool fast = mw + 64 <= a.M && nh + 64 <= a.N && !a.rowbias && !RES && !AUX && a.act2 != VP_ACT_TANH && a.act2 != VP_ACT_SILU &&
                          (a.ysplit <= nh || a.ysplit >= nh + 64) && ((a.ldy | a.yoff) & 7) == 0 &&
                          (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
                          (a.ysplit <= nh || (((a.ldy2 | a.y2off) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y2) & 15) == 0));
        if (fast) {
            const int c8 = (lane & 7) * 8, q8 = lane >> 3;
            const int nc = nh + c8;
            float bs[8], sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { bs[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; }
            if (a.bias) { load4(a.bias + nc, bs); load4(a.bias + nc + 4, bs + 4); }
            if (a.bn_scale) { load4(a.bn_scale + nc, sc); load4(a.bn_scale + nc + 4, sc + 4); }
            if (a.bn_shift) { load4(a.bn_shift + nc, sh); load4(a.bn_shift + nc + 4, sh + 4); }
            const float lo1 = a.act == VP_ACT_RELU ? 0.f : -INFINITY;
            const float lo2 = (a.act2 == VP_ACT_RELU || a.act2 == VP_ACT_HARDTANH20) ? 0.f : -INFINITY;
            const float hi2 = a.act2 == VP_ACT_HARDTANH20 ? 20.f : INFINITY;
            bf16_t* dst = Y + (size_t)(mw + q8) * a.ldy + a.yoff + nc;
            const size_t dstep = (size_t)8 * a.ldy;
            const bool split = a.ysplit > nh;
            bf16_t* dst2 = split ? Y2 + (size_t)(mw + q8) * a.ldy2 + a.y2off + nc : nullptr;
            const size_t dstep2 = (size_t)8 * a.ldy2;
            const bool sums = a.psum != nullptr;
            char* cell = slab + q8 * OROW + c8 * 4;
            // two instances of the row loop: without a second activation (conv -> ReLU -> BN, the TDNN block) the clamp pair is
            // dead work -- 16 of ~44 VALU instructions per 8 values in a loop that is VALU-bound
            auto rows = [&](auto clamp2) {
#pragma unroll 2
                for (int j = 0; j < 8; ++j) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(cell);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(cell + 16);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(a0[e] + bs[e], lo1) * sc[e] + sh[e];
                        v[e + 4] = fmaxf(a1[e] + bs[e + 4], lo1) * sc[e + 4] + sh[e + 4];
                        if constexpr (decltype(clamp2)::value) {
                            v[e] = fminf(fmaxf(v[e], lo2), hi2);
                            v[e + 4] = fminf(fmaxf(v[e + 4], lo2), hi2);
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *reinterpret_cast<bf16x8*>(dst) = o;
                    dst += dstep;
                    if (split) { *reinterpret_cast<bf16x8*>(dst2) = o; dst2 += dstep2; }
                    if (sums) {
                        *reinterpret_cast<f32x4*>(cell) = f32x4{v[0] - sh[0], v[1] - sh[1], v[2] - sh[2], v[3] - sh[3]};
                        *reinterpret_cast<f32x4*>(cell + 16) = f32x4{v[4] - sh[4], v[5] - sh[5], v[6] - sh[6], v[7] - sh[7]};
                    }
                    cell += 8 * OROW;
                }
            };
            if (a.act2 == VP_ACT_NONE) rows(std::false_type{});
            else rows(std::true_type{});
        } else {
        int m = mw + q4;
        int b = (m < a.M ? m : a.M - 1) / a.T_out;
        int t = m - b * a.T_out;                               // may run past T_out for rows >= M: never used then
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
            char* cell = slab + (j * 4 + q4) * OROW + c4 * 4;
            const f32x4 av = *reinterpret_cast<const f32x4*>(cell);
            const bool ok = nvalid && m < a.M;
            float v[4];
            float rbias[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok && a.rowbias) load4(a.rowbias + (size_t)b * a.N + nb, rbias);
            if (ok && RES) load4(RES + (size_t)m * a.ld_res + a.res_off + nb, rs);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = av[r] + bias4[r] + rbias[r];
                if (a.act == VP_ACT_RELU) x = fmaxf(x, 0.f);
                x = x * sc4[r] + sh4[r] + rs[r];
                if (a.act2 == VP_ACT_TANH) x = vp_tanh_for<bf16_t>(x);
                else if (a.act2 == VP_ACT_RELU) x = fmaxf(x, 0.f);
                else if (a.act2 == VP_ACT_HARDTANH20) x = fminf(fmaxf(x, 0.f), 20.f);
                else if (a.act2 == VP_ACT_SILU) x = vp_silu_for<bf16_t>(x);
                v[r] = x;
            }
            if (ok) {
                store4(Y + (size_t)m * a.ldy + a.yoff + nb, v);
                if (nb < a.ysplit) store4(Y2 + (size_t)m * a.ldy2 + a.y2off + nb, v);
                if (AUX) {
                    float ad[4];
                    load4(ADD + (size_t)m * a.ld_add + a.add_off + nb, ad);
                    float s4[4] = {v[0] + ad[0], v[1] + ad[1], v[2] + ad[2], v[3] + ad[3]};
                    store4(AUX + (size_t)m * a.ld_aux + a.aux_off + nb, s4);
                }
            }
            if (a.psum)
                *reinterpret_cast<f32x4*>(cell) = ok ? f32x4{v[0] - sh4[0], v[1] - sh4[1], v[2] - sh4[2], v[3] - sh4[3]}
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            m += 4; t += 4;
            while (t >= a.T_out) { t -= a.T_out; ++b; }
        }
        }
#ifdef VP_TIMING
        if (h == 0) te1 = wall_clock64();
#endif
        if (a.psum) {
            // lane = channel h*64 + lane of this wave's 128.  T_out >= 128 > 64 rows: at most one utterance
            // boundary inside the wave's rows, at the wave-uniform row rb
            const int col = wn * 128 + h * 64 + lane;
            const int bb = (mw < a.M ? mw : a.M - 1) / a.T_out;
            const int rb = min(64, (bb + 1) * a.T_out - mw);     // rows [0, rb) belong to utterance bb
            const char* colp = slab + lane * 4;
            float s1 = 0.f, s2 = 0.f;
            int r = 0;
#pragma unroll 4
            for (; r < rb; ++r) {
                const float d = *reinterpret_cast<const float*>(colp + r * OROW);
                s1 += d; s2 += d * d;
            }
            if (bb - bfirst < 2) {
                red[((0 * 4 + wm) * 2 + (bb - bfirst)) * T2 + col] = s1;
                red[((1 * 4 + wm) * 2 + (bb - bfirst)) * T2 + col] = s2;
            }
            if (rb < 64) {
                s1 = 0.f; s2 = 0.f;
#pragma unroll 4
                for (; r < 64; ++r) {
                    const float d = *reinterpret_cast<const float*>(colp + r * OROW);
                    s1 += d; s2 += d * d;
                }
                if (bb + 1 - bfirst < 2) {
                    red[((0 * 4 + wm) * 2 + (bb + 1 - bfirst)) * T2 + col] = s1;
                    red[((1 * 4 + wm) * 2 + (bb + 1 - bfirst)) * T2 + col] = s2;
                }
            }
        }
    }
    if (a.psum) {
        // per 128-row half (= one M-tile of the 128-wide kernel's psum layout): the two waves' partials.
        // T_out >= 128 (host-checked, nseg == 2): a wave's 64 rows touch at most two utterances and
        // flush each (wave, segment) slot at most once; slots never flushed must read as zero.
        __syncthreads();
        const int col = tid & 255, half = tid >> 8;
        if (n0 + col < a.N && m0 + half * 128 < a.M) {
            const int bf = (m0 + half * 128) / a.T_out;
            for (int sgi = 0; sgi < 2; ++sgi) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int wmx = half * 2 + w;
                    const int mlo = m0 + wmx * 64, mhi = min(mlo + 63, a.M - 1);
                    // did wave wmx own rows of utterance bf + sgi?
                    if (mlo < a.M && mlo / a.T_out <= bf + sgi && bf + sgi <= mhi / a.T_out) {
                        s1 += red[((0 * 4 + wmx) * 2 + sgi) * T2 + col];
                        s2 += red[((1 * 4 + wmx) * 2 + sgi) * T2 + col];
                    }
                }
                const size_t o = ((size_t)(tm * 2 + half) * a.nseg + sgi) * a.N + n0 + col;
                a.psum[o] = s1;
                if (a.psumsq) a.psumsq[o] = s2;
            }
        }
    }
#ifdef VP_TIMING
    if (!a.aux && a.add_in) {          // debug build only: per-workgroup phase stamps (100 MHz counter)
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = (unsigned long long*)a.add_in + (size_t)blockIdx.x * 8;
            o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = wall_clock64(); o[4] = te0; o[5] = ck2 - ck1; o[6] = te1; o[7] = (twait << 32) | tbar;
        }
    }
#endif
}

template <int MODE, int SCHED>
int launch256(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int smem = 8 * 64 * 272 + 2 * 4 * 2 * T2 * 4;      // output slabs + column-sum partials (> the K panels)
    static bool attr_set = false;
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm256_kernel<MODE, SCHED>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm256_kernel<MODE, SCHED>), dim3(a.tiles_m * a.tiles_n), dim3(512), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_gemm256");
    return VP_OK;
}

}  // namespace

// args: ConvArgs with tiles_m / tiles_n / group_m already set for 256-wide tiles
int vp_conv_launch256_bf16(vp_ctx* ctx, const void* args, int mode, int sched, hipStream_t st) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(args);
    if (sched == 0) {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 0>(ctx, a, st);
        if (mode == MODE_TAPS) return launch256<MODE_TAPS, 0>(ctx, a, st);
    } else if (sched == 1) {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 1>(ctx, a, st);
        if (mode == MODE_TAPS) return launch256<MODE_TAPS, 1>(ctx, a, st);
    } else {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 2>(ctx, a, st);
        if (mode == MODE_TAPS) return a.Cin % 64 == 0 ? launch256<MODE_TAPS, 2>(ctx, a, st) : launch256<MODE_TAPS_GEN, 2>(ctx, a, st);
    }
    VP_FAIL(ctx, VP_EUNSUP, "conv256: mode %d not built", mode);
}
