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

// Epilogue of one tile, shared by every K-loop schedule.  Entered by all threads with each wave's 64 x 128 accumulators in
// registers, after a barrier behind which the LDS range [ebase, ebase + NW x 64 x 272 + NW x 2 KB) is dead.
//   y = act2( bn( act( acc + bias + rowbias ) ) + res );  aux = y + add_in;  psum/psumsq over (y - shift)
// (conv_gemm_impl.h semantics, no gate).  Written as ROLLED loops over an LDS image of the accumulators: the fully
// unrolled per-register form of the 128-wide kernel is ~25k instructions for 128 accumulators per lane and ran 15 us per
// tile on instruction fetch alone.  Per 64-channel half h of the wave's 128 channels:
//   A. dump 64 x 64 f32 accumulators into the wave's own LDS slab (row stride 272 B: conflict-free both ways);
//   B. rolled row loop -- fast form (whole block inside the problem, per-channel terms only): lane = (row 8j + lane/8,
//      channels 8 (lane%8)..+7), one 16-B store per lane, 8 lanes = 128 contiguous bytes of a position; general form:
//      lane = (row 4j + lane/16, 4 channels), masks / rowbias / residual / aux / tanh / SiLU; (y - shift) goes back to the slab;
//   C. column sums: the fast rows keep 8 partial sums per lane, transposed through the slab (lane = channel adds 8 rows); the
//      general rows re-read their column.
// Measured and NOT kept (round 3, same-session A/B, 512 -> 512 / 1536 -> 1536 launches): the arithmetic on the accumulators
// (lane = position, 4 channels) writing a bf16 image that a copy-out loop stores without arithmetic -- fewer VALU instructions and
// half the LDS bytes on paper, but 63.8 -> 73.5 us / 309 -> 333 us on the 8-wave kernel (57 -> 60 / equal on the 4-wave one): with
// the arithmetic and the stores in separate phases the waves of a workgroup use the VALU and the store path in turns.  Folding the
// bias into the accumulators' initial value likewise lost: hipcc waits vmcnt(0) for an ordinary load's result while LDS-DMAs are
// in flight, so the first MFMA drained the whole prologue.
// NW = waves of the workgroup: 8 (256-row tile, four row groups of two waves) or 4 (the 128-row tile of the two-per-CU kernel).
// TO: bf16 (inference, training forward) or float (the training engine's data-gradient GEMMs: bf16 operands, f32 gradients; the fast
// rows then also take an f32 residual -- the other path's gradient -- from a coalesced load).
template <int NW, typename TO>
__device__ __forceinline__ void epilogue256(const ConvArgs& a, f32x4 (&acc)[4][8], char* ebase, int tid, int tm, int m0, int n0) {
    constexpr int RS = 64;
    constexpr bool HLO = std::is_same<TO, hl_t>::value;           // split bf16 planes out (conv_gemm_impl.h: hl_t)
    constexpr bool F32O = sizeof(TO) == 4 && !HLO;
    constexpr int OROW = 272;
    constexpr int NP = 64 / RS;                                // row passes per 64-channel half
    constexpr int NWM = NW / 2;                                // row groups (wm) of the tile
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, g = lane >> 4;
    char* slab = ebase + wv * (RS * OROW);
    float* red = reinterpret_cast<float*>(ebase + NW * RS * OROW);    // [2 stats][NWM wm][2 seg][256 col]
    TO* __restrict__ Y = static_cast<TO*>(a.y);
    TO* __restrict__ Y2 = static_cast<TO*>(a.y2);
    const TO* __restrict__ ADD = static_cast<const TO*>(a.add_in);
    const TO* __restrict__ RES = static_cast<const TO*>(a.res);
    TO* __restrict__ AUX = static_cast<TO*>(a.aux);
    const int mw = m0 + wm * 64;                               // first position of this wave
    const int bfirst = (m0 + (wm >> 1) * 128) / a.T_out;       // first utterance of the 128-row half
    const int q4 = lane >> 4, c4 = (lane & 15) * 4;
    // T_out >= 128 > 64 rows (host-checked when psum is set): at most one utterance boundary inside the wave's rows, at
    // the wave-uniform row rb -- rows [0, rb) belong to utterance bb
    const int bb = (mw < a.M ? mw : a.M - 1) / a.T_out;
    const int rb = min(64, (bb + 1) * a.T_out - mw);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nb = n0 + wn * 128 + h * 64 + c4;
        const bool nvalid = nb < a.N;
        const int nh = n0 + wn * 128 + h * 64;
        const int c8 = (lane & 7) * 8, q8 = lane >> 3;
        const int nc = nh + c8;
        // per-channel parameters of both lane layouts (general: 4 channels at nb; fast: 8 channels at nc)
        float bias4[4] = {0.f, 0.f, 0.f, 0.f}, sc4[4] = {1.f, 1.f, 1.f, 1.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvalid) {
            if (a.bias) load4(a.bias + nb, bias4);
            if (a.bn_scale) load4(a.bn_scale + nb, sc4);
            if (a.bn_shift) load4(a.bn_shift + nb, sh4);
        }
        constexpr int AL = HLO ? 31 : F32O ? 3 : 7;            // elements per 16 bytes - 1 (hl32: per 128-byte group)
        const bool res_ok = !RES || (F32O && !a.psum && a.act2 == VP_ACT_NONE && ((a.ld_res | a.res_off) & 3) == 0 &&
                                     (reinterpret_cast<uintptr_t>(a.res) & 15) == 0);
        const bool fast_cols = nh + 64 <= a.N && !a.rowbias && res_ok && !AUX && a.act2 != VP_ACT_TANH && a.act2 != VP_ACT_SILU &&
                               (a.ysplit <= nh || a.ysplit >= nh + 64) && ((a.ldy | a.yoff) & AL) == 0 &&
                               (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
                               (a.ysplit <= nh || (((a.ldy2 | a.y2off) & AL) == 0 && (reinterpret_cast<uintptr_t>(a.y2) & 15) == 0));
        float bs[8], sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; }
        if (fast_cols) {
            if (a.bias) { load4(a.bias + nc, bs); load4(a.bias + nc + 4, bs + 4); }
            if (a.bn_scale) { load4(a.bn_scale + nc, sc); load4(a.bn_scale + nc + 4, sc + 4); }
            if (a.bn_shift) { load4(a.bn_shift + nc, sh); load4(a.bn_shift + nc + 4, sh + 4); }
        }
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;      // general rows: column sums of this lane's channel, utterance bb / bb + 1
        // fast rows: sums of this lane's 8 channels over its rows -- p = every row, q = the rows of utterance bb (only when the
        // wave's rows straddle two utterances); reduced over the 8 lanes of a channel group after the passes
        float p1[8], p2[8], q1[8], q2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { p1[e] = 0.f; p2[e] = 0.f; q1[e] = 0.f; q2[e] = 0.f; }
        const bool fastw = fast_cols && mw + 64 <= a.M;        // wave-uniform: every pass of this half takes the fast rows
#pragma unroll
        for (int rp = 0; rp < NP; ++rp) {
            const int mwp = mw + rp * RS;                      // first position of this pass
#pragma unroll
            for (int mi = 0; mi < RS / 16; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *reinterpret_cast<f32x4*>(slab + (mi * 16 + li) * OROW + (ni * 16 + g * 4) * 4) = acc[rp * (RS / 16) + mi][h * 4 + ni];
            if (fastw) {
                const float lo1 = a.act == VP_ACT_RELU ? 0.f : -INFINITY;
                const float lo2 = (a.act2 == VP_ACT_RELU || a.act2 == VP_ACT_HARDTANH20) ? 0.f : -INFINITY;
                const float hi2 = a.act2 == VP_ACT_HARDTANH20 ? 20.f : INFINITY;
                TO* dst = Y + (size_t)(mwp + q8) * a.ldy + a.yoff + nc;
                const size_t dstep = (size_t)8 * a.ldy;
                const bool split = a.ysplit > nh;
                TO* dst2 = split ? Y2 + (size_t)(mwp + q8) * a.ldy2 + a.y2off + nc : nullptr;
                const size_t dstep2 = (size_t)8 * a.ldy2;
                // hl32: the lane's 8 channels are 16 bytes of the hi plane and 16 bytes of the lo plane of their group
                [[maybe_unused]] const int hlo = (nc >> 5) * 128 + (nc & 31) * 2;
                [[maybe_unused]] char* dsth = reinterpret_cast<char*>(Y) + ((size_t)(mwp + q8) * a.ldy + a.yoff) * 4 + hlo;
                [[maybe_unused]] char* dsth2 = split ? reinterpret_cast<char*>(Y2) + ((size_t)(mwp + q8) * a.ldy2 + a.y2off) * 4 + hlo : nullptr;
                [[maybe_unused]] const TO* rsrc = RES ? RES + (size_t)(mwp + q8) * a.ld_res + a.res_off + nc : nullptr;
                [[maybe_unused]] const size_t rstep = (size_t)8 * a.ld_res;
                const bool sums = a.psum != nullptr;
                char* cell = slab + q8 * OROW + c8 * 4;
                // two instances of the row loop: without a second activation (conv -> ReLU -> BN, the TDNN block) the clamp
                // pair is dead work -- 16 of ~44 VALU instructions per 8 values in a loop that is VALU-bound
                // The column sums are taken over r = act(acc + bias), BEFORE the BN affine, when no second activation follows: sum(y - shift)
                // = scale * sum(r), sum((y - shift)^2) = scale^2 * sum(r^2) -- one add (and one fma for the squares, only when
                // asked for) per value instead of sub / add / fma; the scale is applied once per lane after the passes.
                auto rows = [&](auto clamp2, auto straddle, auto wantsq) {
                    constexpr bool PRE = !decltype(clamp2)::value;
                    constexpr bool SQ = decltype(wantsq)::value;
                    int row = rp * RS + q8;                    // of the wave's 64
#pragma unroll 2
                    for (int j = 0; j < RS / 8; ++j) {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cell);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(cell + 16);
                        float v[8], r8[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            r8[e] = fmaxf(a0[e] + bs[e], lo1);
                            r8[e + 4] = fmaxf(a1[e] + bs[e + 4], lo1);
                            v[e] = r8[e] * sc[e] + sh[e];
                            v[e + 4] = r8[e + 4] * sc[e + 4] + sh[e + 4];
                            if constexpr (decltype(clamp2)::value) {
                                v[e] = fminf(fmaxf(v[e], lo2), hi2);
                                v[e + 4] = fminf(fmaxf(v[e + 4], lo2), hi2);
                            }
                        }
                        if constexpr (HLO) {
                            unsigned short hh[8], ll[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) hl_split(v[e], hh[e], ll[e]);
                            const u32x4 oh = u32x4{(unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16),
                                                   (unsigned)hh[4] | ((unsigned)hh[5] << 16), (unsigned)hh[6] | ((unsigned)hh[7] << 16)};
                            const u32x4 ol = u32x4{(unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16),
                                                   (unsigned)ll[4] | ((unsigned)ll[5] << 16), (unsigned)ll[6] | ((unsigned)ll[7] << 16)};
                            *reinterpret_cast<u32x4*>(dsth) = oh;
                            *reinterpret_cast<u32x4*>(dsth + 64) = ol;
                            dsth += dstep * 4;
                            if (split) {
                                *reinterpret_cast<u32x4*>(dsth2) = oh;
                                *reinterpret_cast<u32x4*>(dsth2 + 64) = ol;
                                dsth2 += dstep2 * 4;
                            }
                        } else if constexpr (F32O) {
                            f32x4 o0 = f32x4{v[0], v[1], v[2], v[3]}, o1 = f32x4{v[4], v[5], v[6], v[7]};
                            if (RES) {                     // (no second activation with a residual here: res_ok)
                                o0 += *reinterpret_cast<const f32x4*>(rsrc);
                                o1 += *reinterpret_cast<const f32x4*>(rsrc + 4);
                                rsrc += rstep;
                            }
                            *reinterpret_cast<f32x4*>(dst) = o0;
                            *reinterpret_cast<f32x4*>(dst + 4) = o1;
                            if (split) { *reinterpret_cast<f32x4*>(dst2) = o0; *reinterpret_cast<f32x4*>(dst2 + 4) = o1; dst2 += dstep2; }
                        } else {
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                            *reinterpret_cast<bf16x8*>(dst) = o;
                            if (split) { *reinterpret_cast<bf16x8*>(dst2) = o; dst2 += dstep2; }
                        }
                        dst += dstep;
                        if (sums) {
                            const float mk = row < rb ? 1.f : 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float d = PRE ? r8[e] : v[e] - sh[e];
                                p1[e] += d;
                                if constexpr (SQ) p2[e] += d * d;
                                if constexpr (decltype(straddle)::value) {
                                    const float dm = d * mk;
                                    q1[e] += dm;
                                    if constexpr (SQ) q2[e] += dm * d;
                                }
                            }
                        }
                        row += 8;
                        cell += 8 * OROW;
                    }
                };
                const bool strad = sums && rb < 64;
                if (a.act2 == VP_ACT_NONE) {
                    if (a.psumsq) {
                        if (strad) rows(std::false_type{}, std::true_type{}, std::true_type{});
                        else rows(std::false_type{}, std::false_type{}, std::true_type{});
                    } else {
                        if (strad) rows(std::false_type{}, std::true_type{}, std::false_type{});
                        else rows(std::false_type{}, std::false_type{}, std::false_type{});
                    }
                } else {
                    if (strad) rows(std::true_type{}, std::true_type{}, std::true_type{});
                    else rows(std::true_type{}, std::false_type{}, std::true_type{});
                }
            } else {
                int m = mwp + q4;
                int b = (m < a.M ? m : a.M - 1) / a.T_out;
                int t = m - b * a.T_out;                       // may run past T_out for rows >= M: never used then
#pragma unroll 1
                for (int j = 0; j < RS / 4; ++j) {
                    char* cell = slab + (j * 4 + q4) * OROW + c4 * 4;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(cell);
                    const bool ok = nvalid && m < a.M;
                    float v[4];
                    float rbias[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
                    if (ok && a.rowbias) load4(a.rowbias + (size_t)b * a.N + nb, rbias);
                    if (ok && RES) ld4(RES, (size_t)m * a.ld_res + a.res_off, nb, rs);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = av[r] + bias4[r] + rbias[r];
                        if (a.act == VP_ACT_RELU) x = fmaxf(x, 0.f);
                        x = x * sc4[r] + sh4[r] + rs[r];
                        if (a.act2 == VP_ACT_TANH) x = vp_tanh_for<TO>(x);
                        else if (a.act2 == VP_ACT_RELU) x = fmaxf(x, 0.f);
                        else if (a.act2 == VP_ACT_HARDTANH20) x = fminf(fmaxf(x, 0.f), 20.f);
                        else if (a.act2 == VP_ACT_SILU) x = vp_silu_for<TO>(x);
                        v[r] = x;
                    }
                    if (ok) {
                        st4(Y, (size_t)m * a.ldy + a.yoff, nb, v);
                        if (nb < a.ysplit) st4(Y2, (size_t)m * a.ldy2 + a.y2off, nb, v);
                        if (AUX) {
                            float ad[4];
                            ld4(ADD, (size_t)m * a.ld_add + a.add_off, nb, ad);
                            float s4[4] = {v[0] + ad[0], v[1] + ad[1], v[2] + ad[2], v[3] + ad[3]};
                            st4(AUX, (size_t)m * a.ld_aux + a.aux_off, nb, s4);
                        }
                    }
                    if (a.psum)
                        *reinterpret_cast<f32x4*>(cell) = ok ? f32x4{v[0] - sh4[0], v[1] - sh4[1], v[2] - sh4[2], v[3] - sh4[3]}
                                                             : f32x4{0.f, 0.f, 0.f, 0.f};
                    m += 4; t += 4;
                    while (t >= a.T_out) { t -= a.T_out; ++b; }
                }
            }
            if (a.psum && !fastw) {
                // lane = channel h*64 + lane of this wave's 128; rows of this pass before / after the utterance boundary
                const int rbp = min(RS, max(0, rb - rp * RS));
                const char* colp = slab + lane * 4;
                int r = 0;
#pragma unroll 4
                for (; r < rbp; ++r) {
                    const float d = *reinterpret_cast<const float*>(colp + r * OROW);
                    s1a += d; s2a += d * d;
                }
#pragma unroll 4
                for (; r < RS; ++r) {
                    const float d = *reinterpret_cast<const float*>(colp + r * OROW);
                    s1b += d; s2b += d * d;
                }
            }
        }
        if (a.psum && fastw) {
            // The 8 lanes of a channel group (equal lane & 7) hold partial sums of their 8 channels over disjoint rows.  Transposed
            // through the wave's slab (free after the last pass; one wave's LDS operations retire in order): every lane stores its
            // 8 partials of a quantity as one slab row segment, then lane = channel adds the 8 rows of its column -- 2 stores +
            // 8 loads + 7 adds per quantity instead of 24 cross-lane shuffles, and the result already sits one channel per lane.
            const bool two = rb < 64;                  // wave-uniform: rows [0, rb) belong to utterance bb, the rest to bb + 1
            const bool sq = a.psumsq != nullptr || a.act2 != VP_ACT_NONE;
            char* wr = slab + (lane >> 3) * OROW + (lane & 7) * 32;
            auto put = [&](int q, const float (&v)[8]) {
                *reinterpret_cast<f32x4*>(wr + q * 8 * OROW) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(wr + q * 8 * OROW + 16) = f32x4{v[4], v[5], v[6], v[7]};
            };
            put(0, p1);
            if (sq) put(1, p2);
            if (two) { put(2, q1); if (sq) put(3, q2); }
            const char* rd = slab + lane * 4;
            auto get = [&](int q) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) t += *reinterpret_cast<const float*>(rd + (q * 8 + r) * OROW);
                return t;
            };
            float P1 = get(0), P2 = sq ? get(1) : 0.f, Q1 = 0.f, Q2 = 0.f;
            if (two) { Q1 = get(2); if (sq) Q2 = get(3); }
            if (a.act2 == VP_ACT_NONE && a.bn_scale) {     // sums were taken before the BN affine (see the row loop)
                const float scl = a.bn_scale[nh + lane];
                P1 *= scl; Q1 *= scl; P2 *= scl * scl; Q2 *= scl * scl;
            }
            if (two) { s1a = Q1; s2a = Q2; s1b = P1 - Q1; s2b = P2 - Q2; }
            else { s1a = P1; s2a = P2; }
        }
        if (a.psum) {
            const int col = wn * 128 + h * 64 + lane;
            if (bb - bfirst < 2) {
                red[((0 * NWM + wm) * 2 + (bb - bfirst)) * T2 + col] = s1a;
                red[((1 * NWM + wm) * 2 + (bb - bfirst)) * T2 + col] = s2a;
            }
            if (rb < 64 && bb + 1 - bfirst < 2) {
                red[((0 * NWM + wm) * 2 + (bb + 1 - bfirst)) * T2 + col] = s1b;
                red[((1 * NWM + wm) * 2 + (bb + 1 - bfirst)) * T2 + col] = s2b;
            }
        }
    }
    if (a.psum) {
        // per 128-row half (= one M-tile of the 128-wide kernel's psum layout): the two waves' partials.
        // T_out >= 128 (host-checked, nseg == 2): a wave's 64 rows touch at most two utterances and
        // flush each (wave, segment) slot at most once; slots never flushed are never read.
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
                        s1 += red[((0 * NWM + wmx) * 2 + sgi) * T2 + col];
                        s2 += red[((1 * NWM + wmx) * 2 + sgi) * T2 + col];
                    }
                }
                const size_t o = ((size_t)(tm * (NWM / 2) + half) * a.nseg + sgi) * a.N + n0 + col;
                a.psum[o] = s1;
                if (a.psumsq) a.psumsq[o] = s2;
            }
        }
    }
}

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

#ifdef VP_TIMING
    const unsigned long long te0 = wall_clock64();
#endif
    __syncthreads();                                           // every wave is done reading the K panels
    epilogue256<8, bf16_t>(a, acc, smem, tid, tm, m0, n0);
#ifdef VP_TIMING
    if (!a.aux && a.add_in) {          // debug build only: per-workgroup phase stamps (100 MHz counter)
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = (unsigned long long*)a.add_in + (size_t)blockIdx.x * 8;
            o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = wall_clock64(); o[4] = te0; o[5] = ck2 - ck1; o[6] = te0; o[7] = (twait << 32) | tbar;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Half-tile ring schedule (schedules 4 / 5).
// A K-step is 64 wide (one 128-byte line per tile row -- a ring of 64-byte rows was measured first and fetches every
// line twice: TCP -> TCC read requests 43.6 M vs 22.0 M per MFA launch).  Per K-step t the operands are FOUR half-tiles of
// 128 rows x 128 B = 16 KB: XA / XB = the first / second 32 of every wave row-group's 64 positions, WA / WB = the first /
// second 64 of every wave column-group's 128 channels.  LDS holds two K-steps of them (8 buffers, 128 KB).  A K-step is
// four phases of 16 MFMAs per wave (one quadrant of the wave's 64 x 128 tile, both k-halves):
//   phase    computes    reads into the idle registers     DMA (2 pieces of 8 rows per wave)
//   P1(t)    XA x WA     XB(t)                             XA(t+2) -> the buffer XA(t) left in P4(t-1)
//   P2(t)    XB x WA     WB(t)                             XB(t+2)
//   P3(t)    XB x WB     WA(t+1)                           WB(t+2)
//   P4(t)    XA x WB     XA(t+1)                           WA(t+3)
// Every fragment is read one phase before its first MFMA, every half-tile is staged seven phases before it is read: at
// the end of a phase only the half-tile issued six phases earlier must have landed (own pieces: vmcnt(12); the barrier
// publishes everyone's).  WAR: a buffer is re-staged in the phase after the one that read it; those reads retired
// (lgkmcnt(0)) before the barrier in between.  Raw s_barrier: __syncthreads() would drain the DMAs in flight.
// 96 fragment registers: WA and WB have a set each; XA / XB swap two sets every K-step (XA(t+1) is read while XA(t) is
// still in use, into the set XB(t) just left), hence the loop body is two K-steps = eight phases.
// Steps past K (odd step counts, the tail's prefetches) are out-of-range DMAs = zeros.
constexpr int HT = 128 * ROWB;               // one half-tile: 128 rows x 128 B
constexpr int K_XA = 0, K_XB = 1, K_WA = 2, K_WB = 3;

// PERSIST: one workgroup per CU walks the tile list with stride gridDim.x (a multiple of 8, so a workgroup stays on the
// XCD whose run of the tile order it serves) instead of one workgroup per tile.
// Every wave runs a phase's MFMA-only half first, then its memory half (measured against memory-half-first and against the
// younger waves at priority 1: 314 / 326 / 322 us on the MFA GEMM).  Giving the two waves of a SIMD opposite orders needs two
// copies of the loop: hipcc then spills fragment registers inside it -- scratch reloads wait vmcnt(0) and drain the DMA ring.
template <int MODE, bool PERSIST>
__global__ __launch_bounds__(512) void conv_gemm256_ring_kernel(const ConvArgs a) {
    constexpr int MI = 4, NI = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, g = lane >> 4;
    const int ntiles = a.tiles_m * a.tiles_n;
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

    // staging: wave wv fills rows [16 wv, 16 wv + 16) of every half-tile, 8 rows x 128 B per DMA; lane l lands at row
    // l >> 3, position l & 7 and therefore fetches chunk position ^ row (the read applies the same XOR)
    const int srow = lane >> 3;
    const unsigned cb = (unsigned)((lane & 7) ^ srow) << 4;
    const bool zero_pad = a.pad_mode == VP_PAD_ZERO;
    const unsigned ldxb = (unsigned)a.ldx * 2u;
    unsigned xo[2][2], wo[2][2];             // [half][piece]
    int xp[2][2];
    int tm, m0, n0;
    // MODE_1X1 (host-checked: T_in == T_out, stride 1, no padding -- source row m for output row m): every source offset is a
    // lane part (row-in-piece x row bytes + swizzled chunk, one VGPR per operand) plus a wave-uniform part (tile, piece row,
    // K-step); rows past M / N and K-steps past K land past num_records = zeros.  Eight offset VGPRs and their selects less.
    const unsigned ldwb = (unsigned)a.K * 2u;
    const unsigned xl = (unsigned)srow * ldxb + cb + (unsigned)a.xoff * 2u;
    const unsigned wl = (unsigned)srow * ldwb + cb;
    unsigned xs0 = 0, ws0 = 0;               // scalar: byte offset of the wave's first row of XA / WA in this tile
    constexpr unsigned PAST = 0xf0000000u;
    // tile `bid` of the XCD-aware grouped order (conv_gemm_impl.h) and this wave's DMA source offsets in it
    auto setup = [&](int bid) {
    const int nblk = ntiles;
    const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int gsz = a.group_m * a.tiles_n;
    const int grp = swz / gsz, rem = swz - grp * gsz;
    const int gm = min(a.group_m, a.tiles_m - grp * a.group_m);
    const int tn = rem / gm;
    tm = grp * a.group_m + (rem - tn * gm);
    m0 = tm * T2; n0 = tn * T2;
    if constexpr (MODE == MODE_1X1) {
        xs0 = (unsigned)(m0 + (wv >> 1) * 64 + (wv & 1) * 16) * ldxb;
        ws0 = (unsigned)(n0 + (wv >> 2) * 128 + (wv & 3) * 16) * ldwb;
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wv * 16 + i * 8 + srow;                          // row of the half-tile
            const int m = m0 + (r >> 5) * 64 + h * 32 + (r & 31);
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            const int b = mm / a.T_out;
            const int t = mm - b * a.T_out;
            xp[h][i] = t * a.stride - a.pad_left;
            const unsigned ro = ok ? ((unsigned)(b * a.T_in) * (unsigned)a.ldx + (unsigned)a.xoff) * 2u : OOB;
            if constexpr (MODE == MODE_1X1) {
                const int traw = xp[h][i];
                int ts = traw < 0 ? -traw : traw;
                ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
                const bool inr = traw >= 0 && traw < a.T_in;
                xo[h][i] = (ok && (inr || !zero_pad)) ? ro + (unsigned)ts * ldxb + cb : OOB;
            } else {
                xo[h][i] = ro;
            }
            const int n = n0 + (r >> 6) * 128 + h * 64 + (r & 63);
            wo[h][i] = n < a.N ? (unsigned)n * (unsigned)a.K * 2u + cb : OOB;
        }
    };
    const int KT = a.KT;
    const int KTp = (KT + 1) & ~1;

    // the wave's two pieces of half-tile `kind` of K-step t
    auto issue_piece = [&](int t, int set, int kind, int i) {
        char* dst = smem + (set * 4 + kind) * HT + (wv * 16 + i * 8) * ROWB;
        const bool tv = t < KT;
        const unsigned kb = (unsigned)t * (unsigned)ROWB;
        const int h = kind & 1;
        if constexpr (MODE == MODE_1X1) {
            if (kind >= K_WA) {
                const unsigned so = tv ? ws0 + (unsigned)(h * 64 + i * 8) * ldwb + kb : PAST;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)dst, 16, wl + so, 0, 0, 0);
            } else {
                const unsigned so = tv ? xs0 + (unsigned)(h * 32 + i * 8) * ldxb + kb : PAST;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, xl + so, 0, 0, 0);
            }
        } else if (kind >= K_WA) {
            const unsigned off = wo[h][i];
            bool ok = tv && off != OOB;
            if constexpr (MODE == MODE_TAPS_GEN) ok = ok && (t * 8 + (int)(cb >> 4)) < a.KC;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)dst, 16, ok ? off + kb : OOB, 0, 0, 0);
        } else if constexpr (MODE == MODE_1X1) {
            const unsigned off = xo[h][i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, (tv && off != OOB) ? off + kb : OOB, 0, 0, 0);
        } else {
            int j;
            unsigned cbyte;
            bool kv = true;
            if constexpr (MODE == MODE_TAPS_GEN) {           // this lane's 16-B chunk q of the K axis: tap q / (Cin / 8)
                const int q = t * 8 + (int)(cb >> 4);
                kv = q < a.KC;
                j = q / a.cpt;
                cbyte = (unsigned)(q - j * a.cpt) << 4;
            } else {                                         // Cin % 64 == 0: one tap per K-step
                const int k0 = t * 64;
                j = k0 / a.Cin;
                cbyte = (unsigned)(k0 - j * a.Cin) * 2u + cb;
            }
            const int traw = xp[h][i] + j * a.dilation;
            int ts = traw < 0 ? -traw : traw;
            ts = ts >= a.T_in ? 2 * (a.T_in - 1) - ts : ts;
            const bool inr = traw >= 0 && traw < a.T_in;
            const bool ok = tv && kv && xo[h][i] != OOB && (inr || !zero_pad);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)dst, 16, ok ? xo[h][i] + (unsigned)ts * ldxb + cbyte : OOB, 0, 0, 0);
        }
    };

    // fragment reads: X half-tile rows wm * 32 + mi2 * 16 + li, W half-tile rows wn * 64 + ni4 * 16 + li
    // (eight base registers: operand x k-half x K-step set -- a set is 64 KB, the reach of a ds_read immediate)
    const int sw0 = (g ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
    const char* const bx0 = smem + (wm * 32 + li) * ROWB + sw0;
    const char* const bx1 = smem + (wm * 32 + li) * ROWB + sw1;
    const char* const bw0 = smem + (wn * 64 + li) * ROWB + sw0;
    const char* const bw1 = smem + (wn * 64 + li) * ROWB + sw1;
    const char* const cx0 = bx0 + 4 * HT;
    const char* const cx1 = bx1 + 4 * HT;
    const char* const cw0 = bw0 + 4 * HT;
    const char* const cw1 = bw1 + 4 * HT;
    auto read_x = [&](int set, int kind, Frag<bf16_t> (&f)[4]) {
        const char* S0 = (set ? cx0 : bx0) + kind * HT;
        const char* S1 = (set ? cx1 : bx1) + kind * HT;
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2) {
            f[mi2 * 2 + 0].v = *reinterpret_cast<const bf16x8*>(S0 + mi2 * 16 * ROWB);
            f[mi2 * 2 + 1].v = *reinterpret_cast<const bf16x8*>(S1 + mi2 * 16 * ROWB);
        }
    };
    auto read_w = [&](int set, int kind, Frag<bf16_t> (&f)[8]) {
        const char* S0 = (set ? cw0 : bw0) + kind * HT;
        const char* S1 = (set ? cw1 : bw1) + kind * HT;
#pragma unroll
        for (int ni4 = 0; ni4 < 4; ++ni4) {
            f[ni4 * 2 + 0].v = *reinterpret_cast<const bf16x8*>(S0 + ni4 * 16 * ROWB);
            f[ni4 * 2 + 1].v = *reinterpret_cast<const bf16x8*>(S1 + ni4 * 16 * ROWB);
        }
    };

    f32x4 acc[MI][NI];

    // MFMAs e0 .. e1 of a quadrant's 16, the eight ks = 0 ones first: the two MFMAs of an accumulator are 8 apart
    auto quad_mma = [&](const Frag<bf16_t> (&x)[4], const Frag<bf16_t> (&w)[8], int mi0, int ni0, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int ks = e >> 3, mi2 = (e >> 2) & 1, ni4 = e & 3;
            mma(w[ni4 * 2 + ks], x[mi2 * 2 + ks], acc[mi0 + mi2][ni0 + ni4]);
        }
    };

    // prologue: the eight half-tiles of K-steps 0 and 1 in the order they are read (K-step 0 = set 0 may have been staged
    // during the previous tile's epilogue), then the first two reads
    constexpr int first4[4] = {K_WA, K_XA, K_XB, K_WB};
    auto stage_set = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            issue_piece(s, s, first4[q], 0);
            issue_piece(s, s, first4[q], 1);
        }
    };
    int bid = blockIdx.x;
    if (bid < ntiles) {
        setup(bid);
        stage_set(0);
    }
    while (bid < ntiles) {
#ifdef VP_TIMING
    const unsigned long long tk0 = wall_clock64();
#endif
    // (acc = bias would save the epilogue's add per value, but hipcc waits vmcnt(0) for an ordinary load's result while LDS-DMAs are in
    // flight: the first MFMA then drains the whole prologue -- measured +10 % on the 512 -> 512 layers.  Zero it is.)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage_set(1);
    Frag<bf16_t> xpf[4], xqf[4], waf[8], wbf[8];
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_w(0, K_WA, waf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_piece(2, 0, K_WA, 0);                               // phase "P4(-1)": WA(0)'s buffer is free, XA(0) has landed
    issue_piece(2, 0, K_WA, 1);
    read_x(0, K_XA, xpf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef VP_TIMING
    const unsigned long long tk1 = wall_clock64();
    const unsigned long long ck1 = clock64();
#endif

    // One phase = a memory half (the reads, one or two per MFMA, then the two DMA pieces two MFMAs apart) and an MFMA-only
    // half; the sched_barrier(0) fences pin the groups (left alone, hipcc sinks the reads and drags MFMAs across the s_barrier).
    //   RX: 4 reads of an X half-tile into xr[]; !RX: 8 reads of a W half-tile into wr[]
    //   rset / dset: the K-step set (0 / 1) read from / staged into -- literals at every call site
    auto phase = [&](auto roleB, auto rx, int rset, int rkind, Frag<bf16_t> (&xr)[4], Frag<bf16_t> (&wr)[8],
                     int td, int dset, int dkind, const Frag<bf16_t> (&xc)[4], const Frag<bf16_t> (&wc)[8], int mi0, int ni0) {
        constexpr bool RB_ = decltype(roleB)::value;
        constexpr bool RX = decltype(rx)::value;
        constexpr int NR = RX ? 4 : 8;
        auto mem_half = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RX) read_x(rset, rkind, xr); else read_w(rset, rkind, wr);
            quad_mma(xc, wc, mi0, ni0, 0, 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                       // MFMA, then one (X) or two (W) reads, four times
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (NR == 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(td, dset, dkind, 0);
            __builtin_amdgcn_sched_barrier(0);
            quad_mma(xc, wc, mi0, ni0, 4, 6);
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(td, dset, dkind, 1);
            __builtin_amdgcn_sched_barrier(0);
            quad_mma(xc, wc, mi0, ni0, 6, 8);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (!RB_) {
            mem_half();
            quad_mma(xc, wc, mi0, ni0, 8, 16);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            quad_mma(xc, wc, mi0, ni0, 8, 16);
            mem_half();
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto steps = [&](auto roleB) {
        constexpr std::true_type X{};
        constexpr std::false_type W{};
        for (int t = 0; t < KTp; t += 2) {
            // even K-step t (set 0): XA in xpf, XB -> xqf
            phase(roleB, X, 0, K_XB, xqf, wbf, t + 2, 0, K_XA, xpf, waf, 0, 0);
            phase(roleB, W, 0, K_WB, xqf, wbf, t + 2, 0, K_XB, xqf, waf, 2, 0);
            phase(roleB, W, 1, K_WA, xqf, waf, t + 2, 0, K_WB, xqf, wbf, 2, 4);
            phase(roleB, X, 1, K_XA, xqf, wbf, t + 3, 1, K_WA, xpf, wbf, 0, 4);
            // odd K-step t + 1 (set 1): XA in xqf, XB -> xpf
            phase(roleB, X, 1, K_XB, xpf, wbf, t + 3, 1, K_XA, xqf, waf, 0, 0);
            phase(roleB, W, 1, K_WB, xpf, wbf, t + 3, 1, K_XB, xpf, waf, 2, 0);
            phase(roleB, W, 0, K_WA, xpf, waf, t + 3, 1, K_WB, xpf, wbf, 2, 4);
            phase(roleB, X, 0, K_XA, xpf, wbf, t + 4, 0, K_WA, xqf, wbf, 0, 4);
        }
    };
    steps(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // zero-fill DMAs of the tail still target the buffers
#ifdef VP_TIMING
    const unsigned long long tk2 = wall_clock64();
    const unsigned long long ck2 = clock64();
    const unsigned long long te0 = tk2;
#endif
    const int ctm = tm, cm0 = m0, cn0 = n0;
    const int nbid = PERSIST ? bid + (int)gridDim.x : ntiles;
    __syncthreads();                                           // every wave is done reading the ring
    static_assert(!PERSIST, "the resident-workgroup variant (round 2, schedule 5: slower than one workgroup per tile) is no longer built");
    epilogue256<8, bf16_t>(a, acc, smem, tid, ctm, cm0, cn0);
#ifdef VP_TIMING
    if (!a.aux && a.add_in) {
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = (unsigned long long*)a.add_in + (size_t)bid * 8;
            o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = wall_clock64(); o[4] = te0; o[5] = ck2 - ck1; o[6] = te0; o[7] = 0;
        }
    }
#endif
    bid = nbid;
    }
}

// ------------------------------------------------------------------------------------------------
// Two workgroups per CU (schedule 6): the half-tile ring on a 128-position x 256-channel tile, 4 waves, 80 KB of LDS.
// Why: with one 8-wave workgroup per CU a tile's prologue (first HBM round trip), K-loop and epilogue (128 KB of stores when
// every CU of the chip stores at once) run back to back and nothing overlaps them -- for a K = 512 layer they are 40 % of the
// tile -- and 596 full tiles on 256 CUs are 2.33 rounds.  Here each SIMD hosts one wave of EACH of two independent
// workgroups: one workgroup's barrier waits, prologue and epilogue run under the other's MFMAs, and the last round is made
// of half-size tiles.  Per wave nothing changes (64 x 128 accumulators, quadrant phases, fragments one phase ahead).
// The ring holds seven half-tiles: XA / XB (64 rows, 8 KB) and WA (128 rows, 16 KB) of two K-steps, WB of ONE:
//   phase    computes    reads                DMA                                  pieces per wave
//   P1(t)    XA x WA     XB(t)                XA(t+2)                              2
//   P2(t)    XB x WA     WB(t)                XB(t+2)                              2
//   P3(t)    XB x WB     WA(t+1)              WB(t+1) -> the buffer P2(t) read     4
//   P4(t)    XA x WB     XA(t+1)              WA(t+3)                              4
// WB (weights: always L2-resident) is the one half-tile with a short lead: issued in P3(t-1), waited for at the end of
// P1(t) -- own pieces by vmcnt(6) (WA(t+2) and XA(t+2) were issued after it), the barrier publishes everyone's; every other
// half-tile is older than WB(t) at that wait, so it is the only counted wait of a K-step.
constexpr int HX = 64 * ROWB;                // X half-tile of the 128-row tile
constexpr int HW = 128 * ROWB;               // W half-tile
constexpr int R2_XA0 = 0, R2_XB0 = HX, R2_XA1 = 2 * HX, R2_XB1 = 3 * HX, R2_WA0 = 4 * HX, R2_WA1 = 4 * HX + HW, R2_WB = 4 * HX + 2 * HW;
constexpr int R2_BYTES = 4 * HX + 3 * HW;    // 81,920 B: two workgroups fill the CU's 160 KB

// X3 (split precision, TO = hl_t or float): both operands are hl32 tensors (conv_gemm_impl.h: hl_t) -- a 128-byte row of a K-step is
// ONE group [32 hi | 32 lo], so the DMA ring, the swizzle and the fragment reads are those of the bf16 kernel with the two k-halves
// re-read as (hi, lo); a K-step is 32 deep and a quadrant phase is 24 MFMAs (lo*hi, hi*lo, hi*hi per accumulator) on the same
// fragments: 1.5 x the matrix-core work per staged byte.
template <typename TO, bool X3>
__global__ __launch_bounds__(256, 2) void conv_gemm128x256_ring_kernel(const ConvArgs a) {
    constexpr int MI = 4, NI = 8;
    constexpr unsigned ESB = X3 ? 4u : 2u;     // bytes per element of an operand row
    constexpr int NMMA = X3 ? 24 : 16;         // MFMAs of a quadrant phase
#ifdef VP_TIMING
    const unsigned long long tk0 = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int li = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

    // tile of the XCD-aware grouped order (conv_gemm_impl.h)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int gsz = a.group_m * a.tiles_n;
    const int grp = swz / gsz, rem = swz - grp * gsz;
    const int gm = min(a.group_m, a.tiles_m - grp * a.group_m);
    const int tn = rem / gm;
    const int tm = grp * a.group_m + (rem - tn * gm);
    const int m0 = tm * 128, n0 = tn * T2;

    // staging: a DMA piece is 8 rows x 128 B; lane l lands at row l >> 3, position l & 7 and fetches chunk position ^ row.
    // Wave wv fills rows [16 wv, +16) of an X half-tile (2 pieces) and rows [32 wv, +32) of a W half-tile (4 pieces).
    // A 1x1 conv with T_in == T_out reads row m of x for output row m (host-checked), so every source offset is
    //   (lane part: row-in-piece x row bytes + swizzled chunk)  +  (wave-uniform part: piece row, K-step)
    // -- two VGPRs for all twelve pieces; rows past M / N and K-steps past K are past num_records = zeros, no select.
    const int srow = lane >> 3;
    const unsigned cb = (unsigned)((lane & 7) ^ srow) << 4;
    const unsigned ldxb = (unsigned)a.ldx * ESB, ldwb = (unsigned)a.K * ESB;
    const unsigned xl = (unsigned)srow * ldxb + cb + (unsigned)a.xoff * ESB;
    const unsigned wl = (unsigned)srow * ldwb + cb;
    const unsigned xs0 = (unsigned)(m0 + (wv >> 1) * 64 + (wv & 1) * 16) * ldxb;      // scalar: the wave's first row of XA
    const unsigned ws0 = (unsigned)(n0 + (wv >> 1) * 128 + (wv & 1) * 32) * ldwb;     // scalar: the wave's first row of WA
    constexpr unsigned PAST = 0xf0000000u;                                           // + lane part: past any operand (host-checked)
    const int KT = a.KT;
    const int KTp = (KT + 1) & ~1;

    // piece i of half h (A = 0 / B = 1) of K-step t into the ring buffer at byte offset `buf`
    auto issue_x = [&](int t, int buf, int h, int i) {
        const unsigned so = t < KT ? xs0 + (unsigned)(h * 32 + i * 8) * ldxb + (unsigned)t * (unsigned)ROWB : PAST;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(smem + buf + (wv * 16 + i * 8) * ROWB), 16, xl + so, 0, 0, 0);
    };
    auto issue_w = [&](int t, int buf, int h, int i) {
        const unsigned so = t < KT ? ws0 + (unsigned)(h * 64 + i * 8) * ldwb + (unsigned)t * (unsigned)ROWB : PAST;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)(smem + buf + (wv * 32 + i * 8) * ROWB), 16, wl + so, 0, 0, 0);
    };

    // fragment reads: X half-tile rows wm * 32 + mi2 * 16 + li, W half-tile rows wn * 64 + ni4 * 16 + li
    const int sw0 = (g ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
    const char* const bx0 = smem + (wm * 32 + li) * ROWB + sw0;
    const char* const bx1 = smem + (wm * 32 + li) * ROWB + sw1;
    const char* const bw0 = smem + (wn * 64 + li) * ROWB + sw0;
    const char* const bw1 = smem + (wn * 64 + li) * ROWB + sw1;
    auto read_x = [&](int buf, Frag<bf16_t> (&f)[4]) {
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2) {
            f[mi2 * 2 + 0].v = *reinterpret_cast<const bf16x8*>(bx0 + buf + mi2 * 16 * ROWB);
            f[mi2 * 2 + 1].v = *reinterpret_cast<const bf16x8*>(bx1 + buf + mi2 * 16 * ROWB);
        }
    };
    auto read_w = [&](int buf, Frag<bf16_t> (&f)[8]) {
#pragma unroll
        for (int ni4 = 0; ni4 < 4; ++ni4) {
            f[ni4 * 2 + 0].v = *reinterpret_cast<const bf16x8*>(bw0 + buf + ni4 * 16 * ROWB);
            f[ni4 * 2 + 1].v = *reinterpret_cast<const bf16x8*>(bw1 + buf + ni4 * 16 * ROWB);
        }
    };

    f32x4 acc[MI][NI];
    // (acc = bias would save the epilogue's add per value, but hipcc waits vmcnt(0) for an ordinary load's result while LDS-DMAs are in
    // flight: the first MFMA then drains the whole prologue -- measured +10 % on the 512 -> 512 layers.  Zero it is.)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto quad_mma = [&](const Frag<bf16_t> (&x)[4], const Frag<bf16_t> (&w)[8], int mi0, int ni0, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const int ks = e >> 3, mi2 = (e >> 2) & 1, ni4 = e & 3;
            if constexpr (X3) {                                 // fragment [.. + 0] = hi plane, [.. + 1] = lo plane; cross terms first
                const int kw = ks == 0 ? 1 : 0, kx = ks == 1 ? 1 : 0;
                mma(w[ni4 * 2 + kw], x[mi2 * 2 + kx], acc[mi0 + mi2][ni0 + ni4]);
            } else {
                mma(w[ni4 * 2 + ks], x[mi2 * 2 + ks], acc[mi0 + mi2][ni0 + ni4]);
            }
        }
    };

    // prologue: the seven half-tiles of K-steps 0 and 1 in the order they are read, 20 pieces per wave
    Frag<bf16_t> xpf[4], xqf[4], waf[8], wbf[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(0, R2_WA0, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_x(0, R2_XA0, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_x(0, R2_XB0, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(0, R2_WB, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(1, R2_WA1, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_x(1, R2_XA1, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_x(1, R2_XB1, 1, i);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // WA(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_w(R2_WA0, waf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");          // XA(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_w(2, R2_WA0, 0, i);      // phase "P4(-1)": every wave has read WA(0)
    read_x(R2_XA0, xpf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // XB(0): WB, WA(1), XA(1), XB(1), WA(2) were issued after it
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // One phase: a memory half (the reads, one or two per MFMA, then the DMA pieces two MFMAs apart) and an MFMA-only half.
    //   RX: 4 reads of an X half-tile (buffer rbuf) into xr[], 2 DMA pieces of an X half-tile; !RX: 8 reads of a W half-tile
    //   into wr[], 4 DMA pieces of a W half-tile.  dh = half (A = 0 / B = 1) of the staged half-tile, td its K-step.
    auto phase = [&](auto rx, int rbuf, Frag<bf16_t> (&xr)[4], Frag<bf16_t> (&wr)[8], auto dx, int td, int dbuf, int dh,
                     const Frag<bf16_t> (&xc)[4], const Frag<bf16_t> (&wc)[8], int mi0, int ni0, auto tight) {
        constexpr bool RX = decltype(rx)::value;
        constexpr bool DX = decltype(dx)::value;
        constexpr int NPC = DX ? 2 : 4;
        __builtin_amdgcn_sched_barrier(0);
        quad_mma(xc, wc, mi0, ni0, 4 + 2 * NPC, NMMA);          // MFMA-only half first
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RX) read_x(rbuf, xr); else read_w(rbuf, wr);
        quad_mma(xc, wc, mi0, ni0, 0, 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                           // MFMA, then one (X) or two (W) reads, four times
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (RX) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            if constexpr (DX) issue_x(td, dbuf, dh, i); else issue_w(td, dbuf, dh, i);
            __builtin_amdgcn_sched_barrier(0);
            quad_mma(xc, wc, mi0, ni0, 4 + 2 * i, 6 + 2 * i);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (decltype(tight)::value) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr std::true_type X{};
    constexpr std::false_type W{};
    constexpr std::true_type TIGHT{};
    constexpr std::false_type LOOSE{};
#ifdef VP_TIMING
    const unsigned long long tk1 = wall_clock64();
    const unsigned long long ck1 = clock64();
#endif
    for (int t = 0; t < KTp; t += 2) {
        // even K-step t (set 0): XA in xpf, XB -> xqf
        phase(X, R2_XB0, xqf, wbf, X, t + 2, R2_XA0, 0, xpf, waf, 0, 0, TIGHT);
        phase(W, R2_WB, xqf, wbf, X, t + 2, R2_XB0, 1, xqf, waf, 2, 0, LOOSE);
        phase(W, R2_WA1, xqf, waf, W, t + 1, R2_WB, 1, xqf, wbf, 2, 4, LOOSE);
        phase(X, R2_XA1, xqf, wbf, W, t + 3, R2_WA1, 0, xpf, wbf, 0, 4, LOOSE);
        // odd K-step t + 1 (set 1): XA in xqf, XB -> xpf
        phase(X, R2_XB1, xpf, wbf, X, t + 3, R2_XA1, 0, xqf, waf, 0, 0, TIGHT);
        phase(W, R2_WB, xpf, wbf, X, t + 3, R2_XB1, 1, xpf, waf, 2, 0, LOOSE);
        phase(W, R2_WA0, xpf, waf, W, t + 2, R2_WB, 1, xpf, wbf, 2, 4, LOOSE);
        phase(X, R2_XA0, xpf, wbf, W, t + 4, R2_WA0, 0, xqf, wbf, 0, 4, LOOSE);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // zero-fill DMAs of the tail still target the buffers
#ifdef VP_TIMING
    const unsigned long long tk2 = wall_clock64();
    const unsigned long long ck2 = clock64();
#endif
    __syncthreads();                                           // every wave is done reading the ring
    epilogue256<4, TO>(a, acc, smem, tid, tm, m0, n0);
#ifdef VP_TIMING
    if (!a.aux && a.add_in) {          // debug build only: per-workgroup phase stamps (100 MHz counter)
        __syncthreads();
        if (tid == 0) {
            unsigned long long* o = (unsigned long long*)a.add_in + (size_t)blockIdx.x * 8;
            o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = wall_clock64(); o[4] = tk2; o[5] = ck2 - ck1; o[6] = tk2; o[7] = 0;
        }
    }
#endif
}

template <typename TO, bool X3>
int launch128x256_ring(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int smem = R2_BYTES;                             // >= the epilogue's 4 x 64 x 272 images + 8 KB of column-sum partials
    static_assert(4 * 64 * 272 + 2 * 2 * 2 * T2 * 4 <= R2_BYTES, "epilogue image must fit the ring");
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm128x256_ring_kernel<TO, X3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm128x256_ring_kernel<TO, X3>), dim3(a.tiles_m * a.tiles_n), dim3(256), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_gemm128x256_ring");
    return VP_OK;
}

template <int MODE>
int launch256_ring(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int smem = 8 * 64 * 272 + 2 * 4 * 2 * T2 * 4;      // output slabs + column-sum partials (> the 128 KB ring)
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm256_ring_kernel<MODE, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_gemm256_ring_kernel<MODE, false>), dim3(a.tiles_m * a.tiles_n), dim3(512), smem, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_gemm256_ring");
    return VP_OK;
}

template <int MODE, int SCHED>
int launch256(vp_ctx* ctx, const ConvArgs& a, hipStream_t st) {
    constexpr int smem = 8 * 64 * 272 + 2 * 4 * 2 * T2 * 4;      // output slabs + column-sum partials (> the K panels)
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
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

// split precision: x and w hl32 tensors (1x1 layer, source row m for output row m), hl32 or f32 out; args as for schedule 6
int vp_conv_launch_ring_x3(vp_ctx* ctx, const void* args, int out_f32, hipStream_t st) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(args);
    return out_f32 ? launch128x256_ring<float, true>(ctx, a, st) : launch128x256_ring<hl_t, true>(ctx, a, st);
}

// args: ConvArgs with tiles_m / tiles_n / group_m already set for 256-wide tiles
// out_f32: bf16 operands, f32 output (the training engine's data-gradient GEMMs) -- schedule 6 only
int vp_conv_launch256_bf16(vp_ctx* ctx, const void* args, int mode, int sched, int out_f32, hipStream_t st) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(args);
    if (out_f32 && (sched != 5 || mode != MODE_1X1)) VP_FAIL(ctx, VP_EUNSUP, "conv256: f32 output needs the 128 x 256 schedule on a 1x1 layer");
    if (sched == 0) {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 0>(ctx, a, st);
        if (mode == MODE_TAPS) return launch256<MODE_TAPS, 0>(ctx, a, st);
    } else if (sched == 1) {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 1>(ctx, a, st);
        if (mode == MODE_TAPS) return launch256<MODE_TAPS, 1>(ctx, a, st);
    } else if (sched == 5) {
        // two workgroups per CU on 128 x 256 tiles (1x1 layers); args carry tiles_m in 128-row units
        if (mode == MODE_1X1) return out_f32 ? launch128x256_ring<float, false>(ctx, a, st) : launch128x256_ring<bf16_t, false>(ctx, a, st);
    } else if (sched == 3 || sched == 4) {
        // half-tile ring: one workgroup per tile (3) or resident workgroups (4)
        if (mode == MODE_1X1) return launch256_ring<MODE_1X1>(ctx, a, st);      // (4 = resident workgroups: no longer built, runs as 3)
        // tapped convs stay on the two-stage schedule: their per-piece tap / reflect arithmetic pushes the ring loop past
        // 256 VGPRs (scratch reloads inside the loop wait vmcnt(0) and drain the ring)
        if (mode == MODE_TAPS) return a.Cin % 64 == 0 ? launch256<MODE_TAPS, 2>(ctx, a, st) : launch256<MODE_TAPS_GEN, 2>(ctx, a, st);
    } else {
        if (mode == MODE_1X1) return launch256<MODE_1X1, 2>(ctx, a, st);
        if (mode == MODE_TAPS) return a.Cin % 64 == 0 ? launch256<MODE_TAPS, 2>(ctx, a, st) : launch256<MODE_TAPS_GEN, 2>(ctx, a, st);
    }
    VP_FAIL(ctx, VP_EUNSUP, "conv256: mode %d not built", mode);
}
