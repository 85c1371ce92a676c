// Weight gradient of the WIDE 1x1 layers (ECAPA tdnn1 / tdnn2: 512 x 512, MFA: 1536 x 1536; round 5: ASP's 128 x 1536 and 1536 x 128) on bf16 operands:
//     dW[n][c] = sum_m dz[m][n] * x[m][c]            (Conv1D backward of models/utils.py:65-93, batch-norm blocks of ecapa_tdnn.py)
// The reduction index is the ROW of both operands, so both MFMA operands are "k-major" in memory.  conv_wgrad_amp_kernel (train_ops.hip)
// transposes them through registers (bit surgery per element, 128 x 128 tiles: on the MFA layer every operand byte is fetched twelve
// times, 816 us); here
//   * tiles are 256 (n) x 256 (c): eight waves of 64 x 128, one workgroup per CU -- the most a CU's accumulators hold; operands are
//     fetched Cout / 256 resp. Cin / 256 times;
//   * the operand rows go to LDS AS THEY LIE in memory (LDS-DMA, 16 B per lane, no VGPRs, no VALU), 32 rows per stage, four stages in
//     flight (128 KB), one counted s_waitcnt vmcnt + one barrier per stage;
//   * fragments come out of that row-major image with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group
//     COLUMN i of the 4 (rows) x 16 (columns) block the group's sixteen 8-byte addresses describe (lanes 4 r .. 4 r + 3 address row r) --
//     two reads = the eight k of a 16x16x32 operand.  A and B are read the same way, so the k order inside an MFMA is whatever the
//     hardware makes it for both.
// LDS image of a stage: [dz 32 rows x 512 B | x 32 rows x 512 B].  The 32-byte segment s (16 columns) of row r sits at segment
// s ^ f(r), f(r) = (r & 3) | ((r >> 3) & 1) << 2: the 32 lanes the LDS serves together read rows {8 g .. 8 g + 3} of two g, one segment
// each -- eight different f, 256 distinct bytes.  The DMA lands lane-linear, so the swizzle is applied to the SOURCE chunk.
// Rows past the split's end and stages past its last are beyond the buffer descriptors' num_records: they arrive as zeros.
// Partials [split][Cout][Cin] f32 are summed in fixed order by the caller (sum_partials_wide_kernel): bit-reproducible, no atomics.
#include "common.h"

namespace {

constexpr int WT = 256;                    // tile edge of dW
constexpr int WT_KS = 32;                  // operand rows per stage = one MFMA k
constexpr int WT_ROWB = 512;               // bytes of an LDS row (256 bf16)
constexpr int WT_HALF = WT_KS * WT_ROWB;   // one operand of a stage
constexpr int WT_STAGE = 2 * WT_HALF;
constexpr int WT_NST = 4;
constexpr int WT_SMEM = WT_NST * WT_STAGE; // 131,072 B

typedef __attribute__((address_space(3))) void* wt_lds_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradTrArgs {
    const bf16_t* x; const bf16_t* dz; float* part;
    int ldx, lddz, M, N, K, rows_per_split;
};

__device__ __forceinline__ bf16x8 wt_frag(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * WT_ROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_tr256_kernel(const WgradTrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wn = wv & 3, wc = wv >> 2;                                 // the wave's 64 outputs (n) x 128 inputs (c) of the tile
    // Workgroup -> (split, n tile, c tile), XCD-aware: the hardware deals consecutive workgroup ids round-robin to the eight XCDs, so id
    // bid runs logical tile (bid & 7) x (blocks per XCD) + (bid >> 3) -- an XCD's resident workgroups are CONSECUTIVE logical tiles:
    // the tiles of one or two row splits, which share their operand rows in that XCD's L2.  (With the plain order every XCD held tiles
    // of every split: 2.8 GB came over the fabric for the MFA layer's 0.47 GB of operands, 5.2 TB/s, and that was the kernel's time.)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qq = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int tx = (a.K + WT - 1) / WT, ty = (a.N + WT - 1) / WT;       // (N or K = 128: one half-used tile on that side, see the host)
    const int split = swz / (tx * ty), rem = swz - split * (tx * ty);
    const int nb = (rem / tx) * WT, cb = (rem % tx) * WT;
    const int m_begin = split * a.rows_per_split;
    const int rows = min(a.M, m_begin + a.rows_per_split) - m_begin;
    float* out = a.part + (size_t)split * a.N * a.K;
    if (rows <= 0) return;                                               // (the host sizes the splits so that none is empty)
    const unsigned lddzb = (unsigned)a.lddz * 2u, ldxb = (unsigned)a.ldx * 2u;
    // A 128-column operand (ASP's attention TDNN: 128 outputs; its logits conv: 128 inputs) takes the same 256-wide tile: the descriptor
    // ends with the operand's LAST valid element, so the missing columns of the last row read as zeros and those of every other row read
    // the next row's first columns -- finite values that only reach accumulators the store below leaves out
    const int ncn = min(WT, a.N - nb), ncc = min(WT, a.K - cb);
    const __amdgpu_buffer_rsrc_t dzr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(a.dz + (size_t)m_begin * a.lddz + nb), 0, (unsigned)(((size_t)(rows - 1) * a.lddz + ncn) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(a.x + (size_t)m_begin * a.ldx + cb), 0, (unsigned)(((size_t)(rows - 1) * a.ldx + ncc) * 2), 0x00020000);

    // staging: a DMA instruction fills two LDS rows (64 lanes x 16 B); wave wv owns rows 4 wv .. 4 wv + 3 of both operands
    unsigned vdz[2], vx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rl = 2 * k + (lane >> 5), r = 4 * wv + rl;
        const int f = (r & 3) | (((r >> 3) & 1) << 2);
        const unsigned chunk = (unsigned)((lane & 31) ^ (f << 1)) << 4;
        vdz[k] = (unsigned)r * lddzb + chunk;
        vx[k] = (unsigned)r * ldxb + chunk;
    }
    const int KT = (rows + WT_KS - 1) / WT_KS;
    auto issue = [&](int t, int slot) {
        char* st = wsm + slot * WT_STAGE + 4 * wv * WT_ROWB;
        // stages past the last: any offset past num_records (the row term alone is, for t >= KT)
        const unsigned rdz = (unsigned)t * (WT_KS * lddzb), rx = (unsigned)t * (WT_KS * ldxb);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dzr, (wt_lds_t)(st + k * 2 * WT_ROWB), 16, vdz[k] + rdz, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (wt_lds_t)(st + WT_HALF + k * 2 * WT_ROWB), 16, vx[k] + rx, 0, 0, 0);
        }
    };

    // fragment addresses inside a stage: row 8 g + (i >> 2) (+ 4 for the second read), segment (block ^ f), 8 bytes at (i & 3)
    const int fl = (i >> 2) | ((g & 1) << 2);
    const int rowoff = (8 * g + (i >> 2)) * WT_ROWB + (i & 3) * 8;
    int aoff[4], boff[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) aoff[p] = rowoff + (((wn * 4 + p) ^ fl) << 5);
#pragma unroll
    for (int q = 0; q < 8; ++q) boff[q] = WT_HALF + rowoff + (((wc * 8 + q) ^ fl) << 5);

    f32x4 acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline inside every wave: iteration t multiplies the fragments of stage t (read during iteration t - 1) while the
    // transposing reads of stage t + 1 are in flight; the DMA runs three stages ahead of the reads.  (Reads and MFMAs of ONE stage back
    // to back left the matrix cores idle through every read burst: all eight waves sit in the same phase between two barriers.)
    auto read = [&](int slot, bf16x8 (&af)[4], bf16x8 (&bf)[8]) {
        const char* st = wsm + slot * WT_STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) af[p] = wt_frag(st + aoff[p]);
#pragma unroll
        for (int q = 0; q < 8; ++q) bf[q] = wt_frag(st + boff[q]);
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[8]) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p], bf[q], acc[p][q], 0, 0, 0);
    };
    // top of iteration t: stage t + 1 has landed when at most the two younger stages' pieces (4 each) are outstanding; past the barrier
    // everyone's pieces of it are in and everyone holds stage t in registers (lgkmcnt: its reads, issued a whole MFMA block ago, have
    // returned -- a raw s_barrier does not wait for them), so stage t's slot takes stage t + 4
    auto top = [&](int t, int slot) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(t + 4, slot);
    };
    bf16x8 afA[4], bfA[8], afB[4], bfB[8];
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read(0, afA, bfA);
    int t = 0;
    for (; t + 1 < KT; t += 2) {                                         // slots: stage t in t & 3
        top(t, t & 3);
        read((t + 1) & 3, afB, bfB);
        mma(afA, bfA);
        top(t + 1, (t + 1) & 3);
        read((t + 2) & 3, afA, bfA);                                     // (stage KT, when t + 2 == KT: zeros, not used)
        mma(afB, bfB);
    }
    if (t < KT) mma(afA, bfA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the zero-fill DMAs of the stages past the end

    const int n0 = nb + wn * 64, c0 = cb + wc * 128;
    if (n0 >= a.N || c0 >= a.K) return;                                 // (wave-uniform: the unused half of a 128-column side)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(n0 + p * 16 + g * 4 + r) * a.K + c0 + q * 16 + i] = acc[p][q][r];
}

}  // namespace

// rows per split of the 256-tile kernel for an (N, K) layer over M rows: one workgroup per CU
int vp_wgrad_tr256_splits(long long M, int N, int K) {
    const int tiles = ((N + WT - 1) / WT) * ((K + WT - 1) / WT);
    int S = 256 / tiles;
    if (S < 1) S = 1;
    if ((long long)S * WT_KS > M) S = (int)((M + WT_KS - 1) / WT_KS);
    long long rps = (M + S - 1) / S;
    rps = (rps + WT_KS - 1) / WT_KS * WT_KS;
    return (int)((M + rps - 1) / rps);
}

// part: [splits][N][K] f32.  VP_EUNSUP when the layer is not covered (the caller runs the 128-tile kernel).
int vp_wgrad_tr256_bf16(vp_ctx* ctx, const void* x, int ldx, int xoff, const void* dz, int lddz, long long M, int N, int K, float* part,
                        int* splits_out, hipStream_t st) {
    // sides: multiples of 256, or exactly 128 (a half-used tile: wave columns wn < 2 / wc < 1 hold the outputs, the rest is discarded --
    // 64 x 128 per wave, so 128 is the one narrower width whose valid part is whole waves on either side)
    const bool n_ok = N % WT == 0 || N == 128, k_ok = K % WT == 0 || K == 128;
    if (N <= 0 || K <= 0 || !n_ok || !k_ok || (N == 128 && K == 128) || (ldx | xoff | lddz) % 8 || M < WT_KS) return VP_EUNSUP;
    // small problems: one workgroup per CU leaves each split a handful of stages and the partial sums cost more than the GEMM
    // (512 x 512 over 9536 rows: 37 us here against 28 us on the 128-tile kernel; 1536 x 1536 over the same rows: 78 against 101)
    // (a 128-column side executes a 256-wide tile: counted as such -- ASP's 128 x 1536 over 76 288 rows is 2.9998e10 flop as stated)
    const int Ne = N < WT ? WT : N, Ke = K < WT ? WT : K;
    if (2.0 * (double)M * Ne * Ke < 3e10) return VP_EUNSUP;
    if (((uintptr_t)x | (uintptr_t)dz) & 15) return VP_EUNSUP;
    const int S = vp_wgrad_tr256_splits(M, N, K);
    long long rps = (M + S - 1) / S;
    rps = (rps + WT_KS - 1) / WT_KS * WT_KS;
    if (rps * (long long)(lddz > ldx ? lddz : ldx) * 2 >= 0x70000000LL) return VP_EUNSUP;      // 32-bit buffer offsets
    static bool attr_dev[64] = {};                    // the attribute is per DEVICE (a process may drive several GPUs)
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tr256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WT_SMEM));
        attr_set = true;
    }
    WgradTrArgs a;
    a.x = static_cast<const bf16_t*>(x) + xoff; a.dz = static_cast<const bf16_t*>(dz); a.part = part;
    a.ldx = ldx; a.lddz = lddz; a.M = (int)M; a.N = N; a.K = K; a.rows_per_split = (int)rps;
    hipLaunchKernelGGL(conv_wgrad_tr256_kernel, dim3(((K + WT - 1) / WT) * ((N + WT - 1) / WT) * S), dim3(512), WT_SMEM, st, a);
    VP_LAUNCH_CHECK(ctx, "conv_wgrad_tr256");
    *splits_out = S;
    return VP_OK;
}
