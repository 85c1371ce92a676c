// conv1d / conv2d as an implicit GEMM on the gfx950 matrix cores, with the layer epilogue fused in.
//
// Replaces per launch: Conv1d.forward (ppvector/models/utils.py:65-93) / nn.Conv1D (tdnn.py:13-21,
// campplus.py:53-58,78-86,126) / nn.Conv2D (campplus.py:216-236,254-260) -> activation ->
// BatchNorm eval (utils.py:96-119, :147-148), the Res2Net hand-off x_{i+1} + y_i (ecapa_tdnn.py:36-47),
// the time sums SEBlock / ASP need (ecapa_tdnn.py:69-78, pooling.py:97-104), CAM++'s pre-activation
// BN-ReLU on the layer input (campplus.py:137-143,186-189), its context gate (campplus.py:92-94) and the
// ResBlock shortcut add (campplus.py:238-243).
//
// Layout: activations are position-major (B*T[*F], C) -- K (channels) contiguous for BOTH operands,
// so a conv tap is a row shift (reflect / zero padding handled in the row index), never an im2col.
//   M = B*T_out[*F_out] rows,  N = Cout,  K = taps*Cin  (k = tap*Cin + channel).
// Tile: 128 x {128, 64, 32} per 256-thread workgroup (4 waves), K staged 128 B per row per stage
// (64 bf16 / 32 f32), double-buffered LDS (XOR swizzle of the 16-B chunk index by row & 7: no bank
// conflicts -- SQ_LDS_BANK_CONFLICT = 0 measured) fed by a TWO-deep register prefetch of buffer
// loads (wave-uniform SRDs, out-of-range offsets for everything that must read as zero).
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16 engine) or v_mfma_f32_16x16x4_f32 (exact-f32 engine), weights as
// the A operand and activations as the B operand so each lane ends up with 4 CONSECUTIVE output
// channels of one position: 8/16-byte epilogue loads and stores, float4 parameter reads.
// Roofline: MFMA-bound (dense contraction); algorithmic flops = 2*M*N*K per launch.
#include "common.h"

// mirrors of the kernel-side definitions in conv_gemm_impl.h (kept in one header there; the host
// only needs the struct layout and the mode ids)
#include "conv_gemm_impl.h"

int vp_conv_launch_bf16_bf16(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_bf16_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_f32_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_amp_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_x3_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_x3w_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_x3_hl(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_hl_hl(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st);
int vp_conv_launch_ring_x3(vp_ctx* ctx, const void* args, int out_f32, hipStream_t st);
int vp_conv_launch256_bf16(vp_ctx* ctx, const void* args, int mode, int sched, int out_f32, hipStream_t st);

// Schedule of the 256-wide kernel: 0 pins the 128-wide kernel; 1 = interleaved DMA, 2 = ping-pong phases, 3 = role-split
// DMA (two 64 KB stages; the tapped convs always take it), 4 = half-tile ring, 5 = half-tile ring with resident workgroups.
// 6 = half-tile ring on 128 x 256 tiles, two 4-wave workgroups per CU (1x1 layers; the rest as 4), 7 = 6 for K <= 1024, else 4.
// Default (-1) = 6.  Measured on MI355X, 1536 -> 1536 / 512 -> 512 launches at 256 x 298 rows (round 3, one session, with the fused
// time sums): 4: 322 / 70.6 us, 5: 362 / 83, 6: 332 / 64.4; without sums 4: 313 / 62.0, 6: 320 / 55.2.  Inside the two-stream step
// (bench.py) 6 for every 1x1 layer beats 7: 1.282 vs 1.301 ms (4: 1.313) -- a 4-wave workgroup with 80 KB of LDS leaves half a CU to
// the other launch sequence's memory-bound kernels.
// VPMI_CONV256 presets it; vp_conv256_select() switches at run time (A/B in one process).
static int g_conv256 = -2;
// narrowest layer the 256-column LDS-DMA tiles take (columns past Cout are zero-filled operands and masked stores: a 128-channel layer
// pays for 256); VPMI_RING_MIN_COUT overrides for A/B
static int ring_min_cout() {
    static const int v = [] { const char* e = getenv("VPMI_RING_MIN_COUT"); return e ? atoi(e) : 256; }();
    return v;
}

static int use_conv256() {
    if (g_conv256 == -2) { const char* e = getenv("VPMI_CONV256"); g_conv256 = e ? atoi(e) : -1; }
    return g_conv256;
}

extern "C" {

int vp_conv256_select(int schedule) {
    const int prev = use_conv256();
    if (schedule >= -1 && schedule <= 7) g_conv256 = schedule;
    return prev;
}

int vp_conv1d_tiles_m(int B, int T_out) { return (int)(((long long)B * T_out + BM - 1) / BM); }
int vp_conv1d_nseg(int T_out) { return T_out > 0 ? (BM - 1) / T_out + 2 : 0; }

int vp_conv1d_fwd(vp_ctx* ctx, const vp_conv1d_desc* d, vp_stream stream) {
    if (!ctx || !d || !d->x || !d->w || !d->y) VP_FAIL(ctx, VP_EINVAL, "conv1d: null argument");
    if ((d->dtype_in != VP_F32 && d->dtype_in != VP_BF16 && d->dtype_in != VP_HL32) ||
        (d->dtype_out != VP_F32 && d->dtype_out != VP_BF16 && d->dtype_out != VP_HL32))
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad dtype");
    if (d->dtype_in == VP_F32 && d->dtype_out == VP_BF16) VP_FAIL(ctx, VP_EUNSUP, "conv1d: f32 -> bf16 not built");
    const bool hl_in = d->dtype_in == VP_HL32, hl_out = d->dtype_out == VP_HL32;
    if (hl_in || hl_out) {
        // split bf16 planes (vpmi.h: VP_HL32): 32-channel groups; f32 x (split while staging, mfma_bf16 = 2) or hl32 x, hl32 w with hl32 x
        if ((hl_in && (d->Cin % 32 || d->ldx % 32 || d->xoff % 32)) ||
            (hl_out && (d->Cout % 32 || d->ldy % 32 || d->yoff % 32 || d->ysplit % 32 || d->ldy2 % 32 || d->y2off % 32 || d->ld_add % 32 ||
                        d->add_off % 32 || d->ld_aux % 32 || d->aux_off % 32 || d->ld_res % 32 || d->res_off % 32)))
            VP_FAIL(ctx, VP_EINVAL, "conv1d: hl32 tensors need channel counts, leading dimensions and offsets that are multiples of 32");
        if (d->dtype_in == VP_BF16 || d->dtype_out == VP_BF16 || (d->dtype_in == VP_F32 && d->mfma_bf16 != 2) || d->gate || d->pro_scale ||
            d->KF > 1 || d->F_in > 1 || d->F_out > 1)
            VP_FAIL(ctx, VP_EUNSUP, "conv1d: hl32 is built for 1-D layers of the split-precision engine (f32 / hl32 in with mfma_bf16 = 2 semantics)");
    }
    const int epc = d->dtype_in == VP_BF16 ? 8 : 4;
    if (d->B <= 0 || d->T_in <= 0 || d->T_out <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->dilation <= 0)
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad shape");
    if (d->Cin % epc || d->ldx % epc || d->xoff % epc) VP_FAIL(ctx, VP_EINVAL, "conv1d: Cin/ldx/xoff must be multiples of %d", epc);
    if (d->Cout % 4 || d->ldy % 4 || d->yoff % 4) VP_FAIL(ctx, VP_EINVAL, "conv1d: Cout/ldy/yoff must be multiples of 4");
    if (d->ysplit) {
        if (!d->y2 || d->ysplit % 4 || d->ldy2 % 4 || d->y2off % 4 || d->ysplit > d->Cout)
            VP_FAIL(ctx, VP_EINVAL, "conv1d: bad y2 split");
    }
    if (d->aux && (!d->add_in || d->ld_add % 4 || d->add_off % 4 || d->ld_aux % 4 || d->aux_off % 4))
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad aux/add_in");
    if (d->res && (d->ld_res % 4 || d->res_off % 4)) VP_FAIL(ctx, VP_EINVAL, "conv1d: bad residual");
    const bool two_d = d->KF > 1 || d->F_in > 1 || d->F_out > 1;
    const bool pro = d->pro_scale != nullptr;
    if (pro && (!d->pro_shift || d->KW != 1 || two_d)) VP_FAIL(ctx, VP_EINVAL, "conv1d: input prologue needs a 1x1 conv");
    const int span = d->dilation * ((two_d ? d->KW / (d->KF > 0 ? d->KF : 1) : d->KW) - 1);
    if (two_d) {
        if (d->KF < 1 || d->KW % d->KF || d->F_in < 1 || d->F_out < 1 || d->stride_f < 1 || d->pad_mode != VP_PAD_ZERO)
            VP_FAIL(ctx, VP_EINVAL, "conv2d: bad geometry (zero padding only)");
        if (d->rowbias || d->psum || d->gate) VP_FAIL(ctx, VP_EUNSUP, "conv2d: per-utterance epilogue terms are 1-D only");
    } else if (d->pad_mode == VP_PAD_NONE) {
        if ((d->T_out - 1) * d->stride + span > d->T_in - 1)
            VP_FAIL(ctx, VP_EINVAL, "conv1d: un-padded window leaves the input (T_in %d, T_out %d)", d->T_in, d->T_out);
        if (d->pad_left != 0) VP_FAIL(ctx, VP_EINVAL, "conv1d: pad_left with PAD_NONE");
    } else if (d->pad_mode == VP_PAD_REFLECT) {
        const int right = (d->T_out - 1) * d->stride - d->pad_left + span - (d->T_in - 1);
        if (d->pad_left >= d->T_in || right >= d->T_in) VP_FAIL(ctx, VP_EINVAL, "conv1d: reflect pad >= T_in");
    } else if (d->pad_mode != VP_PAD_ZERO) {
        VP_FAIL(ctx, VP_EINVAL, "conv1d: bad pad_mode");
    }
    if (d->gate && (d->gate_len < 1 || d->gate_nseg < 1)) VP_FAIL(ctx, VP_EINVAL, "conv1d: bad gate segmentation");
    const int F_in = two_d ? d->F_in : 1, F_out = two_d ? d->F_out : 1;
    const long long Mll = (long long)d->B * d->T_out * F_out;
    if (Mll > 0x7fffffffLL / 2) VP_FAIL(ctx, VP_EINVAL, "conv1d: too many output positions");
    const size_t es = d->dtype_in == VP_BF16 ? 2 : 4;
    const unsigned long long xbytes = ((unsigned long long)d->B * d->T_in * F_in - 1) * d->ldx * es + (d->xoff + d->Cin) * es;
    // (mfma_bf16 = 3: the weights are split bf16 planes whose rows are zero-padded to whole 32-element groups)
    const unsigned long long wrow = d->mfma_bf16 == 3 ? ((unsigned long long)d->KW * d->Cin + 31) / 32 * 32 : (unsigned long long)d->KW * d->Cin;
    const unsigned long long wbytes = (unsigned long long)d->Cout * wrow * es;
    if (wbytes >= 0xffffff00ull) VP_FAIL(ctx, VP_EUNSUP, "conv1d: weights larger than 4 GiB (32-bit buffer offsets)");
    if (xbytes >= 0xffffff00ull) {
        // The kernels address x through a 32-bit buffer offset.  Utterances are independent rows of the GEMM, so a larger activation
        // tensor (ERes2Net-large at 128 utterances per GPU: BASELINE configs[4]) runs as consecutive launches over batch slices; a
        // slice of the fused time sums must start on an M-tile boundary of the partial-sum arrays.
        const unsigned long long per_utt = (unsigned long long)d->T_in * F_in * d->ldx * es;
        long long bc = (long long)(0xe0000000ull / (per_utt ? per_utt : 1));
        if (d->psum) {
            int ga = d->T_out, gb = BM;                         // slices of bc utterances with bc * T_out a multiple of the M-tile
            while (gb) { const int t_ = ga % gb; ga = gb; gb = t_; }
            const long long q = BM / ga;
            bc = bc / q * q;
        }
        if (bc < 1 || d->B < 2) VP_FAIL(ctx, VP_EUNSUP, "conv1d: one utterance's activations exceed 4 GiB (32-bit buffer offsets)");
        const size_t eo = d->dtype_out == VP_BF16 ? 2 : 4;
        for (long long b0 = 0; b0 < d->B; b0 += bc) {
            vp_conv1d_desc s = *d;
            s.B = (int)(d->B - b0 < bc ? d->B - b0 : bc);
            const size_t rin = (size_t)b0 * d->T_in * F_in, rout = (size_t)b0 * d->T_out * F_out;
            s.x = static_cast<const char*>(d->x) + rin * d->ldx * es;
            s.y = static_cast<char*>(d->y) + rout * d->ldy * eo;
            if (d->y2) s.y2 = static_cast<char*>(d->y2) + rout * d->ldy2 * eo;
            if (d->add_in) s.add_in = static_cast<const char*>(d->add_in) + rout * d->ld_add * eo;
            if (d->aux) s.aux = static_cast<char*>(d->aux) + rout * d->ld_aux * eo;
            if (d->res) s.res = static_cast<const char*>(d->res) + rout * d->ld_res * eo;
            if (d->rowbias) s.rowbias = d->rowbias + (size_t)b0 * d->Cout;
            if (d->gate) s.gate = d->gate + (size_t)b0 * d->gate_nseg * d->Cout;
            const size_t pso = rout / BM * (size_t)vp_conv1d_nseg(d->T_out) * d->Cout;
            if (d->psum) s.psum = d->psum + pso;
            if (d->psumsq) s.psumsq = d->psumsq + pso;
            const int rc = vp_conv1d_fwd(ctx, &s, stream);
            if (rc) return rc;
        }
        return VP_OK;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.rowbias = d->rowbias;
    a.x_bytes = (unsigned)xbytes; a.w_bytes = (unsigned)wbytes; a.y2 = d->y2;
    a.bn_scale = d->bn_scale; a.bn_shift = d->bn_shift; a.y = d->y; a.add_in = d->add_in; a.aux = d->aux;
    a.pro_scale = d->pro_scale; a.pro_shift = d->pro_shift; a.gate = d->gate; a.res = d->res;
    a.psum = d->psum; a.psumsq = d->psumsq;
    a.ldx = d->ldx; a.xoff = d->xoff; a.ldy2 = d->ldy2; a.y2off = d->y2off; a.ysplit = d->ysplit;
    a.ldy = d->ldy; a.yoff = d->yoff; a.ld_add = d->ld_add; a.add_off = d->add_off; a.ld_aux = d->ld_aux;
    a.aux_off = d->aux_off; a.ld_res = d->ld_res; a.res_off = d->res_off;
    a.M = (int)Mll; a.N = d->Cout; a.K = d->KW * d->Cin; a.Kw = a.K; a.cpt = d->Cin / epc; a.KC = a.K / epc; a.Cin = d->Cin;
    a.KT = (a.KC + 7) / 8;
    a.T_in = d->T_in; a.T_out = d->T_out; a.dilation = d->dilation; a.stride = d->stride; a.pad_left = d->pad_left;
    a.pad_mode = d->pad_mode; a.act = d->act; a.act2 = d->act2;
    a.F_in = F_in; a.F_out = F_out; a.KF = two_d ? d->KF : 1; a.stride_f = two_d ? d->stride_f : 1; a.pad_f = two_d ? d->pad_f : 0;
    a.gate_len = d->gate_len > 0 ? d->gate_len : 1; a.gate_nseg = d->gate_nseg;
    const int mode = two_d ? MODE_2D : (d->KW == 1 ? (pro ? MODE_1X1_PRO : MODE_1X1) : MODE_TAPS);
    int bn = d->Cout <= 32 ? 32 : (d->Cout <= 64 ? 64 : 128);
    if (mode == MODE_1X1_PRO && bn > 64) bn = 64;
    if (d->dtype_in == VP_BF16 && d->dtype_out == VP_F32) bn = 128;
    if (hl_in || hl_out) bn = (hl_in && hl_out && mode == MODE_1X1 && d->Cout <= 128 && getenv("VPMI_HL_BN128") == nullptr) ? 64 : 128;
    // (Round 3 pinned the bf16-input 128-column tile to 64 columns: kernels running beside it returned wrong lanes.  Round 4 found the
    // cause in the VICTIMS, not here -- packed-f32 VALU instructions reading freshly loaded registers next to an MFMA-heavy wave,
    // DESIGN.md section 8 -- and the library is now built without packed-f32 instructions; VPMI_BN64=1 keeps the narrow tile for A/B.)
    { static const bool bn64 = getenv("VPMI_BN64") != nullptr; if (bn64 && d->dtype_in == VP_BF16 && bn == 128) bn = 64; }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + bn - 1) / bn;
    a.nseg = vp_conv1d_nseg(d->T_out);
    a.group_m = 96 / a.tiles_n;
    if (a.group_m < 1) a.group_m = 1;
    if (a.group_m > 16) a.group_m = 16;
    if (d->psum && a.nseg > NSEG_MAX) VP_FAIL(ctx, VP_EUNSUP, "conv1d: T_out %d too short for fused time sums", d->T_out);
    hipStream_t st = (hipStream_t)stream;
    // wide bf16 layers: 128 x 256 / 256 x 256 tiles fed by LDS-DMA (conv_gemm256.hip).  bf16 -> f32 (the training engine's
    // data-gradient GEMMs over bf16 dz) is built on the default schedule (6) for plain 1x1 layers.
    if (d->dtype_in == VP_BF16 && (mode == MODE_1X1 || mode == MODE_TAPS) &&
        (d->Cin % 64 == 0 || (mode == MODE_TAPS && (use_conv256() >= 3 || use_conv256() < 0))) &&
        d->Cout >= ring_min_cout() && !d->gate && (!d->psum || d->T_out >= 128) && use_conv256()) {
        int sched = use_conv256() < 0 ? 6 : use_conv256();
        if (sched == 7) sched = a.K <= 1024 ? 6 : 4;
        if (sched == 5) sched = 4;              // resident workgroups: no longer built
        if (sched == 6 && mode != MODE_1X1) sched = 4;
        // the ring kernels address a 1x1 layer as "source row m for output row m" with wave-uniform piece offsets and let the
        // buffer range check zero what lies past M / N / K: anything else (strided / padded 1x1, operands near 4 GiB) takes the
        // two-stage schedule with its per-row offsets
        if (sched >= 4 && mode == MODE_1X1 &&
            (d->stride != 1 || d->pad_left != 0 || d->T_in != d->T_out || xbytes >= 0xe0000000ull || wbytes >= 0xe0000000ull))
            sched = 3;
        const bool out_f32 = d->dtype_out == VP_F32;
        const long long min_rows = sched == 6 ? 128 * 32 : 256 * 64;          // enough tiles to be worth a wide-tile launch
        if ((!out_f32 || sched == 6) && a.M >= min_rows) {
            a.tiles_m = (a.M + 255) / 256;
            a.tiles_n = (a.N + 255) / 256;
            a.group_m = 32 / a.tiles_n;
            if (a.group_m < 1) a.group_m = 1;
            if (a.group_m > 16) a.group_m = 16;
            { static int gm = -1; if (gm < 0) { const char* e = getenv("VPMI_GROUP_M"); gm = e ? atoi(e) : 0; } if (gm > 0) a.group_m = gm; }
            if (sched == 6) {                   // 128-row tiles, two workgroups per CU: ~64 workgroups of an XCD share a group's panels
                a.tiles_m = (a.M + 127) / 128;
                a.group_m = 64 / a.tiles_n;
                if (a.group_m < 1) a.group_m = 1;
                if (a.group_m > 32) a.group_m = 32;
            }
            return vp_conv_launch256_bf16(ctx, &a, mode, sched - 1, out_f32 ? 1 : 0, st);
        }
    }
    if (hl_in) {
        // wide 1x1 layers: the 128 x 256 LDS-DMA ring in split precision (conv_gemm256.hip); the rest on the 128 x 128 kernel
        if (mode == MODE_1X1 && d->Cout >= ring_min_cout() && (!d->psum || d->T_out >= 128) && d->stride == 1 && d->pad_left == 0 &&
            d->T_in == d->T_out && xbytes < 0xe0000000ull && wbytes < 0xe0000000ull && a.M >= 128 * 32 && use_conv256()) {
            a.tiles_m = (a.M + 127) / 128;
            a.tiles_n = (a.N + 255) / 256;
            a.group_m = 64 / a.tiles_n;
            if (a.group_m < 1) a.group_m = 1;
            if (a.group_m > 32) a.group_m = 32;
            { static int gm = -1; if (gm < 0) { const char* e = getenv("VPMI_GROUP_M"); gm = e ? atoi(e) : 0; } if (gm > 0) a.group_m = gm; }
            return vp_conv_launch_ring_x3(ctx, &a, hl_out ? 0 : 1, st);
        }
        if (!hl_out) VP_FAIL(ctx, VP_EUNSUP, "conv1d: hl32 -> f32 is built on the ring kernel only (wide 1x1 layers)");
        return vp_conv_launch_hl_hl(ctx, &a, bn, mode, st);
    }
    if (hl_out) return vp_conv_launch_x3_hl(ctx, &a, bn, mode, st);
    if (d->dtype_in == VP_BF16 && d->dtype_out == VP_BF16) return vp_conv_launch_bf16_bf16(ctx, &a, bn, mode, st);
    if (d->dtype_in == VP_BF16 && d->dtype_out == VP_F32) return vp_conv_launch_bf16_f32(ctx, &a, bn, mode, st);
    if (d->mfma_bf16 == 3) {                 // split precision with the weights given as split planes, rows padded to 32-element groups
        if (d->dtype_in != VP_F32 || d->dtype_out != VP_F32) VP_FAIL(ctx, VP_EINVAL, "conv1d: mfma_bf16 = 3 takes f32 tensors");
        a.Kw = (int)wrow;
        return vp_conv_launch_x3w_f32(ctx, &a, bn, mode, st);
    }
    if (d->mfma_bf16 == 2) return vp_conv_launch_x3_f32(ctx, &a, bn, mode, st);
    if (d->mfma_bf16) return vp_conv_launch_amp_f32(ctx, &a, bn, mode, st);
    return vp_conv_launch_f32_f32(ctx, &a, bn, mode, st);
}

}  // extern "C"
