// Split-precision instantiations of the conv GEMM: f32 tensors, each operand split into bf16 hi + lo while staging, three bf16 MFMAs
// per k-step (hi*hi + hi*lo + lo*hi), f32 accumulate, f32 out.  Kernel: conv_gemm_impl.h (x3_t).
#include "conv_gemm_impl.h"

int vp_conv_launch_x3_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<x3_t, float, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}

// f32 activations split while staging, weights already split (hl32 planes, rows zero-padded to 32-element groups): mfma_bf16 = 3
int vp_conv_launch_x3w_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<x3w_t, float, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}

// hl32 tensors (split bf16 planes in memory, conv_gemm_impl.h: hl_t): the layers of the ECAPA split-precision fast path that are not on
// the LDS-DMA ring kernel -- blocks[0] (f32 features in, hl32 out) and the ASP attention TDNN (hl32 in, hl32 out).  128-column tiles only.
int vp_conv_launch_x3_hl(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<x3_t, hl_t, false>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}
int vp_conv_launch_hl_hl(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    // 64-column tiles for the narrow 1x1 layers (the ASP attention TDNN, 1536 -> 128: 596 M-tiles are 1.16 rounds of 128-wide tiles)
    if (bn == 64 && mode == MODE_1X1) return launch_conv<hl_t, hl_t, 64, MODE_1X1>(ctx, *static_cast<const ConvArgs*>(args), st);
    return dispatch_conv<hl_t, hl_t, false>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}
