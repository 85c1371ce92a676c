// bf16-input instantiations of the conv GEMM (bf16 or f32 output).  Kernel: conv_gemm_impl.h.
#include "conv_gemm_impl.h"

int vp_conv_launch_bf16_bf16(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<bf16_t, bf16_t, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}
int vp_conv_launch_bf16_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<bf16_t, float, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}
