// f32 instantiations of the conv GEMM (exact-f32 matrix cores).  Kernel: conv_gemm_impl.h.
#include "conv_gemm_impl.h"

int vp_conv_launch_f32_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<float, float, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}

// f32 tensors, bf16 matrix cores (operands rounded while staging; f32 accumulate, f32 out): the mixed-precision training engine
int vp_conv_launch_amp_f32(vp_ctx* ctx, const void* args, int bn, int mode, hipStream_t st) {
    return dispatch_conv<amp_t, float, true>(ctx, *static_cast<const ConvArgs*>(args), bn, mode, st);
}
