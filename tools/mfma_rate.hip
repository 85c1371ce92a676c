// Sustained MFMA issue rate on this GPU: every wave runs ITERS x 16 independent v_mfma_f32_16x16x32_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c[16];
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int waves : {4, 8}) {
        for (int grid : {256, 512, 2048}) {
            k<<<grid, waves * 64>>>(out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<grid, waves * 64>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * waves * iters * 16 * 16384.0;
            printf("waves/WG %d grid %4d: %8.3f ms  %8.1f TFLOP/s\n", waves, grid, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
