// Context management for libvpmi.
#include "common.h"

#include <stdlib.h>

extern "C" {

int vp_version(void) { return VPMI_VERSION; }

vp_ctx* vp_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    vp_ctx* c = (vp_ctx*)calloc(1, sizeof(vp_ctx));
    if (!c) return nullptr;
    c->device = device;
    // grid-barrier counters (res2_train.hip): allocated here so that no launcher allocates during a stream capture
    if (hipMalloc(&c->grid_bar, 2048) != hipSuccess || hipMemset(c->grid_bar, 0, 2048) != hipSuccess) c->grid_bar = nullptr;
    c->grid_bar_own = c->grid_bar;
    return c;
}

void vp_destroy(vp_ctx* ctx) {
    if (!ctx) return;
    vp_fbank_release_tables(ctx);
    vp_mel_release_tables(ctx);
    if (ctx->grid_bar_own) (void)hipFree(ctx->grid_bar_own);
    free(ctx);
}

const char* vp_last_error(vp_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int vp_set_margin_table(vp_ctx* ctx, const float* table) {
    if (!ctx) return VP_EINVAL;
    ctx->margin_table = table;
    return VP_OK;
}

int vp_set_grid_barrier_words(vp_ctx* ctx, void* words) {
    if (!ctx) return VP_EINVAL;
    if (words && (reinterpret_cast<uintptr_t>(words) & 255)) VP_FAIL(ctx, VP_EINVAL, "grid-barrier words must be 256-byte aligned");
    ctx->grid_bar = words ? static_cast<unsigned*>(words) : ctx->grid_bar_own;
    return VP_OK;
}

int vp_set_grid_reserve_cus(vp_ctx* ctx, int n_cus) {
    if (!ctx || n_cus < 0) return VP_EINVAL;
    ctx->grid_reserve_cus = n_cus;
    return VP_OK;
}

}  // extern "C"

// A workgroup that holds its CU slot (and lds_bytes of LDS) for ~usec microseconds: the stand-in for a persistent kernel of another
// queue (a collective) in the co-residency tests of the grid-barrier kernels.
namespace {
__global__ __launch_bounds__(256) void occupy_cus_kernel(long long ticks, unsigned* sink) {
    extern __shared__ unsigned lds_hold[];
    if (threadIdx.x == 0) lds_hold[0] = blockIdx.x;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && threadIdx.x == 0 && lds_hold[0] == 0xffffffffu) *sink = 1u;        // (keeps the LDS allocation alive)
}
}  // namespace

extern "C" int vp_occupy_cus(vp_ctx* ctx, int n_workgroups, int lds_bytes, int usec, vp_stream stream) {
    if (!ctx || n_workgroups < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024 || usec < 0 || usec > 2000000)
        VP_FAIL(ctx, VP_EINVAL, "occupy_cus: bad arguments");
    static bool attr_dev[64] = {};
    bool& attr_set = attr_dev[ctx->device & 63];
    if (!attr_set) {
        VP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // wall_clock64 ticks at 100 MHz on gfx9
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(n_workgroups), dim3(256), (size_t)(lds_bytes < 4 ? 4 : lds_bytes), (hipStream_t)stream,
                       (long long)usec * 100, (unsigned*)nullptr);
    VP_LAUNCH_CHECK(ctx, "occupy_cus");
    return VP_OK;
}
