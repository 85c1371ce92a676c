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
    return c;
}

void vp_destroy(vp_ctx* ctx) {
    if (!ctx) return;
    vp_fbank_release_tables(ctx);
    vp_mel_release_tables(ctx);
    if (ctx->grid_bar) (void)hipFree(ctx->grid_bar);
    free(ctx);
}

const char* vp_last_error(vp_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int vp_set_margin_table(vp_ctx* ctx, const float* table) {
    if (!ctx) return VP_EINVAL;
    ctx->margin_table = table;
    return VP_OK;
}

}  // extern "C"
