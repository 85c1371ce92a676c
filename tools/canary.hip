// Canary kernels for the co-run screen (tools/stress_canary.py): a victim that keeps KNOWN state in one resource class and
// re-checks it for ~15 us, so that a hit says WHAT a misbehaving neighbour damaged -- registers at rest, LDS at rest, the data of
// a vector-memory load, or the result of an LDS broadcast read.  Not part of libvpmi; built by tools/build_canary.sh into
// voiceprintrecognition-paddlepaddle_amd/lib/libcanary.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// report layout (u32): [0] mismatch count, [1..7] first mismatch: block, thread, slot, iteration, got, expected, kind
__device__ __forceinline__ void report(unsigned* rep, unsigned kind, unsigned slot, unsigned it, unsigned got, unsigned want) {
    const unsigned n = atomicAdd(rep, 1u);
    if (n == 0) {
        rep[1] = blockIdx.x; rep[2] = threadIdx.x; rep[3] = slot; rep[4] = it; rep[5] = got; rep[6] = want; rep[7] = kind;
    }
    if (n < 64) {   // a short log of (thread, slot) pairs behind the header
        rep[8 + 2 * n] = (blockIdx.x << 16) | threadIdx.x;
        rep[9 + 2 * n] = (slot << 16) | (it & 0xffff);
    }
}

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}

// kind 0: registers at rest.  NR values per lane stay in VGPRs (the empty asm keeps the compiler from folding them) while the wave
// sleeps and wakes; every value is re-derived and compared each round.
template <int NR>
__global__ __launch_bounds__(1024) void canary_vgpr(unsigned* rep, int iters) {
    unsigned r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = hash3(blockIdx.x, threadIdx.x, i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(r[i]));
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned want = hash3(blockIdx.x, threadIdx.x, i);
            if (r[i] != want) { report(rep, 0, i, it, r[i], want); r[i] = want; }
        }
    }
}

// kind 1: LDS at rest.  The workgroup fills `words` dwords of dynamic LDS, then every thread re-reads ITS 16-byte chunks each round.
__global__ __launch_bounds__(1024) void canary_lds(unsigned* rep, int iters, int words) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = hash3(blockIdx.x, i, 77);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int c = threadIdx.x * 4; c + 3 < words; c += blockDim.x * 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + c);
            const unsigned got[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned want = hash3(blockIdx.x, c + j, 77);
                if (got[j] != want) report(rep, 1, c + j, it, got[j], want);
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// kind 2: vector-memory load data.  `buf` holds hash3(7, index, 3) per dword; every round a thread has eight 16-byte loads in
// flight (the shape of se_gate's weight stream: consecutive lanes read consecutive float4) and checks all 32 dwords.
__global__ __launch_bounds__(1024) void canary_vmem(unsigned* rep, int iters, const unsigned* __restrict__ buf, int words) {
    const int nvec = words / 4;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
        int idx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            idx[k] = (int)(((unsigned)threadIdx.x + 1024u * (unsigned)(it * 8 + k) + 4096u * blockIdx.x) % (unsigned)nvec);
            v[k] = *reinterpret_cast<const uint4*>(buf + 4 * idx[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned got[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned want = hash3(7, 4 * idx[k] + j, 3);
                if (got[j] != want) report(rep, 2, (k << 2) | j, it, got[j], want);
            }
        }
    }
}

// kind 3: LDS broadcast reads + FMA chain (se_gate's matvec inner loop without the weight stream): acc += in[k] * c_k with
// in[] in LDS read by every lane at the same address; the expected sum is recomputed from the generating formula.
__global__ __launch_bounds__(1024) void canary_bcast(unsigned* rep, int iters) {
    __shared__ float in[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) in[i] = (float)((hash3(blockIdx.x, i, 5) >> 8) & 0xffff) * (1.f / 65536.f);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, ref[4] = {0.f, 0.f, 0.f, 0.f};
        const int k0 = (threadIdx.x >> 7) * 16;
#pragma unroll 8
        for (int k = k0; k < k0 + 16; ++k) {
            const float x = in[k];
            const float xr = (float)((hash3(blockIdx.x, k, 5) >> 8) & 0xffff) * (1.f / 65536.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = (float)((threadIdx.x * 4 + j + k) & 255) * (1.f / 256.f);
                acc[j] = fmaf(x, w, acc[j]);
                ref[j] = fmaf(xr, w, ref[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (acc[j] != ref[j]) report(rep, 3, j, it, __float_as_uint(acc[j]), __float_as_uint(ref[j]));
        __builtin_amdgcn_s_sleep(1);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// kinds 5-8: se_gate's inner loop, piece by piece.  acc(x,y,z,w) += in[k] * W[k][4v..4v+3] over 16 k per round, all values small
// integers (exact in f32 in any order); the expected sums are taken in INTEGER arithmetic.
//   PK = 1: v_pk_fma_f32 with op_sel_hi:[1,0,1] / op_sel:[0,1,0] exactly as hipcc emits it in se_matvec; 0: four v_fma_f32
//   SRC = 0: both operands generated in registers; 1: in[] read from LDS (ds_read2_b32); 2: LDS + W streamed from global memory
//   (eight 16-byte loads in flight, consecutive lanes = consecutive float4: the weight stream of se_gate)
//   VAR (PK = 1 only): 1 = no op_sel (x broadcast into a pair by v_mov), 2 = "s_nop 2" ahead of every packed FMA, 3 = v_mov_b64 of the
//   loaded pair + four v_fma_f32 on the copy;  SRC = 3: in[] generated in registers, W streamed from global memory
template <int PK, int SRC, int VAR = 0>
__global__ __launch_bounds__(1024) void canary_fma(unsigned* rep, int iters, const float* __restrict__ W, int N) {
    __shared__ float in[128];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) in[i] = (float)(hash3(blockIdx.x, i, 9) & 15);
    __syncthreads();
    const int nv = N >> 2, v = threadIdx.x % nv, sl = threadIdx.x / nv;      // N = 512: 128 column groups x 8 K-slices of 16
    const int k0 = sl * 16;
    for (int it = 0; it < iters; ++it) {
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        float4 w[16];
        float x[16];
        if (SRC >= 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + k) * N + 4 * v);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                w[k].x = (float)(hash3(k0 + k, 4 * v + 0, 11) & 15); w[k].y = (float)(hash3(k0 + k, 4 * v + 1, 11) & 15);
                w[k].z = (float)(hash3(k0 + k, 4 * v + 2, 11) & 15); w[k].w = (float)(hash3(k0 + k, 4 * v + 3, 11) & 15);
            }
        }
        if (SRC == 1 || SRC == 2) {
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                f32x2 t;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"((unsigned)(uintptr_t)(in + k0)), "i"(k), "i"(k + 1));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
                x[k] = t.x; x[k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) { x[k] = (float)(hash3(blockIdx.x, k0 + k, 9) & 15); asm volatile("" : "+v"(x[k])); }
        }
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            f32x2 xx = {x[k], x[k + 1]};
            f32x2 w01 = {w[k].x, w[k].y}, w23 = {w[k].z, w[k].w}, u01 = {w[k + 1].x, w[k + 1].y}, u23 = {w[k + 1].z, w[k + 1].w};
            if (PK && VAR == 1) {
                f32x2 x0 = {xx.x, xx.x}, x1 = {xx.y, xx.y};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(x0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(x0));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(u01), "v"(x1));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(u23), "v"(x1));
            } else if (PK && VAR == 2) {
                asm volatile("s_nop 2\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("s_nop 2\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a23) : "v"(w23), "v"(xx));
                asm volatile("s_nop 2\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a01) : "v"(u01), "v"(xx));
                asm volatile("s_nop 2\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a23) : "v"(u23), "v"(xx));
            } else if (PK && VAR == 3) {
                f32x2 c01, c23, d01, d23;
                asm volatile("v_mov_b64 %0, %1" : "=v"(c01) : "v"(w01));
                asm volatile("v_mov_b64 %0, %1" : "=v"(c23) : "v"(w23));
                asm volatile("v_mov_b64 %0, %1" : "=v"(d01) : "v"(u01));
                asm volatile("v_mov_b64 %0, %1" : "=v"(d23) : "v"(u23));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(c01.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(c01.y), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(c23.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(c23.y), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(d01.x), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(d01.y), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(d23.x), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(d23.y), "v"(xx.y));
            } else if (PK) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a23) : "v"(w23), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a01) : "v"(u01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a23) : "v"(u23), "v"(xx));
            } else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(w01.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(w01.y), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(w23.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(w23.y), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(u01.x), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(u01.y), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(u23.x), "v"(xx.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(u23.y), "v"(xx.y));
            }
        }
        unsigned want[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned xi = hash3(blockIdx.x, k0 + k, 9) & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) want[j] += xi * (hash3(k0 + k, 4 * v + j, 11) & 15);
        }
        const float got[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (got[j] != (float)want[j]) report(rep, 5 + PK * 4 + SRC, j, it, __float_as_uint(got[j]), __float_as_uint((float)want[j]));
    }
}


// kinds 12-19: WHICH consumer of a freshly returned load is exposed?  Sixteen 16-byte loads in flight per round (the weight stream
// of canary_fma<*, 2>), consumed in order under the compiler's counted s_waitcnt vmcnt(N):
//   0 v_pk_fma_f32 behind "s_nop 2"      1 v_pk_fma_f32 behind "s_nop 7"      2 v_mov_b64 of the loaded pair, then v_fma_f32 on the copy
//   3 v_mov_b32_dpp row_shr:1 of the loaded registers      4 every load landed (vmcnt(0)), then 8 NEW loads in flight, then v_pk_fma_f32
//   5 v_pk_mul_f32 (no accumulate)        6 v_lshl_add_u64 on the loaded pair (64-bit integer VALU)
template <int MODE>
__global__ __launch_bounds__(512) void canary_use(unsigned* rep, int iters, const float* __restrict__ W, int N) {
    __shared__ float in[128];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) in[i] = (float)(hash3(blockIdx.x, i, 9) & 15);
    __syncthreads();
    const int nv = N >> 2, v = threadIdx.x % nv, sl = threadIdx.x / nv;
    const int k0 = sl * 16;
    for (int it = 0; it < iters; ++it) {
        float4 w[16];
        float x[16];
        // x first: the consumers below must issue the moment their s_waitcnt releases (a wave that was never blocked at the wait
        // cannot lose the race)
#pragma unroll
        for (int k = 0; k < 16; ++k) { x[k] = (float)(hash3(blockIdx.x, k0 + k, 9) & 15); asm volatile("" : "+v"(x[k])); }
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + k) * N + 4 * v);
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(x[k]));
        float4 w2[8];
        if (MODE == 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) w2[k] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + k) * N + 4 * ((v + 1) % nv));
        }
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        unsigned long long isum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x2 xx = {x[k], x[k]};
            f32x2 w01 = {w[k].x, w[k].y}, w23 = {w[k].z, w[k].w};
            if (MODE == 0) {
                asm volatile("s_nop 2\n\tv_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(xx));
            } else if (MODE == 7) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(xx));
            } else if (MODE == 8) {
                asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(xx));
            } else if (MODE == 1) {
                asm volatile("s_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(xx));
            } else if (MODE == 2) {
                f32x2 c01, c23;
                asm volatile("v_mov_b64 %0, %1" : "=v"(c01) : "v"(w01));
                asm volatile("v_mov_b64 %0, %1" : "=v"(c23) : "v"(w23));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(c01.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(c01.y), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(c23.x), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(c23.y), "v"(xx.x));
            } else if (MODE == 3) {
                // d = value of lane - 1 within the 16-lane row (lane 0 of a row keeps 0): acc += x * d, expected below with v - 1
                float d[4];
                d[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[k].x), 0x111, 0xf, 0xf, false));
                d[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[k].y), 0x111, 0xf, 0xf, false));
                d[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[k].z), 0x111, 0xf, 0xf, false));
                d[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w[k].w), 0x111, 0xf, 0xf, false));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.x) : "v"(d[0]), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a01.y) : "v"(d[1]), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.x) : "v"(d[2]), "v"(xx.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a23.y) : "v"(d[3]), "v"(xx.x));
            } else if (MODE == 4) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(w23), "v"(xx));
            } else if (MODE == 5) {
                f32x2 p01, p23;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p01) : "v"(w01), "v"(xx));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p23) : "v"(w23), "v"(xx));
                a01.x += p01.x; a01.y += p01.y; a23.x += p23.x; a23.y += p23.y;
                asm volatile("" : "+v"(a01.x), "+v"(a01.y), "+v"(a23.x), "+v"(a23.y));
            } else {
                unsigned long long q01 = __builtin_bit_cast(unsigned long long, w01), q23 = __builtin_bit_cast(unsigned long long, w23), r01, r23;
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r01) : "v"(q01), "v"(isum));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r23) : "v"(q23), "v"(r01));
                isum = r23;
            }
        }
        if (MODE == 6) {
            unsigned long long want = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const f32x2 e01 = {(float)(hash3(k0 + k, 4 * v + 0, 11) & 15), (float)(hash3(k0 + k, 4 * v + 1, 11) & 15)};
                const f32x2 e23 = {(float)(hash3(k0 + k, 4 * v + 2, 11) & 15), (float)(hash3(k0 + k, 4 * v + 3, 11) & 15)};
                want += __builtin_bit_cast(unsigned long long, e01) + __builtin_bit_cast(unsigned long long, e23);
            }
            if (isum != want) report(rep, 12 + MODE, (unsigned)((isum ^ want) >> 32 ? 1 : 0), it, (unsigned)isum, (unsigned)want);
            continue;
        }
        unsigned want[4] = {0, 0, 0, 0};
        const int vs = MODE == 3 ? v - 1 : v;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned xi = hash3(blockIdx.x, k0 + k, 9) & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) want[j] += (MODE == 3 && (v & 15) == 0) ? 0u : xi * (hash3(k0 + k, 4 * vs + j, 11) & 15);
        }
        const float got[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (got[j] != (float)want[j]) report(rep, 12 + MODE, j, it, __float_as_uint(got[j]), __float_as_uint((float)want[j]));
        if (MODE == 4) {            // the second batch, consumed by scalar FMAs
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned wb[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                b[0] = fmaf(w2[k].x, x[k], b[0]); b[1] = fmaf(w2[k].y, x[k], b[1]); b[2] = fmaf(w2[k].z, x[k], b[2]); b[3] = fmaf(w2[k].w, x[k], b[3]);
                const unsigned xi = hash3(blockIdx.x, k0 + k, 9) & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j) wb[j] += xi * (hash3(k0 + k, 4 * ((v + 1) % nv) + j, 11) & 15);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b[j] != (float)wb[j]) report(rep, 12 + MODE, 4 + j, it, __float_as_uint(b[j]), __float_as_uint((float)wb[j]));
        }
    }
}

// kind 20: MFMA operands straight from loads (csrc/pointwise.hip feeds its activation fragments that way).  Lane l loads 16 bytes =
// A[i = l & 15][k = 8 (l >> 4) ..] of eight 16 x 32 bf16 tiles (values 0..3), B[k][j] = (k + j) & 3 generated in registers;
// D = A B per tile under counted waits, every D element checked against integer arithmetic.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(1024) void canary_mfma(unsigned* rep, int iters, const unsigned short* __restrict__ A, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    bf16x8_t bfrag;
#pragma unroll
    for (int e = 0; e < 8; ++e) bfrag[e] = (__bf16)(float)(((8 * g + e) + i) & 3);          // B[k = 8 g + e][j = i]
    for (int it = 0; it < iters; ++it) {
        bf16x8_t a[8];
        int t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            t[q] = (int)((blockIdx.x * 16u + wave + 7u * (unsigned)(it * 8 + q)) % (unsigned)tiles);
            a[q] = *reinterpret_cast<const bf16x8_t*>(A + ((size_t)t[q] * 16 + i) * 32 + 8 * g);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q], bfrag, d, 0, 0, 0);          // D[row = 4 g + r][col = i]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned want = 0;
                for (int k = 0; k < 32; ++k) want += (hash3(t[q], (4 * g + r) * 32 + k, 21) & 3) * ((k + i) & 3);
                if (d[r] != (float)want) report(rep, 20, (q << 2) | r, it, __float_as_uint(d[r]), __float_as_uint((float)want));
            }
        }
    }
}

}  // namespace

extern "C" {

// W[k][n] = hash3(k, n, 11) & 15 as float (128 x N), the weight stream of canary_fma<*, 2>
__global__ void canary_fill_w(float* W, int N) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 128 * N; i += gridDim.x * blockDim.x) W[i] = (float)(hash3(i / N, i % N, 11) & 15);
}
// A tiles for canary_mfma: tile t, row i, k: bf16(hash3(t, i * 32 + k, 21) & 3)
__global__ void canary_fill_a(unsigned short* A, int tiles) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tiles * 512; i += gridDim.x * blockDim.x) {
        const float f = (float)(hash3(i / 512, i % 512, 21) & 3);
        A[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}
int canary_fill_a_launch(void* A, int tiles, void* stream) {
    hipLaunchKernelGGL(canary_fill_a, dim3(256), dim3(256), 0, (hipStream_t)stream, (unsigned short*)A, tiles);
    return (int)hipGetLastError();
}
int canary_fill_w_launch(void* W, int N, void* stream) {
    hipLaunchKernelGGL(canary_fill_w, dim3(256), dim3(256), 0, (hipStream_t)stream, (float*)W, N);
    return (int)hipGetLastError();
}

// fills a device buffer with the pattern canary_vmem expects (host-side helper: run once)
__global__ void canary_fill(unsigned* buf, int words) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) buf[i] = hash3(7, i, 3);
}
int canary_fill_launch(void* buf, int words, void* stream) {
    hipLaunchKernelGGL(canary_fill, dim3(256), dim3(256), 0, (hipStream_t)stream, (unsigned*)buf, words);
    return (int)hipGetLastError();
}

// kind: 0 VGPR (40 regs), 1 LDS (`words` dwords), 2 VMEM loads (buf, words), 3 LDS broadcast + FMA, 4 VGPR (200 regs, 256 threads)
int canary_launch(int kind, int blocks, int threads, int iters, void* buf, int words, void* rep, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    unsigned* r = (unsigned*)rep;
    switch (kind) {
        case 0: hipLaunchKernelGGL((canary_vgpr<40>), dim3(blocks), dim3(threads), 0, st, r, iters); break;
        case 1: hipLaunchKernelGGL(canary_lds, dim3(blocks), dim3(threads), (size_t)words * 4, st, r, iters, words); break;
        case 2: hipLaunchKernelGGL(canary_vmem, dim3(blocks), dim3(threads), 0, st, r, iters, (const unsigned*)buf, words); break;
        case 3: hipLaunchKernelGGL(canary_bcast, dim3(blocks), dim3(threads), 0, st, r, iters); break;
        case 4: hipLaunchKernelGGL((canary_vgpr<100>), dim3(blocks), dim3(threads), 0, st, r, iters); break;
        case 5: hipLaunchKernelGGL((canary_fma<0, 0>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 6: hipLaunchKernelGGL((canary_fma<0, 1>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 7: hipLaunchKernelGGL((canary_fma<0, 2>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 9: hipLaunchKernelGGL((canary_fma<1, 0>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 10: hipLaunchKernelGGL((canary_fma<1, 1>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 11: hipLaunchKernelGGL((canary_fma<1, 2>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 12: hipLaunchKernelGGL((canary_use<0>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 13: hipLaunchKernelGGL((canary_use<1>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 14: hipLaunchKernelGGL((canary_use<2>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 15: hipLaunchKernelGGL((canary_use<3>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 16: hipLaunchKernelGGL((canary_use<4>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 17: hipLaunchKernelGGL((canary_use<5>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 18: hipLaunchKernelGGL((canary_use<6>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 19: hipLaunchKernelGGL((canary_use<7>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 21: hipLaunchKernelGGL((canary_use<8>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 22: hipLaunchKernelGGL((canary_fma<1, 2, 1>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 23: hipLaunchKernelGGL((canary_fma<1, 2, 2>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 24: hipLaunchKernelGGL((canary_fma<1, 2, 3>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 25: hipLaunchKernelGGL((canary_fma<1, 3, 0>), dim3(blocks), dim3(threads), 0, st, r, iters, (const float*)buf, 512); break;
        case 20: hipLaunchKernelGGL(canary_mfma, dim3(blocks), dim3(threads), 0, st, r, iters, (const unsigned short*)buf, words); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

}  // extern "C"
