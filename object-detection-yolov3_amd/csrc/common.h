// Shared helpers for the gfx950 kernels of libyolo3hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/yolo3hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void y3_set_error(const char* fmt, ...);

#define Y3_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            y3_set_error(__VA_ARGS__);   \
            return Y3_EINVAL;            \
        }                                \
    } while (0)

#define Y3_CHECK_LAUNCH(what)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess) {                                            \
            y3_set_error("%s: %s", what, hipGetErrorString(e_));          \
            return Y3_ELAUNCH;                                             \
        }                                                                  \
    } while (0)

static inline int y3_ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline bool y3_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int y3_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// TF padding='same': pad_before for one spatial axis
static inline int y3_same_pad_before(int size, int k, int s) {
    int out = (size + s - 1) / s;
    int total = (out - 1) * s + k - size;
    if (total < 0) total = 0;
    return total / 2;
}

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2): remap so
// that every XCD works on one contiguous run of tile ids (bijective for any grid).
__device__ __forceinline__ int y3_xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = orig & 7, j = orig >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it (a plain
// __syncthreads() also waits for vmcnt(0)).
__device__ __forceinline__ void y3_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float y3_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double y3_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// x / d for 0 <= x < 2^31 without a divide: the compiler's sequence for a run-time divisor is ~30 dependent instructions
// (float reciprocal + two correction steps), and the index decode of a workgroup needs five of them before its first load.
//   d == 1: mul == 0 (identity);  d == 2^k: shift = k - 1, mul = 2^31 + 1;  else shift = floor(log2 d), mul = floor(2^(32+shift) / d) + 1
// (error term mul * d - 2^(32+shift) <= d, so floor is exact while x * d < 2^(32+shift), i.e. for every x < 2^31).
struct Y3Div {
    unsigned mul;
    int shift;
};
static inline Y3Div y3_make_div(int d) {
    Y3Div r = {0u, 0};
    if (d <= 1) return r;
    int s = 0;
    while ((2LL << s) <= d) ++s;      // floor(log2 d)
    if ((1LL << s) == d) {
        r.shift = s - 1;
        r.mul = 0x80000001u;
    } else {
        r.shift = s;
        r.mul = (unsigned)(((1ULL << (32 + s)) / (unsigned long long)d) + 1ULL);
    }
    return r;
}
__device__ __forceinline__ int y3_div(int x, const Y3Div d) { return d.mul ? (int)(__umulhi((unsigned)x, d.mul) >> d.shift) : x; }
#define Y3_PIN_S(x) asm volatile("" : "+s"(x))

