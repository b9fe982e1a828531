// Probe: fp32 MFMA issue rate with (a) operands in registers, (b) operands re-read from LDS every group, at 1/2/3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.25f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = lane * 0.01f, a1 = 0.5f, b0 = 0.25f, b1 = lane * 0.02f;
    const float* as = lds + (threadIdx.x >> 6) * 1024 + (lane & 31) * 20 + (lane >> 5) * 4;
    const float* bs = lds + 4096 + (lane >> 5) * 4 * 128 + (lane & 31);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 av0 = *(const f32x4*)(as + kk * 8), av1 = *(const f32x4*)(as + 640 + kk * 8);
                float bv[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[j][q] = bs[(kk * 8 + q) * 128 + j * 32];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], bv[0][q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], bv[1][q], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], bv[0][q], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], bv[1][q], acc[3], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs = 1; wgs <= 3; ++wgs) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wgs), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256 * wgs), dim3(256), 0, 0, out, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            double flop = 256.0 * wgs * 4 /*waves*/ * iters * 32.0 * 4096.0;
            printf("mode %d (%s) %d WG/CU: %.3f ms  %.1f TFLOP/s\n", mode, mode ? "LDS operands" : "register operands", wgs, ms, flop / ms / 1e9);
        }
    return 0;
}
