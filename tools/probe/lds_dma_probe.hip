// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS or leave it untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, unsigned bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4; i += 64) lds[i] = -7.f;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    // even lanes in range, odd lanes out of range
    const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4; i += 64) out[i] = lds[i];
}
int main() {
    float h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, d, (unsigned)sizeof(h), o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane0: %g %g %g %g | lane1 (OOB): %g %g %g %g | lane2: %g %g %g %g | lane3 (OOB): %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12]);
    return 0;
}
