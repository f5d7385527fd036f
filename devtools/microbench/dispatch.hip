// Workgroup / wave dispatch rate on MI355X: kernels that do (almost) nothing, at the grids the SpMM launches use.
#include <hip/hip_runtime.h>
#include <stdint.h>
template <int VG>
__global__ void empty_kernel(float *out, int spin) {
    // VG: keep roughly VG VGPRs live so the launch has to allocate them
    float a[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) a[i] = (float)(threadIdx.x + i);
    for (int s = 0; s < spin; ++s) {
#pragma unroll
        for (int i = 0; i < VG; ++i) a[i] = a[i] * 1.0001f + 1.f;
        __builtin_amdgcn_s_sleep(32);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) t += a[i];
    if (t == 12345.678f) out[0] = t;
}
template <int LDSB>
__global__ void lds_kernel(float *out, int spin) {
    __shared__ float buf[LDSB / 4];
    buf[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(32);
    if (buf[(threadIdx.x + 1) & 63] == 12345.678f) out[0] = 1.f;
}
extern "C" int mb_dispatch(float *out, int grid, int block, int vg, int spin, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (vg == 4) hipLaunchKernelGGL(empty_kernel<4>, dim3(grid), dim3(block), 0, s, out, spin);
    else if (vg == 32) hipLaunchKernelGGL(empty_kernel<32>, dim3(grid), dim3(block), 0, s, out, spin);
    else if (vg == 60) hipLaunchKernelGGL(empty_kernel<60>, dim3(grid), dim3(block), 0, s, out, spin);
    else if (vg == 120) hipLaunchKernelGGL(empty_kernel<120>, dim3(grid), dim3(block), 0, s, out, spin);
    else if (vg == -16) hipLaunchKernelGGL(lds_kernel<16384>, dim3(grid), dim3(block), 0, s, out, spin);
    else return -1;
    return (int)hipGetLastError();
}
