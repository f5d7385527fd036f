// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  A = a subnormal value everywhere, B = 1: the accumulator should hold 16 x that value.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float *out) {
    f16x8 A, B;
    for (int j = 0; j < 8; ++j) A[j] = (_Float16)a, B[j] = (_Float16)b;
    f32x16 x = {0};
    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, x, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = x[0];
}
int main() {
    float *d, h;
    hipMalloc(&d, 4);
    const float vals[] = {1.0f, 6.103515625e-05f /* 2^-14: smallest normal */, 3.0517578125e-05f /* 2^-15 */, 5.9604644775390625e-08f /* 2^-24 */};
    for (float a : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1.0f, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("{\"a\": %.10e, \"b\": 1, \"acc\": %.10e, \"expected\": %.10e}\n", a, h, 16.0 * a);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1.0f, a, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("{\"a\": 1, \"b\": %.10e, \"acc\": %.10e, \"expected\": %.10e}\n", a, h, 16.0 * a);
    }
    return 0;
}
