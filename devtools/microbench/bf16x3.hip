// 3 x bf16 split-precision GEMM probe: S = U I^T with every fp32 operand split into three bf16 terms (hi + mid + lo, RNE)
// and the six products of order <= 2^-16 formed on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Frag3 { bf16x8 h, m, l; };

__device__ __forceinline__ Frag3 split8(const float *x) {
    Frag3 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 v = {x[2 * j], x[2 * j + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        f.h[2 * j] = h[0]; f.h[2 * j + 1] = h[1];
        f.m[2 * j] = m[0]; f.m[2 * j + 1] = m[1];
        f.l[2 * j] = l[0]; f.l[2 * j + 1] = l[1];
    }
    return f;
}

__device__ __forceinline__ f32x16 mma6(const Frag3 &a, const Frag3 &b, f32x16 acc, int terms) {
    if (terms >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    }
    if (terms >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
    return acc;
}

// one wave per 32x32 tile, operands straight from global memory (correctness probe, not a fast kernel); d % 16 == 0
__global__ void gemm_kernel(const float *U, const float *I, float *S, int B, int n, int d, int terms) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int tu = blockIdx.y, ti = blockIdx.x;
    const int ur = min(tu * 32 + i, B - 1), ir = min(ti * 32 + i, n - 1);
    f32x16 acc = {0};
    for (int k = 0; k < d; k += 16) {
        float a[8], b[8];
        for (int j = 0; j < 8; ++j) {
            a[j] = U[(int64_t)ur * d + k + 8 * h + j];
            b[j] = I[(int64_t)ir * d + k + 8 * h + j];
        }
        if (terms == 0) {  // exact fp32 MFMA: 8 instructions for the same 16 k (lane-half h supplies k = 8h + j at step j)
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        } else {
            acc = mma6(split8(a), split8(b), acc, terms);
        }
    }
    const int item = ti * 32 + i;
    for (int r = 0; r < 16; ++r) {
        const int u = tu * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (u < B && item < n) S[(int64_t)u * n + item] = acc[r];
    }
}

__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, int mode) {
    Frag3 a, b;
    for (int j = 0; j < 8; ++j) {
        a.h[j] = (__bf16)(float)(threadIdx.x + j); a.m[j] = (__bf16)0.5f; a.l[j] = (__bf16)0.25f;
        b.h[j] = (__bf16)1.0f; b.m[j] = (__bf16)0.125f; b.l[j] = (__bf16)0.0625f;
    }
    f32x16 acc = {0};
    float fa = threadIdx.x, fb = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        } else {
            acc = mma6(a, b, acc, 6);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}

extern "C" int mb_bf16x3_gemm(const float *U, const float *I, float *S, int B, int n, int d, int terms, void *stream) {
    hipLaunchKernelGGL(gemm_kernel, dim3((n + 31) / 32, (B + 31) / 32), dim3(64), 0, (hipStream_t)stream, U, I, S, B, n, d, terms);
    return (int)hipGetLastError();
}
extern "C" int mb_bf16x3_rate(float *out, int blocks, int iters, int mode, void *stream) {
    hipLaunchKernelGGL(rate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, mode);
    return (int)hipGetLastError();
}
