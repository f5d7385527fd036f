// mfma_common.h — pieces shared by the fp32-MFMA kernels (score, top-k, lse / InfoNCE, NGCF dense layers).
//
// All of them use v_mfma_f32_32x32x2_f32 with the same operand walk: the k-sum is order-free, so lane-half h = lane >> 5
// walks k = 64 c + 32 h + s (s = 0..31) and a lane's A or B fragment of a 64-wide k chunk is ONE contiguous 32-float run
// of "its" row (lane & 31).  The accumulator (C/D) layout is col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
#pragma once

#include <hip/hip_runtime.h>

namespace rbg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// row of the 32x32 accumulator tile held by register `reg` of lane-half `h`
__host__ __device__ constexpr int mfma_rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// The 32-float run p[k0 .. k0+31] of a row, zero beyond d (and all zero when !ok: an out-of-range row).
//   RUN_FAST : eight unconditional float4 loads (the run is inside the row, 16-byte aligned)
//   RUN_VEC  : float4 loads, each guarded by k < d  (d % 4 == 0, aligned rows)
//   RUN_ANY  : scalar guarded loads
enum { RUN_ANY = 0, RUN_VEC = 1, RUN_FAST = 2 };
template <int MODE>
__device__ __forceinline__ void load_run32f(const float *p, bool ok, int k0, int d, float (&r)[32]) {
    if (MODE == RUN_FAST) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(p + k0 + 4 * q);
            r[4 * q + 0] = v.x;
            r[4 * q + 1] = v.y;
            r[4 * q + 2] = v.z;
            r[4 * q + 3] = v.w;
        }
    } else if (MODE == RUN_VEC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && k0 + 4 * q < d) v = *reinterpret_cast<const float4 *>(p + k0 + 4 * q);
            r[4 * q + 0] = v.x;
            r[4 * q + 1] = v.y;
            r[4 * q + 2] = v.z;
            r[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) r[s] = (ok && k0 + s < d) ? p[k0 + s] : 0.f;
    }
}

}  // namespace rbg
