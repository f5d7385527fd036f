// score.hip — the dense user x item scoring GEMM of full_sort_predict on the gfx950 matrix cores.
//
// Replaces  scores = torch.matmul(u_embeddings, self.restore_item_e.transpose(0, 1))
//   recbole_gnn/model/general_recommender/lightgcn.py:131 (ngcf.py:147, sgl.py:240).
//
// S[B, n] = U[B, d] · I[n, d]^T in exact fp32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain per
// output, MI355X guide §3).  This is the only MFMA use on the path: at d = 64 the GEMM has 32 flop
// per output byte, i.e. it sits at the fp32-MFMA / HBM-write balance point, so the kernel streams
// item rows once per 128-user block and writes each score exactly once.
//
// Mapping: a wavefront owns a 32-user x 32-item tile.  For v_mfma_f32_32x32x2_f32 lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; since the sum over k is order-free we let lane-half
// h = l>>5 walk k = kc + 32h + s (s = 0..31), so each lane reads ONE contiguous 128-byte run of its
// user row and of its item row per 64-wide k chunk (8 x global_load_dwordx4), no LDS.  The 4 waves of
// a workgroup hold 4 different user tiles and walk the same item tiles, so item rows are fetched from
// L2 once per workgroup and hit in L1 for the other three waves.

#include <hip/hip_runtime.h>

#include <algorithm>

#include "internal.h"

namespace rbg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTilesPerWave = 8;  // item tiles walked by one workgroup

// 32 floats of row `row` starting at k0 (zero beyond d or when the row is out of range).
template <bool VEC>
__device__ __forceinline__ void load_run(const float *base, int64_t ld, int64_t row, bool row_ok, int k0, int d,
                                         float (&r)[32]) {
    const float *p = base + row * ld + k0;
    if (VEC) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && k0 + 4 * q < d) v = *reinterpret_cast<const float4 *>(p + 4 * q);
            r[4 * q + 0] = v.x;
            r[4 * q + 1] = v.y;
            r[4 * q + 2] = v.z;
            r[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) r[s] = (row_ok && k0 + s < d) ? p[s] : 0.f;
    }
}

// NCHUNK > 0: d <= 64*NCHUNK and the user fragment stays in registers across item tiles.
// NCHUNK == 0: any d, the user fragment is re-read per item tile (L1/L2 resident).
template <int NCHUNK, bool VEC>
__global__ __launch_bounds__(256) void score_kernel(const float *__restrict__ U, int64_t ldu,
                                                    const float *__restrict__ I, int64_t ldi, float *__restrict__ S,
                                                    int64_t B, int64_t n, int d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t user_tile = (int64_t)blockIdx.y * 4 + wave;
    const int64_t ur = user_tile * 32 + i;
    if (user_tile * 32 >= B) return;  // whole wave out of range (no barriers below)
    const bool u_ok = ur < B;
    const int nchunk = (d + 63) / 64;

    float a[(NCHUNK > 0 ? NCHUNK : 1)][32];
    if (NCHUNK > 0) {
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) load_run<VEC>(U, ldu, ur, u_ok, c * 64 + h * 32, d, a[c]);
    }

    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * kTilesPerWave;
    for (int64_t t = t0; t < t0 + kTilesPerWave && t < n_tiles; ++t) {
        const int64_t jr = t * 32 + i;
        const bool j_ok = jr < n;
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (NCHUNK > 0) {
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                float b[32];
                load_run<VEC>(I, ldi, jr, j_ok, c * 64 + h * 32, d, b);
#pragma unroll
                for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s], b[s], acc, 0, 0, 0);
            }
        } else {
            for (int c = 0; c < nchunk; ++c) {
                float b[32];
                load_run<VEC>(U, ldu, ur, u_ok, c * 64 + h * 32, d, a[0]);
                load_run<VEC>(I, ldi, jr, j_ok, c * 64 + h * 32, d, b);
#pragma unroll
                for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][s], b[s], acc, 0, 0, 0);
            }
        }
        // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int64_t item = t * 32 + i;
        if (item < n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t u = user_tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (u < B) S[u * n + item] = acc[r];
            }
        }
    }
}

template <int NCHUNK>
static int launch_score(const float *U, int64_t ldu, const float *I, int64_t ldi, float *S, int64_t B, int64_t n, int d,
                        bool vec, hipStream_t s) {
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t gx = (n_tiles + kTilesPerWave - 1) / kTilesPerWave;
    const int64_t gy = (B + 127) / 128;
    if (gx > INT32_MAX || gy > 65535) return fail(RBG_EUNSUPPORTED, "score grid too large (B = %lld)", (long long)B);
    dim3 grid((unsigned)gx, (unsigned)gy);
    if (vec)
        hipLaunchKernelGGL((score_kernel<NCHUNK, true>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d);
    else
        hipLaunchKernelGGL((score_kernel<NCHUNK, false>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" int rbg_score_f32(const float *U, int64_t ldu, const float *I, int64_t ldi, float *S, int64_t B, int64_t n,
                             int d, void *stream) {
    clear_error();
    if (B < 0 || n < 0 || d <= 0) return fail(RBG_ESHAPE, "B = %lld, n = %lld, d = %d", (long long)B, (long long)n, d);
    if (ldu < d || ldi < d) return fail(RBG_ESHAPE, "row stride smaller than d");
    if (B == 0 || n == 0) return RBG_OK;
    if (!U || !I || !S) return fail(RBG_EINVAL, "NULL pointer");
    const bool vec = (d % 4 == 0) && (ldu % 4 == 0) && (ldi % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(I)) & 15u) == 0;
    hipStream_t s = (hipStream_t)stream;
    if (d <= 64) return launch_score<1>(U, ldu, I, ldi, S, B, n, d, vec, s);
    if (d <= 128) return launch_score<2>(U, ldu, I, ldi, S, B, n, d, vec, s);
    if (d <= 256) return launch_score<4>(U, ldu, I, ldi, S, B, n, d, vec, s);
    return launch_score<0>(U, ldu, I, ldi, S, B, n, d, vec, s);
}
