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

// One 64-wide k chunk of one 32x32 tile: 32 exact-fp32 MFMAs.
__device__ __forceinline__ f32x16 mfma_chunk(const float (&a)[32], const float (&b)[32], f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// C/D layout of the 32x32 MFMA: col = lane&31 (item), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (user).
__device__ __forceinline__ void store_tile(float *__restrict__ S, int64_t n, int64_t B, int64_t user0, int64_t item, int h,
                                           const f32x16 &acc) {
    if (item < n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (u < B) S[u * n + item] = acc[r];
        }
    }
}

// NCHUNK > 0: d <= 64*NCHUNK and the user fragment stays in registers across item tiles; the item
// fragment of the NEXT (tile, chunk) is fetched into the other buffer BEFORE the current tile's stores are
// issued.  gfx950's vmcnt counts stores too, so an un-pipelined loop would drain every tile's 16 stores
// (HBM write latency) before its next operand load could complete — measured MFMA busy 30 %.
// NCHUNK == 0: any d, un-pipelined (user fragment re-read per tile).
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
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * kTilesPerWave;
    const int64_t t1 = (t0 + kTilesPerWave < n_tiles) ? t0 + kTilesPerWave : n_tiles;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if constexpr (NCHUNK == 0) {
        const int nchunk = (d + 63) / 64;
        for (int64_t t = t0; t < t1; ++t) {
            const int64_t jr = t * 32 + i;
            f32x16 acc = zero;
            for (int c = 0; c < nchunk; ++c) {
                float a[32], b[32];
                load_run<VEC>(U, ldu, ur, u_ok, c * 64 + h * 32, d, a);
                load_run<VEC>(I, ldi, jr, jr < n, c * 64 + h * 32, d, b);
                acc = mfma_chunk(a, b, acc);
            }
            store_tile(S, n, B, user_tile * 32, jr, h, acc);
        }
    } else {
        float a[NCHUNK][32];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) load_run<VEC>(U, ldu, ur, u_ok, c * 64 + h * 32, d, a[c]);
        float b0[32], b1[32];
        load_run<VEC>(I, ldi, t0 * 32 + i, t0 * 32 + i < n, h * 32, d, b0);
        if constexpr (NCHUNK == 1) {
            for (int64_t t = t0; t < t1; t += 2) {  // two tiles per trip so the buffer choice is static
                const int64_t j0 = t * 32 + i, j1 = j0 + 32, j2 = j0 + 64;
                if (t + 1 < t1) load_run<VEC>(I, ldi, j1, j1 < n, h * 32, d, b1);
                f32x16 acc = mfma_chunk(a[0], b0, zero);
                store_tile(S, n, B, user_tile * 32, j0, h, acc);
                if (t + 2 < t1) load_run<VEC>(I, ldi, j2, j2 < n, h * 32, d, b0);
                if (t + 1 < t1) {
                    acc = mfma_chunk(a[0], b1, zero);
                    store_tile(S, n, B, user_tile * 32, j1, h, acc);
                }
            }
        } else {  // NCHUNK even: chunk c uses buffer c & 1; the next tile's chunk 0 lands in b0 during the last chunk
            for (int64_t t = t0; t < t1; ++t) {
                const int64_t jr = t * 32 + i, jn = jr + 32;
                f32x16 acc = zero;
#pragma unroll
                for (int c = 0; c < NCHUNK; c += 2) {
                    load_run<VEC>(I, ldi, jr, jr < n, (c + 1) * 64 + h * 32, d, b1);
                    acc = mfma_chunk(a[c], b0, acc);
                    if (c + 2 < NCHUNK) load_run<VEC>(I, ldi, jr, jr < n, (c + 2) * 64 + h * 32, d, b0);
                    else if (t + 1 < t1) load_run<VEC>(I, ldi, jn, jn < n, h * 32, d, b0);
                    acc = mfma_chunk(a[c + 1], b1, acc);
                }
                store_tile(S, n, B, user_tile * 32, jr, h, acc);
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
