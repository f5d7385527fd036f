// mfma_common.h — pieces shared by the fp32-MFMA kernels (score, top-k, lse / InfoNCE, NGCF dense layers).
//
// All of them use v_mfma_f32_32x32x2_f32 with the same operand walk: the k-sum is order-free, so lane-half h = lane >> 5
// walks k = 64 c + 32 h + s (s = 0..31) and a lane's A or B fragment of a 64-wide k chunk is ONE contiguous 32-float run
// of "its" row (lane & 31).  The accumulator (C/D) layout is col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
#pragma once

#include <hip/hip_runtime.h>

namespace rbg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// A 32-row x 64 NCHUNK operand tile staged through LDS: the 256 threads of a workgroup fetch it with coalesced float4
// loads (consecutive threads = consecutive 16 bytes; "every lane reads its own 128-byte run" would touch 64 cache lines
// per instruction), publish it with row stride LD = 64 NCHUNK + 4 floats (16-byte aligned rows: full-rate b128 reads of
// the fragments), and every wave multiplies it against its register-resident A fragments.  Rows >= n_rows are clamped
// to the last row: callers never use what they produce.
template <int NCHUNK, int MODE>
struct RowTile {
    static constexpr int LD = NCHUNK * 64 + 4;
    float4 stage[NCHUNK * 2];

    __device__ __forceinline__ void fetch(const float *base, int64_t ld, int64_t n_rows, int d, int64_t tile, int tid) {
        fetch_rows(base, ld, n_rows, d, tile * 32, tid);
    }
    // rows [row0, row0 + 32), clamped into [0, n_rows) on both sides (row0 may be negative: a tile shifted to a line boundary)
    __device__ __forceinline__ void fetch_rows(const float *base, int64_t ld, int64_t n_rows, int d, int64_t row0, int tid) {
#pragma unroll
        for (int k = 0; k < NCHUNK * 2; ++k) {
            const int f = tid + 256 * k, row = f / (NCHUNK * 16), c4 = (f % (NCHUNK * 16)) * 4;
            const int64_t r = row0 + row;
            const float *src = base + (r < n_rows ? (r < 0 ? 0 : r) : n_rows - 1) * ld + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == RUN_FAST) {
                v = *reinterpret_cast<const float4 *>(src);
            } else if (MODE == RUN_VEC) {
                if (c4 < d) v = *reinterpret_cast<const float4 *>(src);
            } else {
                if (c4 + 0 < d) v.x = src[0];
                if (c4 + 1 < d) v.y = src[1];
                if (c4 + 2 < d) v.z = src[2];
                if (c4 + 3 < d) v.w = src[3];
            }
            stage[k] = v;
        }
    }
    __device__ __forceinline__ void publish(float (*tile)[LD], int tid) const {
#pragma unroll
        for (int k = 0; k < NCHUNK * 2; ++k) {
            const int f = tid + 256 * k, row = f / (NCHUNK * 16), c4 = (f % (NCHUNK * 16)) * 4;
            *reinterpret_cast<float4 *>(&tile[row][c4]) = stage[k];
        }
    }
    // acc[i = A row][j = tile row]: lane (i, h) walks k = 64c + 32h + s of tile row i, as its A fragment does
    static __device__ __forceinline__ f32x16 product(const float (*tile)[LD], const float (&a)[NCHUNK][32], int i, int h) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const float4 b = *reinterpret_cast<const float4 *>(&tile[i][c * 64 + h * 32 + s4 * 4]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s4 * 4 + 0], b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s4 * 4 + 1], b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s4 * 4 + 2], b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s4 * 4 + 3], b.w, acc, 0, 0, 0);
            }
        return acc;
    }
};

// ---- split-precision operands: fp32 = hi + mid + lo in bf16 -------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (1/16 of the bf16 matrix rate).  Splitting every fp32 operand into
// three bf16 terms (round-to-nearest residuals: x = h + m + l up to 2^-24 |x|) and forming the SIX products whose order is
// >= 2^-16 (h h, h m, m h, h l, l h, m m — the dropped ones are <= 2^-23 relative) on v_mfma_f32_32x32x16_bf16 gives
// the same accuracy as the exact-fp32 chain (measured: 2.1e-7 vs 2.3e-7 relative to float64 at d = 64, 5.8e-7 vs 6.3e-7
// at d = 256, profiles/r02_bf16x3_probe.jsonl) at 2.3x its throughput (6 x 32 instead of 8 x 64 cycles per 16 k).
// The k-sum is order-free, so lane-half h keeps its contiguous run k = 64c + 32h + [0, 32): MFMA (c, m) consumes elements
// [8m, 8m + 8) of that run from both halves, for A and B alike.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_bf16(float x0, float x1, bf16x2 &h, bf16x2 &m, bf16x2 &l) {
    const f32x2v v = {x0, x1};
    h = __builtin_convertvector(v, bf16x2);                       // v_cvt_pk_bf16_f32 (RNE)
    const f32x2v r1 = v - __builtin_convertvector(h, f32x2v);    // exact
    m = __builtin_convertvector(r1, bf16x2);
    const f32x2v r2 = r1 - __builtin_convertvector(m, f32x2v);
    l = __builtin_convertvector(r2, bf16x2);
}

template <int NCHUNK>
struct AFrag3 {  // a lane's A operand: NCHUNK runs of 32 floats, split; 48 NCHUNK registers
    bf16x8 h[NCHUNK * 4], m[NCHUNK * 4], l[NCHUNK * 4];
};

template <int NCHUNK>
__device__ __forceinline__ void split_a(const float (&a)[NCHUNK][32], AFrag3<NCHUNK> &f) {
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x2 h, m, l;
                split2_bf16(a[c][8 * q + 2 * j], a[c][8 * q + 2 * j + 1], h, m, l);
                f.h[c * 4 + q][2 * j] = h[0];
                f.h[c * 4 + q][2 * j + 1] = h[1];
                f.m[c * 4 + q][2 * j] = m[0];
                f.m[c * 4 + q][2 * j + 1] = m[1];
                f.l[c * 4 + q][2 * j] = l[0];
                f.l[c * 4 + q][2 * j + 1] = l[1];
            }
}

// RowTile with the tile published as three bf16 planes (row stride 64 NCHUNK + 8 elements = 128 NCHUNK + 16 bytes:
// 16-byte aligned rows, conflict-free b128 fragment reads).  Same fetch as RowTile; 3 x 32 x (64 NCHUNK + 8) x 2 bytes of LDS.
template <int NCHUNK, int MODE, int THREADS = 256>
struct RowTile3 {
    static constexpr int LDH = NCHUNK * 64 + 8;
    static constexpr int NSTAGE = NCHUNK * 512 / THREADS;  // float4s a thread fetches per tile (THREADS threads share it)
    typedef __bf16 Planes[3][32][LDH];
    float4 stage[NSTAGE];

    __device__ __forceinline__ void fetch(const float *base, int64_t ld, int64_t n_rows, int d, int64_t tile, int tid) {
        fetch_rows(base, ld, n_rows, d, tile * 32, tid);
    }
    __device__ __forceinline__ void fetch_rows(const float *base, int64_t ld, int64_t n_rows, int d, int64_t row0, int tid) {
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int f = tid + THREADS * k, row = f / (NCHUNK * 16), c4 = (f % (NCHUNK * 16)) * 4;
            const int64_t r = row0 + row;
            const float *src = base + (r < n_rows ? (r < 0 ? 0 : r) : n_rows - 1) * ld + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == RUN_FAST) {
                v = *reinterpret_cast<const float4 *>(src);
            } else if (MODE == RUN_VEC) {
                if (c4 < d) v = *reinterpret_cast<const float4 *>(src);
            } else {
                if (c4 + 0 < d) v.x = src[0];
                if (c4 + 1 < d) v.y = src[1];
                if (c4 + 2 < d) v.z = src[2];
                if (c4 + 3 < d) v.w = src[3];
            }
            stage[k] = v;
        }
    }
    __device__ __forceinline__ void publish(Planes &tile, int tid) const {
#pragma unroll
        for (int k = 0; k < NSTAGE; ++k) {
            const int f = tid + THREADS * k, row = f / (NCHUNK * 16), c4 = (f % (NCHUNK * 16)) * 4;
            bf16x2 h0, m0, l0, h1, m1, l1;
            split2_bf16(stage[k].x, stage[k].y, h0, m0, l0);
            split2_bf16(stage[k].z, stage[k].w, h1, m1, l1);
            const bf16x4 hv = {h0[0], h0[1], h1[0], h1[1]}, mv = {m0[0], m0[1], m1[0], m1[1]}, lv = {l0[0], l0[1], l1[0], l1[1]};
            *reinterpret_cast<bf16x4 *>(&tile[0][row][c4]) = hv;
            *reinterpret_cast<bf16x4 *>(&tile[1][row][c4]) = mv;
            *reinterpret_cast<bf16x4 *>(&tile[2][row][c4]) = lv;
        }
    }
    // acc[i = A row][j = tile row]; small terms first
    static __device__ __forceinline__ f32x16 product(const Planes &tile, const AFrag3<NCHUNK> &a, int i, int h) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = c * 64 + h * 32 + q * 8;
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(&tile[0][i][off]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8 *>(&tile[1][i][off]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(&tile[2][i][off]);
                const int s = c * 4 + q;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s], bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s], bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s], bh, acc, 0, 0, 0);
            }
        return acc;
    }
};

// ---- split-precision operands, fp16 form (r06): fp32 = hi + lo in fp16 after a power-of-two scale -----------------------
// For operands of known magnitude (unit rows, weights in [0, 1]) two fp16 terms (11 significand bits each: x = h + l up to
// 2^-22 |x|) and the THREE products h h, h l, l h on v_mfma_f32_32x32x16_f16 stand in for the six bf16 products above at half the
// matrix-core time, a third less LDS traffic and a third fewer fragment registers; the price is two bits (relative error of a
// product <= 3 x 2^-22 instead of ~ 2^-23).  The power-of-two `scale` (exact; the caller folds its inverse into what consumes the
// accumulator) keeps the LOW term a normal fp16 number wherever that matters: l is below fp16's smallest normal 2^-14 only for
// |x| scale < 2^-3, and is then rounded with an ABSOLUTE error <= 2^-25 / scale (1.2e-10 at scale = 2^8) instead of 2^-11 |l| —
// (the matrix core does NOT flush subnormal fp16 inputs: devtools/microbench/mfma_f16_denorm.hip, profiles/r06_mfma_f16_denorm.jsonl).
// Overflow needs |x| scale > 65 504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_f16(float x0, float x1, float scale, f16x2 &h, f16x2 &l) {
    const f32x2v v = {x0 * scale, x1 * scale};                    // (power of two: exact)
    h = __builtin_convertvector(v, f16x2);                        // v_cvt_pk_f16_f32 (RNE)
    const f32x2v r = v - __builtin_convertvector(h, f32x2v);     // exact
    l = __builtin_convertvector(r, f16x2);
}

template <int NCHUNK>
struct AFrag2 {  // a lane's operand: NCHUNK runs of 32 floats as two fp16 terms; 32 NCHUNK registers
    f16x8 h[NCHUNK * 4], l[NCHUNK * 4];
};

template <int NCHUNK>
__device__ __forceinline__ void split_a_f16(const float (&a)[NCHUNK][32], float scale, AFrag2<NCHUNK> &f) {
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x2 h, l;
                split2_f16(a[c][8 * q + 2 * j], a[c][8 * q + 2 * j + 1], scale, h, l);
                f.h[c * 4 + q][2 * j] = h[0];
                f.h[c * 4 + q][2 * j + 1] = h[1];
                f.l[c * 4 + q][2 * j] = l[0];
                f.l[c * 4 + q][2 * j + 1] = l[1];
            }
}

}  // namespace rbg
