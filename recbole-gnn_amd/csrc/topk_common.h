// topk_common.h — pieces shared by the fused top-k kernels (topk.hip: the exact passes; topk_screen.hip: the bf16 screen + exact
// rescoring of r06): the total order of (score, item), the 64-lane bitonic sort, the history test against a graph row.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbg {

constexpr float kNegInf = -__builtin_inff();

// (va, ia) "better than" (vb, ib): higher score first, lower item id on ties (a total order -> deterministic)
__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// 64-lane bitonic sort, best first.
__device__ __forceinline__ void wave_sort_desc(float &v, int &idx, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float ov = __shfl_xor(v, j);
            const int oi = __shfl_xor(idx, j);
            const bool up = (lane & k) == 0;        // this block sorts best-first
            const bool lower = (lane & j) == 0;     // I hold the earlier position of the pair
            const bool other_better = better(ov, oi, v, idx);
            const bool take = (up == lower) ? other_better : !other_better;
            if (take) {
                v = ov;
                idx = oi;
            }
        }
    }
}

// Is `item` in the training history of `user` (= a column of the user's graph row)?  Binary search; called by
// up to 64 lanes at once so the chain of dependent loads is paid once per batch, not once per candidate.
__device__ __forceinline__ bool in_history(const int32_t *rowptr, const int32_t *col, int64_t n_users, int64_t user, int item) {
    if (!rowptr || user < 0 || item < 0) return false;
    int lo = rowptr[user], hi = rowptr[user + 1];
    const int target = (int)(item + n_users);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == target) return true;
        if (c < target) lo = mid + 1; else hi = mid;
    }
    return false;
}

// The same test against a copy of the row's head in LDS (r05, the merge and threshold kernels: one wavefront per user).  The
// binary search of in_history() is a chain of ~ log2(degree) dependent global loads (~ 0.5 us each) paid by every kernel that
// masks; staged, the chain is two loads (rowptr, the row) and the search runs at LDS latency.  Columns past the staged head are
// still read from global memory.
constexpr int kHistStage = 128;  // (merge kernel: 32 KB of staged lists + 2 KB of history per workgroup = four workgroups per CU, all 4 096 users in one round)
struct HistRow {
    const __attribute__((address_space(1))) int32_t *col;  // (global, not generic: a flat load in the search loop would make every wait a full one)
    const int *lds;
    int lo, hi, staged;
    int64_t n_users;
    // two steps, so that a caller can put its own loads between them (they then travel with the row's)
    __device__ __forceinline__ void begin(const int32_t *rowptr, const int32_t *col_, int64_t n_users_, int64_t user, int *buf) {
        col = (const __attribute__((address_space(1))) int32_t *)col_, lds = buf, n_users = n_users_;
        lo = hi = staged = 0;
        if (!rowptr || user < 0) return;
        lo = rowptr[user], hi = rowptr[user + 1];
    }
    __device__ __forceinline__ void finish(int *buf, int lane) {
        staged = hi - lo < kHistStage ? hi - lo : kHistStage;
        for (int e = lane; e < staged; e += 64) buf[e] = col[lo + e];
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void stage(const int32_t *rowptr, const int32_t *col_, int64_t n_users_, int64_t user, int *buf, int lane) {
        begin(rowptr, col_, n_users_, user, buf);
        finish(buf, lane);
    }
    __device__ __forceinline__ bool has(int item) const {
        if (item < 0) return false;
        int a = lo, b = hi;
        const int target = (int)(item + n_users);
        while (a < b) {
            const int mid = (a + b) >> 1;
            const int c = (mid - lo < staged) ? lds[mid - lo] : col[mid];
            if (c == target) return true;
            if (c < target) a = mid + 1; else b = mid;
        }
        return false;
    }
};

}  // namespace rbg
