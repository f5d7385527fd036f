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

// The same network with the partner's value taken by DPP where a lane's partner sits in its row of 16 (strides 1, 2: quad_perm;
// 4: row_shr / row_shl + select; 8: row_ror) — 18 of the 21 stages — instead of ds_bpermute: no LDS round trip per stage (r06: the
// fold phase of topk_screen.hip's merge kernel was bound by them with 16 waves per CU sorting at once).  Same comparisons, same result.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false); }
template <int J>
__device__ __forceinline__ int lane_xor(int x, int lane) {
    if constexpr (J == 1) return dpp_mov<0xB1>(x);        // quad_perm [1, 0, 3, 2]
    else if constexpr (J == 2) return dpp_mov<0x4E>(x);   // quad_perm [2, 3, 0, 1]
    else if constexpr (J == 4) {
        const int up = dpp_mov<0x114>(x), dn = dpp_mov<0x104>(x);  // row_shr:4 (from lane - 4), row_shl:4 (from lane + 4)
        return (lane & 4) ? up : dn;
    } else if constexpr (J == 8) return dpp_mov<0x128>(x);  // row_ror:8
    else return __shfl_xor(x, J);
}
template <int K, int J>
__device__ __forceinline__ void sort_stage(float &v, int &idx, int lane) {
    const float ov = __int_as_float(lane_xor<J>(__float_as_int(v), lane));
    const int oi = lane_xor<J>(idx, lane);
    const bool up = (lane & K) == 0;
    const bool lower = (lane & J) == 0;
    const bool other_better = ov > v || (ov == v && oi < idx);
    const bool take = (up == lower) ? other_better : !other_better;
    if (take) {
        v = ov;
        idx = oi;
    }
    if constexpr (J > 1) sort_stage<K, J / 2>(v, idx, lane);
}
__device__ __forceinline__ void wave_sort_desc_dpp(float &v, int &idx, int lane) {
    sort_stage<2, 1>(v, idx, lane);
    sort_stage<4, 2>(v, idx, lane);
    sort_stage<8, 4>(v, idx, lane);
    sort_stage<16, 8>(v, idx, lane);
    sort_stage<32, 16>(v, idx, lane);
    sort_stage<64, 32>(v, idx, lane);
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
