// ordered.h — row scatters without float atomics (option "deterministic").
//
// The mini-batch kernels of train.hip and lse.hip add one contribution per batch occurrence onto the row of its id; ids repeat inside
// a batch, so by default the adds are float atomics (what torch's GPU index_put_(accumulate=True) / index_add_ do) and the low
// bits of those rows depend on the order the hardware serves them in.  With the option on, a row is written by ONE wavefront — the
// one whose occurrence is the first with that key — which applies all the occurrences of the key in batch order with plain
// read-add-writes: bit-identical from run to run.  Finding the occurrences is a brute-force scan of the batch's keys by every
// wavefront (n / 64 coalesced steps for n occurrences: ~ 10 us at 3 x 4096 ids); no sort, no scratch memory, capturable.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbg {

template <bool ORDERED>
__device__ __forceinline__ void row_add(float *p, float v) {
    if constexpr (ORDERED) *p += v;  // this wavefront owns the row
    else atomicAdd(p, v);
}

// One wavefront per occurrence w of [0, n).  C supplies
//   int segments(int64_t w, KeySeg (&seg)[2], int64_t &mine) const   the id arrays occurrence w can share its id with, in occurrence order
//        (users scan the batch's users; items its positives, then its negatives), each with the occurrence number of its first
//        element, and w's own id
//   void apply(int64_t m, int lane) const  add occurrence m's contribution with row_add<true> (called by the whole wavefront)
// The scan is O(n^2 / 64) wavefront steps and instruction-bound, so a step is kept to a load, a compare and a scalar OR: the ids in
// front of w only answer "is there an earlier occurrence" (one test after all of them), the ids from w on are applied where they match.
// (First form: a three-way select per id through generic pointers, a wave index the compiler took for divergent, two branches per
// 64 ids: 37-39 us per launch at 3 x 2048 ids; this form: 29 us — what remains is 6 144 wavefronts reading the same 48 KB of ids out
// of L2, 160 MB in all; staging them in LDS per workgroup is the next step if this mode ever matters for speed.)
struct KeySeg {
    const int64_t *keys;
    int64_t n, first;
};

template <class C>
__global__ __launch_bounds__(256) void ordered_scatter_kernel(const C c, int64_t n) {
    typedef const __attribute__((address_space(1))) int64_t *gkeys;  // (global loads: through the struct the pointers are generic = flat)
    const int lane = threadIdx.x & 63;
    // wave-uniform, and known to be: the whole scan is scalar control flow around one vector load and compare per 64 ids
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= n) return;
    KeySeg seg[2];
    int64_t mine;
    const int nseg = c.segments(w, seg, mine);
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const gkeys keys = (gkeys)seg[sgi].keys;
        const int64_t cnt = seg[sgi].n, first = seg[sgi].first;
        const int64_t wrel = w - first;
        const int64_t nb = wrel <= 0 ? 0 : (wrel >= cnt ? cnt : wrel);  // ids of this array in front of w
        unsigned long long earlier = 0ull;
        int64_t j = 0;
        for (; j + 256 <= nb; j += 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) earlier |= __ballot(keys[j + 64 * u + lane] == mine);
        }
        for (; j < nb; j += 64) earlier |= __ballot(j + lane < nb && keys[j + lane] == mine);
        if (earlier) return;  // an earlier occurrence of the id owns the row
        for (j = nb; j < cnt; j += 64) {
            unsigned long long mask = __ballot(j + lane < cnt && keys[j + lane] == mine);
            while (mask) {
                const int bit = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                c.apply(first + j + bit, lane);
            }
        }
    }
}

template <class C>
inline void launch_ordered_scatter(const C &c, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((ordered_scatter_kernel<C>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, c, n);
}

}  // namespace rbg
