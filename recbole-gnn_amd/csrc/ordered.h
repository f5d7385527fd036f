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
//   int64_t key(int64_t m) const          the row an occurrence adds onto (occurrences with equal keys are served together)
//   void range(int64_t w, int64_t &lo, int64_t &hi) const   the occurrences that can share w's key (a sub-range of [0, n))
//   void apply(int64_t m, int lane) const  add occurrence m's contribution with row_add<true> (called by the whole wavefront)
template <class C>
__global__ __launch_bounds__(256) void ordered_scatter_kernel(const C c, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n) return;
    int64_t lo, hi;
    c.range(w, lo, hi);
    const int64_t mine = c.key(w);
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t m = base + lane;
        unsigned long long mask = __ballot(m < hi && c.key(m) == mine);
        if (base < w) {  // an earlier occurrence of the key owns the row
            const unsigned long long below = (w - base >= 64) ? ~0ull : ((1ull << (w - base)) - 1ull);
            if (mask & below) return;
        }
        while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            c.apply(base + bit, lane);
        }
    }
}

template <class C>
inline void launch_ordered_scatter(const C &c, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((ordered_scatter_kernel<C>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, c, n);
}

}  // namespace rbg
