// sell_kernel.h — device code of the column-slab propagation: one wave per unit, one launch per layer (sell.hip).  DESIGN 2.1c.
// (Measured and moved out of the product: two gather batches in flight per wave, units walked by fewer / longer waves, one
// launch per row class, a persistent K-layer launch, a resident round of waves with a longest-first schedule — devtools/experiments.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "internal.h"

namespace rbg {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int kSellPast = 0x7ffffff0;  // padding slots: past every table this path accepts, a buffer load returns zeros

struct SellParams {
    const v4i *ent;          // pairs of entries {internal column * W * 4, bits of val}
    const int4 *head;        // unit headers {first entry, first row, slots << 16, log2(parts) | rows << 8 | wide << 16}
    int32_t unit_base[2], n_units[2], n_class[2];
    const float *xs;         // gathered operand, slab layout
    float *ys;               // result, slab layout (last = 0)
    int64_t slab_off[2][4];  // float offset of (class, slab)
    int32_t last;            // 1: out[orig[row]] = (sum_i prev[i] + acc) / denom, row-major [N, NS W]; 0: ys = acc (+ prev[0] if n_prev)
    int32_t n_prev;
    const float *prev[RBG_MAX_FUSED_LAYERS + 1];  // slab layout
    float denom;
    float *out;
    const int32_t *orig;     // original node id of (class, internal row)
    // row-major operands in the REFERENCE's numbering (class 0 = the user table [n_class[0], NS W], class 1 = the item table):
    const v4i *ent0;         // x_rm: the same entries with the column offset = original class-local row * 2 W * 4
    const float *rm[2];      // x_rm: the gathered operand
    const float *prm[2];     // prev0_rm: prev[0] (E0's two tables, the incoming gradient, or Y itself for Y += A X), read through orig[]
    int32_t x_rm;            // 1: the gathered operand is rm[] (entries ent0) — E0 / the incoming gradient is never converted
    int32_t prev0_rm;
    int32_t prev_rm_all;     // 1: prev[1..] are row-major [N, NS W] arrays in the reference's numbering as well
    float *out2;             // last: also store the layer itself (acc), row-major (RBG_FWD_KEEP_LAST_LAYER)
    // factored chain (val_ij = r_i r_j, the symmetric normalisation): the slabs between the layers hold z = r (.) y, a launch that
    // gathers z reads COLUMN OFFSETS ONLY (entc: 4 bytes per entry instead of 8) and scales its row sums by r_i
    const int32_t *entc;     // compact: the offsets column of ent
    const int32_t *entc16;   // compact, or NULL: the same as PAIRS of 16-bit slab-row numbers (r05: half the entry bytes again)
    int32_t c16_shift;       // log2(W * 4): row number -> byte offset
    int32_t c16_ok[2];       // rows of class c gather class 1 - c: 16-bit numbers serve them when THAT class has < 65 535 rows
    const float *rs, *irs;   // r_i and 1 / r_i (0 for an empty row), the plan's numbering
    int32_t compact;         // 1: gather through entc (the operand is a scaled slab), acc *= r_i
    int32_t store_scaled;    // 1: ys = r_i * (...): the next launch is compact
    int32_t prev_scaled;     // 1 (last): prev[1..] are scaled slabs: their sum is multiplied by 1 / r_i
    int32_t nt;              // option "sell_nt"
    int32_t rm_ld;           // x_rm: floats between the rows of rm[] (NS W when contiguous; a column block of a wider buffer otherwise)
    int32_t rm_shift;        // x_rm: log2(rm_ld / (2 W)) (-1: rm_ld = W): ent0's offsets are rows of 2 W floats
    int32_t rm_rows[2];      // x_rm: rows of rm[c] (n_class[c]; a rectangular plan: both = the one table's rows)
    const float *noise;      // last (row-major out): out = y + sign(y) * noise / max(|noise row|, 1e-12) * eps   (simgcl.py:30-33)
    float eps;
    float *wide_part;        // [wide units][128]: partial sums of the units of wide rows (the plan's scratch)
    uint32_t *wide_ctr;      // [wide units][4]: arrivals per (row, slab)
};

// What the waves of XCD x need to know about their (class, slab) role, worked out on the host (sell.hip sell_launch): ONE
// 64-byte scalar load indexed by blockIdx alone (r06).  Read out of the parameter block the same values took a chain of
// dependent scalar round trips — x_rm -> table pointer -> slab offset -> row count -> stride -> unit count -> header — because
// every choice between the row-major and the slab operand became a branch around a load.
struct alignas(64) SellRoleK {
    const float *xtab;     // the table this role gathers, at its column piece (slab s of class 1 - cls, or rm[1 - cls] + s W)
    const int4 *heads;     // unit headers of the role's row class
    const void *ents;      // the entries this launch reads for the class: ent / ent0 (valued), entc or entc16 (compact)
    int64_t ybase;         // float offset of the role's result slab
    uint32_t tab_bytes;    // the table's extent: gathers past it return zeros
    int32_t n_units;
    int32_t cbase;         // first row of the class in the plan's numbering (0 / n_class[0])
    int32_t c16;           // compact: the entries are 16-bit slab-row numbers
    int32_t wbase;         // first scratch slot of the class's wide units (0 / n_wide_units[0])
    int32_t pad[3];
};
static_assert(sizeof(SellRoleK) == 64, "one s_load_dwordx16 per role");
struct SellLaunch {
    SellParams p;
    SellRoleK role[8];
};

// The parameter block is read where the launch put it — the kernel-argument segment (constant address space) — through this
// reference type: a by-value copy handed to an inlined function by reference stayed in scratch memory (432 bytes per lane).
typedef const __attribute__((address_space(4))) SellParams SellParamsK;
typedef const __attribute__((address_space(4))) SellLaunch SellLaunchK;
__device__ __forceinline__ SellLaunchK &sell_kernarg() {
    return *(SellLaunchK *)__builtin_amdgcn_kernarg_segment_ptr();
}
// the role record in registers (wave-uniform: scalar)
struct SellRole {
    const int4 *heads;
    const void *ents;
    int64_t ybase;
    int32_t n_units, cbase, c16, wbase;
};

// One layer of a chain as the host describes it (sell.hip's chain builders)
struct SellChainLayer {
    const float *xs;
    float *ys;
    int32_t x_rm, compact, store_scaled, last, n_prev, prev0_rm, prev_scaled, pad;
};

// What changes from layer to layer of a chain (wave-uniform: scalar registers); the one-launch-per-layer kernel fills it from
// the parameter block (a persistent K-layer kernel would fill it from a per-layer table).
struct SellLayer {
    const float *xs;
    float *ys;
    int32_t x_rm, store_scaled, last, n_prev, prev0_rm, prev_scaled;
};
__device__ __forceinline__ SellLayer sell_layer_of(SellParamsK &p) {
    return SellLayer{p.xs, p.ys, p.x_rm, p.store_scaled, p.last, p.n_prev, p.prev0_rm, p.prev_scaled};
}


// Per-wave clock (devtools/microbench/sell_trace.hip builds sell.hip with RBG_SELL_TRACE; the product does not): 16 words per
// wave — [0] s_memrealtime at entry, [1] at exit (100 MHz, chip-wide), [2] units walked, [3] slots gathered, [4..9] s_memtime
// cycles per phase: 0 entry -> the first unit's header is there and its first entries are requested, 1 -> a unit's first batch
// of gathers is issued (its entries have arrived), 2 -> the unit's last batch is consumed, 3 -> reduction + epilogue, the store
// is issued, 4 hand-over to the next unit; [10] XCC id, [11] HW_ID; four regions of 32 768 waves (by launch kind).  In the product
// every method is empty.
#ifdef RBG_SELL_TRACE
__device__ unsigned long long *g_sell_trace = nullptr;
// what-if switches of the diagnostic build (results are wrong, timings tell what a part costs): 1 = every gather falls into the
// first 16 KB of its table (an L1-resident table), 2 = no epilogue (no addend loads, no stores), 4 = entries are not loaded
// (synthesised offsets), 8 = gathers out of range (the address units work, no cache access)
__device__ int g_sell_debug = 0;
#define RBG_SELL_DBG(bit) ((g_sell_debug & (bit)) != 0)
struct SellClock {
    unsigned long long rt0, last, acc[6], units, slots;
    __device__ __forceinline__ void start() {
        rt0 = __builtin_amdgcn_s_memrealtime();
        last = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < 6; ++k) acc[k] = 0;
        units = slots = 0;
    }
    __device__ __forceinline__ void lap(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[k] += now - last;
        last = now;
    }
    __device__ __forceinline__ void count(int n) { ++units, slots += n; }
    __device__ __forceinline__ void dump(int region) {  // region: 0 valued launch, 2 compact, + 1 for a chain's last launch
        if (!g_sell_trace || (threadIdx.x & 63)) return;
        const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wv >= 32768) return;
        unsigned long long *tr = g_sell_trace + ((size_t)region * 32768 + wv) * 16;
        tr[0] = rt0, tr[1] = __builtin_amdgcn_s_memrealtime(), tr[2] = units, tr[3] = slots;
        for (int k = 0; k < 6; ++k) tr[4 + k] = acc[k];
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        tr[10] = xcc & 0xf, tr[11] = hw;
    }
};
#else
#define RBG_SELL_DBG(bit) false
struct SellClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void lap(int) {}
    __device__ __forceinline__ void count(int) {}
    __device__ __forceinline__ void dump(int) {}
};
#endif

template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // lane K of every quad, in all its lanes
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);
}
// entry J (0..7) of the 8 a quad holds: lane J / 2, components (J & 1) * 2 + {0, 1}
template <int J>
__device__ __forceinline__ int ent_col(const v4i &w) { return quad_bcast<J / 2>((J & 1) ? w.z : w.x); }
template <int J>
__device__ __forceinline__ float ent_val(const v4i &w) { return __int_as_float(quad_bcast<J / 2>((J & 1) ? w.w : w.y)); }
// COMPACT: an entry is its column offset alone (the operand is pre-scaled by the column's factor)
template <int J>
__device__ __forceinline__ int ent_col(const v2i &w) { return quad_bcast<J / 2>((J & 1) ? w.y : w.x); }

template <int J, int N>
struct SellFor {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, J>{});
        SellFor<J + 1, N>::run(f);
    }
};
template <int N>
struct SellFor<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

// epilogue accesses with an optional non-temporal hint (option "sell_nt": 1 = stores, 2 = the mean's addend loads)
__device__ __forceinline__ void st4(float *p, const float4 v, const bool nt) {
    v4f w = {v.x, v.y, v.z, v.w};
    if (nt) __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p));
    else *reinterpret_cast<v4f *>(p) = w;
}
__device__ __forceinline__ float4 ld4(const float *p, const bool nt) {
    const v4f w = nt ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)) : *reinterpret_cast<const v4f *>(p);
    return make_float4(w.x, w.y, w.z, w.w);
}

struct SellAcc {
    v2f lo, hi;
};
__device__ __forceinline__ void fma_row(SellAcc &a, float v, v4f x) {
    const v2f vv = {v, v};
    a.lo = __builtin_elementwise_fma(vv, __builtin_shufflevector(x, x, 0, 1), a.lo);
    a.hi = __builtin_elementwise_fma(vv, __builtin_shufflevector(x, x, 2, 3), a.hi);
}

// Eight gathered rows in registers: named members (an array indexed through lambdas went to scratch: hipcc kept it in memory)
struct SellRows {
    v4f r0, r1, r2, r3, r4, r5, r6, r7;
    template <int J>
    __device__ __forceinline__ v4f &at() {
        if constexpr (J == 0) return r0;
        else if constexpr (J == 1) return r1;
        else if constexpr (J == 2) return r2;
        else if constexpr (J == 3) return r3;
        else if constexpr (J == 4) return r4;
        else if constexpr (J == 5) return r5;
        else if constexpr (J == 6) return r6;
        else return r7;
    }
};

// N gathers of one batch: the index broadcast is folded into the address add (v_add_u32_dpp), padding slots read zeros past the table
template <int J, int N>
struct SellIssue {
    template <class WT>
    static __device__ __forceinline__ void run(SellRows &x, const WT &w, const __amdgpu_buffer_rsrc_t rs, const int lane_off) {
        x.template at<J>() = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, ent_col<J>(w) + lane_off, 0, 0));
        SellIssue<J + 1, N>::run(x, w, rs, lane_off);
    }
};
template <int N>
struct SellIssue<N, N> {
    template <class WT>
    static __device__ __forceinline__ void run(SellRows &, const WT &, const __amdgpu_buffer_rsrc_t, const int) {}
};
template <int J, int N>
struct SellConsume {
    template <class WT>
    static __device__ __forceinline__ void run(SellAcc &acc, SellRows &x, const WT &w) {
        const v4f v = x.template at<J>();
        if constexpr (std::is_same<WT, v2i>::value) {
            acc.lo += __builtin_shufflevector(v, v, 0, 1);
            acc.hi += __builtin_shufflevector(v, v, 2, 3);
        } else {
            fma_row(acc, ent_val<J>(w), v);
        }
        SellConsume<J + 1, N>::run(acc, x, w);
    }
};
template <int N>
struct SellConsume<N, N> {
    template <class WT>
    static __device__ __forceinline__ void run(SellAcc &, SellRows &, const WT &) {}
};
// n in {2, 4, 6, 8} slots: pairs issued / consumed behind nested tests.  (Four separate bodies selected by n — the r03 form —
// let hipcc merge their common tails into one block that picks the register by a pointer phi: the rows went to scratch.)
template <class WT>
__device__ __forceinline__ void sell_issue_n(const int n, SellRows &x, const WT &w, const __amdgpu_buffer_rsrc_t rs, const int lane_off) {
    SellIssue<0, 2>::run(x, w, rs, lane_off);
    if (n > 2) {
        SellIssue<2, 4>::run(x, w, rs, lane_off);
        if (n > 4) {
            SellIssue<4, 6>::run(x, w, rs, lane_off);
            if (n > 6) SellIssue<6, 8>::run(x, w, rs, lane_off);
        }
    }
}
template <class WT>
__device__ __forceinline__ void sell_consume_n(const int n, SellAcc &acc, SellRows &x, const WT &w) {
    SellConsume<0, 2>::run(acc, x, w);
    if (n > 2) {
        SellConsume<2, 4>::run(acc, x, w);
        if (n > 4) {
            SellConsume<4, 6>::run(acc, x, w);
            if (n > 6) SellConsume<6, 8>::run(acc, x, w);
        }
    }
}

// offsets of a row-major table whose rows are 2 W << sh floats apart (sh = -1: W floats) from ent0's (rows of 2 W floats);
// padding stays out of range (the shifted padding offset of a 4 W row wrapped into the table for lanes 2.. of a
// lane-group — harmless only while row 0 is finite)
template <class WT>
__device__ __forceinline__ void sell_widen(WT &e, const int sh) {
    if constexpr (std::is_same<WT, v4i>::value) {
        if (sh > 0) {
            if (e.x != kSellPast) e.x <<= sh;
            if (e.z != kSellPast) e.z <<= sh;
        } else if (sh < 0) {  // half the stride; kSellPast / 2 is still past every table that has row-major entries
            e.x >>= 1;
            e.z >>= 1;
        }
    }
}

// The gathers of one unit: batches of 8 slots per lane-group (the last one of nc % 8, even); the pair of entries a lane holds
// for batch k sits at base + (LGW k) / 2 + lg (sb / 2) + q4.  One batch of gathers in flight per wave.
// (r04, measured and removed: the unit's first batch from a fixed-stride block requested together with the header instead of
// after it — 93.4 vs 93.5 us per propagation at the Gowalla shape, 126.0 vs 126.2 at Yelp2018: the header -> entries round trip
// is not on the critical path; profiles/r04_launch_forms.jsonl)
// (c16: the unit's entries as pairs of 16-bit slab-row numbers, or NULL — uniform over the launch)
template <int W, int NS, class WT>
__device__ __forceinline__ void sell_gather1(SellAcc &acc, const WT *base, const int32_t *c16, const int c16_shift, const int nc, const int lg,
                                             const int q4, const __amdgpu_buffer_rsrc_t rs, const int lane_off, const int sh, SellClock &clk) {
    constexpr int LGW = 64 / (W / 4);
    if (nc <= 0) return;
    int sb = min(8, nc);
    auto entries = [&](const int idx) __attribute__((always_inline)) -> WT {
        if constexpr (std::is_same<WT, v2i>::value) {
            if (c16) {  // 0xffff (padding) << shift is past every table with < 65 536 rows
                const unsigned t = (unsigned)c16[idx];
                return WT{(int)((t & 0xffffu) << c16_shift), (int)((t >> 16) << c16_shift)};
            }
        }
        return base[idx];
    };
    // (plain loads: with the non-temporal hint on the entry stream the layer measured 38.6 us instead of 31)
    WT w = {};
    if (2 * q4 < sb) w = entries(lg * (sb >> 1) + q4);
    sell_widen(w, sh);
    for (int k = 0; k < nc; k += 8) {
        const int sbn = min(8, nc - k - 8);  // slots of the next batch (<= 0: none)
        WT wn = {};
        SellRows x;
        sell_issue_n(sb, x, w, rs, lane_off);
        if (sbn > 0 && 2 * q4 < sbn) wn = entries(((LGW * (k + 8)) >> 1) + lg * (sbn >> 1) + q4);
        if (k == 0) clk.lap(1);
        sell_consume_n(sb, acc, x, w);
        sell_widen(wn, sh);
        w = wn;
        sb = sbn;
    }
}

__device__ __forceinline__ float sell_sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }  // torch.sign

// One unit: the gathers of its lane-groups, the reduction of split rows, the epilogue.  h = the unit's header (unit t of its
// class); rs = the buffer resource of the gathered table (class 1 - cls, slab s).
template <int W, int NS, bool COMPACT>
__device__ __forceinline__ void sell_unit(SellParamsK &p, const SellLayer &L, const SellRole &R, const int cls, const int s, const unsigned t,
                                          const int4 h, const __amdgpu_buffer_rsrc_t rs, SellClock &clk) {
    constexpr int G = W / 4;      // lanes per lane-group
    constexpr int LGW = 64 / G;   // lane-groups per wave = pieces per unit
    constexpr int D = NS * W;     // row width: NS slabs
    using WT = std::conditional_t<COMPACT, v2i, v4i>;
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G, q4 = lane & 3;
    const int lane_off = sl * 16;
    const int row0 = h.y, nc = (int)((unsigned)h.z >> 16), lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
    const bool wide = (h.w >> 16) & 1;  // one of the U units of a wide row (wave-uniform)
    // the epilogue's row-indexed scalars are requested before the gathers (they would otherwise be two dependent round trips
    // at the end of the wave: orig[] -> the row-major addend)
    const int r = lg >> lp;
    const int row = row0 + (r < nrows ? r : 0);
    const int cbase = R.cbase;
    const int64_t ybase = R.ybase;
    int node = 0;
    float r_i = 1.f;
    if constexpr (COMPACT) {  // (the valued instantiation has no register to spare: it asks at the end)
        if (L.last || L.prev0_rm) node = p.orig[cbase + row];
        if (COMPACT || L.store_scaled) r_i = p.rs[cbase + row];
    }
    SellAcc acc = {{0.f, 0.f}, {0.f, 0.f}};
    // (ent0's offsets are rows of 2 W floats: a 4 W or a W row-major operand rescales them)
    const WT *ebase;
    const int32_t *e16 = nullptr;
    if constexpr (COMPACT) {
        ebase = reinterpret_cast<const v2i *>(R.ents) + (h.x >> 1);
        if (R.c16) e16 = reinterpret_cast<const int32_t *>(R.ents) + (h.x >> 1);
    } else {
        ebase = reinterpret_cast<const v4i *>(R.ents) + (h.x >> 1);
    }
    const int sh = (!COMPACT && L.x_rm) ? p.rm_shift : 0;
    sell_gather1<W, NS, WT>(acc, ebase, e16, p.c16_shift, nc, lg, q4, rs, lane_off, sh, clk);
    // "these values are in their registers HERE": the row scalars requested above are waited for behind the gathers — hipcc
    // otherwise sign-extends `node` right behind its load and the wave sits out a whole vector round trip (s_waitcnt vmcnt(0))
    // before it has requested its first entries (r06; the per-wave clock's 5 200-cycle prologue)
    if constexpr (COMPACT) asm volatile("" : "+v"(node), "+v"(r_i));
    clk.count(nc * LGW);
    clk.lap(2);
    // the pieces of a split row sit in adjacent lane-groups: butterfly, fixed order
    const int parts = 1 << lp;
    if (lp > 0) {
#pragma unroll
        for (int off = 1; off < LGW; off <<= 1) {
            const float a0 = __shfl_xor(acc.lo.x, off * G), a1 = __shfl_xor(acc.lo.y, off * G);
            const float a2 = __shfl_xor(acc.hi.x, off * G), a3 = __shfl_xor(acc.hi.y, off * G);
            if (off < parts) { acc.lo.x += a0; acc.lo.y += a1; acc.hi.x += a2; acc.hi.y += a3; }
        }
    }
    // A wide row = U units (waves anywhere on the role's XCDs).  Every unit publishes its sum WRITE-THROUGH (agent-scope relaxed
    // stores: sc1, no L2-flushing release fence), drains, and bumps the row's arrival counter; the LAST to arrive re-reads the U
    // partials with agent-scope loads and adds them in unit order, so the result does not depend on the arrival order (the
    // idiom of spmm.hip's split rows; MI355X guide 6 G16, form R1).  It leaves the counter at zero for the next launch.
    bool emit = true;
    if (wide) {
        const int U = (int)((unsigned)h.w >> 17), j = h.z & 0xffff;
        const int64_t slot0 = (int64_t)R.wbase + (int64_t)t - j;  // the row's first unit
        if (lg == 0) {
            float *dst = p.wide_part + (slot0 + j) * 128 + s * W + sl * 4;
            __hip_atomic_store(dst + 0, acc.lo.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, acc.lo.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2, acc.hi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 3, acc.hi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t *arrivals = p.wide_ctr + slot0 * 4 + s;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        emit = old == (unsigned)(U - 1);
        if (emit) {
            if (lane == 0) __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning
            if (lg == 0) {
                float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int i = 0; i < U; ++i) {
                    const float *src = p.wide_part + (slot0 + i) * 128 + s * W + sl * 4;
                    tsum.x += __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tsum.y += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tsum.z += __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tsum.w += __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                acc.lo.x = tsum.x; acc.lo.y = tsum.y; acc.hi.x = tsum.z; acc.hi.y = tsum.w;
            }
        }
    }
    const bool owner = (lg & (parts - 1)) == 0 && r < nrows && emit;
    // the noise row's norm spans all NS slabs: every lane-group reads the whole row (all lanes take part in the shuffles)
    float nsc = 0.f;
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.last && p.noise) {
        if constexpr (!COMPACT) node = p.orig[cbase + row];
        const float *nrow = p.noise + (int64_t)node * D + sl * 4;
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(nrow + q * W);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (q == s) nz = v;
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) ss += __shfl_xor(ss, off);
        nsc = p.eps / fmaxf(sqrtf(ss), 1e-12f);
    }
    if (owner) {
    const int64_t o = ybase + (int64_t)row * W + sl * 4;
    if constexpr (!COMPACT) {
        if (L.last || L.prev0_rm) node = p.orig[cbase + row];
    }
    const int64_t orm = (int64_t)node * D + s * W + sl * 4;  // row-major [N, D], the reference's numbering
    const float *prev0 = L.prev0_rm ? p.prm[cls] + (orm - (int64_t)cbase * D) : p.prev[0] + o;
    float4 y = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
    if constexpr (!COMPACT) {
        if (L.store_scaled) r_i = p.rs[cbase + row];
    }
    if (COMPACT) { y.x *= r_i; y.y *= r_i; y.z *= r_i; y.w *= r_i; }  // y = r_i sum_j z_j
    if (L.last) {
        const bool ntl = (p.nt & 2) != 0, nts = (p.nt & 1) != 0;
        // the addends are requested AG at a time (4; 2 in the valued kernel, which has no register to spare), not one per wait (r05: the per-wave clock showed the mean's epilogue at 4 500
        // cycles per wave against 700 for a layer without addends), and summed in the old order: sum = prev0 (+ prev[1] + ...), or
        // prev0 + (z_1 + z_2 + ...) / r_i when the layers in between are stored scaled
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), zs = make_float4(0.f, 0.f, 0.f, 0.f);
        const int np = L.n_prev;
        const int64_t oprev = (L.prev_scaled || !p.prev_rm_all) ? o : orm;
        constexpr int AG = COMPACT ? 4 : 2;
        for (int i0 = 0; i0 < np; i0 += AG) {
            float4 a[AG];
#pragma unroll
            for (int j = 0; j < AG; ++j) {
                a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + j < np) a[j] = ld4((i0 + j) == 0 ? prev0 : p.prev[i0 + j] + oprev, ntl);
            }
#pragma unroll
            for (int j = 0; j < AG; ++j) {
                if (i0 + j >= np) continue;
                if (i0 + j == 0) sum = a[j];
                else if (L.prev_scaled) { zs.x += a[j].x; zs.y += a[j].y; zs.z += a[j].z; zs.w += a[j].w; }
                else { sum.x += a[j].x; sum.y += a[j].y; sum.z += a[j].z; sum.w += a[j].w; }
            }
        }
        if (L.prev_scaled) {  // E_k = z_k / r_i
            const float ir = p.irs[cbase + row];
            sum.x += zs.x * ir; sum.y += zs.y * ir; sum.z += zs.z * ir; sum.w += zs.w * ir;
        }
        if (p.out2) st4(p.out2 + orm, y, nts);
        sum.x = (sum.x + y.x) / p.denom; sum.y = (sum.y + y.y) / p.denom;
        sum.z = (sum.z + y.z) / p.denom; sum.w = (sum.w + y.w) / p.denom;
        if (p.noise) {  // (a plain layer: n_prev = 0, denom = 1: sum = y)
            sum.x = fmaf(sell_sgn(sum.x) * nz.x, nsc, sum.x); sum.y = fmaf(sell_sgn(sum.y) * nz.y, nsc, sum.y);
            sum.z = fmaf(sell_sgn(sum.z) * nz.z, nsc, sum.z); sum.w = fmaf(sell_sgn(sum.w) * nz.w, nsc, sum.w);
        }
        st4(p.out + orm, sum, nts);
    } else {
        if (L.n_prev) {  // a step of the backward chain: y = g + A x
            const float4 q = *reinterpret_cast<const float4 *>(prev0);
            y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
        }
        if (L.store_scaled) { y.x *= r_i; y.y *= r_i; y.z *= r_i; y.w *= r_i; }
        st4(L.ys + o, y, (p.nt & 1) != 0);
    }
    }
}

}  // namespace rbg
