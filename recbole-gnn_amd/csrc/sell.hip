// sell.hip — the propagation of LightGCN.forward / SGL.forward (lightgcn.py:70-81, sgl.py:128-145) over COLUMN SLABS with a
// sliced-ELL graph (r03).  Used by rbg_lightgcn_forward_f32 at d = 64 when a plan is attached
// (rbg_graph_attach_sell; planner recbole-gnn_amd/sell.py) and the caller does not read the intermediate layers.
//
// Why: the binned kernel (spmm.hip) is bound by L2 misses — an XCD's 4 MB L2 cannot hold the table its rows gather from, and
// a layer moves 230 MB over the fabric for 53 MB of algorithmic bytes at the Gowalla shape.  Here
//   * the dense operand is kept as two column slabs [2][row][32] between the layers, rows renumbered per class; XCD x of a
//     row class gathers ONE slab of the other class's table (5.2 MB instead of 10.5 MB of item rows; 128-byte gathers = whole
//     L2 lines), so the fabric traffic falls to 134 MB per layer; the CSR is read once per slab;
//   * the graph is SELL-C-sigma over lane-groups: a unit = the 8 lane-groups (8 lanes x float4) of one wave on consecutive
//     rows of similar length, its entries stored unit-major and padded to the unit's longest piece, so one wave-wide 16-byte
//     load fetches a batch of 8 slots per lane-group and nothing is masked; the index broadcast is a DPP quad_perm (a
//     lane-group is two quads), the gather a buffer load whose padded slots read zeros past the table: 4 VALU + 1 VMEM per
//     gathered row against 13 + 1 in the binned kernel;
//   * rows longer than 128 entries are cut into up to 8 pieces in adjacent lane-groups (butterfly), rows longer than 1 024 into
//     32 pieces over the four waves of a workgroup (LDS): without the latter the longest row is one wave's serial chain of 48
//     gather batches and sets the duration of the whole launch (38.3 -> 29.7 us per layer);
//   * heaviest units first, one wave per unit: the hardware dispatcher balances the load.
// E0 is converted to slabs once per propagation; the last layer's epilogue adds the layer mean and writes it row-major in
// the reference's numbering.  Summation order is fixed by the plan: results are bit-stable run to run, no float atomics.
// Measured (profiles/r03_slab_wide_rows_*): layer 40.3 -> 29.7 us, propagation 131 -> 101 us at the Gowalla shape; 59 -> 44 us
// per layer at the Yelp2018 shape, 134 -> 100 us at Amazon-Book, 1 188 -> 1 065 us at 1.3 M nodes.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <new>
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
    int32_t last;            // 1: out[orig[row]] = (sum_i prev[i] + acc) / denom, row-major [N, 2 W]; 0: ys = acc (+ prev[0] if n_prev)
    int32_t n_prev;
    const float *prev[RBG_MAX_FUSED_LAYERS + 1];  // slab layout
    float denom;
    float *out;
    const int32_t *orig;     // original node id of (class, internal row)
    // row-major operands in the REFERENCE's numbering (class 0 = the user table [n_class[0], 2 W], class 1 = the item table):
    const v4i *ent0;         // x_rm: the same entries with the column offset = original class-local row * 2 W * 4
    const float *rm[2];      // x_rm: the gathered operand
    const float *prm[2];     // prev0_rm: prev[0] (E0's two tables, the incoming gradient, or Y itself for Y += A X), read through orig[]
    int32_t x_rm;            // 1: the gathered operand is rm[] (entries ent0) — E0 / the incoming gradient is never converted
    int32_t prev0_rm;
    int32_t prev_rm_all;     // 1: prev[1..] are row-major [N, 2 W] arrays in the reference's numbering as well
    float *out2;             // last: also store the layer itself (acc), row-major (RBG_FWD_KEEP_LAST_LAYER)
    // factored chain (val_ij = r_i r_j, the symmetric normalisation): the slabs between the layers hold z = r (.) y, a launch that
    // gathers z reads COLUMN OFFSETS ONLY (entc: 4 bytes per entry instead of 8) and scales its row sums by r_i
    const int32_t *entc;     // compact: the offsets column of ent
    const float *rs, *irs;   // r_i and 1 / r_i (0 for an empty row), the plan's numbering
    int32_t compact;         // 1: gather through entc (the operand is a scaled slab), acc *= r_i
    int32_t store_scaled;    // 1: ys = r_i * (...): the next launch is compact
    int32_t prev_scaled;     // 1 (last): prev[1..] are scaled slabs: their sum is multiplied by 1 / r_i
    int32_t nt;              // option "sell_nt"
};

template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // lane K of every quad, in all its lanes
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);
}
// entry J (0..7) of the 8 a quad holds: lane J / 2, components (J & 1) * 2 + {0, 1}
template <int J>
__device__ __forceinline__ int ent_col(const v4i &w) { return quad_bcast<J / 2>((J & 1) ? w.z : w.x); }
template <int J>
__device__ __forceinline__ float ent_val(const v4i &w) { return __int_as_float(quad_bcast<J / 2>((J & 1) ? w.w : w.y)); }

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

// The gathers of one unit: batches of 8 slots per lane-group (the last one of nc % 8, even); the pair of entries a lane holds
// for batch k sits at base + (LGW k) / 2 + lg (sb / 2) + q4.  COMPACT: an entry is its column offset alone (the operand is
// pre-scaled by the column's factor) — 8 bytes per lane and batch, an add instead of an FMA.
template <int J>
__device__ __forceinline__ int ent_col(const v2i &w) { return quad_bcast<J / 2>((J & 1) ? w.y : w.x); }

template <int W, bool COMPACT, int SHIFT>
__device__ __forceinline__ void sell_gather(SellAcc &acc, const std::conditional_t<COMPACT, v2i, v4i> *base, const int nc, const int lg,
                                            const int q4, const __amdgpu_buffer_rsrc_t rs, const int lane_off, const bool shift) {
    constexpr int LGW = 64 / (W / 4);
    using WT = std::conditional_t<COMPACT, v2i, v4i>;
    if (nc <= 0) return;
    int sb = min(8, nc);
    WT w = {};
    // (plain loads: with the non-temporal hint on the entry stream the layer measured 38.6 us instead of 31)
    if (2 * q4 < sb) w = base[lg * (sb >> 1) + q4];
    auto widen = [&](WT &e) __attribute__((always_inline)) {  // offsets of a row-major table twice as wide (padding stays out of range)
        if constexpr (SHIFT && !COMPACT) {
            if (shift) { e.x = (int)((unsigned)e.x << SHIFT); e.z = (int)((unsigned)e.z << SHIFT); }
        }
    };
    widen(w);
    for (int k = 0; k < nc; k += 8) {
        const int sbn = min(8, nc - k - 8);  // slots of the next batch (<= 0: none)
        WT wn = {};
        auto batch = [&](auto nc_) __attribute__((always_inline)) {
            constexpr int n = decltype(nc_)::value;
            v4f xv[n];
            SellFor<0, n>::run([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                xv[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, ent_col<j>(w) + lane_off, 0, 0));
            });
            if (sbn > 0 && 2 * q4 < sbn) wn = base[((LGW * (k + 8)) >> 1) + lg * (sbn >> 1) + q4];
            SellFor<0, n>::run([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (COMPACT) {
                    acc.lo += __builtin_shufflevector(xv[j], xv[j], 0, 1);
                    acc.hi += __builtin_shufflevector(xv[j], xv[j], 2, 3);
                } else {
                    fma_row(acc, ent_val<j>(w), xv[j]);
                }
            });
        };
        if (sb == 8) batch(std::integral_constant<int, 8>{});
        else if (sb == 6) batch(std::integral_constant<int, 6>{});
        else if (sb == 4) batch(std::integral_constant<int, 4>{});
        else batch(std::integral_constant<int, 2>{});
        widen(wn);
        w = wn;
        sb = sbn;
    }
}

// W = slab width (32 at d = 64).  One wave per unit; workgroup b runs on XCD b & 7: XCDs 0-3 take user rows, 4-7 item rows,
// XCD pair (x & 1) owns slab x & 1 of its class.
template <int W, int NS, bool COMPACT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8))) void sell_spmm_kernel(const SellParams p) {
    constexpr int G = W / 4;      // lanes per lane-group
    constexpr int LGW = 64 / G;   // lane-groups per wave = pieces per unit
    constexpr int D = NS * W;     // row width: NS slabs (2: two XCDs share a (class, slab) role; 4: one XCD per role)
    __shared__ float s_wide[4][W];
    const int x = blockIdx.x & 7, cls = x >> 2, s = x & (NS - 1), xi = NS == 2 ? (x & 3) >> 1 : 0;
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G, q4 = lane & 3, wave = threadIdx.x >> 6;
    // (kernel arguments first, all of them, then the unit test: an early exit in front of them serialises four dependent
    // scalar-load round trips per wave — n_units, pointers, header, offsets)
    // (a row-major table is read as its column half s: 128-byte (W = 32) pieces at a 2 W stride — whole L2 lines, the same
    // footprint per XCD as a slab)
    const float *xtab = p.x_rm ? p.rm[1 - cls] + s * W : p.xs + p.slab_off[1 - cls][s];
    const int n_tab = p.n_class[1 - cls];
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xtab), 0, p.x_rm ? (unsigned)n_tab * (D * 4) - s * W * 4 : n_tab * W * 4, 0x00020000);
    const int lane_off = sl * 16;
    const unsigned nun = (unsigned)p.n_units[cls];
    const int4 *heads = p.head + p.unit_base[cls];
    const v4i *ents = p.x_rm ? p.ent0 : p.ent;
    const int64_t ybase = p.slab_off[cls][s];
    constexpr int XR = NS == 2 ? 2 : 1;             // XCDs per role
    const unsigned n_w = (gridDim.x >> 3) * 4 * XR;  // waves of this role: the grid covers the units, so the loop body runs at most once
    for (unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)((((blockIdx.x >> 3) * XR + xi) * 4 + wave))); t < nun; t += n_w) {
    const int4 h = heads[t];
    const int row0 = h.y, nc = h.z >> 16, lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
    const bool wide = (h.w >> 16) & 1;  // uniform over the workgroup: the plan aligns a wide row to four units
    // the epilogue's row-indexed scalars are requested before the gathers (they would otherwise be two dependent round trips
    // at the end of the wave: orig[] -> the row-major addend)
    const int r = lg >> lp;
    const int row = row0 + (r < nrows ? r : 0);
    const int cbase = cls ? p.n_class[0] : 0;
    int node = 0;
    float r_i = 1.f;
    if constexpr (COMPACT) {  // (the valued instantiation has no register to spare: it asks at the end)
        if (p.last || p.prev0_rm) node = p.orig[cbase + row];
        r_i = p.rs[cbase + row];
    }
    SellAcc acc = {{0.f, 0.f}, {0.f, 0.f}};
    // (ent0's offsets are rows of 2 W floats: a 4 W row-major operand doubles them)
    if constexpr (COMPACT) sell_gather<W, true, 0>(acc, reinterpret_cast<const v2i *>(p.entc) + (h.x >> 1), nc, lg, q4, rs, lane_off, false);
    else sell_gather<W, false, (NS == 4 ? 1 : 0)>(acc, ents + (h.x >> 1), nc, lg, q4, rs, lane_off, p.x_rm != 0);
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
    if (wide) {  // 4 waves x LGW pieces of ONE row: per-wave partial sums through LDS, added in wave order
        if (lg == 0) *reinterpret_cast<float4 *>(&s_wide[wave][sl * 4]) = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
        __syncthreads();
        if (wave == 0 && lg == 0) {
            float4 tsum = *reinterpret_cast<const float4 *>(&s_wide[0][sl * 4]);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float4 o4 = *reinterpret_cast<const float4 *>(&s_wide[q][sl * 4]);
                tsum.x += o4.x; tsum.y += o4.y; tsum.z += o4.z; tsum.w += o4.w;
            }
            acc.lo.x = tsum.x; acc.lo.y = tsum.y; acc.hi.x = tsum.z; acc.hi.y = tsum.w;
        }
        __syncthreads();  // (a wave that walks on to another wide unit must not overwrite s_wide under wave 0's reads)
    }
    if ((lg & (parts - 1)) == 0 && r < nrows && (!wide || wave == 0)) {
    const int64_t o = ybase + (int64_t)row * W + sl * 4;
    if constexpr (!COMPACT) {
        if (p.last || p.prev0_rm) node = p.orig[cbase + row];
    }
    const int64_t orm = (int64_t)node * D + s * W + sl * 4;  // row-major [N, D], the reference's numbering
    const float *prev0 = p.prev0_rm ? p.prm[cls] + (orm - (int64_t)cbase * D) : p.prev[0] + o;
    float4 y = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
    if constexpr (!COMPACT) {
        if (p.store_scaled) r_i = p.rs[cbase + row];
    }
    if (COMPACT) { y.x *= r_i; y.y *= r_i; y.z *= r_i; y.w *= r_i; }  // y = r_i sum_j z_j
    if (p.last) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ntl = (p.nt & 2) != 0, nts = (p.nt & 1) != 0;
        if (p.n_prev) sum = ld4(prev0, ntl);
        if (p.prev_scaled) {  // the layers in between are stored scaled: E_k = z_k / r_i
            float4 zs = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = 1; i < p.n_prev; ++i) {
                const float4 q = ld4(p.prev[i] + o, ntl);
                zs.x += q.x; zs.y += q.y; zs.z += q.z; zs.w += q.w;
            }
            const float ir = p.irs[cbase + row];
            sum.x += zs.x * ir; sum.y += zs.y * ir; sum.z += zs.z * ir; sum.w += zs.w * ir;
        } else {
            for (int i = 1; i < p.n_prev; ++i) {
                const float4 q = ld4(p.prev[i] + (p.prev_rm_all ? orm : o), ntl);
                sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
            }
        }
        if (p.out2) st4(p.out2 + orm, y, nts);
        sum.x = (sum.x + y.x) / p.denom; sum.y = (sum.y + y.y) / p.denom;
        sum.z = (sum.z + y.z) / p.denom; sum.w = (sum.w + y.w) / p.denom;
        st4(p.out + orm, sum, nts);
    } else {
        if (p.n_prev) {  // a step of the backward chain: y = g + A x
            const float4 q = *reinterpret_cast<const float4 *>(prev0);
            y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
        }
        if (p.store_scaled) { y.x *= r_i; y.y *= r_i; y.z *= r_i; y.w *= r_i; }
        st4(p.ys + o, y, (p.nt & 1) != 0);
    }
    }
    }
}

// the two embedding tables, row-major [n, NS W] in the reference's numbering -> slabs in the plan's numbering
template <int W, int NS>
__global__ __launch_bounds__(256) void sell_to_slab_kernel(const float *user_emb, const float *item_emb, int64_t n_users, float *dst,
                                                           const int32_t *orig, int n0, int n1, int64_t off0, int64_t off1) {
    constexpr int D = NS * W;
    const int g = (blockIdx.x * 256 + threadIdx.x) / (D / 4), c4 = ((blockIdx.x * 256 + threadIdx.x) % (D / 4)) * 4;
    if (g >= n0 + n1) return;
    const int cls = g >= n0, row = cls ? g - n0 : g;
    const int64_t node = orig[g];
    const float *src = node < n_users ? user_emb + node * D : item_emb + (node - n_users) * D;
    const int s = c4 / W;
    const int64_t so = (cls ? off1 + (int64_t)s * n1 * W : off0 + (int64_t)s * n0 * W) + (int64_t)row * W + (c4 - s * W);
    *reinterpret_cast<float4 *>(dst + so) = *reinterpret_cast<const float4 *>(src + c4);
}

// memory safety of a plan (its content is the planner's business: parity tests pin it): every index the kernel dereferences
__global__ void sell_check_units_kernel(const int4 *head, int n_units_total, int unit_base1, int n0, int n1, int64_t n_ent, int lgw,
                                        int *err) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_units_total) return;
    const int4 h = head[t];
    const int cls = t >= unit_base1, nc = h.z >> 16, lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff, wide = (h.w >> 16) & 1;
    const int n_c = cls ? n1 : n0;
    bool bad = h.x < 0 || (h.x & 1) || (h.z & 0xffff) != 0 || nc < 0 || (nc & 1) || (int64_t)h.x + (int64_t)lgw * nc > n_ent;
    bad = bad || lp < 0 || (1 << lp) > lgw || nrows < 0 || nrows > (lgw >> lp) || h.y < 0 || h.y + nrows > n_c;
    const int tl = t - (cls ? unit_base1 : 0);
    if (wide) {  // a wide row = units 4 j .. 4 j + 3 of its class, all flagged, one row
        const int4 h0 = head[t - (tl & 3)];
        bad = bad || !((h0.w >> 16) & 1) || h0.y != h.y || nrows != 1 || (1 << lp) != lgw;
    } else if (tl & 3) {
        bad = bad || ((head[t - (tl & 3)].w >> 16) & 1);
    }
    if (bad) atomicExch(err, 1 + t);
}
__global__ void sell_check_entries_kernel(const int2 *ent, int64_t n_ent, int64_t first_ent1, int n0, int n1, int W, int *err) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
        const int off = ent[e].x;
        const int64_t lim = (int64_t)(e >= first_ent1 ? n0 : n1) * W * 4;  // class 0 rows gather the class 1 table and vice versa
        if (off != kSellPast && (off < 0 || off >= lim || off % (W * 4) != 0)) atomicExch(err, -1);
    }
}
__global__ void sell_check_orig_kernel(const int32_t *orig, int n, int n_users, int n0, int *err) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int v = orig[g];
    if (v < 0 || v >= n || ((g < n0) != (v < n_users))) atomicExch(err, -2);
}

// ent0: the entries for a launch that gathers a ROW-MAJOR table in the reference's numbering (E0, the incoming gradient)
__global__ void sell_first_entries_kernel(const int2 *ent, int2 *ent0, int64_t n_ent, int64_t first_ent1, const int32_t *orig, int n0, int W) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
        int2 v = ent[e];
        if (v.x != kSellPast) {
            const int obase = e >= first_ent1 ? 0 : n0;  // class 0 rows gather the item table (nodes n0 ..), class 1 rows the user table
            v.x = (orig[obase + v.x / (W * 4)] - obase) * (2 * W * 4);
        }
        ent0[e] = v;
    }
}

__global__ void sell_compact_entries_kernel(const int2 *ent, int32_t *entc, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) entc[e] = ent[e].x;
}

// factor check: one thread per (unit, lane-group) walks its slots; every stored value must be r[row] * r[col] to 1e-6 relative
__global__ void sell_check_factors_kernel(const int2 *ent, const int4 *head, int n_units_total, int unit_base1, int n0, int W, int lgw,
                                          const float *r, int *err) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)(gid / lgw), lg = (int)(gid % lgw);
    if (t >= n_units_total) return;
    const int4 h = head[t];
    const int cls = t >= unit_base1, nc = h.z >> 16, lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
    const int rr = lg >> lp;
    if (rr >= nrows) return;
    const int rbase = cls ? n0 : 0, cbase = cls ? 0 : n0;
    const float ri = r[rbase + h.y + rr];
    for (int k = 0; k < nc; k += 8) {
        const int sb = min(8, nc - k);
        const int2 *b = ent + h.x + (int64_t)lgw * k + lg * sb;
        for (int j = 0; j < sb; ++j) {
            const int2 e = b[j];
            if (e.x == kSellPast) continue;
            const float v = __int_as_float(e.y), f = ri * r[cbase + e.x / (W * 4)];
            if (!(fabsf(v - f) <= 1e-6f * fabsf(v))) atomicExch(err, 1);
        }
    }
}
__global__ void sell_inverse_kernel(const float *r, float *ir, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ir[i] = r[i] > 0.f ? 1.f / r[i] : 0.f;
}

void free_sell(SellDev *sw) {
    if (!sw) return;
    if (sw->ent) (void)hipFree(sw->ent);
    if (sw->ent0) (void)hipFree(sw->ent0);
    if (sw->entc) (void)hipFree(sw->entc);
    if (sw->rs) (void)hipFree(sw->rs);  // (irs is its second half)
    if (sw->head) (void)hipFree(sw->head);
    if (sw->orig) (void)hipFree(sw->orig);
    if (sw->bwd) (void)hipFree(sw->bwd);
    delete sw;
}

bool sell_applicable(const rbg_graph *g, int d) {
    return opt_sell() && g && g->sell && (g->sell->W * 2 == d || (g->sell->W == 32 && d == 128));
}

// the chains run factored (compact entries from the second launch on) when the plan carries row factors
static bool sell_factored(const SellDev *sw) { return sw->rs && sw->entc && opt_sell_factored(); }

const char *sell_kernel_name(const rbg_graph *g, int d, bool compact) {
    const int W = g->sell->W;
    if (W == 32 && d == 64) return compact ? "sell_spmm_kernel<32, 2, true>" : "sell_spmm_kernel<32, 2, false>";
    if (W == 32) return compact ? "sell_spmm_kernel<32, 4, true>" : "sell_spmm_kernel<32, 4, false>";
    return compact ? "sell_spmm_kernel<64, 2, true>" : "sell_spmm_kernel<64, 2, false>";
}
bool sell_chain_factored(const rbg_graph *g) { return g && g->sell && sell_factored(g->sell); }

template <int W, int NS>
static int sell_launch(const SellDev *sw, const SellParams &p, hipStream_t s) {
    const int64_t max_units = std::max(sw->n_units[0], sw->n_units[1]);
    constexpr int per = NS == 2 ? 8 : 4;  // units per 8 workgroups: two XCDs (NS = 2) or one (NS = 4) per (class, slab), four waves each
    const int64_t upw = std::max(1, opt_sell_units_per_wave());  // > 1: a wave walks units t, t + n_w, ... (fewer, longer waves)
    const unsigned grid = (unsigned)(8 * std::max<int64_t>(1, ((max_units + per - 1) / per + upw - 1) / upw));
    if (p.compact) hipLaunchKernelGGL((sell_spmm_kernel<W, NS, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((sell_spmm_kernel<W, NS, false>), dim3(grid), dim3(256), 0, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

static void sell_fill(const SellDev *sw, int W, int NS, SellParams &p) {
    const int64_t off1 = (int64_t)sw->n_class[0] * NS * W;
    p.ent = reinterpret_cast<const v4i *>(sw->ent);
    p.ent0 = reinterpret_cast<const v4i *>(sw->ent0);
    p.head = reinterpret_cast<const int4 *>(sw->head);
    p.orig = sw->orig;
    p.entc = sw->entc;
    p.nt = opt_sell_nt();
    p.rs = sw->rs;
    p.irs = sw->irs;
    for (int c = 0; c < 2; ++c) {
        p.unit_base[c] = sw->unit_base[c];
        p.n_units[c] = sw->n_units[c];
        p.n_class[c] = sw->n_class[c];
        for (int q = 0; q < NS; ++q) p.slab_off[c][q] = (c ? off1 : 0) + (int64_t)q * sw->n_class[c] * W;
    }
}

template <int W, int NS>
static int sell_to_slab(const SellDev *sw, const float *user_emb, const float *item_emb, float *dst, hipStream_t s) {
    const int n0 = sw->n_class[0], n1 = sw->n_class[1];
    const int64_t work = (int64_t)(n0 + n1) * (NS * W / 4);
    hipLaunchKernelGGL((sell_to_slab_kernel<W, NS>), dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, user_emb, item_emb, (int64_t)n0, dst,
                       sw->orig, n0, n1, (int64_t)0, (int64_t)n0 * NS * W);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// K launches.  With ent0 the first one gathers E0 where it lies (two row-major tables) and the mean's epilogue reads E0
// through orig[]: no conversion (7.7 us of a 101 us propagation at the Gowalla shape); without it E0 is converted to slabs
// in layers[K - 1] first.
template <int W, int NS>
static int sell_forward_w(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int K,
                          hipStream_t s) {
    const SellDev *sw = g->sell;
    const int64_t nd = g->n_rows * NS * W;
    const bool rm = sw->ent0 != nullptr && opt_sell_rowmajor();
    const bool fac = sell_factored(sw);
    float *e0s = layers + (int64_t)(K - 1) * nd;
    if (!rm) {
        const int rc = sell_to_slab<W, NS>(sw, user_emb, item_emb, e0s, s);
        if (rc) return rc;
    }
    for (int k = 0; k < K; ++k) {
        SellParams p{};
        sell_fill(sw, W, NS, p);
        p.rm[0] = p.prm[0] = user_emb;
        p.rm[1] = p.prm[1] = item_emb;
        p.x_rm = (rm && k == 0) ? 1 : 0;
        p.xs = (k == 0) ? e0s : layers + (int64_t)(k - 1) * nd;
        p.compact = (fac && k > 0) ? 1 : 0;          // layers[k - 1] holds r (.) E_k
        p.store_scaled = (fac && k < K - 1) ? 1 : 0;
        if (k == K - 1) {
            p.last = 1;
            p.n_prev = K;
            p.prev_scaled = fac ? 1 : 0;
            p.prev0_rm = rm ? 1 : 0;
            p.prev[0] = e0s;
            for (int i = 1; i < K; ++i) p.prev[i] = layers + (int64_t)(i - 1) * nd;
            p.denom = (float)(K + 1);
            p.out = out_mean;
        } else {
            p.ys = layers + (int64_t)k * nd;
        }
        if (int rc = sell_launch<W, NS>(sw, p, s)) return rc;
    }
    return RBG_OK;
}

// Every layer row-major in the reference's numbering (a caller that reads `layers`: NCL, keep_layers): K launches that gather
// the previous layer where it lies and write layers[k] through orig[]; the last one adds the mean (and keeps its own layer
// when asked).  ~2 us per layer slower than the slab chain, no scratch layout.
template <int W, int NS>
static int sell_forward_rowmajor_w(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int K,
                                   bool keep_last, hipStream_t s) {
    const SellDev *sw = g->sell;
    const int64_t nd = g->n_rows * NS * W;
    const int n0 = sw->n_class[0];
    for (int k = 0; k < K; ++k) {
        SellParams p{};
        sell_fill(sw, W, NS, p);
        const float *x = k ? layers + (int64_t)(k - 1) * nd : nullptr;
        p.rm[0] = k ? x : user_emb;
        p.rm[1] = k ? x + (int64_t)n0 * NS * W : item_emb;
        p.x_rm = 1;
        p.last = 1;
        if (k == K - 1) {
            p.n_prev = K;
            p.prev0_rm = 1;
            p.prm[0] = user_emb;
            p.prm[1] = item_emb;
            p.prev_rm_all = 1;
            for (int i = 1; i < K; ++i) p.prev[i] = layers + (int64_t)(i - 1) * nd;
            p.denom = (float)(K + 1);
            p.out = out_mean;
            p.out2 = keep_last ? layers + (int64_t)k * nd : nullptr;
        } else {
            p.denom = 1.f;
            p.out = layers + (int64_t)k * nd;
        }
        if (int rc = sell_launch<W, NS>(sw, p, s)) return rc;
    }
    return RBG_OK;
}

bool sell_rowmajor_applicable(const rbg_graph *g, int d) { return sell_applicable(g, d) && g->sell->ent0 && opt_sell_rowmajor(); }

int sell_forward_rowmajor(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int d, int K,
                          bool keep_last, hipStream_t s) {
    const int W = g->sell->W;
    if (W == 32 && d == 64) return sell_forward_rowmajor_w<32, 2>(g, user_emb, item_emb, out_mean, layers, K, keep_last, s);
    if (W == 32 && d == 128) return sell_forward_rowmajor_w<32, 4>(g, user_emb, item_emb, out_mean, layers, K, keep_last, s);
    if (W == 64 && d == 128) return sell_forward_rowmajor_w<64, 2>(g, user_emb, item_emb, out_mean, layers, K, keep_last, s);
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

// Y = A X (accumulate: Y += A X), X and Y row-major [N, d] in the reference's numbering: rbg_spmm_f32 over the plan.
template <int W, int NS>
static int sell_spmm_w(const rbg_graph *g, const float *X, float *Y, int accumulate, hipStream_t s) {
    const SellDev *sw = g->sell;
    const int n0 = sw->n_class[0];
    SellParams p{};
    sell_fill(sw, W, NS, p);
    p.rm[0] = X;
    p.rm[1] = X + (int64_t)n0 * NS * W;
    p.x_rm = 1;
    p.last = 1;
    p.denom = 1.f;
    p.out = Y;
    if (accumulate) {  // a thread reads the piece of Y it then overwrites
        p.n_prev = 1;
        p.prev0_rm = 1;
        p.prm[0] = Y;
        p.prm[1] = Y + (int64_t)n0 * NS * W;
    }
    return sell_launch<W, NS>(sw, p, s);
}

int sell_spmm(const rbg_graph *g, const float *X, float *Y, int d, int accumulate, hipStream_t s) {
    const int W = g->sell->W;
    if (W == 32 && d == 64) return sell_spmm_w<32, 2>(g, X, Y, accumulate, s);
    if (W == 32 && d == 128) return sell_spmm_w<32, 4>(g, X, Y, accumulate, s);
    if (W == 64 && d == 128) return sell_spmm_w<64, 2>(g, X, Y, accumulate, s);
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

int sell_forward(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int d, int K,
                 hipStream_t s) {
    const int W = g->sell->W;
    if (W == 32 && d == 64) return sell_forward_w<32, 2>(g, user_emb, item_emb, out_mean, layers, K, s);
    if (W == 32 && d == 128) return sell_forward_w<32, 4>(g, user_emb, item_emb, out_mean, layers, K, s);
    if (W == 64 && d == 128) return sell_forward_w<64, 2>(g, user_emb, item_emb, out_mean, layers, K, s);
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

template <int W, int NS>
static int sell_backward_w(const rbg_graph *g, const float *grad_out, float *grad_e0, int K, hipStream_t s) {
    SellDev *sw = g->sell;
    const int64_t n = g->n_rows, nd = n * NS * W;
    // the incoming gradient is gathered and added where it lies unless the result overwrites it (in-place call) or the plan
    // has no row-major entries: then it is converted to slabs first
    const bool rm = sw->ent0 != nullptr && opt_sell_rowmajor() && grad_out != grad_e0;
    const bool fac = sell_factored(sw);
    if (!sw->bwd && (K > 1 || !rm)) {  // slab scratch (g, ping, pong), allocated by the first backward on this handle — never inside a capture
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return RBG_EUNSUPPORTED;
        std::lock_guard<std::mutex> lock(sw->bwd_mutex);
        if (!sw->bwd) {
            float *b = nullptr;
            if (hipMalloc(&b, sizeof(float) * 3 * (size_t)nd) != hipSuccess) {
                (void)hipGetLastError();
                return RBG_EUNSUPPORTED;  // the caller runs the binned chain
            }
            sw->bwd = b;
        }
    }
    float *gs = sw->bwd, *ping = sw->bwd + nd, *pong = sw->bwd + 2 * nd;
    const int n0 = sw->n_class[0];
    if (!rm) {
        const int rc = sell_to_slab<W, NS>(sw, grad_out, grad_out + (int64_t)n0 * NS * W, gs, s);
        if (rc) return rc;
    }
    // dE0 = (g + A (g + A (... (g + A g)))) / (K + 1): K launches, every one adds g in its epilogue; the last divides and
    // writes row-major (A symmetric: rbg_lightgcn_backward_f32 is called with the transposed handles, the handle itself here)
    const float *x = gs;
    for (int i = 0; i < K; ++i) {
        SellParams p{};
        sell_fill(sw, W, NS, p);
        p.rm[0] = p.prm[0] = grad_out;
        p.rm[1] = p.prm[1] = grad_out + (int64_t)n0 * NS * W;
        p.x_rm = (rm && i == 0) ? 1 : 0;
        p.compact = (fac && i > 0) ? 1 : 0;
        p.store_scaled = (fac && i < K - 1) ? 1 : 0;
        p.prev0_rm = rm ? 1 : 0;
        p.xs = x;
        p.n_prev = 1;
        p.prev[0] = gs;
        if (i == K - 1) {
            p.last = 1;
            p.denom = (float)(K + 1);
            p.out = grad_e0;
        } else {
            p.ys = (i & 1) ? pong : ping;
            x = p.ys;
        }
        if (int rc = sell_launch<W, NS>(sw, p, s)) return rc;
    }
    return RBG_OK;
}

// RBG_EUNSUPPORTED = "not this time" (no scratch yet and the stream is capturing, or the allocation failed): the caller
// runs the binned chain instead.
int sell_backward(const rbg_graph *g, const float *grad_out, float *grad_e0, int d, int K, hipStream_t s) {
    const int W = g->sell->W;
    if (W == 32 && d == 64) return sell_backward_w<32, 2>(g, grad_out, grad_e0, K, s);
    if (W == 32 && d == 128) return sell_backward_w<32, 4>(g, grad_out, grad_e0, K, s);
    if (W == 64 && d == 128) return sell_backward_w<64, 2>(g, grad_out, grad_e0, K, s);
    return RBG_EUNSUPPORTED;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_graph_attach_sell(rbg_graph *g, int W, const int32_t *ent, int64_t n_ent, const int32_t *head, const int32_t *unit_base,
                          const int32_t *n_units, const int32_t *orig) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "a SELL plan needs a device graph");
    if (g->base) return fail(RBG_EUNSUPPORTED, "a re-weighted view cannot carry a SELL plan (the plan holds the values)");
    if (W != 32 && W != 64) return fail(RBG_EINVAL, "W = %d (32 or 64)", W);
    if (g->n_users <= 0 || g->n_users >= g->n_rows || g->n_rows != g->n_cols)
        return fail(RBG_EUNSUPPORTED, "a SELL plan needs a square graph with a user / item boundary");
    if (!ent || !head || !unit_base || !n_units || !orig || n_ent < 0 || (n_ent & 1)) return fail(RBG_EINVAL, "NULL or malformed plan array");
    if (g->n_rows > INT32_MAX || n_ent > INT32_MAX - 256) return fail(RBG_EUNSUPPORTED, "graph too large for a SELL plan");
    const int n0 = (int)g->n_users, n1 = (int)(g->n_rows - g->n_users);
    if ((int64_t)std::max(n0, n1) * W * 4 >= kSellPast) return fail(RBG_EUNSUPPORTED, "table too large for 32-bit slab offsets");
    if (unit_base[0] != 0 || n_units[0] < 0 || n_units[1] < 0 || unit_base[1] != n_units[0])
        return fail(RBG_EINVAL, "unit_base / n_units malformed");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    const int n_total = n_units[0] + n_units[1];
    // ---- validate on the device (the arrays are device arrays) ---------------------------------------------------------------
    int *d_err = nullptr;
    RBG_HIP(hipMalloc(&d_err, sizeof(int)));
    RBG_HIP(hipMemset(d_err, 0, sizeof(int)));
    const int lgw = 64 / (W / 4);
    int64_t first_ent1 = n_ent;
    if (n_units[1] > 0) {
        int4 h1;
        if (hipMemcpy(&h1, reinterpret_cast<const int4 *>(head) + n_units[0], sizeof(int4), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipFree(d_err);
            return fail(RBG_EHIP, "reading the plan failed");
        }
        first_ent1 = h1.x;
    }
    if (n_total) hipLaunchKernelGGL(sell_check_units_kernel, dim3((n_total + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const int4 *>(head), n_total,
                                    n_units[0], n0, n1, n_ent, lgw, d_err);
    if (n_ent) hipLaunchKernelGGL(sell_check_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(ent), n_ent, first_ent1, n0,
                                  n1, W, d_err);
    hipLaunchKernelGGL(sell_check_orig_kernel, dim3((unsigned)((g->n_rows + 255) / 256)), dim3(256), 0, 0, orig, (int)g->n_rows, n0, n0, d_err);
    int h_err = 0;
    const hipError_t ce = hipMemcpy(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_err);
    if (ce != hipSuccess) return fail(RBG_EHIP, "plan validation failed to run: %s", hipGetErrorString(ce));
    if (h_err > 0) return fail(RBG_EINVAL, "SELL plan: unit %d is out of range or misaligned", h_err - 1);
    if (h_err == -1) return fail(RBG_EINVAL, "SELL plan: an entry's column offset is out of range");
    if (h_err == -2) return fail(RBG_EINVAL, "SELL plan: orig[] is out of range or crosses the user / item boundary");
    // ---- adopt copies ---------------------------------------------------------------------------------------------------------
    if ((rc = rbg_graph_detach_sell(g))) return rc;
    SellDev *sw = new (std::nothrow) SellDev();
    if (!sw) return fail(RBG_ENOMEM, "out of host memory");
    sw->W = W;
    sw->n_ent = n_ent;
    for (int c = 0; c < 2; ++c) {
        sw->unit_base[c] = unit_base[c];
        sw->n_units[c] = n_units[c];
    }
    sw->n_class[0] = n0;
    sw->n_class[1] = n1;
    const size_t ent_bytes = sizeof(int32_t) * 2 * (size_t)(n_ent + 128), head_bytes = sizeof(int32_t) * 4 * (size_t)std::max(n_total, 1);
    bool ok = hipMalloc(&sw->ent, ent_bytes) == hipSuccess && hipMalloc(&sw->head, head_bytes) == hipSuccess &&
              hipMalloc(&sw->orig, sizeof(int32_t) * (size_t)g->n_rows) == hipSuccess;
    ok = ok && hipMemset(sw->ent, 0, ent_bytes) == hipSuccess;  // (the 128 entries of slack a wave's last 16-byte loads may touch)
    ok = ok && hipMemcpy(sw->ent, ent, sizeof(int32_t) * 2 * (size_t)n_ent, hipMemcpyDeviceToDevice) == hipSuccess;
    ok = ok && (n_total == 0 || hipMemcpy(sw->head, head, sizeof(int32_t) * 4 * (size_t)n_total, hipMemcpyDeviceToDevice) == hipSuccess);
    ok = ok && hipMemcpy(sw->orig, orig, sizeof(int32_t) * (size_t)g->n_rows, hipMemcpyDeviceToDevice) == hipSuccess;
    if (!ok) {
        free_sell(sw);
        return fail(RBG_ENOMEM, "device allocation / copy of the SELL plan failed");
    }
    // the offsets column alone (the factored chain's launches read 4 bytes per entry)
    {
        const size_t cb = sizeof(int32_t) * (size_t)(n_ent + 256);
        if (hipMalloc(&sw->entc, cb) == hipSuccess && hipMemset(sw->entc, 0, cb) == hipSuccess) {
            if (n_ent) hipLaunchKernelGGL(sell_compact_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent), sw->entc,
                                          n_ent);
        } else {
            (void)hipGetLastError();
            if (sw->entc) (void)hipFree(sw->entc);
            sw->entc = nullptr;
        }
    }
    // the row-major twin of the entries (used under option 'sell_rowmajor', default 1; without it E0 is converted to slabs per propagation)
    if ((int64_t)std::max(n0, n1) * 2 * W * 4 < kSellPast) {
        if (hipMalloc(&sw->ent0, ent_bytes) == hipSuccess && hipMemset(sw->ent0, 0, ent_bytes) == hipSuccess) {
            if (n_ent) hipLaunchKernelGGL(sell_first_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent),
                                          reinterpret_cast<int2 *>(sw->ent0), n_ent, first_ent1, sw->orig, n0, W);
            if (hipDeviceSynchronize() != hipSuccess) {
                free_sell(sw);
                return fail(RBG_EHIP, "building the row-major entries failed");
            }
        } else {
            (void)hipGetLastError();
            if (sw->ent0) (void)hipFree(sw->ent0);
    if (sw->entc) (void)hipFree(sw->entc);
    if (sw->rs) (void)hipFree(sw->rs);  // (irs is its second half)
            sw->ent0 = nullptr;
        }
    }
    g->sell = sw;
    return RBG_OK;
}

int rbg_graph_sell_set_factors(rbg_graph *g, const float *r) {
    clear_error();
    if (!g || !g->sell) return fail(RBG_EINVAL, "no SELL plan attached");
    if (!r) return fail(RBG_EINVAL, "r is NULL");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    SellDev *sw = g->sell;
    if (!sw->entc) return fail(RBG_EUNSUPPORTED, "the plan has no compact entries");
    const int n = (int)g->n_rows, n_total = sw->n_units[0] + sw->n_units[1], lgw = 64 / (sw->W / 4);
    (void)hipDeviceSynchronize();  // (not concurrently with launches on this handle)
    if (sw->rs) (void)hipFree(sw->rs);
    sw->rs = sw->irs = nullptr;
    float *buf = nullptr;
    int *d_err = nullptr;
    if (hipMalloc(&buf, sizeof(float) * 2 * (size_t)n) != hipSuccess || hipMalloc(&d_err, sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        if (buf) (void)hipFree(buf);
        return fail(RBG_ENOMEM, "device allocation of the row factors failed");
    }
    bool ok = hipMemcpy(buf, r, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice) == hipSuccess && hipMemset(d_err, 0, sizeof(int)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(sell_inverse_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, buf, buf + n, n);
        const int64_t work = (int64_t)n_total * lgw;
        if (work) hipLaunchKernelGGL(sell_check_factors_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, 0,
                                     reinterpret_cast<const int2 *>(sw->ent), reinterpret_cast<const int4 *>(sw->head), n_total, sw->n_units[0],
                                     sw->n_class[0], sw->W, lgw, buf, d_err);
    }
    int h_err = 0;
    ok = ok && hipMemcpy(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_err);
    if (!ok || h_err) {
        (void)hipFree(buf);
        if (!ok) return fail(RBG_EHIP, "checking the row factors failed to run");
        return fail(RBG_EINVAL, "the plan's values are not r[row] * r[col]");
    }
    sw->rs = buf;
    sw->irs = buf + n;
    return RBG_OK;
}

int rbg_graph_detach_sell(rbg_graph *g) {
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->sell) {
        if (g->device >= 0) {
            int rc = set_device_for(g->device);
            if (rc) return rc;
            (void)hipDeviceSynchronize();
        }
        free_sell(g->sell);
        g->sell = nullptr;
    }
    return RBG_OK;
}

int rbg_graph_has_sell(const rbg_graph *g, int d) { return (g && g->sell && (g->sell->W * 2 == d || (g->sell->W == 32 && d == 128))) ? 1 : 0; }

}  // extern "C"
