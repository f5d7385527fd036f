// sell_stream.h — the column-slab propagation as ONE RESIDENT ROUND of waves per launch (r05).
//
// What the r04 probes said about sell_spmm_kernel (one short wave per unit): with every gather masked off a layer still costs
// 23 of its 31 us at the Gowalla shape — the time is a chain of dependent round trips per wave (eight scalar waits on the
// parameter block, unit header -> orig[] -> first entries -> 1-4 gather batches -> the epilogue's addends one after the other),
// paid by 22 000 waves of which 8 192 are resident at a time.  Most rows of a power-law graph are short (median 12-17 entries:
// 2 batches), so the fixed part of a wave's life is longer than its gathers.
//
// Here the grid is what the chip holds at once and a wave WALKS units t_0, t_1, ... of its (class, slab) role:
//   * the role (class, slab, tables, unit range) is one 64-byte record per XCD in the kernel-argument segment: one scalar load,
//     nothing indexed by a loaded value;
//   * unit headers are scalar loads (the unit index is wave-uniform), the NEXT unit's header is requested when a unit starts;
//   * behind a unit's LAST gather batch the wave requests the unit's row scalars (orig[], r_i) and the next unit's first entries
//     (in the registers the unit's next batch would have used): when the epilogue's store has been issued the next unit's
//     gathers issue at once — per unit the wave waits for its gather batches and nothing else;
//   * the last layer's addends are requested together, not one per wait;
//   * units are dealt in snake order over the degree-sorted list (tier k: k n_w + w for even k, k n_w + n_w - 1 - w for odd k),
//     so every wave gets the same mix of heavy and light units; wide rows (four units, the four waves of a workgroup) sit at
//     the front of the list: tier 0 keeps them in one workgroup, in wave order.
// The plan (units, entries, numbering) and the summation order are those of sell_spmm_kernel: results are bit-identical.
#pragma once

#include "sell_kernel.h"

namespace rbg {

// what XCD x needs to know about its role
struct SellRoleK {
    int64_t ybase;   // float offset of the role's result slab (slab_off[cls][s])
    int64_t xoff;    // float offset of the gathered slab (slab_off[1 - cls][s])
    int32_t n_units, unit_base;
    int32_t n_tab;   // rows of the gathered table (n_class[1 - cls])
    int32_t cbase;   // first row of the class in the plan's numbering (0 / n_class[0])
    int32_t cls, s;
    int32_t xi, xr;  // this XCD's index among the xr XCDs of the role
    // the role's schedule (NULL: units dealt in snake order): wave w walks sched_head[sched_off[w] .. sched_off[w + 1]) — copies of
    // the unit headers in the order of a longest-first deal by gather batches (the host's, sell.hip sell_schedule)
    const int4 *sched_head;
    const int32_t *sched_off;
};
static_assert(sizeof(SellRoleK) == 64, "one s_load_dwordx16 per role");

struct SellStreamParams {
    SellParams p;
    SellRoleK role[8];
};
typedef const __attribute__((address_space(4))) SellStreamParams SellStreamParamsK;
typedef const __attribute__((address_space(4))) v4i SellHeadK;  // unit headers through the scalar cache (the plan is immutable)

// the first batch of a unit's entries (a request: no wait, no arithmetic on the result — the caller widens later)
template <class WT>
__device__ __forceinline__ WT sell_first_batch(const WT *ebase_all, const int4 h, const int lg, const int q4) {
    WT w = {};
    const int nc = (int)((unsigned)h.z >> 16), sb = min(8, nc);
    if (RBG_SELL_DBG(4)) {
        w.x = ((h.x * 37 + lg * 1031 + q4 * 7) & 0x7fff) << 7;
        if constexpr (std::is_same<WT, v4i>::value) w.z = w.x ^ 0x5580; else w.y = w.x ^ 0x5580;
        return w;
    }
    if (2 * q4 < sb) w = (ebase_all + (h.x >> 1))[lg * (sb >> 1) + q4];
    return w;
}
// "this value is in its registers HERE": the wait for its load is placed at this point and not at a later join of paths
// (the unit loop's head would otherwise wait for vmcnt(0) — behind the previous unit's epilogue stores)
template <class T>
__device__ __forceinline__ void sell_pin(T &v) { asm volatile("" : "+v"(v)); }

// the diagnostic build's what-if switches on a batch of entries (nothing in the product)
template <class WT>
__device__ __forceinline__ void sell_debug_entries(WT &e) {
#ifdef RBG_SELL_TRACE
    if (RBG_SELL_DBG(1)) {
        e.x &= 0x3f80;
        if constexpr (std::is_same<WT, v4i>::value) e.z &= 0x3f80; else e.y &= 0x3f80;
    }
    if (RBG_SELL_DBG(8)) {
        e.x = kSellPast;
        if constexpr (std::is_same<WT, v4i>::value) e.z = kSellPast; else e.y = kSellPast;
    }
#endif
}

// the row of lane-group lg in unit h (the plan's numbering, class-local)
__device__ __forceinline__ int sell_row_of(const int4 h, const int lg) {
    const int lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff, r = lg >> lp;
    return h.y + (r < nrows ? r : 0);
}

// reduction of split rows + epilogue of one unit (sell_unit's second half; the addends of the mean are requested together)
template <int W, int NS, bool COMPACT>
__device__ __forceinline__ void sell_finish(SellParamsK &p, const SellLayer &L, const int cls, const int cbase, const int s, const int4 h, SellAcc acc,
                                            const int node, const float r_i, const int64_t ybase, float (*s_wide)[W]) {
    constexpr int G = W / 4;
    constexpr int LGW = 64 / G;
    constexpr int D = NS * W;
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G, wave = (threadIdx.x >> 6) & 3;
    const int row0 = h.y, lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
    const bool wide = (h.w >> 16) & 1;
    const int r = lg >> lp;
    const int row = row0 + (r < nrows ? r : 0);
    const int parts = 1 << lp;
    if (lp > 0) {
#pragma unroll
        for (int off = 1; off < LGW; off <<= 1) {
            const float a0 = __shfl_xor(acc.lo.x, off * G), a1 = __shfl_xor(acc.lo.y, off * G);
            const float a2 = __shfl_xor(acc.hi.x, off * G), a3 = __shfl_xor(acc.hi.y, off * G);
            if (off < parts) { acc.lo.x += a0; acc.lo.y += a1; acc.hi.x += a2; acc.hi.y += a3; }
        }
    }
    if (wide) {  // 4 waves x LGW pieces of ONE row (tier 0 only: the four waves of the workgroup hold the row's four units)
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
        __syncthreads();
    }
    const bool owner = (lg & (parts - 1)) == 0 && r < nrows && (!wide || wave == 0);
    float nsc = 0.f;
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L.last && p.noise) {  // the noise row's norm spans all NS slabs: every lane-group reads the whole row
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
    if (!owner) return;
    const int64_t o = ybase + (int64_t)row * W + sl * 4;
    const int64_t orm = (int64_t)node * D + s * W + sl * 4;  // row-major [N, D], the reference's numbering
    const float *prev0 = L.prev0_rm ? p.prm[cls] + (orm - (int64_t)cbase * D) : p.prev[0] + o;
    float4 y = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
    if (COMPACT) { y.x *= r_i; y.y *= r_i; y.z *= r_i; y.w *= r_i; }  // y = r_i sum_j z_j
    if (L.last) {
        const bool ntl = (p.nt & 2) != 0, nts = (p.nt & 1) != 0;
        // the addends are requested four at a time (sell_unit waits for them one by one) and summed in sell_unit's order:
        // sum = prev0 (+ prev[1] + ...), or prev0 + (z_1 + z_2 + ...) / r_i when the layers in between are stored scaled
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), zs = make_float4(0.f, 0.f, 0.f, 0.f);
        const int np = L.n_prev;
        const int64_t oprev = (L.prev_scaled || !p.prev_rm_all) ? o : orm;
        for (int i0 = 0; i0 < np; i0 += 4) {
            float4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 + j < np) a[j] = ld4((i0 + j) == 0 ? prev0 : p.prev[i0 + j] + oprev, ntl);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
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
        if (p.noise) {
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

// One resident round: workgroup b runs on XCD b & 7 with the role the host wrote into role[b & 7]; wave w of the role's n_w
// walks units w, 2 n_w - 1 - w, 2 n_w + w, ...
template <int W, int NS, bool COMPACT, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void sell_stream_kernel(const SellStreamParams q_) {
    SellStreamParamsK &q = *(SellStreamParamsK *)__builtin_amdgcn_kernarg_segment_ptr();  // (= q_, read in place)
    SellParamsK &p = q.p;
    constexpr int G = W / 4;
    constexpr int LGW = 64 / G;
    using WT = std::conditional_t<COMPACT, v2i, v4i>;
    __shared__ float s_wide[4][W];
    SellClock clk;
    clk.start();
    // ---- everything wave-uniform, requested at once: the role record, the layer, the table -----------------------------------------
    const auto &R = q.role[blockIdx.x & 7];
    const int cls = R.cls, s = R.s, cbase = R.cbase;
    const unsigned nun = (unsigned)R.n_units;
    const int64_t ybase = R.ybase;
    const SellLayer L = sell_layer_of(p);
    const int n_tab = R.n_tab;
    const float *tab = L.x_rm ? (cls ? p.rm[0] : p.rm[1]) + s * W : L.xs + R.xoff;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(tab), 0, L.x_rm ? (unsigned)n_tab * (unsigned)(p.rm_ld * 4) - s * W * 4 : n_tab * W * 4, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lg = lane / G, q4 = lane & 3, lane_off = (lane % G) * 16;
    const unsigned n_w = (gridDim.x >> 3) * 4 * (unsigned)R.xr;
    const unsigned wi = (unsigned)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x >> 3) * R.xr + R.xi) * 4 + wave));
    const WT *ebase_all;
    if constexpr (COMPACT) ebase_all = reinterpret_cast<const v2i *>(p.entc);
    else ebase_all = L.x_rm ? p.ent0 : p.ent;
    const int sh = (!COMPACT && L.x_rm) ? p.rm_shift : 0;
    const bool need_node = L.last || L.prev0_rm, need_r = COMPACT || L.store_scaled;
    // the units of this wave: a slice of the role's schedule, or every n_w-th unit of the plan's list in snake order
    const bool sched = R.sched_off != nullptr;
    if (!sched && wi >= nun) {
        clk.dump((COMPACT ? 2 : 0) + (L.last ? 1 : 0));
        return;
    }
    unsigned t = wi, t_end = nun;
    if (sched) {
        const auto *off = (const __attribute__((address_space(4))) int32_t *)R.sched_off;
        t = (unsigned)off[wi], t_end = (unsigned)off[wi + 1];
        if (t >= t_end) {
            clk.dump((COMPACT ? 2 : 0) + (L.last ? 1 : 0));
            return;
        }
    }
    SellHeadK *heads = sched ? (SellHeadK *)R.sched_head : (SellHeadK *)(p.head + R.unit_base);
    auto head_of = [&](const unsigned u) __attribute__((always_inline)) {
        const v4i v = heads[u];
        return make_int4(v.x, v.y, v.z, v.w);
    };
    int4 h = head_of(t);
    unsigned tier = 0;
    unsigned tn = sched ? t + 1 : 2 * n_w - 1 - wi;  // (snake: tier 1)
    WT w = sell_first_batch<WT>(ebase_all, h, lg, q4);
    sell_widen(w, sh);
    sell_pin(w);
    clk.lap(0);
    for (;;) {
        const bool more = tn < t_end;
        int4 hn = h;
        if (more) hn = head_of(tn);  // scalar: back long before the last batch
        const int nc = (int)((unsigned)h.z >> 16);
        clk.count(nc * LGW);
        SellAcc acc = {{0.f, 0.f}, {0.f, 0.f}};
        // what rides behind the unit's LAST batch of gathers: its row scalars (orig[], r_i: the epilogue's) and the next unit's
        // first batch of entries — in the registers the next batch of this unit would have taken
        int node = 0;
        float r_i = 1.f;
        WT wn = {};
        auto tail_requests = [&]() __attribute__((always_inline)) {
            const int row = cbase + sell_row_of(h, lg);
            if (need_node) node = p.orig[row];
            if (need_r) r_i = p.rs[row];
            if (more) wn = sell_first_batch<WT>(ebase_all, hn, lg, q4);
        };
        if (nc <= 0) {
            tail_requests();
        } else {
            const WT *base = ebase_all + (h.x >> 1);
            int sb = min(8, nc);
            for (int k = 0; k < nc; k += 8) {
                const int sbn = min(8, nc - k - 8);  // slots of the next batch (<= 0: this is the last one)
                SellRows x;
                // (the rows a short batch does not load are "defined" here, by an empty statement: left undefined, hipcc hoists their
                // implicit definitions to the kernel's entry and spills them across the unit loop — 31 registers through scratch)
                asm volatile("" : "=v"(x.r0), "=v"(x.r1), "=v"(x.r2), "=v"(x.r3), "=v"(x.r4), "=v"(x.r5), "=v"(x.r6), "=v"(x.r7));
                sell_issue_n(sb, x, w, rs, lane_off);
                wn = WT{};
                if (sbn > 0) {
                    if (RBG_SELL_DBG(4)) {
                        wn.x = ((h.x * 37 + k * 131 + lg * 1031 + q4 * 7) & 0x7fff) << 7;
                        if constexpr (std::is_same<WT, v4i>::value) wn.z = wn.x ^ 0x5580; else wn.y = wn.x ^ 0x5580;
                    } else if (2 * q4 < sbn) wn = base[((LGW * (k + 8)) >> 1) + lg * (sbn >> 1) + q4];
                } else {
                    tail_requests();
                }
                if (k == 0) clk.lap(1);
                sell_consume_n(sb, acc, x, w);
                sell_widen(wn, sh);
                sell_debug_entries(wn);
                w = wn;
                sb = sbn;
            }
        }
        if (nc <= 0) {
            sell_widen(wn, sh);
            w = wn;
        }
        sell_pin(w);  // (the next unit's first batch: it arrived right behind this unit's last gathers)
        clk.lap(2);
        if (RBG_SELL_DBG(2)) {
            if (acc.lo.x == 12345.678f && node == -77 && r_i == -3.f) L.ys[0] = acc.hi.y;  // (never: keeps the sums and the row scalars alive)
        } else
        sell_finish<W, NS, COMPACT>(p, L, cls, cbase, s, h, acc, node, r_i, ybase, s_wide);
        clk.lap(3);
        if (!more) break;
        h = hn;
        ++tier;
        tn = sched ? tn + 1 : (tier + 1) * n_w + (((tier + 1) & 1) ? n_w - 1 - wi : wi);
        clk.lap(4);
    }
    clk.dump((COMPACT ? 2 : 0) + (L.last ? 1 : 0));
}

}  // namespace rbg
