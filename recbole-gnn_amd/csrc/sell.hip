// sell.hip — the propagation of LightGCN.forward / SGL.forward (lightgcn.py:70-81, sgl.py:128-145) over COLUMN SLABS with a
// sliced-ELL graph (r03).  Used by rbg_lightgcn_forward_f32 at d = 64 when a plan is attached
// (rbg_graph_attach_sell; planner sell_plan.hip, specification tests/sell_spec.py) and the caller does not read the intermediate layers.
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

#include <algorithm>
#include <mutex>
#include <new>
#include <type_traits>

#include "internal.h"
#include "sell_kernel.h"

namespace rbg {

// W = slab width (32).  One wave per unit; workgroup b runs on XCD b & 7: XCDs 0-3 take user rows, 4-7 item rows, XCD x of a class
// owns slab x % NS; the 4 / NS XCDs of a (class, slab) role share its units.
template <int W, int NS, bool COMPACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void sell_spmm_kernel(const SellLaunch a_) {
    SellLaunchK &A = sell_kernarg();  // (= a_, read in place)
    SellParamsK &p = A.p;
    SellClock clk;
    clk.start();
    const int x = blockIdx.x & 7;
    const int cls = x >> 2, xr = x & 3;
    const int s = xr & (NS - 1), xi = xr / NS;
    constexpr int XR = 4 / NS;  // XCDs per role
    // the role record (one scalar load, indexed by blockIdx alone) and the layer's switches: everything a wave needs before its
    // header, requested together
    const float *xtab = A.role[x].xtab;
    const unsigned tab_bytes = A.role[x].tab_bytes;
    const SellRole R{A.role[x].heads, A.role[x].ents, A.role[x].ybase, A.role[x].n_units, A.role[x].cbase, A.role[x].c16, A.role[x].wbase};
    const SellLayer L = sell_layer_of(p);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xtab), 0, (int)tab_bytes, 0x00020000);
    // (the grid covers the units: one unit per single-wave workgroup — a retired wave's slot is refilled on its own (r06; r03-r05:
    // four-wave workgroups, because a wide row's four units met in LDS) — heaviest first, dealt by the hardware dispatcher)
    const unsigned t = ((blockIdx.x >> 3) * XR + xi) * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (t < (unsigned)R.n_units) {
        const int4 h = R.heads[t];
        clk.lap(0);
        sell_unit<W, NS, COMPACT>(p, L, R, cls, s, t, h, rs, clk);
        clk.lap(3);
    }
    clk.dump((COMPACT ? 2 : 0) + (p.last ? 1 : 0));
}

// the two embedding tables, row-major [n, NS W] in the reference's numbering -> slabs in the plan's numbering
template <int W, int NS>
__global__ __launch_bounds__(256) void sell_to_slab_kernel(const float *user_emb, const float *item_emb, int64_t n_users, float *dst,
                                                           const int32_t *orig, int n0, int n1, int64_t off0, int64_t off1) {
    constexpr int D = NS * W;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t g = gid / (D / 4);
    const int c4 = (int)(gid % (D / 4)) * 4;
    if (g >= (int64_t)n0 + n1) return;
    const int cls = g >= n0, row = (int)(cls ? g - n0 : g);
    const int64_t node = orig[g];
    const float *src = node < n_users ? user_emb + node * D : item_emb + (node - n_users) * D;
    const int s = c4 / W;
    const int64_t so = (cls ? off1 + (int64_t)s * n1 * W : off0 + (int64_t)s * n0 * W) + (int64_t)row * W + (c4 - s * W);
    *reinterpret_cast<float4 *>(dst + so) = *reinterpret_cast<const float4 *>(src + c4);
}

// memory safety of a plan (its content is the planner's business: parity tests pin it): every index the kernel dereferences
// wide units of each class (a plan the caller attached: the planner knows its own)
__global__ void sell_count_wide_kernel(const int4 *head, int n_units_total, int unit_base1, int *n_wide) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_units_total && ((head[t].w >> 16) & 1)) atomicAdd(&n_wide[t >= unit_base1], 1);
}
__global__ void sell_check_units_kernel(const int4 *head, int n_units_total, int unit_base1, int n0, int n1, int64_t n_ent, int lgw, int nw0,
                                        int nw1, int *err) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_units_total) return;
    const int4 h = head[t];
    const int cls = t >= unit_base1, nc = (int)((unsigned)h.z >> 16), lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff, wide = (h.w >> 16) & 1;
    const int n_c = cls ? n1 : n0, nw = cls ? nw1 : nw0;
    const int U = (int)((unsigned)h.w >> 17), j = h.z & 0xffff;
    bool bad = h.x < 0 || (h.x & 1) || (nc & 1) || nc > 65534 || (int64_t)h.x + (int64_t)lgw * nc > n_ent;
    bad = bad || lp < 0 || (1 << lp) > lgw || nrows < 0 || nrows > (lgw >> lp) || h.y < 0 || h.y + nrows > n_c;
    const int tl = t - (cls ? unit_base1 : 0);
    bad = bad || (wide != 0) != (tl < nw);  // the wide units are the first units of their class: their number is their scratch slot
    if (wide) {  // a wide row = U consecutive units, all flagged, one row; unit j of it knows j and U
        bad = bad || U < 1 || j >= U || tl - j < 0 || tl - j + U > nw || nrows != 1 || (1 << lp) != lgw;
        if (!bad) {
            const int4 h0 = head[t - j];
            bad = !((h0.w >> 16) & 1) || h0.y != h.y || (h0.z & 0xffff) != 0 || (int)((unsigned)h0.w >> 17) != U;
        }
    } else {
        bad = bad || j != 0 || U != 0;
    }
    if (bad) atomicExch(err, 1 + t);
}
// (stride = W 4, the rows of a slab; a rectangular plan: n0 = n1 = the table's rows, stride = 2 W 4)
__global__ void sell_check_entries_kernel(const int2 *ent, int64_t n_ent, int64_t first_ent1, int n0, int n1, int stride, int *err) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
        const int off = ent[e].x;
        const int64_t lim = (int64_t)(e >= first_ent1 ? n0 : n1) * stride;  // class 0 rows gather the class 1 table and vice versa
        if (off != kSellPast && (off < 0 || off >= lim || off % stride != 0)) atomicExch(err, -1);
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
// ... and as 16-bit slab-row numbers (offset / (W 4); padding = 0xffff), [n, n_alloc) filled with padding
__global__ void sell_compact16_kernel(const int2 *ent, uint16_t *entc16, int64_t n, int64_t n_alloc, int shift) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_alloc; e += (int64_t)gridDim.x * blockDim.x) {
        const int off = e < n ? ent[e].x : kSellPast;
        entc16[e] = (off == kSellPast || (off >> shift) >= 0xffff) ? (uint16_t)0xffff : (uint16_t)(off >> shift);  // (a class too large for 16 bits never reads its half)
    }
}

// a re-weighted view's values: ent0v[pos].y = vals[src[pos]]
__global__ void sell_refresh_values_kernel(int2 *ent, const int32_t *src, const float *vals, int64_t n_ent) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_ent; e += (int64_t)gridDim.x * blockDim.x) {
        const int sidx = src[e];
        if (sidx >= 0) ent[e].y = __float_as_int(vals[sidx]);
    }
}

// factor check: one thread per (unit, lane-group) walks its slots; every stored value must be r[row] * r[col] to 1e-6 relative
__global__ void sell_check_factors_kernel(const int2 *ent, const int4 *head, int n_units_total, int unit_base1, int n0, int W, int lgw,
                                          const float *r, int *err) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)(gid / lgw), lg = (int)(gid % lgw);
    if (t >= n_units_total) return;
    const int4 h = head[t];
    const int cls = t >= unit_base1, nc = (int)((unsigned)h.z >> 16), lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
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
    if (!sw->borrowed) {  // (a view owns its valued row-major entries and their first-batch block only)
        if (sw->ent) (void)hipFree(sw->ent);
        if (sw->entc) (void)hipFree(sw->entc);
        if (sw->entc16) (void)hipFree(sw->entc16);
        if (sw->rs) (void)hipFree(sw->rs);  // (irs is its second half)
        if (sw->head) (void)hipFree(sw->head);
        if (sw->orig) (void)hipFree(sw->orig);
        if (sw->src) (void)hipFree(sw->src);
    }
    if (sw->ent0 && sw->ent0 != sw->ent) (void)hipFree(sw->ent0);  // (a rectangular plan's ent0 IS its ent)
    if (sw->wide_part) (void)hipFree(sw->wide_part);
    if (sw->wide_ctr) (void)hipFree(sw->wide_ctr);
    if (sw->bwd) (void)hipFree(sw->bwd);
    delete sw;
}

static bool sell_width_ok(const SellDev *sw, int d) { return sw->W * 2 == d || (sw->W == 32 && (d == 128 || d == 32)); }
// a re-weighted view runs its plan only after the caller has refreshed it once (rbg_graph_refresh_values: the plan holds a COPY
// of the values, the binned kernel reads the caller's array at launch time)
static bool sell_usable(const rbg_graph *g) { return g && g->sell && (!g->sell->borrowed || g->sell->view_fresh); }

bool sell_applicable(const rbg_graph *g, int d) {  // the slab chains (never a rectangular block: it has no slab layout)
    return opt_sell() && sell_usable(g) && !g->sell->borrowed && !g->sell->rect && sell_width_ok(g->sell, d);
}

// the chains run factored (compact entries from the second launch on) when the plan carries row factors
static bool sell_factored(const SellDev *sw) { return sw->rs && sw->entc && opt_sell_factored() && !sw->borrowed; }

const char *sell_kernel_name(const rbg_graph *g, int d, bool compact) {
    static thread_local char buf[64];
    const int W = g->sell->W, ns = d / W;
    snprintf(buf, sizeof buf, "sell_spmm_kernel<%d, %d, %s>", W, ns, compact ? "true" : "false");
    return buf;
}
bool sell_chain_factored(const rbg_graph *g) { return g && g->sell && sell_factored(g->sell); }

// one launch: both row classes (user rows on XCDs 0-3, item rows on 4-7), one wave per unit.
// Measured and moved out of the product (DESIGN results log 6.9, 6.10, 6.12; devtools/experiments): two gather batches in flight per
// wave (occupancy 5: +7 %), waves that walk several units (+7 .. +30 %), one launch per row class (+4 % at Amazon-Book, -0.4 % at
// 1.3 M nodes), one resident round of waves with cross-unit prefetch and a longest-first schedule (+10 %), 64-byte slabs (+45 %).
template <int W, int NS>
static int sell_launch(const SellDev *sw, SellParams &p, hipStream_t s) {
    const int64_t units = std::max(sw->n_units[0], sw->n_units[1]);
    const int wpb = opt_sell_wpb();  // waves per workgroup (1, 2 or 4: nothing in the kernel is workgroup-wide)
    const int per = wpb * (4 / NS);  // units per 8 workgroups: the XCDs of a (class, slab) role
    const unsigned grid = (unsigned)(8 * std::max<int64_t>(1, (units + per - 1) / per));
    SellLaunch a;
    a.p = p;
    for (int x = 0; x < 8; ++x) {  // what sell_spmm_kernel's workgroups on XCD x work on (x = blockIdx & 7)
        const int cls = x >> 2, sl = (x & 3) & (NS - 1);
        SellRoleK &r = a.role[x];
        r = SellRoleK{};
        r.xtab = p.x_rm ? p.rm[1 - cls] + sl * W : p.xs + p.slab_off[1 - cls][sl];
        r.tab_bytes = p.x_rm ? (uint32_t)p.rm_rows[1 - cls] * (uint32_t)(p.rm_ld * 4) - (uint32_t)(sl * W * 4) : (uint32_t)p.n_class[1 - cls] * (uint32_t)(W * 4);
        r.heads = p.head + p.unit_base[cls];
        r.n_units = p.n_units[cls];
        r.cbase = cls ? p.n_class[0] : 0;
        r.wbase = cls ? sw->n_wide_units[0] : 0;
        r.ybase = p.slab_off[cls][sl];
        if (p.compact) {
            r.c16 = (p.entc16 && p.c16_ok[cls]) ? 1 : 0;
            r.ents = r.c16 ? (const void *)p.entc16 : (const void *)p.entc;
        } else {
            r.ents = p.x_rm ? (const void *)p.ent0 : (const void *)p.ent;
        }
    }
    if (p.compact) hipLaunchKernelGGL((sell_spmm_kernel<W, NS, true>), dim3(grid), dim3(64 * wpb), 0, s, a);
    else hipLaunchKernelGGL((sell_spmm_kernel<W, NS, false>), dim3(grid), dim3(64 * wpb), 0, s, a);
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
    p.entc16 = (sw->entc16 && opt_sell_c16()) ? reinterpret_cast<const int32_t *>(sw->entc16) : nullptr;
    p.c16_shift = W == 64 ? 8 : 7;
    p.c16_ok[0] = sw->n_class[1] < 65535, p.c16_ok[1] = sw->n_class[0] < 65535;
    p.nt = opt_sell_nt();
    p.rs = sw->rs;
    p.irs = sw->irs;
    p.rm_ld = NS * W;
    p.rm_shift = NS == 4 ? 1 : (NS == 1 ? -1 : 0);
    p.rm_rows[0] = sw->rect ? sw->n_tab : sw->n_class[0];
    p.rm_rows[1] = sw->rect ? sw->n_tab : sw->n_class[1];
    p.wide_part = sw->wide_part;
    p.wide_ctr = sw->wide_ctr;
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

static void sell_layer_params(SellParams &p, const SellParams &base, const SellChainLayer &l) {
    p = base;
    p.xs = l.xs, p.ys = l.ys, p.x_rm = l.x_rm, p.compact = l.compact, p.store_scaled = l.store_scaled;
    p.last = l.last, p.n_prev = l.n_prev, p.prev0_rm = l.prev0_rm, p.prev_scaled = l.prev_scaled;
}

// a chain = the parameter block its layers share + what differs per layer, issued as K launches.
// (r04, measured and removed from the build: ONE persistent launch for the K layers — workgroup tickets or a static deal, a layer
// barrier across the XCDs — is 27 - 110 % SLOWER than the K launches in every form tried: contended same-address atomics cost
// 30 - 100 ns each on this chip, and the hardware's kernel boundary + dispatcher beat the software barrier + deal;
// devtools/experiments/sell_persist.hip, profiles/r04_persist_probe.jsonl, DESIGN 6.10.  Also measured and removed: the K layers as
// the TWO independent per-class launch chains the bipartite graph allows (U1 -> I2 -> U3, I1 -> U2 -> I3) on two streams, so that
// one chain's waves fill the other's ramp / tail: 111 - 120 vs 92 us — profiles/r04_two_chains_probe.jsonl.)
template <int W, int NS>
static int sell_run_chain(const rbg_graph *g, const SellParams &base, const SellChainLayer *lay, int K, hipStream_t s) {
    const SellDev *sw = g->sell;
    for (int k = 0; k < K; ++k) {
        SellParams p;
        sell_layer_params(p, base, lay[k]);
        if (int rc = sell_launch<W, NS>(sw, p, s)) return rc;
    }
    return RBG_OK;
}

// K layers.  With ent0 the first one gathers E0 where it lies (two row-major tables) and the mean's epilogue reads E0
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
    SellParams base{};
    sell_fill(sw, W, NS, base);
    base.rm[0] = base.prm[0] = user_emb;
    base.rm[1] = base.prm[1] = item_emb;
    base.prev[0] = e0s;  // (the last layer's addends)
    for (int i = 1; i < K; ++i) base.prev[i] = layers + (int64_t)(i - 1) * nd;
    base.denom = (float)(K + 1);
    base.out = out_mean;
    SellChainLayer lay[RBG_MAX_FUSED_LAYERS + 1] = {};
    for (int k = 0; k < K; ++k) {
        lay[k].x_rm = (rm && k == 0) ? 1 : 0;
        lay[k].xs = (k == 0) ? e0s : layers + (int64_t)(k - 1) * nd;
        lay[k].compact = (fac && k > 0) ? 1 : 0;          // layers[k - 1] holds r (.) E_k
        lay[k].store_scaled = (fac && k < K - 1) ? 1 : 0;
        if (k == K - 1) {
            lay[k].last = 1;
            lay[k].n_prev = K;
            lay[k].prev_scaled = fac ? 1 : 0;
            lay[k].prev0_rm = rm ? 1 : 0;
        } else {
            lay[k].ys = layers + (int64_t)k * nd;
        }
    }
    return sell_run_chain<W, NS>(g, base, lay, K, s);
}

// Every layer row-major in the reference's numbering (a caller that reads `layers`: NCL, keep_layers; a list of per-layer
// graphs — SGL's RW views, sgl.py:89-91 — since every plan has its own row numbering): K launches that gather the previous
// layer where it lies and write layers[k] through orig[]; the last one adds the mean (and keeps its own layer when asked).
// ~2 us per layer slower than the slab chain, no scratch layout.
template <int W, int NS>
static int sell_forward_rowmajor_w(const rbg_graph *const *graphs, int n_graphs, const float *user_emb, const float *item_emb, float *out_mean,
                                   float *layers, int K, bool keep_last, hipStream_t s) {
    const int64_t nd = graphs[0]->n_rows * NS * W;
    for (int k = 0; k < K; ++k) {
        const SellDev *sw = graphs[n_graphs == 1 ? 0 : k]->sell;
        const int n0 = sw->n_class[0];
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

bool sell_rowmajor_applicable(const rbg_graph *g, int d) {
    return opt_sell() && sell_usable(g) && sell_width_ok(g->sell, d) && g->sell->ent0 && opt_sell_rowmajor();
}

// the plain layer Y (+)= A X on the plan: through the row-major entries, or — a plan without them (tables beyond 32-bit byte
// offsets) on contiguous X — through the handle's slab scratch
bool sell_plain_applicable(const rbg_graph *g, int d, int64_t ldx) {
    if (sell_rowmajor_applicable(g, d)) return sell_stride_ok(g, d, ldx);
    return sell_applicable(g, d) && ldx == d;
}

#define RBG_SELL_DISPATCH(W_, d_, CALL)                              \
    do {                                                             \
        if ((W_) == 32 && (d_) == 64) return CALL(32, 2);            \
        if ((W_) == 32 && (d_) == 128) return CALL(32, 4);           \
        if ((W_) == 32 && (d_) == 32) return CALL(32, 1);            \
        if ((W_) == 64 && (d_) == 128) return CALL(64, 2);           \
    } while (0)

int sell_forward_rowmajor(const rbg_graph *const *graphs, int n_graphs, const float *user_emb, const float *item_emb, float *out_mean,
                          float *layers, int d, int K, bool keep_last, hipStream_t s) {
    const int W = graphs[0]->sell->W;
    for (int i = 1; i < n_graphs; ++i)
        if (graphs[i]->sell->W != W) return fail(RBG_EUNSUPPORTED, "per-layer plans of different slab widths");
#define CALL(W_, NS_) sell_forward_rowmajor_w<W_, NS_>(graphs, n_graphs, user_emb, item_emb, out_mean, layers, K, keep_last, s)
    RBG_SELL_DISPATCH(W, d, CALL);
#undef CALL
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

// Y = A X (accumulate: Y += A X), X and Y row-major [N, d] in the reference's numbering: rbg_spmm_f32 over the plan.
// noise != NULL: Y = A X + sign(A X) * noise / |noise row| * eps (rbg_spmm_noise_f32; simgcl.py:29-33).
// The per-handle slab scratch of the launches of a plan WITHOUT row-major entries (tables beyond 32-bit byte offsets, or option
// "sell_rowmajor" = 0): at least `floats` floats, (re)allocated outside captures only; ONE such launch sequence at a time on
// the handle.  RBG_EUNSUPPORTED = not available now (the caller runs the binned kernels).
static int sell_scratch(SellDev *sw, int64_t floats, hipStream_t s) {
    if (sw->bwd && sw->bwd_floats >= floats) return RBG_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return RBG_EUNSUPPORTED;
    std::lock_guard<std::mutex> lock(sw->bwd_mutex);
    if (sw->bwd && sw->bwd_floats >= floats) return RBG_OK;
    if (sw->bwd) {  // a wider table than the first call's: launches in flight still read the old block
        (void)hipDeviceSynchronize();
        (void)hipFree(sw->bwd);
        sw->bwd = nullptr, sw->bwd_floats = 0;
    }
    float *b = nullptr;
    if (dev_malloc(&b, sizeof(float) * (size_t)floats) != hipSuccess) {
        (void)hipGetLastError();
        return RBG_EUNSUPPORTED;
    }
    sw->bwd = b, sw->bwd_floats = floats;
    return RBG_OK;
}

template <int W, int NS>
static int sell_spmm_w(const rbg_graph *g, const float *X, int64_t ldx, float *Y, int accumulate, const float *noise, float eps, hipStream_t s) {
    const SellDev *sw = g->sell;
    const int n0 = sw->n_class[0];
    SellParams p{};
    sell_fill(sw, W, NS, p);
    if (!sw->ent0 || !opt_sell_rowmajor()) {  // no row-major entries: X goes through the handle's slab scratch (ldx == d here)
        const int rc = sell_scratch(g->sell, g->n_rows * NS * W, s);
        if (rc) return rc;
        const int rc2 = sell_to_slab<W, NS>(sw, X, X + (int64_t)n0 * ldx, sw->bwd, s);
        if (rc2) return rc2;
        p.xs = sw->bwd;
    } else {
        p.rm[0] = X;
        p.rm[1] = sw->rect ? X : X + (int64_t)n0 * ldx;  // (rect: both classes gather the one table)
        p.x_rm = 1;
    }
    if (ldx != NS * W) {  // X is a column block of a wider row-major buffer (NGCF's concatenated output, ngcf.py:100)
        p.rm_ld = (int32_t)ldx;
        int sh = 0;
        while (((int64_t)2 * W << sh) < ldx) ++sh;
        p.rm_shift = sh;
    }
    p.last = 1;
    p.denom = 1.f;
    p.out = Y;
    p.noise = noise;
    p.eps = eps;
    if (accumulate) {  // a thread reads the piece of Y it then overwrites
        p.n_prev = 1;
        p.prev0_rm = 1;
        p.prm[0] = Y;
        p.prm[1] = Y + (int64_t)n0 * NS * W;
    }
    return sell_launch<W, NS>(sw, p, s);
}

// out = (srcs[0] + ... + srcs[n - 1] (+ partial) + A X) / (n + 1): the last layer of a propagation whose layers the caller keeps
// row-major (rbg_spmm_mean_f32: the sharded forward's last launch — lightgcn.py:76-78 on a rank's rows)
template <int W, int NS>
static int sell_spmm_mean_w(const rbg_graph *g, const float *X, const float *partial, const float *const *srcs, int n_srcs, float *out_mean,
                            float denom, hipStream_t s) {
    const SellDev *sw = g->sell;
    const int n0 = sw->n_class[0];
    SellParams p{};
    sell_fill(sw, W, NS, p);
    p.rm[0] = X;
    p.rm[1] = sw->rect ? X : X + (int64_t)n0 * NS * W;
    p.x_rm = 1;
    p.last = 1;
    p.prev0_rm = 1;
    p.prev_rm_all = 1;
    p.prm[0] = srcs[0];
    p.prm[1] = srcs[0] + (int64_t)n0 * NS * W;
    for (int i = 1; i < n_srcs; ++i) p.prev[i] = srcs[i];
    p.n_prev = n_srcs;
    if (partial) p.prev[p.n_prev++] = partial;
    p.denom = denom;  // (n_srcs + 1 for the mean; 1 for rbg_spmm_add_f32's Y = Z + A X)
    p.out = out_mean;
    return sell_launch<W, NS>(sw, p, s);
}

int sell_spmm_mean(const rbg_graph *g, const float *X, const float *partial, const float *const *srcs, int n_srcs, float *out_mean, int d,
                   float denom, hipStream_t s) {
    if (n_srcs + (partial ? 1 : 0) > RBG_MAX_FUSED_LAYERS + 1 || !g->sell->ent0 || !opt_sell_rowmajor()) return RBG_EUNSUPPORTED;
    const int W = g->sell->W;
#define CALL(W_, NS_) sell_spmm_mean_w<W_, NS_>(g, X, partial, srcs, n_srcs, out_mean, denom, s)
    RBG_SELL_DISPATCH(W, d, CALL);
#undef CALL
    return RBG_EUNSUPPORTED;
}

// the row strides of X the plan's row-major entries reach: d itself, or 2 W << k (k <= 3) while 32-bit offsets hold every row
bool sell_stride_ok(const rbg_graph *g, int d, int64_t ldx) {
    const SellDev *sw = g->sell;
    if (ldx == d) return !sw->rect || (int64_t)sw->n_tab * d * 4 < kSellPast;  // (rect plans are cut for 2 W; 4 W doubles the offsets)
    const int64_t w2 = 2 * sw->W;
    if (ldx < d || ldx % w2 || ldx / w2 > 8 || ((ldx / w2) & (ldx / w2 - 1))) return false;
    return (int64_t)(sw->rect ? sw->n_tab : std::max(sw->n_class[0], sw->n_class[1])) * ldx * 4 < kSellPast;
}

int sell_spmm(const rbg_graph *g, const float *X, int64_t ldx, float *Y, int d, int accumulate, const float *noise, float eps, hipStream_t s) {
    const int W = g->sell->W;
#define CALL(W_, NS_) sell_spmm_w<W_, NS_>(g, X, ldx, Y, accumulate, noise, eps, s)
    RBG_SELL_DISPATCH(W, d, CALL);
#undef CALL
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

int sell_forward(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int d, int K,
                 hipStream_t s) {
    const int W = g->sell->W;
#define CALL(W_, NS_) sell_forward_w<W_, NS_>(g, user_emb, item_emb, out_mean, layers, K, s)
    RBG_SELL_DISPATCH(W, d, CALL);
#undef CALL
    return fail(RBG_EUNSUPPORTED, "sell path at d = %d", d);
}

template <int W, int NS>
static int sell_backward_w(const rbg_graph *g, const float *grad_out, float *grad_e0, float *work, int K, hipStream_t s) {
    SellDev *sw = g->sell;
    const int64_t n = g->n_rows, nd = n * NS * W;
    // the incoming gradient is gathered and added where it lies unless the plan has no row-major entries: then it is converted
    // to slabs first
    const bool rm = sw->ent0 != nullptr && opt_sell_rowmajor() && grad_out != grad_e0 && (K < 2 || work);
    const bool fac = sell_factored(sw);
    // Scratch.  With row-major entries the K - 1 slab intermediates alternate between the caller's `work` and grad_e0 itself
    // (the last launch gathers `work` and writes grad_e0 row-major): nothing of the handle is written, so chains on different
    // streams do not meet.  Without them: a per-handle scratch (g, ping, pong), allocated by the first such
    // backward — never inside a capture — and ONE chain at a time on the handle.
    float *gs = nullptr, *ping = nullptr, *pong = nullptr;
    if (!rm) {
        const int rc = sell_scratch(sw, 3 * nd, s);
        if (rc) return rc;  // (RBG_EUNSUPPORTED: the caller runs the binned chain)
        gs = sw->bwd, ping = sw->bwd + nd, pong = sw->bwd + 2 * nd;
    }
    const int n0 = sw->n_class[0];
    if (!rm) {
        const int rc = sell_to_slab<W, NS>(sw, grad_out, grad_out + (int64_t)n0 * NS * W, gs, s);
        if (rc) return rc;
    }
    // dE0 = (g + A (g + A (... (g + A g)))) / (K + 1): K layers, every one adds g in its epilogue; the last divides and
    // writes row-major (A symmetric: rbg_lightgcn_backward_f32 is called with the transposed handles, the handle itself here)
    SellParams base{};
    sell_fill(sw, W, NS, base);
    base.rm[0] = base.prm[0] = grad_out;
    base.rm[1] = base.prm[1] = grad_out + (int64_t)n0 * NS * W;
    base.prev[0] = gs;
    base.denom = (float)(K + 1);
    base.out = grad_e0;
    SellChainLayer lay[RBG_MAX_FUSED_LAYERS + 1] = {};
    const float *x = gs;
    for (int i = 0; i < K; ++i) {
        lay[i].x_rm = (rm && i == 0) ? 1 : 0;
        lay[i].compact = (fac && i > 0) ? 1 : 0;
        lay[i].store_scaled = (fac && i < K - 1) ? 1 : 0;
        lay[i].prev0_rm = rm ? 1 : 0;
        lay[i].xs = x;
        lay[i].n_prev = 1;
        if (i == K - 1) {
            lay[i].last = 1;
        } else {
            if (rm) lay[i].ys = ((K - 2 - i) % 2 == 0) ? work : grad_e0;
            else lay[i].ys = (i & 1) ? pong : ping;
            x = lay[i].ys;
        }
    }
    return sell_run_chain<W, NS>(g, base, lay, K, s);
}

// RBG_EUNSUPPORTED = "not this time" (no scratch yet and the stream is capturing, or the allocation failed): the caller
// runs the binned chain instead.
int sell_backward(const rbg_graph *g, const float *grad_out, float *grad_e0, float *work, int d, int K, hipStream_t s) {
    const int W = g->sell->W;
#define CALL(W_, NS_) sell_backward_w<W_, NS_>(g, grad_out, grad_e0, work, K, s)
    RBG_SELL_DISPATCH(W, d, CALL);
#undef CALL
    return RBG_EUNSUPPORTED;
}

// ---- adoption: validation, derived arrays ---------------------------------------------------------------------------------------
static int sell_validate(const rbg_graph *g, const SellDev *sw) {
    const int n0 = sw->n_class[0], n1 = sw->n_class[1], n_total = sw->n_units[0] + sw->n_units[1], lgw = 64 / (sw->W / 4);
    int *d_err = nullptr;
    if (dev_malloc(&d_err, sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RBG_ENOMEM, "device allocation failed");
    }
    hipError_t ce = hipMemset(d_err, 0, sizeof(int));
    if (n_total) hipLaunchKernelGGL(sell_check_units_kernel, dim3((n_total + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const int4 *>(sw->head),
                                    n_total, sw->n_units[0], n0, n1, sw->n_ent, lgw, sw->n_wide_units[0], sw->n_wide_units[1], d_err);
    if (sw->n_ent) hipLaunchKernelGGL(sell_check_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent), sw->n_ent,
                                      sw->first_ent1, sw->rect ? sw->n_tab : n0, sw->rect ? sw->n_tab : n1, sw->rect ? 2 * sw->W * 4 : sw->W * 4, d_err);
    hipLaunchKernelGGL(sell_check_orig_kernel, dim3((unsigned)((g->n_rows + 255) / 256)), dim3(256), 0, 0, sw->orig, (int)g->n_rows, n0, n0, d_err);
    int h_err = 0;
    if (ce == hipSuccess) ce = hipMemcpy(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_err);
    if (ce != hipSuccess) return fail(RBG_EHIP, "plan validation failed to run: %s", hipGetErrorString(ce));
    if (h_err > 0) return fail(RBG_EINVAL, "SELL plan: unit %d is out of range or misaligned", h_err - 1);
    if (h_err == -1) return fail(RBG_EINVAL, "SELL plan: an entry's column offset is out of range");
    if (h_err == -2) return fail(RBG_EINVAL, "SELL plan: orig[] is out of range or crosses the user / item boundary");
    return RBG_OK;
}

// optional array: a failed allocation leaves the plan without it (and the error state clean)
template <class T>
static bool sell_opt_alloc(T **p, size_t bytes, bool zero) {
    *p = nullptr;
    if (dev_malloc(p, bytes) != hipSuccess || (zero && hipMemset(*p, 0, bytes) != hipSuccess)) {
        (void)hipGetLastError();
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        return false;
    }
    return true;
}

// the launch scratch of the wide rows' units (one 128-float slot and four arrival counters per unit)
static int sell_wide_scratch(SellDev *sw) {
    const int64_t nw = (int64_t)sw->n_wide_units[0] + sw->n_wide_units[1];
    if (!nw) return RBG_OK;
    if (dev_malloc(&sw->wide_part, sizeof(float) * 128 * (size_t)nw) != hipSuccess || dev_malloc(&sw->wide_ctr, sizeof(uint32_t) * 4 * (size_t)nw) != hipSuccess ||
        hipMemset(sw->wide_ctr, 0, sizeof(uint32_t) * 4 * (size_t)nw) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RBG_ENOMEM, "device allocation of the wide rows' scratch (%lld units) failed", (long long)nw);
    }
    return RBG_OK;
}

int sell_adopt(rbg_graph *g, SellDev *sw, bool validate) {
    int rc = RBG_OK;
    if (!sw->native) {  // an attached plan: count its wide units (the planner knows its own)
        int *d_nw = nullptr, h_nw[2] = {0, 0};
        const int n_total = sw->n_units[0] + sw->n_units[1];
        if (dev_malloc(&d_nw, sizeof(int) * 2) != hipSuccess || hipMemset(d_nw, 0, sizeof(int) * 2) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(RBG_ENOMEM, "device allocation failed");
        } else {
            if (n_total) hipLaunchKernelGGL(sell_count_wide_kernel, dim3((n_total + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const int4 *>(sw->head),
                                            n_total, sw->n_units[0], d_nw);
            if (hipMemcpy(h_nw, d_nw, sizeof(h_nw), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RBG_EHIP, "reading the plan failed");
            sw->n_wide_units[0] = h_nw[0], sw->n_wide_units[1] = h_nw[1];
        }
        if (d_nw) (void)hipFree(d_nw);
    }
    if (rc == RBG_OK && validate) rc = sell_validate(g, sw);
    if (rc == RBG_OK) rc = sell_wide_scratch(sw);
    if (rc == RBG_OK) rc = rbg_graph_detach_sell(g);
    if (rc) {
        free_sell(sw);
        return rc;
    }
    const int n0 = sw->n_class[0], n1 = sw->n_class[1], W = sw->W;
    const int64_t n_ent = sw->n_ent;
    if (sw->rect) {  // row-major offsets already: the entries are their own row-major twin; no compact / 16-bit forms (no slab chain)
        sw->ent0 = sw->ent;
        g->sell = sw;
        return RBG_OK;
    }
    // the offsets column alone (the factored chain's launches read 4 bytes per entry)
    if (sell_opt_alloc(&sw->entc, sizeof(int32_t) * (size_t)(n_ent + 256), true) && n_ent)
        hipLaunchKernelGGL(sell_compact_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent), sw->entc, n_ent);
    // r05: the same column as 16-bit slab-row numbers; the rows of class c use them when class 1 - c has < 65 535 rows (Gowalla,
    // Yelp2018: both classes; Amazon-Book: the item rows, which gather the 52 644 user rows)
    if (sw->entc && std::min(n0, n1) < 65535 && sell_opt_alloc(&sw->entc16, sizeof(uint16_t) * (size_t)(n_ent + 512), true))
        hipLaunchKernelGGL(sell_compact16_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent), sw->entc16, n_ent, n_ent + 512,
                           W == 64 ? 8 : 7);
    // the row-major twin of the entries (used under option 'sell_rowmajor', default 1; without it E0 is converted to slabs per
    // propagation).  A failed allocation leaves the plan without the twin — and touches nothing else (r03: this branch freed
    // entc and rs without clearing them)
    if ((int64_t)std::max(n0, n1) * 2 * W * 4 < kSellPast && sell_opt_alloc(&sw->ent0, sizeof(int32_t) * 2 * (size_t)(n_ent + 128), true) && n_ent)
        hipLaunchKernelGGL(sell_first_entries_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const int2 *>(sw->ent),
                           reinterpret_cast<int2 *>(sw->ent0), n_ent, sw->first_ent1, sw->orig, n0, W);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
        free_sell(sw);
        return fail(RBG_EHIP, "building the derived arrays of the SELL plan failed");
    }
    g->sell = sw;
    return RBG_OK;
}

int sell_set_factors(rbg_graph *g, const float *r) {
    if (!g || !g->sell) return fail(RBG_EINVAL, "no SELL plan attached");
    if (!r) return fail(RBG_EINVAL, "r is NULL");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    SellDev *sw = g->sell;
    if (sw->borrowed) return fail(RBG_EUNSUPPORTED, "a re-weighted view has no row factors");
    if (!sw->entc) return fail(RBG_EUNSUPPORTED, "the plan has no compact entries");
    const int n = (int)g->n_rows, n_total = sw->n_units[0] + sw->n_units[1], lgw = 64 / (sw->W / 4);
    (void)hipDeviceSynchronize();  // (not concurrently with launches on this handle)
    if (sw->rs) (void)hipFree(sw->rs);
    sw->rs = sw->irs = nullptr;
    float *buf = nullptr;
    int *d_err = nullptr;
    if (dev_malloc(&buf, sizeof(float) * 2 * (size_t)n) != hipSuccess || dev_malloc(&d_err, sizeof(int)) != hipSuccess) {
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

// A re-weighted view of a planned graph: the structure (units, offsets, numbering) is the base's; the view owns a copy of the
// row-major entries whose values rbg_graph_refresh_values rewrites from the caller's array through src[].
int sell_make_view(rbg_graph *view, const rbg_graph *base) {
    const SellDev *b = base->sell;
    if (!b || b->borrowed || !b->ent0 || !b->src) return RBG_EUNSUPPORTED;
    SellDev *sw = new (std::nothrow) SellDev();
    if (!sw) return fail(RBG_ENOMEM, "out of host memory");
    sw->borrowed = b;
    sw->rect = b->rect;
    sw->n_tab = b->n_tab;
    sw->W = b->W;
    sw->chunk = b->chunk;
    sw->n_ent = b->n_ent;
    sw->first_ent1 = b->first_ent1;
    for (int c = 0; c < 2; ++c)
        sw->unit_base[c] = b->unit_base[c], sw->n_units[c] = b->n_units[c], sw->n_class[c] = b->n_class[c];
    sw->ent = b->ent, sw->entc = b->entc, sw->head = b->head, sw->orig = b->orig, sw->src = b->src;
    sw->n_wide_units[0] = b->n_wide_units[0], sw->n_wide_units[1] = b->n_wide_units[1];
    if (sell_wide_scratch(sw) != RBG_OK) {  // (the view's own launch scratch: it may run beside its base graph)
        clear_error();
        free_sell(sw);
        return RBG_EUNSUPPORTED;
    }
    const size_t bytes = sizeof(int32_t) * 2 * (size_t)(b->n_ent + 128);
    if (dev_malloc(&sw->ent0, bytes) != hipSuccess || hipMemcpy(sw->ent0, b->ent0, bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
        (void)hipGetLastError();
        free_sell(sw);
        return RBG_EUNSUPPORTED;  // (the view keeps the binned kernel)
    }
    view->sell = sw;
    base->sell_views.fetch_add(1);  // (released by the view's destruction: graph_build.cpp free_device)
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_graph_attach_sell(rbg_graph *g, int W, const int32_t *ent, int64_t n_ent, const int32_t *head, const int32_t *unit_base,
                          const int32_t *n_units, const int32_t *orig) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "a SELL plan needs a device graph");
    if (g->base) return fail(RBG_EUNSUPPORTED, "a re-weighted view cannot carry a SELL plan of its own (it borrows its base graph's)");
    if (g->sell && !g->sell->borrowed && g->sell_views.load() > 0)
        return fail(RBG_EUNSUPPORTED, "%d re-weighted view(s) borrow this handle's column-slab plan: destroy them before attaching another", g->sell_views.load());
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
    int64_t first_ent1 = n_ent;
    if (n_units[1] > 0) {
        int4 h1;
        if (hipMemcpy(&h1, reinterpret_cast<const int4 *>(head) + n_units[0], sizeof(int4), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(RBG_EHIP, "reading the plan failed");
        first_ent1 = h1.x;
    }
    // ---- a copy of the caller's arrays, validated on the device before it is adopted ------------------------------------------------
    SellDev *sw = new (std::nothrow) SellDev();
    if (!sw) return fail(RBG_ENOMEM, "out of host memory");
    sw->W = W;
    sw->n_ent = n_ent;
    sw->first_ent1 = first_ent1;
    for (int c = 0; c < 2; ++c) {
        sw->unit_base[c] = unit_base[c];
        sw->n_units[c] = n_units[c];
    }
    sw->n_class[0] = n0;
    sw->n_class[1] = n1;
    const size_t ent_bytes = sizeof(int32_t) * 2 * (size_t)(n_ent + 128), head_bytes = sizeof(int32_t) * 4 * (size_t)std::max(n_total, 1);
    bool ok = dev_malloc(&sw->ent, ent_bytes) == hipSuccess && dev_malloc(&sw->head, head_bytes) == hipSuccess &&
              dev_malloc(&sw->orig, sizeof(int32_t) * (size_t)g->n_rows) == hipSuccess;
    ok = ok && hipMemset(sw->ent, 0, ent_bytes) == hipSuccess;  // (the 128 entries of slack a wave's last 16-byte loads may touch)
    ok = ok && hipMemcpy(sw->ent, ent, sizeof(int32_t) * 2 * (size_t)n_ent, hipMemcpyDeviceToDevice) == hipSuccess;
    ok = ok && (n_total == 0 || hipMemcpy(sw->head, head, sizeof(int32_t) * 4 * (size_t)n_total, hipMemcpyDeviceToDevice) == hipSuccess);
    ok = ok && hipMemcpy(sw->orig, orig, sizeof(int32_t) * (size_t)g->n_rows, hipMemcpyDeviceToDevice) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        free_sell(sw);
        return fail(RBG_ENOMEM, "device allocation / copy of the SELL plan failed");
    }
    return sell_adopt(g, sw, true);
}

int rbg_graph_sell_set_factors(rbg_graph *g, const float *r) {
    clear_error();
    return sell_set_factors(g, r);
}

int rbg_graph_detach_sell(rbg_graph *g) {
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    // re-weighted views hold raw pointers into this plan (ent, entc, head, orig, src): it stays until they are destroyed
    if (g->sell && !g->sell->borrowed && g->sell_views.load() > 0)
        return fail(RBG_EUNSUPPORTED, "%d re-weighted view(s) borrow this handle's column-slab plan: destroy them before detaching or re-planning",
                    g->sell_views.load());
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

int rbg_graph_has_sell(const rbg_graph *g, int d) { return (sell_usable(g) && sell_width_ok(g->sell, d)) ? 1 : 0; }

int rbg_graph_sell_info(const rbg_graph *g, int *W, int *chunk, int64_t *n_ent, int32_t *n_units, int *factored, int *rowmajor) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (!g->sell) return fail(RBG_EINVAL, "no SELL plan on this handle");
    const SellDev *sw = g->sell;
    if (W) *W = sw->W;
    if (chunk) *chunk = sw->chunk;
    if (n_ent) *n_ent = sw->n_ent;
    if (n_units) n_units[0] = sw->n_units[0], n_units[1] = sw->n_units[1];
    if (factored) *factored = (sw->rs && sw->entc && !sw->borrowed) ? 1 : 0;
    if (rowmajor) *rowmajor = sw->ent0 ? 1 : 0;
    return RBG_OK;
}

int rbg_graph_sell_arrays(const rbg_graph *g, const int32_t **ent, const int32_t **head, const int32_t **orig, const float **factors,
                          const int32_t **src) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (!g->sell) return fail(RBG_EINVAL, "no SELL plan on this handle");
    if (ent) *ent = g->sell->ent;
    if (head) *head = g->sell->head;
    if (orig) *orig = g->sell->orig;
    if (factors) *factors = g->sell->rs;
    if (src) *src = g->sell->src;
    return RBG_OK;
}

int rbg_graph_refresh_values(rbg_graph *g, void *stream) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (!g->base) return fail(RBG_EINVAL, "rbg_graph_refresh_values: not a re-weighted view");
    SellDev *sw = g->sell;
    if (!sw || !sw->borrowed) return RBG_OK;  // no plan on the base graph: the view's launches read the caller's array directly
    int rc = set_device_for(g->device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (sw->n_ent) {
        hipLaunchKernelGGL(sell_refresh_values_kernel, dim3((unsigned)std::min<int64_t>((sw->n_ent + 255) / 256, 16384)), dim3(256), 0, s,
                           reinterpret_cast<int2 *>(sw->ent0), sw->src, g->d_val, sw->n_ent);
        RBG_HIP(hipGetLastError());
    }
    sw->view_fresh = true;
    return RBG_OK;
}

}  // extern "C"
