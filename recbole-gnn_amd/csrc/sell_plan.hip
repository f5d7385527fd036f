// sell_plan.hip — the planner of the column-slab propagation (csrc/sell.hip), on the device (r04).
//
// rbg_graph_plan_sell(g, W, chunk) cuts the handle's normalized CSR (the product of get_norm_adj_mat, dataset.py:49-79, or of
// an SGL view rebuild, sgl.py:107-126) into the SELL-C-sigma form sell_spmm_kernel reads, entirely in HBM: rocPRIM sorts and
// scans plus a few one-pass kernels; the host sees two small result blocks (segment counts, entry totals).  It is called by
// rbg_graph_create* itself (option "sell_auto", default 1) for every device graph with a user / item boundary, so a caller that
// binds the C ABI alone (rbg_graph_create -> rbg_lightgcn_forward_f32) runs the column-slab kernel.
//
// The layout is specified by tests/sell_spec.py (torch ops; kept as the executable specification: the tests compare this
// planner's arrays with it bit for bit).  Per row class (user rows / item rows):
//   1. rows sorted by degree, descending, stable (= (parts, degree) descending: parts is monotone in the degree)
//      -> the plan's row numbering `orig`; rocPRIM radix_sort_pairs_desc
//   2. every entry (row, col) -> key (internal row, internal column of the OTHER class), payload = its CSR position; one
//      radix_sort_pairs over the class's entries: a row's entries by ascending internal column
//   3. rows with equal `parts` form segments (counts by binary search in the sorted degrees: the one host round trip); a
//      segment's units are consecutive groups of LGW / parts rows; a WIDE row (more than chunk LGW entries) has
//      U = ceil(degree / (chunk LGW)) units of its own (r06: any number — their partial sums meet in the plan's scratch and the
//      last to arrive adds them in unit order; r03-r05: exactly four, one workgroup, and longer hub rows were refused); the
//      first unit of every wide row comes from an exclusive scan of U over the class's wide rows (`woff`)
//   4. per unit: longest piece -> slots (rounded to 2); exclusive scan -> the unit's first entry
//   5. per sorted entry: (unit, lane-group, batch, slot) in closed form -> ent[pos] = {column offset, val}, src[pos] = CSR position
// No step is proportional to N or nnz on the host; at the config-#5 shape (15 M rows, 400 M entries) the temporaries are
// two 2.4 GB key / payload double buffers per class, freed before the next class is cut.
//
// r06 — the RECTANGULAR form (plan_sell(..., rect = true); SellDev::rect): a block whose rows are cut into the two classes at
// row_split but whose columns index ONE row-major table of n_cols rows — a shard's [owned | halo] product, or its halo block
// alone (sharded.py / shard.hip: rows = the rank's nodes, the same operator as layers.py:19-20 restricted to them).  Same
// steps; the sort key's column is the table row itself, the entries hold row-major offsets (col * 2 W * 4) and there is no
// slab chain: every launch gathers the table where it lies.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <cstring>
#include <new>

#include <rocprim/rocprim.hpp>

#include "internal.h"

namespace rbg {

namespace {

constexpr int kPast = 0x7ffffff0;  // = kSellPast (sell.hip)
constexpr int kMaxSegs = 6;        // wide + parts LGW, LGW / 2, ..., 1 (LGW <= 16)
constexpr int kMaxUnits = 32767;   // sell_spec.py MAX_UNITS: units of one wide row (15 bits of the header)

struct Seg {
    int32_t pp;      // pieces per row (4 LGW for wide rows)
    int32_t lp;      // log2(pieces of a row inside one unit)
    int32_t row_b, row_e;
    int32_t unit_b;  // first unit (class-local)
    int32_t per;     // rows per unit (wide: 1 row = 4 units)
    int32_t wide;
    int32_t pad;
};
struct ClassSegs {
    int32_t n, n_units;
    int32_t n_wide_rows, n_wide_units;  // the wide segment (always s[0] when there is one)
    const int32_t *woff;                // [n_wide_rows + 1]: first unit of wide row r (class-local)
    int32_t cw;                         // chunk * lgw: entries per wide unit (at most)
    int32_t pad;
    Seg s[kMaxSegs];
};

struct Buf {  // frees on scope exit
    void *p = nullptr;
    ~Buf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    int alloc(size_t bytes) {
        release();
        if (dev_malloc(&p, bytes ? bytes : 1) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(RBG_ENOMEM, "SELL planner: device allocation of %zu bytes failed", bytes);
        }
        return RBG_OK;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

int bits_for(int64_t n) {  // bits that hold every value in [0, n)
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}

__global__ void plan_degree_kernel(const int32_t *__restrict__ rowptr, int base, int n, int32_t *__restrict__ deg, uint32_t *__restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    deg[i] = rowptr[base + i + 1] - rowptr[base + i];
    ids[i] = (uint32_t)i;
}

__global__ void plan_inverse_kernel(const uint32_t *__restrict__ order, int n, int base, int32_t *__restrict__ inv, int32_t *__restrict__ orig) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    inv[order[i]] = i;
    orig[base + i] = (int32_t)order[i] + base;
}

// rdeg: the class's degrees, descending.  out[j] = rows with degree > thr[j] (j < n_thr); out[n_thr] = the largest degree;
// out[n_thr + 1] = rowptr[split] (the first entry of class 1).
struct Thresholds {
    int32_t n;
    int32_t t[kMaxSegs];
};
__global__ void plan_count_kernel(const int32_t *__restrict__ rdeg, int n, Thresholds thr, const int32_t *__restrict__ rowptr, int split,
                                  int32_t *__restrict__ out) {
    const int j = threadIdx.x;
    if (j < thr.n) {
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rdeg[mid] > thr.t[j]) lo = mid + 1; else hi = mid;
        }
        out[j] = lo;
    } else if (j == thr.n) {
        out[j] = n > 0 ? rdeg[0] : 0;
    } else if (j == thr.n + 1) {
        out[j] = rowptr[split];
    }
}

__device__ __forceinline__ int seg_of_unit(const ClassSegs &segs, int u) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < kMaxSegs; ++i)
        if (i < segs.n && u >= segs.s[i].unit_b) q = i;
    return q;
}
__device__ __forceinline__ int seg_of_row(const ClassSegs &segs, int r) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < kMaxSegs; ++i)
        if (i < segs.n && r >= segs.s[i].row_b) q = i;
    return q;
}

// one thread per unit: header {0, first row, slots << 16, log2(parts) | rows << 8 | wide << 16} and the unit's entry count
__global__ void plan_units_kernel(const ClassSegs segs, int lgw, const int32_t *__restrict__ rdeg, int4 *__restrict__ head,
                                  int64_t *__restrict__ slots, int *__restrict__ max_nc) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= segs.n_units) return;
    Seg sg = segs.s[seg_of_unit(segs, u)];
    int row0, nrows, pbase, wj = 0, wu = 0;
    if (sg.wide) {  // the row whose units include u: woff[row] <= u < woff[row + 1]
        int lo = 0, hi = segs.n_wide_rows;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (segs.woff[mid + 1] <= u) lo = mid + 1; else hi = mid;
        }
        row0 = lo;
        nrows = 1;
        wj = u - segs.woff[lo];
        wu = segs.woff[lo + 1] - segs.woff[lo];
        pbase = wj * lgw;
        sg.pp = wu * lgw;
    } else {
        row0 = sg.row_b + (u - sg.unit_b) * sg.per;
        nrows = min(sg.per, sg.row_e - row0);
        pbase = 0;
    }
    int mx = 0;
    for (int lg = 0; lg < lgw; ++lg) {
        const int sub = lg >> sg.lp;
        if (sub >= nrows) continue;
        const int64_t dg = rdeg[row0 + sub], part = pbase + (lg & ((1 << sg.lp) - 1));
        mx = max(mx, (int)(dg * (part + 1) / sg.pp - dg * part / sg.pp));
    }
    const int nc = (mx + 1) / 2 * 2;
    head[u] = make_int4(0, row0, (int)(((unsigned)nc << 16) | (unsigned)wj), sg.lp | (nrows << 8) | (sg.wide << 16) | (wu << 17));
    slots[u] = (int64_t)lgw * nc;
    if (nc > 32767) atomicMax(max_nc, nc);  // (rare: only then is the atomic worth issuing)
}

// U of every wide row (rdeg: the class's degrees, descending; the first n_wide are the wide rows)
__global__ void plan_wide_units_kernel(const int32_t *__restrict__ rdeg, int n_wide, int cw, int32_t *__restrict__ u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_wide) u[i] = (int32_t)(((int64_t)rdeg[i] + cw - 1) / cw);
    else if (i == n_wide) u[i] = 0;
}

__global__ void plan_head_offsets_kernel(int4 *__restrict__ head, const int64_t *__restrict__ uoff, int n_units) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_units) head[u].x = (int)uoff[u];
}

__global__ void plan_fill_kernel(int2 *__restrict__ ent, int32_t *__restrict__ src, int64_t n_ent, int64_t n_alloc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_alloc; i += (int64_t)gridDim.x * blockDim.x) {
        ent[i] = i < n_ent ? make_int2(kPast, 0) : make_int2(0, 0);
        if (i < n_ent) src[i] = -1;
    }
}

// one thread per CSR entry of the class: key = (internal row << shift) | internal column, payload = the entry's CSR position
__global__ void plan_keys_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t e_b, int64_t n_e, int row_b,
                                 int n_c, int obase, int n_o, const int32_t *__restrict__ inv_c, const int32_t *__restrict__ inv_o, int shift,
                                 unsigned long long *__restrict__ keys, uint32_t *__restrict__ pay, int *__restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_e; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e_b + i;
        int lo = row_b, hi = row_b + n_c;  // the row that holds entry e: rowptr[r] <= e < rowptr[r + 1]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rowptr[mid + 1] <= e) lo = mid + 1; else hi = mid;
        }
        int c = col[e] - obase;
        if (c < 0 || c >= n_o) {  // a user row that lists a user (or an item row an item): not the bipartite adjacency (rect: out of the table)
            atomicExch(bad, 1);
            c = 0;
        }
        keys[i] = ((unsigned long long)(uint32_t)inv_c[lo - row_b] << shift) | (unsigned long long)(uint32_t)(inv_o ? inv_o[c] : c);
        pay[i] = (uint32_t)e;
    }
}

// one thread per sorted entry: its slot in closed form (sell_spec.py: pos = u_off + LGW 8 k + lg sb + j)
__global__ void plan_scatter_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ pay, int64_t n_e, int shift,
                                    const ClassSegs segs, int lgw, const int32_t *__restrict__ ptr, const int32_t *__restrict__ rdeg,
                                    const int4 *__restrict__ head, int stride, const float *__restrict__ val, int2 *__restrict__ ent,
                                    int32_t *__restrict__ src) {
    const unsigned long long mask = (1ull << shift) - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_e; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        const int ir = (int)(key >> shift), ci = (int)(key & mask);
        const int64_t t = i - ptr[ir], dg = rdeg[ir];
        Seg sg = segs.s[seg_of_row(segs, ir)];
        if (sg.wide) sg.pp = (segs.woff[ir + 1] - segs.woff[ir]) * lgw;
        const int64_t q = ((t + 1) * sg.pp + dg - 1) / dg - 1;  // the piece that holds entry t: floor(dg q / pp) <= t < floor(dg (q + 1) / pp)
        int u, lg;
        if (sg.wide) {
            u = segs.woff[ir] + (int)(q / lgw);
            lg = (int)(q % lgw);
        } else {
            u = sg.unit_b + (ir - sg.row_b) / sg.per;
            lg = ((ir - sg.row_b) % sg.per) * sg.pp + (int)q;
        }
        const int4 h = head[u];
        const int nc = (int)((unsigned)h.z >> 16);
        const int i_sec = (int)(t - dg * q / sg.pp), k = i_sec >> 3, j = i_sec & 7;
        const int sb = min(8, nc - 8 * k);
        const int64_t pos = (int64_t)h.x + (int64_t)lgw * 8 * k + lg * sb + j;
        const uint32_t e = pay[i];
        ent[pos] = make_int2(ci * stride, __float_as_int(val[e]));  // stride = W 4 (a slab row) or 2 W 4 (rect: a row-major table row)
        src[pos] = (int32_t)e;
    }
}

// r = deg^-1/2 in the plan's numbering (0 for an empty row): the factors of the symmetric normalisation (dataset.py:41-79)
__global__ void plan_factors_kernel(const int32_t *__restrict__ rdeg, int n, float *__restrict__ r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = rdeg[i] > 0 ? (float)(1.0 / sqrt((double)rdeg[i])) : 0.f;
}

unsigned grid_for_n(int64_t n, int cap = 16384) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, cap)); }

}  // namespace

// RBG_EUNSUPPORTED = the graph is outside what the slab path serves (g->sell_note says why); the caller keeps the binned kernel.
int plan_sell(rbg_graph *g, int W, int chunk, bool rect) {
    auto na = [&](const char *why) {
        g->sell_note = why;
        return fail(RBG_EUNSUPPORTED, "SELL plan not applicable: %s", why);
    };
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "a SELL plan needs a device graph");
    if (g && g->sell && !g->sell->borrowed && g->sell_views.load() > 0)
        return fail(RBG_EUNSUPPORTED, "%d re-weighted view(s) borrow this handle's column-slab plan: destroy them before re-planning", g->sell_views.load());
    if (W != 32 && W != 64) return fail(RBG_EINVAL, "W = %d (32 or 64)", W);
    if (chunk == 0) chunk = 128;  // sell_spec.py CHUNK
    if (chunk < 1 || chunk > (1 << 20)) return fail(RBG_EINVAL, "chunk = %d", chunk);
    if (g->base) return na("a re-weighted view borrows its base graph's plan (rbg_graph_refresh_values)");
    const int64_t split = g->n_users >= 0 ? g->n_users : g->row_split;
    if (rect) {
        if (split <= 0 || split >= g->n_rows) return na("a rectangular block without two row classes (rbg_graph_create_csr_classes)");
        if (g->n_cols <= 0 || g->n_cols > INT32_MAX) return na("no columns, or more than 2^31");
        if (W != 32) return na("rectangular blocks are planned at W = 32");
    } else if (g->n_rows != g->n_cols || split <= 0 || split >= g->n_rows) {
        return na("no user / item boundary (a square bipartite graph is needed)");
    }
    if (g->n_rows > INT32_MAX) return na("more than 2^31 rows");
    const int n[2] = {(int)split, (int)(g->n_rows - split)}, base[2] = {0, (int)split};
    const int lgw = 64 / (W / 4), lp_full = lgw == 8 ? 3 : (lgw == 4 ? 2 : 4);
    if (!rect && (int64_t)std::max(n[0], n[1]) * W * 4 >= kPast) return na("table too large for 32-bit slab offsets");
    if (rect && g->n_cols * 2 * W * 4 >= kPast) return na("table too large for 32-bit row offsets");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    hipStream_t s = nullptr;
    const int N = (int)g->n_rows;

    // ---- 1. rows by degree, descending -----------------------------------------------------------------------------------------
    Buf b_deg, b_ids, b_rdeg, b_order, b_inv, b_ptr, b_info, b_tmp, b_orig;
    if ((rc = b_deg.alloc(sizeof(int32_t) * (size_t)N)) || (rc = b_ids.alloc(sizeof(uint32_t) * (size_t)N)) ||
        (rc = b_rdeg.alloc(sizeof(int32_t) * ((size_t)N + 2))) || (rc = b_order.alloc(sizeof(uint32_t) * (size_t)N)) ||
        (rc = b_inv.alloc(sizeof(int32_t) * (size_t)N)) || (rc = b_ptr.alloc(sizeof(int32_t) * ((size_t)N + 2))) ||
        (rc = b_info.alloc(sizeof(int32_t) * 64)) || (rc = b_orig.alloc(sizeof(int32_t) * (size_t)N)))
        return rc;
    // class c's sorted degrees live at rdeg + rb[c] with one zero after them (the scan's last element), its ptr at ptr + rb[c]
    const int rb[2] = {0, n[0] + 1};
    RBG_HIP(hipMemsetAsync(b_rdeg.p, 0, sizeof(int32_t) * ((size_t)N + 2), s));
    size_t tmp_bytes = 0;
    for (int c = 0; c < 2; ++c) {
        size_t t1 = 0, t2 = 0;
        RBG_HIP(rocprim::radix_sort_pairs_desc(nullptr, t1, b_deg.as<unsigned int>(), b_rdeg.as<unsigned int>(), b_ids.as<uint32_t>(),
                                               b_order.as<uint32_t>(), (size_t)n[c], 0, 32, s));
        RBG_HIP(rocprim::exclusive_scan(nullptr, t2, b_rdeg.as<int32_t>(), b_ptr.as<int32_t>(), 0, (size_t)n[c] + 1, rocprim::plus<int32_t>(), s));
        tmp_bytes = std::max(tmp_bytes, std::max(t1, t2));
    }
    if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
    Thresholds thr{};
    for (int64_t t = (int64_t)chunk * lgw; thr.n < kMaxSegs && t >= chunk; t >>= 1) thr.t[thr.n++] = (int32_t)std::min<int64_t>(t, INT32_MAX);
    // (chunk * lgw, chunk * lgw / 2, ..., chunk: lp_full + 1 thresholds)
    for (int c = 0; c < 2; ++c) {
        int32_t *deg = b_deg.as<int32_t>() + base[c], *rdeg = b_rdeg.as<int32_t>() + rb[c];
        uint32_t *ids = b_ids.as<uint32_t>() + base[c], *order = b_order.as<uint32_t>() + base[c];
        hipLaunchKernelGGL(plan_degree_kernel, dim3((n[c] + 255) / 256), dim3(256), 0, s, g->d_rowptr, base[c], n[c], deg, ids);
        size_t tb = tmp_bytes;
        RBG_HIP(rocprim::radix_sort_pairs_desc(b_tmp.p, tb, reinterpret_cast<unsigned int *>(deg), reinterpret_cast<unsigned int *>(rdeg), ids, order,
                                               (size_t)n[c], 0, 32, s));
        hipLaunchKernelGGL(plan_inverse_kernel, dim3((n[c] + 255) / 256), dim3(256), 0, s, order, n[c], base[c], b_inv.as<int32_t>() + base[c],
                           b_orig.as<int32_t>());
        tb = tmp_bytes;
        RBG_HIP(rocprim::exclusive_scan(b_tmp.p, tb, rdeg, b_ptr.as<int32_t>() + rb[c], 0, (size_t)n[c] + 1, rocprim::plus<int32_t>(), s));
        hipLaunchKernelGGL(plan_count_kernel, dim3(1), dim3(64), 0, s, rdeg, n[c], thr, g->d_rowptr, (int)split, b_info.as<int32_t>() + 16 * c);
    }
    RBG_HIP(hipGetLastError());
    int32_t info[32];
    RBG_HIP(hipMemcpyAsync(info, b_info.p, sizeof(info), hipMemcpyDeviceToHost, s));
    RBG_HIP(hipStreamSynchronize(s));
    b_deg.release();
    b_ids.release();
    b_order.release();

    // ---- 2. segments and units (host: a handful of integers) -------------------------------------------------------------------
    const int64_t max_deg = std::max(info[thr.n], info[16 + thr.n]);
    const int64_t cw = (int64_t)chunk * lgw;  // entries per wide unit (at most)
    if (max_deg > kMaxUnits * cw) {
        char why[160];
        snprintf(why, sizeof why, "a row of %lld entries is longer than %d units of %lld", (long long)max_deg, kMaxUnits, (long long)cw);
        return na(why);
    }
    // the wide rows' units: U = ceil(degree / cw) per row, exclusive scan -> woff (device); the totals come back with one copy
    Buf b_woff[2], b_wu;
    int32_t n_wide_units[2] = {0, 0};
    {
        const int nw_max = std::max(info[0], info[16]);
        if ((rc = b_wu.alloc(sizeof(int32_t) * ((size_t)nw_max + 1)))) return rc;
        for (int c = 0; c < 2; ++c) {
            const int nw = info[16 * c];
            if ((rc = b_woff[c].alloc(sizeof(int32_t) * ((size_t)nw + 1)))) return rc;
            hipLaunchKernelGGL(plan_wide_units_kernel, dim3((nw + 256) / 256), dim3(256), 0, s, b_rdeg.as<int32_t>() + rb[c], nw, (int)cw, b_wu.as<int32_t>());
            size_t tb = 0;
            RBG_HIP(rocprim::exclusive_scan(nullptr, tb, b_wu.as<int32_t>(), b_woff[c].as<int32_t>(), 0, (size_t)nw + 1, rocprim::plus<int32_t>(), s));
            if (tb > tmp_bytes) {
                RBG_HIP(hipStreamSynchronize(s));
                if ((rc = b_tmp.alloc(tb))) return rc;
                tmp_bytes = tb;
            }
            RBG_HIP(rocprim::exclusive_scan(b_tmp.p, tb, b_wu.as<int32_t>(), b_woff[c].as<int32_t>(), 0, (size_t)nw + 1, rocprim::plus<int32_t>(), s));
            RBG_HIP(hipMemcpyAsync(&n_wide_units[c], b_woff[c].as<int32_t>() + nw, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            RBG_HIP(hipStreamSynchronize(s));  // (b_wu is reused by the next class)
        }
    }
    const int64_t ent_split = info[thr.n + 1];  // rowptr[split]
    ClassSegs segs[2] = {};
    for (int c = 0; c < 2; ++c) {
        const int32_t *cnt = info + 16 * c;
        int row = 0, unit = 0;
        auto push = [&](int pp, int lp, int per, int wide, int row_e) {
            if (row_e <= row) return;
            Seg &sg = segs[c].s[segs[c].n++];
            sg.pp = pp, sg.lp = lp, sg.row_b = row, sg.row_e = row_e, sg.unit_b = unit, sg.per = per, sg.wide = wide;
            unit += (row_e - row + per - 1) / per;
            row = row_e;
        };
        segs[c].n_wide_rows = cnt[0];
        segs[c].n_wide_units = n_wide_units[c];
        segs[c].woff = b_woff[c].as<int32_t>();
        segs[c].cw = (int32_t)cw;
        if (cnt[0] > 0) {  // degree > chunk lgw: U units of lgw pieces per row (pp is the row's own: U lgw, set where a row is looked at)
            Seg &sg = segs[c].s[segs[c].n++];
            sg.pp = 0, sg.lp = lp_full, sg.row_b = 0, sg.row_e = cnt[0], sg.unit_b = 0, sg.per = 1, sg.wide = 1;
            unit = n_wide_units[c];
            row = cnt[0];
        }
        for (int j = 1; j < thr.n; ++j) push(lgw >> (j - 1), lp_full - (j - 1), 1 << (j - 1), 0, cnt[j]);  // parts = lgw >> (j - 1)
        push(1, 0, lgw, 0, n[c]);                                                 // degree <= chunk: whole rows
        segs[c].n_units = unit;
    }
    const int n_units[2] = {segs[0].n_units, segs[1].n_units}, n_total = n_units[0] + n_units[1];

    // ---- 3. unit headers and entry offsets ---------------------------------------------------------------------------------------
    SellDev *sw = new (std::nothrow) SellDev();
    if (!sw) return fail(RBG_ENOMEM, "out of host memory");
    struct Guard {  // the plan under construction is freed on every early return
        SellDev *&sw;
        ~Guard() { free_sell(sw); }
    } guard{sw};
    sw->W = W;
    sw->chunk = chunk;
    sw->native = true;
    sw->rect = rect;
    sw->n_tab = rect ? (int32_t)g->n_cols : 0;
    for (int c = 0; c < 2; ++c) {
        sw->unit_base[c] = c ? n_units[0] : 0;
        sw->n_units[c] = n_units[c];
        sw->n_class[c] = n[c];
        sw->n_wide_units[c] = n_wide_units[c];
    }
    sw->orig = b_orig.as<int32_t>();
    b_orig.p = nullptr;  // (owned by the plan from here on)
    Buf b_slots, b_uoff, b_flag;
    if (dev_malloc(&sw->head, sizeof(int32_t) * 4 * (size_t)std::max(n_total, 1)) != hipSuccess) return fail(RBG_ENOMEM, "SELL planner: unit headers");
    if ((rc = b_slots.alloc(sizeof(int64_t) * ((size_t)n_total + 1))) || (rc = b_uoff.alloc(sizeof(int64_t) * ((size_t)n_total + 1))) ||
        (rc = b_flag.alloc(sizeof(int) * 2)))
        return rc;
    RBG_HIP(hipMemsetAsync(b_slots.p, 0, sizeof(int64_t) * ((size_t)n_total + 1), s));
    RBG_HIP(hipMemsetAsync(b_flag.p, 0, sizeof(int) * 2, s));
    int4 *head = reinterpret_cast<int4 *>(sw->head);
    for (int c = 0; c < 2; ++c)
        if (n_units[c])
            hipLaunchKernelGGL(plan_units_kernel, dim3((n_units[c] + 255) / 256), dim3(256), 0, s, segs[c], lgw, b_rdeg.as<int32_t>() + rb[c],
                               head + sw->unit_base[c], b_slots.as<int64_t>() + sw->unit_base[c], b_flag.as<int>());
    {
        size_t t3 = 0;
        RBG_HIP(rocprim::exclusive_scan(nullptr, t3, b_slots.as<int64_t>(), b_uoff.as<int64_t>(), (int64_t)0, (size_t)n_total + 1,
                                        rocprim::plus<int64_t>(), s));
        if (t3 > tmp_bytes) {
            if ((rc = b_tmp.alloc(t3))) return rc;
            tmp_bytes = t3;
        }
        RBG_HIP(rocprim::exclusive_scan(b_tmp.p, t3, b_slots.as<int64_t>(), b_uoff.as<int64_t>(), (int64_t)0, (size_t)n_total + 1,
                                        rocprim::plus<int64_t>(), s));
    }
    if (n_total) hipLaunchKernelGGL(plan_head_offsets_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s, head, b_uoff.as<int64_t>(), n_total);
    RBG_HIP(hipGetLastError());
    int64_t tot[2] = {0, 0};
    int flag[2] = {0, 0};
    RBG_HIP(hipMemcpyAsync(&tot[0], b_uoff.as<int64_t>() + n_units[0], sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RBG_HIP(hipMemcpyAsync(&tot[1], b_uoff.as<int64_t>() + n_total, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RBG_HIP(hipMemcpyAsync(flag, b_flag.p, sizeof(flag), hipMemcpyDeviceToHost, s));
    RBG_HIP(hipStreamSynchronize(s));
    b_slots.release();
    b_uoff.release();
    const int64_t n_ent = tot[1];
    if (n_ent >= ((int64_t)1 << 31) - 256) return na("more than 2^31 plan entries");
    if (flag[0] > 65534) return na("a unit of more than 65 534 slots");
    sw->n_ent = n_ent;
    sw->first_ent1 = tot[0];

    // ---- 4. entries ------------------------------------------------------------------------------------------------------------
    const int64_t n_alloc = n_ent + 128;  // (the slack a wave's last 16-byte loads may touch)
    if (dev_malloc(&sw->ent, sizeof(int32_t) * 2 * (size_t)n_alloc) != hipSuccess ||
        dev_malloc(&sw->src, sizeof(int32_t) * (size_t)std::max<int64_t>(n_ent, 1)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RBG_ENOMEM, "SELL planner: device allocation of the entry array (%lld entries) failed", (long long)n_ent);
    }
    hipLaunchKernelGGL(plan_fill_kernel, dim3(grid_for_n(n_alloc)), dim3(256), 0, s, reinterpret_cast<int2 *>(sw->ent), sw->src, n_ent, n_alloc);
    const int64_t e_b[2] = {0, ent_split}, e_n[2] = {ent_split, g->nnz - ent_split};
    for (int c = 0; c < 2; ++c) {
        if (e_n[c] <= 0) continue;
        const int n_o = rect ? (int)g->n_cols : n[1 - c];  // rect: the key's column is the table row itself
        const int shift = bits_for(std::max(n_o, 2)), key_bits = shift + bits_for(std::max(n[c], 2));
        Buf k_in, k_out, p_in, p_out;
        if ((rc = k_in.alloc(8 * (size_t)e_n[c])) || (rc = k_out.alloc(8 * (size_t)e_n[c])) || (rc = p_in.alloc(4 * (size_t)e_n[c])) ||
            (rc = p_out.alloc(4 * (size_t)e_n[c])))
            return rc;
        hipLaunchKernelGGL(plan_keys_kernel, dim3(grid_for_n(e_n[c])), dim3(256), 0, s, g->d_rowptr, g->d_col, e_b[c], e_n[c], base[c], n[c],
                           rect ? 0 : base[1 - c], n_o, b_inv.as<int32_t>() + base[c], rect ? (const int32_t *)nullptr : b_inv.as<int32_t>() + base[1 - c],
                           shift, k_in.as<unsigned long long>(), p_in.as<uint32_t>(), b_flag.as<int>() + 1);
        size_t t4 = 0;
        RBG_HIP(rocprim::radix_sort_pairs(nullptr, t4, k_in.as<unsigned long long>(), k_out.as<unsigned long long>(), p_in.as<uint32_t>(),
                                          p_out.as<uint32_t>(), (size_t)e_n[c], 0, (unsigned)key_bits, s));
        if (t4 > tmp_bytes) {
            RBG_HIP(hipStreamSynchronize(s));
            if ((rc = b_tmp.alloc(t4))) return rc;
            tmp_bytes = t4;
        }
        RBG_HIP(rocprim::radix_sort_pairs(b_tmp.p, t4, k_in.as<unsigned long long>(), k_out.as<unsigned long long>(), p_in.as<uint32_t>(),
                                          p_out.as<uint32_t>(), (size_t)e_n[c], 0, (unsigned)key_bits, s));
        hipLaunchKernelGGL(plan_scatter_kernel, dim3(grid_for_n(e_n[c])), dim3(256), 0, s, k_out.as<unsigned long long>(), p_out.as<uint32_t>(), e_n[c],
                           shift, segs[c], lgw, b_ptr.as<int32_t>() + rb[c], b_rdeg.as<int32_t>() + rb[c], head + sw->unit_base[c],
                           rect ? 2 * W * 4 : W * 4, g->d_val,
                           reinterpret_cast<int2 *>(sw->ent), sw->src);
        RBG_HIP(hipGetLastError());
        RBG_HIP(hipStreamSynchronize(s));  // (the temporaries of this class are freed here)
    }
    RBG_HIP(hipMemcpy(flag, b_flag.p, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag[1]) return na(rect ? "a column index outside the table" : "a user row lists a user or an item row an item: not the bipartite adjacency");

    // ---- 5. r = deg^-1/2 in the plan's numbering, then adopt: validation, derived arrays, factors ------------------------------
    if (rect) {  // (no slab chain on a rectangular block: no factors — a shard's rows do not even hold their columns' degrees)
        SellDev *adopt = sw;
        sw = nullptr;
        if ((rc = sell_adopt(g, adopt, true))) return rc;
        g->sell_note = "planned";
        return RBG_OK;
    }
    Buf b_r;
    if ((rc = b_r.alloc(sizeof(float) * (size_t)N))) return rc;
    for (int c = 0; c < 2; ++c)
        hipLaunchKernelGGL(plan_factors_kernel, dim3((n[c] + 255) / 256), dim3(256), 0, s, b_rdeg.as<int32_t>() + rb[c], n[c], b_r.as<float>() + base[c]);
    RBG_HIP(hipGetLastError());
    RBG_HIP(hipStreamSynchronize(s));
    SellDev *adopt = sw;
    sw = nullptr;  // (sell_adopt owns it now: freed there on failure)
    if ((rc = sell_adopt(g, adopt, true))) return rc;
    g->sell_note = "planned";
    if (sell_set_factors(g, b_r.as<float>()) != RBG_OK) clear_error();  // values that are not r_i r_j (a CSR the caller weighted): the valued chain
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_graph_plan_sell(rbg_graph *g, int W, int chunk) {
    clear_error();
    if (g && g->n_rows != g->n_cols) return plan_sell(g, W, chunk, true);
    return plan_sell(g, W, chunk, false);
}

int rbg_graph_sell_status(const rbg_graph *g, char *buf, int len) {
    if (!g || !buf || len <= 0) return fail(RBG_EINVAL, "NULL argument");
    snprintf(buf, (size_t)len, "%s", g->sell ? (g->sell->borrowed ? "view of a planned graph" : (g->sell->native ? "planned" : "attached"))
                                             : (g->sell_note.empty() ? "no plan" : g->sell_note.c_str()));
    return RBG_OK;
}

}  // extern "C"
