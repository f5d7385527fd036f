// graph_build.cpp — host side of librbgnn.so: error plumbing, the normalized-adjacency builder
// (replaces GeneralGraphDataset.get_norm_adj_mat, recbole_gnn/data/dataset.py:49-79, and SGL's view
// rebuild, sgl.py:107-126), degree binning for the SpMM launch, upload to HBM, export.
//
// Layout produced (see DESIGN.md "Data layout in HBM"):
//   rowptr int32 [N+1], col int32 [nnz] (ascending within a row), val fp32 [nnz]
//   desc  RowDesc[]       {row, beg, end} of wavefront / lane-group rows, by degree descending per row class
//   tasks BlockTask[]     one per workgroup row segment
// Users come first, items after (col = iid + n_users, dataset.py:61).  Duplicated interactions stay
// separate entries, each counted in the degree (PyG gcn_norm does the same).

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <thread>

#include "internal.h"

namespace rbg {

static thread_local std::string t_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_error = buf;
    return code;
}
void clear_error() { t_error.clear(); }

static std::atomic<int> g_short_max{64}, g_wave_max{256}, g_seg_len{4096};

static std::atomic<int> g_spmm_unroll{8}, g_xcd_split{4}, g_nt_store{1}, g_topk_sample{8192}, g_score_tiles{0}, g_col_split{-1}, g_mfma_split{1}, g_bignn_dma{1};
int spmm_unroll() { return g_spmm_unroll.load(); }
int opt_xcd_split() { return g_xcd_split.load(); }
int opt_nt_store() { return g_nt_store.load(); }
int opt_topk_sample() { return g_topk_sample.load(); }
int opt_score_tiles() { return g_score_tiles.load(); }
static std::atomic<int> g_score_uniform{1};
int opt_score_uniform() { return g_score_uniform.load(); }
static std::atomic<int> g_topk_short{1};
int opt_topk_short_lists() { return g_topk_short.load(); }
static std::atomic<int> g_topk_image{1};
int opt_topk_image() { return g_topk_image.load(); }
static std::atomic<int> g_topk_screen{1};
int opt_topk_screen() { return g_topk_screen.load(); }
static std::atomic<int> g_lse_onepass{1};
int opt_lse_onepass() { return g_lse_onepass.load(); }
static std::atomic<int> g_lse_tr_read{1};
int opt_lse_tr_read() { return g_lse_tr_read.load(); }
static std::atomic<int> g_lse_f16{3};
int opt_lse_f16() { return g_lse_f16.load(); }
static std::atomic<int> g_lse_image{0};  // (measured neutral to slightly slower, profiles/r06_lse_image.jsonl: off)
int opt_lse_image() { return g_lse_image.load(); }
static std::atomic<int> g_sell_c16{1};
int opt_sell_c16() { return g_sell_c16.load(); }
static std::atomic<int> g_sell_wpb{1};
int opt_sell_wpb() { return g_sell_wpb.load(); }
static std::atomic<int> g_deterministic{0};
int opt_deterministic() { return g_deterministic.load(); }
int opt_col_split() { return g_col_split.load(); }
static std::atomic<int> g_shard_single_stream{0};
static std::atomic<int> g_shard_fused{1};
int opt_shard_fused() { return g_shard_fused.load(); }
static std::atomic<int> g_slab{0};
static std::atomic<int> g_sell{1};
int opt_sell() { return g_sell.load(); }
static std::atomic<int> g_sell_nt{0};
int opt_sell_nt() { return g_sell_nt.load(); }
static std::atomic<int> g_sell_factored{1};
int opt_sell_factored() { return g_sell_factored.load(); }
static std::atomic<int> g_sell_rowmajor{1};
int opt_sell_rowmajor() { return g_sell_rowmajor.load(); }
static std::atomic<int> g_sell_auto{1};
int opt_sell_auto() { return g_sell_auto.load(); }

// fault injection for the tests: the (n + 1)-th dev_malloc from now fails once (option "fail_alloc_after" = n; -1 = off)
static std::atomic<int64_t> g_fail_alloc_after{-1};
hipError_t dev_malloc(void **p, size_t bytes) {
    int64_t left = g_fail_alloc_after.load();
    while (left >= 0) {
        if (g_fail_alloc_after.compare_exchange_weak(left, left - 1)) {
            if (left == 0) {
                *p = nullptr;
                return hipErrorOutOfMemory;
            }
            break;
        }
    }
    return hipMalloc(p, bytes ? bytes : 1);
}
int opt_slab() { return g_slab.load(); }
int opt_shard_single_stream() { return g_shard_single_stream.load(); }
int opt_bignn_dma() { return g_bignn_dma.load(); }
int opt_mfma_split() { return g_mfma_split.load(); }

Tuning current_tuning() { return Tuning{g_short_max.load(), g_wave_max.load(), g_seg_len.load()}; }

int set_device_for(int device) {
    int cur = -1;
    RBG_HIP(hipGetDevice(&cur));
    if (cur != device) RBG_HIP(hipSetDevice(device));
    return RBG_OK;
}

// Run fn(t, n_threads) on a small pool; used for the row-parallel passes of the builder.
template <class F>
static void parallel_run(int64_t work, F &&fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int n = (int)std::min<int64_t>(hw ? hw : 1, std::max<int64_t>(1, work / (1 << 16)));
    if (n <= 1) {
        fn(0, 1);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(n);
    for (int t = 0; t < n; ++t) th.emplace_back([&fn, t, n] { fn(t, n); });
    for (auto &x : th) x.join();
}

// Scatter pass: append (other + offset) to row `key[e]` while visiting interactions in ascending
// order of `other` (counting sort), so every row comes out column-sorted without a per-row sort.
static void fill_side(int64_t n_inter, const int64_t *key, const int64_t *other, const uint8_t *keep,
                      int64_t key_offset, int64_t other_offset, int64_t n_other,
                      const std::vector<int32_t> &rowptr, std::vector<int32_t> &col) {
    std::vector<int64_t> start((size_t)n_other + 1, 0);
    for (int64_t e = 0; e < n_inter; ++e)
        if (!keep || keep[e]) start[(size_t)other[e] + 1]++;
    for (int64_t i = 0; i < n_other; ++i) start[(size_t)i + 1] += start[(size_t)i];
    const int64_t kept = start[(size_t)n_other];
    std::vector<int64_t> order((size_t)kept);
    for (int64_t e = 0; e < n_inter; ++e)
        if (!keep || keep[e]) order[(size_t)start[(size_t)other[e]]++] = e;
    // write cursors of the destination rows (only rows [key_offset, key_offset + n_key) are advanced)
    std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
    for (int64_t k = 0; k < kept; ++k) {
        const int64_t e = order[(size_t)k];
        const int64_t r = key[e] + key_offset;
        col[(size_t)cur[(size_t)r]++] = (int32_t)(other[e] + other_offset);
    }
}

int build_host_csr(rbg_graph *g, int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                   const int64_t *iid, const uint8_t *keep) {
    if (n_users < 0 || n_items < 0 || n_inter < 0) return fail(RBG_EINVAL, "negative size");
    if (n_inter > 0 && (!uid || !iid)) return fail(RBG_EINVAL, "uid/iid is NULL");
    const int64_t n = n_users + n_items;
    if (n >= (int64_t)INT32_MAX) return fail(RBG_EUNSUPPORTED, "node count %lld >= 2^31", (long long)n);
    int64_t kept = 0;
    for (int64_t e = 0; e < n_inter; ++e) {
        if (keep && !keep[e]) continue;
        if (uid[e] < 0 || uid[e] >= n_users)
            return fail(RBG_EINVAL, "uid[%lld] = %lld out of [0,%lld)", (long long)e, (long long)uid[e], (long long)n_users);
        if (iid[e] < 0 || iid[e] >= n_items)
            return fail(RBG_EINVAL, "iid[%lld] = %lld out of [0,%lld)", (long long)e, (long long)iid[e], (long long)n_items);
        ++kept;
    }
    if (2 * kept >= (int64_t)INT32_MAX)
        return fail(RBG_EUNSUPPORTED, "nnz %lld >= 2^31 (int32 rowptr build)", (long long)(2 * kept));
    try {
        g->n_rows = g->n_cols = n;
        g->n_users = n_users;
        g->nnz = 2 * kept;
        g->h_rowptr.assign((size_t)n + 1, 0);
        for (int64_t e = 0; e < n_inter; ++e) {
            if (keep && !keep[e]) continue;
            g->h_rowptr[(size_t)uid[e] + 1]++;
            g->h_rowptr[(size_t)(iid[e] + n_users) + 1]++;
        }
        for (int64_t r = 0; r < n; ++r) g->h_rowptr[(size_t)r + 1] += g->h_rowptr[(size_t)r];
        g->h_col.assign((size_t)g->nnz, 0);
        g->h_val.assign((size_t)g->nnz, 0.f);
        // user rows list items (ascending), item rows list users (ascending)
        fill_side(n_inter, uid, iid, keep, 0, n_users, n_items, g->h_rowptr, g->h_col);
        fill_side(n_inter, iid, uid, keep, n_users, 0, n_users, g->h_rowptr, g->h_col);
        // gcn_norm(add_self_loops=False): deg = row sums of ones; dis = deg^-0.5 (inf -> 0), computed
        // as the IEEE 1/sqrt ATen's CPU pow(x,-0.5) uses; val = (dis[row] * 1) * dis[col], fp32.
        std::vector<float> dis((size_t)n);
        for (int64_t r = 0; r < n; ++r) {
            const float deg = (float)(g->h_rowptr[(size_t)r + 1] - g->h_rowptr[(size_t)r]);
            const float s = 1.0f / sqrtf(deg);
            dis[(size_t)r] = isinf(s) ? 0.0f : s;
        }
        const int32_t *rp = g->h_rowptr.data();
        const int32_t *cp = g->h_col.data();
        float *vp = g->h_val.data();
        const float *dp = dis.data();
        parallel_run(g->nnz, [=](int t, int nt) {
            const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
            for (int64_t r = r0; r < r1; ++r) {
                const float dr = dp[r] * 1.0f;
                for (int32_t e = rp[r]; e < rp[r + 1]; ++e) vp[e] = dr * dp[cp[e]];
            }
        });
    } catch (const std::bad_alloc &) {
        return fail(RBG_ENOMEM, "host allocation failed while building the CSR");
    }
    return RBG_OK;
}

// Degree-bin the given rows (ascending ids) of the CSR into one GroupPlan (appends to plan.desc / plan.tasks).
static void plan_group(const rbg_graph *g, const std::vector<int32_t> &rows, BinPlan &plan, GroupPlan &gp) {
    const int32_t *rp = g->h_rowptr.data();
    const Tuning tn = g->tuning;
    const int64_t n = (int64_t)rows.size();
    int32_t max_deg = 0;
    for (int32_t r : rows) max_deg = std::max(max_deg, rp[r + 1] - rp[r]);
    plan.max_deg = std::max(plan.max_deg, max_deg);
    // counting sort by degree, descending; ties keep ascending row id
    std::vector<int32_t> order((size_t)n);
    std::vector<int64_t> start((size_t)max_deg + 2, 0);
    for (int32_t r : rows) start[(size_t)(max_deg - (rp[r + 1] - rp[r])) + 1]++;
    for (int32_t k = 0; k <= max_deg; ++k) start[(size_t)k + 1] += start[(size_t)k];
    for (int32_t r : rows) order[(size_t)start[(size_t)(max_deg - (rp[r + 1] - rp[r]))]++] = r;
    gp = GroupPlan{};
    gp.task_base = (int32_t)plan.tasks.size();
    int64_t k = 0;
    for (; k < n; ++k) {  // workgroup rows
        const int32_t r = order[(size_t)k];
        const int32_t deg = rp[r + 1] - rp[r];
        if (deg <= tn.wave_max) break;
        ++plan.n_block_rows;
        const int32_t nseg = (deg + tn.seg_len - 1) / tn.seg_len;
        int32_t len = (deg + nseg - 1) / nseg;  // equal-length segments, a multiple of 64 entries
        len = (len + 63) / 64 * 64;
        const int32_t real = (deg + len - 1) / len;
        for (int32_t s = 0; s < real; ++s) {
            BlockTask t{};
            t.row = r;
            t.beg = rp[r] + s * len;
            t.end = std::min(rp[r + 1], t.beg + len);
            t.seg = s;
            t.nseg = real;
            t.part_base = real > 1 ? (int32_t)plan.n_slots : 0;
            t.ctr = real > 1 ? (int32_t)plan.n_split : 0;
            plan.tasks.push_back(t);
        }
        if (real > 1) {
            plan.n_slots += real;
            ++plan.n_split;
        }
    }
    gp.n_tasks = (int32_t)plan.tasks.size() - gp.task_base;
    gp.pos_wave = (int32_t)plan.desc.size();
    for (; k < n; ++k) {  // wavefront rows
        const int32_t r = order[(size_t)k];
        if (rp[r + 1] - rp[r] <= tn.short_max) break;
        plan.desc.push_back(RowDesc{r, rp[r], rp[r + 1], 0});
    }
    gp.n_wave = (int32_t)plan.desc.size() - gp.pos_wave;
    gp.pos_short = (int32_t)plan.desc.size();
    for (; k < n; ++k) {  // lane-group rows
        const int32_t r = order[(size_t)k];
        plan.desc.push_back(RowDesc{r, rp[r], rp[r + 1], 0});
    }
    gp.n_short = (int32_t)plan.desc.size() - gp.pos_short;
    plan.n_wave += gp.n_wave;
    plan.n_short += gp.n_short;
}

int plan_bins(const rbg_graph *g, BinPlan &plan) {
    const int64_t n = g->n_rows;
    plan = BinPlan{};
    plan.desc.reserve((size_t)n);
    if (g->flags & RBG_GRAPH_NATURAL_ORDER) {  // no binning: every row is a lane-group row, in id order
        const int32_t *rp = g->h_rowptr.data();
        for (int64_t r = 0; r < n; ++r) {
            plan.desc.push_back(RowDesc{(int32_t)r, rp[r], rp[r + 1], 0});
            plan.max_deg = std::max(plan.max_deg, rp[r + 1] - rp[r]);
        }
        plan.groups[0].n_short = (int32_t)n;
        plan.n_short = n;
        return RBG_OK;
    }
    auto range = [](int64_t a, int64_t b) {
        std::vector<int32_t> v((size_t)(b - a));
        for (int64_t r = a; r < b; ++r) v[(size_t)(r - a)] = (int32_t)r;
        return v;
    };
    if (!g->h_part.empty()) {
        // caller-supplied communities: part p runs on XCDs [p*(8/P), (p+1)*(8/P))
        const int P = g->n_parts, per = 8 / P;
        plan.n_groups = P;
        std::vector<std::vector<int32_t>> rows((size_t)P);
        for (int64_t r = 0; r < n; ++r) rows[(size_t)g->h_part[(size_t)r]].push_back((int32_t)r);
        for (int q = 0; q < P; ++q) plan_group(g, rows[(size_t)q], plan, plan.groups[q]);
        for (int x = 0; x < 8; ++x) {
            plan.xmap.grp[x] = (uint8_t)(x / per);
            plan.xmap.idx[x] = (uint8_t)(x % per);
            plan.xmap.cnt[x] = (uint8_t)per;
        }
        return RBG_OK;
    }
    // default XCD specialisation needs the user/item boundary (graphs built from interactions) and both classes
    // (or, for a CSR-built block such as a shard's interior / halo part, the caller-declared row_split: the two row
    // classes reference disjoint column sets there too)
    const int S = std::min(7, opt_xcd_split());
    const int64_t cut = g->row_split >= 0 ? g->row_split : ((g->n_rows == g->n_cols) ? g->n_users : -1);
    const bool split = S > 0 && cut > 0 && cut < n;
    if (split) {
        plan.n_groups = 2;
        plan_group(g, range(0, cut), plan, plan.groups[0]);
        plan_group(g, range(cut, n), plan, plan.groups[1]);
        for (int x = 0; x < 8; ++x) {
            const bool second = x >= S;
            plan.xmap.grp[x] = second ? 1 : 0;
            plan.xmap.idx[x] = (uint8_t)(second ? x - S : x);
            plan.xmap.cnt[x] = (uint8_t)(second ? 8 - S : S);
        }
    } else {
        plan_group(g, range(0, n), plan, plan.groups[0]);
    }
    return RBG_OK;
}

static void free_device(rbg_graph *g) {
    if (g->device < 0) return;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (cur != g->device && hipSetDevice(g->device) != hipSuccess) return;
    if (g->base && g->sell && g->sell->borrowed) g->base->sell_views.fetch_sub(1);  // the view lets go of its base's plan
    free_sell(g->sell);
    g->sell = nullptr;
    if (g->base) {  // a view owns only its split-row scratch
        (void)hipFree(g->d_partials);
        (void)hipFree(g->d_counters);
        g->d_partials = nullptr;
        g->d_counters = nullptr;
        if (cur != g->device) (void)hipSetDevice(cur);
        return;
    }
    (void)hipFree(g->d_rowptr);
    (void)hipFree(g->d_col);
    (void)hipFree(g->d_val);
    (void)hipFree(g->d_desc);
    (void)hipFree(g->d_tasks);
    (void)hipFree(g->d_partials);
    (void)hipFree(g->d_counters);
    g->d_rowptr = g->d_col = nullptr;
    g->d_desc = nullptr;
    g->d_val = g->d_partials = nullptr;
    g->d_tasks = nullptr;
    g->d_counters = nullptr;
    if (cur != g->device) (void)hipSetDevice(cur);
}

template <class T>
static int to_device(T **dst, const T *src, size_t count) {
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)dst, bytes);
    if (e != hipSuccess) return fail(RBG_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (count) RBG_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RBG_OK;
}

int to_device_raw(void **dst, const void *src, size_t bytes) {
    *dst = nullptr;
    hipError_t e = hipMalloc(dst, bytes ? bytes : 1);
    if (e != hipSuccess) {
        *dst = nullptr;
        return fail(RBG_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    if (src && bytes) RBG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RBG_OK;
}

// Launch plan (row classes, degree bins, split-row scratch) from the host copy of rowptr.
int upload_plan(rbg_graph *g) {
    int rc;
    BinPlan plan;
    try {
        rc = plan_bins(g, plan);
    } catch (const std::bad_alloc &) {
        return fail(RBG_ENOMEM, "host allocation failed while binning rows");
    }
    if (rc) return rc;
    g->n_groups = plan.n_groups;
    g->xmap = plan.xmap;
    for (int q = 0; q < kMaxGroups; ++q) g->groups[q] = plan.groups[q];
    std::vector<int8_t>().swap(g->h_part);
    g->n_block_rows = plan.n_block_rows;
    g->n_wave = plan.n_wave;
    g->n_short = plan.n_short;
    g->n_tasks = (int64_t)plan.tasks.size();
    g->n_split_rows = plan.n_split;
    g->n_partial_slots = plan.n_slots;
    g->max_degree = plan.max_deg;
    if ((rc = to_device(&g->d_desc, plan.desc.data(), plan.desc.size()))) return rc;
    if ((rc = to_device(&g->d_tasks, plan.tasks.data(), plan.tasks.size()))) return rc;
    const size_t pb = std::max<size_t>((size_t)plan.n_slots, 1) * kPartialSlotFloats * sizeof(float);
    hipError_t e = hipMalloc((void **)&g->d_partials, pb);
    if (e != hipSuccess) return fail(RBG_ENOMEM, "hipMalloc(%zu bytes) failed: %s", pb, hipGetErrorString(e));
    // two arrival counters per split row: the column-half SpMM mode finishes each half of a row on its own
    const size_t cb = 2 * std::max<size_t>((size_t)plan.n_split, 1) * sizeof(uint32_t);
    e = hipMalloc((void **)&g->d_counters, cb);
    if (e != hipSuccess) return fail(RBG_ENOMEM, "hipMalloc(%zu bytes) failed: %s", cb, hipGetErrorString(e));
    RBG_HIP(hipMemset(g->d_counters, 0, cb));
    RBG_HIP(hipDeviceSynchronize());
    return RBG_OK;
}

int upload_graph(rbg_graph *g) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(RBG_ENODEV, "a device graph was requested but no GPU is visible");
    if (g->device >= n_dev) return fail(RBG_EINVAL, "device %d out of range (%d visible)", g->device, n_dev);
    int rc = set_device_for(g->device);
    if (rc) return rc;
    if ((rc = to_device(&g->d_rowptr, g->h_rowptr.data(), g->h_rowptr.size()))) return rc;
    if ((rc = to_device(&g->d_col, g->h_col.data(), g->h_col.size()))) return rc;
    if ((rc = to_device(&g->d_val, g->h_val.data(), g->h_val.size()))) return rc;
    if ((rc = upload_plan(g))) return rc;
    if (!(g->flags & RBG_GRAPH_KEEP_HOST)) {
        std::vector<int32_t>().swap(g->h_rowptr);
        std::vector<int32_t>().swap(g->h_col);
        std::vector<float>().swap(g->h_val);
    }
    return RBG_OK;
}

// Every device graph with a user / item boundary gets the column-slab plan at creation (option "sell_auto"): the fast kernel is
// a property of the handle, not of an adapter above the C ABI.  A graph the plan does not serve (a hub row beyond its reach,
// a table beyond 32-bit offsets, no boundary, out of memory) keeps the binned kernel: the reason is kept for
// rbg_graph_sell_status, the creation itself never fails because of it.
static void auto_plan(rbg_graph *g) {
    if (g->device < 0 || !opt_sell_auto() || !opt_sell()) {
        g->sell_note = g->device < 0 ? "host graph" : "planning disabled (options \"sell_auto\" / \"sell\")";
        return;
    }
    if (g->n_parts > 1) {
        g->sell_note = "rows are pinned to XCDs by the caller's community partition";
        return;
    }
    // a rectangular block with two row classes (a shard's [owned | halo] product or its halo block): the rectangular form;
    // a square one that is not the bipartite adjacency (a halo block with as many slots as rows): the same
    int rc = g->n_rows != g->n_cols ? plan_sell(g, 32, 0, true) : plan_sell(g, 32, 0, false);
    if (rc == RBG_EUNSUPPORTED && g->n_rows == g->n_cols && g->n_users < 0 && g->row_split > 0 && g->row_split < g->n_rows &&
        g->sell_note.find("not the bipartite") != std::string::npos) {
        clear_error();
        rc = plan_sell(g, 32, 0, true);
    }
    if (rc != RBG_OK) {
        if (g->sell_note.empty() || rc != RBG_EUNSUPPORTED) g->sell_note = t_error;
        clear_error();
    }
}

static int finish_create(rbg_graph **out, rbg_graph *g, int rc) {
    if (rc == RBG_OK && g->device >= 0) rc = upload_graph(g);
    if (rc == RBG_OK) auto_plan(g);
    if (rc != RBG_OK) {
        free_device(g);
        delete g;
        return rc;
    }
    *out = g;
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_abi_version(void) { return RBG_ABI_VERSION; }

const char *rbg_last_error(void) { return rbg::t_error.c_str(); }

int rbg_device_count(int *count) {
    if (!count) return fail(RBG_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return RBG_OK;
}

int rbg_set_tuning(int short_max, int wave_max, int seg_len) {
    const Tuning cur = current_tuning();
    const int s = short_max < 0 ? cur.short_max : short_max;
    const int w = wave_max < 0 ? cur.wave_max : wave_max;
    const int l = seg_len < 0 ? cur.seg_len : seg_len;
    if (w < s) return fail(RBG_EINVAL, "wave_max (%d) < short_max (%d)", w, s);
    if (l < 64) return fail(RBG_EINVAL, "seg_len (%d) < 64", l);
    g_short_max = s;
    g_wave_max = w;
    g_seg_len = l;
    return RBG_OK;
}

int rbg_get_tuning(int *short_max, int *wave_max, int *seg_len) {
    const Tuning cur = current_tuning();
    if (short_max) *short_max = cur.short_max;
    if (wave_max) *wave_max = cur.wave_max;
    if (seg_len) *seg_len = cur.seg_len;
    return RBG_OK;
}

int rbg_set_option(const char *key, int64_t value) {
    if (!key) return fail(RBG_EINVAL, "key is NULL");
    if (!strcmp(key, "spmm_unroll")) {
        if (value != 4 && value != 8) return fail(RBG_EINVAL, "spmm_unroll must be 4 or 8");
        g_spmm_unroll = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "xcd_split")) {
        if (value < 0 || value > 7) return fail(RBG_EINVAL, "xcd_split must be 0 (off) or 1..7 XCDs for user rows");
        g_xcd_split = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "nt_store")) {
        g_nt_store = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "col_split")) {
        if (value < -1 || value > 1) return fail(RBG_EINVAL, "col_split must be -1 (auto), 0 (off) or 1 (on)");
        g_col_split = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "shard_single_stream")) {
        g_shard_single_stream = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_wpb")) {
        if (value != 1 && value != 2 && value != 4) return fail(RBG_EINVAL, "sell_wpb must be 1, 2 or 4 waves per workgroup");
        g_sell_wpb = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "lse_image")) {
        g_lse_image = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "lse_tr_read")) {
        g_lse_tr_read = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "lse_f16")) {
        g_lse_f16 = value < 0 ? 0 : (value > 3 ? 3 : (int)value);
        return RBG_OK;
    }
    if (!strcmp(key, "topk_image")) {
        g_topk_image = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        return RBG_OK;
    }
    if (!strcmp(key, "topk_screen")) {
        g_topk_screen = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
        return RBG_OK;
    }
    if (!strcmp(key, "shard_fused")) {
        g_shard_fused = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "slab")) {
        g_slab = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell")) {
        g_sell = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_rowmajor")) {
        g_sell_rowmajor = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_auto")) {
        g_sell_auto = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "fail_alloc_after")) {
        g_fail_alloc_after = value < 0 ? -1 : value;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_factored")) {
        g_sell_factored = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_nt")) {
        g_sell_nt = (int)(value & 3);
        return RBG_OK;
    }
    if (!strcmp(key, "bignn_dma")) {
        g_bignn_dma = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "mfma_split")) {
        g_mfma_split = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "score_uniform")) {
        g_score_uniform = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "lse_onepass")) {
        g_lse_onepass = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "deterministic")) {
        g_deterministic = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "sell_c16")) {
        g_sell_c16 = value ? 1 : 0;
        return RBG_OK;
    }
    if (!strcmp(key, "topk_short_lists")) {
        if (value < 0 || value > 2) return fail(RBG_EINVAL, "topk_short_lists must be 0, 1 or 2");
        g_topk_short = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "score_tiles")) {
        if (value < 0 || value > 4096) return fail(RBG_EINVAL, "score_tiles must be 0 (auto) or 1..4096");
        g_score_tiles = (int)value;
        return RBG_OK;
    }
    if (!strcmp(key, "topk_sample")) {
        if (value < 1024 || value > (1 << 20) || value % 128) return fail(RBG_EINVAL, "topk_sample must be a multiple of 128 in [1024, 2^20]");
        g_topk_sample = (int)value;
        return RBG_OK;
    }
    return fail(RBG_EINVAL, "unknown option '%s'", key);
}

int rbg_get_option(const char *key, int64_t *value) {
    if (!key || !value) return fail(RBG_EINVAL, "NULL argument");
    if (!strcmp(key, "spmm_unroll")) {
        *value = g_spmm_unroll.load();
        return RBG_OK;
    }
    if (!strcmp(key, "xcd_split")) {
        *value = g_xcd_split.load();
        return RBG_OK;
    }
    if (!strcmp(key, "nt_store")) {
        *value = g_nt_store.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_wpb")) {
        *value = g_sell_wpb.load();
        return RBG_OK;
    }
    if (!strcmp(key, "lse_image")) {
        *value = g_lse_image.load();
        return RBG_OK;
    }
    if (!strcmp(key, "lse_tr_read")) {
        *value = g_lse_tr_read.load();
        return RBG_OK;
    }
    if (!strcmp(key, "lse_f16")) {
        *value = g_lse_f16.load();
        return RBG_OK;
    }
    if (!strcmp(key, "topk_image")) {
        *value = g_topk_image.load();
        return RBG_OK;
    }
    if (!strcmp(key, "topk_screen")) {
        *value = g_topk_screen.load();
        return RBG_OK;
    }
    if (!strcmp(key, "shard_fused")) {
        *value = g_shard_fused.load();
        return RBG_OK;
    }
    if (!strcmp(key, "shard_single_stream")) {
        *value = g_shard_single_stream.load();
        return RBG_OK;
    }
    if (!strcmp(key, "slab")) {
        *value = g_slab.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell")) {
        *value = g_sell.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_rowmajor")) {
        *value = g_sell_rowmajor.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_auto")) {
        *value = g_sell_auto.load();
        return RBG_OK;
    }
    if (!strcmp(key, "fail_alloc_after")) {
        *value = g_fail_alloc_after.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_factored")) {
        *value = g_sell_factored.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_nt")) {
        *value = g_sell_nt.load();
        return RBG_OK;
    }
    if (!strcmp(key, "col_split")) {
        *value = g_col_split.load();
        return RBG_OK;
    }
    if (!strcmp(key, "bignn_dma")) {
        *value = g_bignn_dma.load();
        return RBG_OK;
    }
    if (!strcmp(key, "mfma_split")) {
        *value = g_mfma_split.load();
        return RBG_OK;
    }
    if (!strcmp(key, "score_uniform")) {
        *value = g_score_uniform.load();
        return RBG_OK;
    }
    if (!strcmp(key, "deterministic")) {
        *value = g_deterministic.load();
        return RBG_OK;
    }
    if (!strcmp(key, "sell_c16")) {
        *value = g_sell_c16.load();
        return RBG_OK;
    }
    if (!strcmp(key, "lse_onepass")) {
        *value = g_lse_onepass.load();
        return RBG_OK;
    }
    if (!strcmp(key, "topk_short_lists")) {
        *value = g_topk_short.load();
        return RBG_OK;
    }
    if (!strcmp(key, "score_tiles")) {
        *value = g_score_tiles.load();
        return RBG_OK;
    }
    if (!strcmp(key, "topk_sample")) {
        *value = g_topk_sample.load();
        return RBG_OK;
    }
    return fail(RBG_EINVAL, "unknown option '%s'", key);
}

int rbg_graph_create_masked(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter,
                            const int64_t *uid, const int64_t *iid, const uint8_t *keep, int device,
                            uint32_t flags) {
    return rbg_graph_create_partitioned(out, n_users, n_items, n_inter, uid, iid, keep, nullptr, 0, device, flags);
}

int rbg_graph_create_partitioned(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter,
                                 const int64_t *uid, const int64_t *iid, const uint8_t *keep, const int32_t *part,
                                 int n_parts, int device, uint32_t flags) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (device < -1) return fail(RBG_EINVAL, "device %d", device);
    if (part && n_parts != 1 && n_parts != 2 && n_parts != 4 && n_parts != 8)
        return fail(RBG_EINVAL, "n_parts = %d (must be 1, 2, 4 or 8: parts are pinned to whole XCDs)", n_parts);
    if ((flags & RBG_GRAPH_INPUTS_ON_DEVICE) && (device < 0 || (flags & RBG_GRAPH_BUILD_ON_HOST)))
        return fail(RBG_EINVAL, "RBG_GRAPH_INPUTS_ON_DEVICE needs a device graph and the device builder");
    rbg_graph *g = new (std::nothrow) rbg_graph();
    if (!g) return fail(RBG_ENOMEM, "graph handle allocation failed");
    g->device = device;
    g->flags = flags | (device < 0 ? RBG_GRAPH_KEEP_HOST : 0u);
    g->tuning = current_tuning();
    if (part && n_parts > 1 && n_users >= 0 && n_items >= 0) {
        const int64_t n = n_users + n_items;
        try {
            g->h_part.resize((size_t)n);
        } catch (const std::bad_alloc &) {
            delete g;
            return fail(RBG_ENOMEM, "host allocation failed");
        }
        for (int64_t r = 0; r < n; ++r) {
            if (part[r] < 0 || part[r] >= n_parts) {
                const int bad = part[r];
                delete g;
                return fail(RBG_EINVAL, "part[%lld] = %d out of [0,%d)", (long long)r, bad, n_parts);
            }
            g->h_part[(size_t)r] = (int8_t)part[r];
        }
        g->n_parts = n_parts;
    }
    if (device >= 0 && !(flags & RBG_GRAPH_BUILD_ON_HOST)) {
        // device builder: sort / scan / weights in HBM, only rowptr returns for the launch plan
        int rc = build_device_csr(g, n_users, n_items, n_inter, uid, iid, keep);
        if (rc == RBG_OK) rc = upload_plan(g);
        if (rc == RBG_OK && (flags & RBG_GRAPH_KEEP_HOST)) {
            g->h_col.resize((size_t)g->nnz);
            g->h_val.resize((size_t)g->nnz);
            if (g->nnz && (hipMemcpy(g->h_col.data(), g->d_col, sizeof(int32_t) * (size_t)g->nnz, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(g->h_val.data(), g->d_val, sizeof(float) * (size_t)g->nnz, hipMemcpyDeviceToHost) != hipSuccess))
                rc = fail(RBG_EHIP, "D2H copy of the CSR failed");
        } else if (rc == RBG_OK) {
            std::vector<int32_t>().swap(g->h_rowptr);
        }
        if (rc != RBG_OK) {
            free_device(g);
            delete g;
            return rc;
        }
        auto_plan(g);
        *out = g;
        return RBG_OK;
    }
    return finish_create(out, g, build_host_csr(g, n_users, n_items, n_inter, uid, iid, keep));
}

int rbg_graph_create(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                     const int64_t *iid, int device, uint32_t flags) {
    return rbg_graph_create_masked(out, n_users, n_items, n_inter, uid, iid, nullptr, device, flags);
}

int rbg_graph_create_csr(rbg_graph **out, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                         const int32_t *col, const float *val, int device, uint32_t flags) {
    return rbg_graph_create_csr_classes(out, n_rows, n_cols, rowptr, col, val, -1, device, flags);
}

int rbg_graph_create_csr_classes(rbg_graph **out, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                                 const int32_t *col, const float *val, int64_t n_class0_rows, int device, uint32_t flags) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_rows < 0 || n_cols < 0 || !rowptr) return fail(RBG_EINVAL, "bad shape or NULL rowptr");
    if (n_class0_rows > n_rows) return fail(RBG_EINVAL, "n_class0_rows = %lld > n_rows", (long long)n_class0_rows);
    if (device < -1) return fail(RBG_EINVAL, "device %d", device);
    if (n_rows >= (int64_t)INT32_MAX || n_cols >= (int64_t)INT32_MAX)
        return fail(RBG_EUNSUPPORTED, "dimension >= 2^31");
    if (rowptr[0] != 0) return fail(RBG_EINVAL, "rowptr[0] != 0");
    for (int64_t r = 0; r < n_rows; ++r)
        if (rowptr[r + 1] < rowptr[r]) return fail(RBG_EINVAL, "rowptr not monotone at row %lld", (long long)r);
    const int64_t nnz = rowptr[n_rows];
    if (nnz >= (int64_t)INT32_MAX) return fail(RBG_EUNSUPPORTED, "nnz %lld >= 2^31", (long long)nnz);
    if (nnz > 0 && (!col || !val)) return fail(RBG_EINVAL, "col/val is NULL");
    for (int64_t e = 0; e < nnz; ++e)
        if (col[e] < 0 || col[e] >= n_cols)
            return fail(RBG_EINVAL, "col[%lld] = %d out of [0,%lld)", (long long)e, col[e], (long long)n_cols);
    if (n_class0_rows < 0 && n_rows == n_cols && nnz > 0) {  // a square CSR without stated classes: the bipartite boundary, if there is one
        int64_t s_lo = 0, s_hi = n_rows;
        for (int64_t r = 0; r < n_rows && s_lo <= s_hi; ++r)
            for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                s_lo = std::max<int64_t>(s_lo, std::min<int64_t>(r, col[e]) + 1);
                s_hi = std::min<int64_t>(s_hi, std::max<int64_t>(r, col[e]));
            }
        if (s_lo <= s_hi && s_lo > 0 && s_lo < n_rows) n_class0_rows = s_lo;
    }
    rbg_graph *g = new (std::nothrow) rbg_graph();
    if (!g) return fail(RBG_ENOMEM, "graph handle allocation failed");
    g->device = device;
    g->flags = flags | (device < 0 ? RBG_GRAPH_KEEP_HOST : 0u);
    g->tuning = current_tuning();
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->row_split = n_class0_rows < 0 ? -1 : n_class0_rows;
    int rc = RBG_OK;
    try {
        g->h_rowptr.resize((size_t)n_rows + 1);
        for (int64_t r = 0; r <= n_rows; ++r) g->h_rowptr[(size_t)r] = (int32_t)rowptr[r];
        g->h_col.assign(col, col + nnz);
        g->h_val.assign(val, val + nnz);
    } catch (const std::bad_alloc &) {
        rc = fail(RBG_ENOMEM, "host allocation failed while copying the CSR");
    }
    return finish_create(out, g, rc);
}

int rbg_graph_create_coo(rbg_graph **out, int64_t n_nodes, int64_t nnz, const int64_t *edge_index,
                         const float *edge_weight, int device, uint32_t flags) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_nodes < 0 || nnz < 0) return fail(RBG_EINVAL, "negative size");
    if (nnz > 0 && (!edge_index || !edge_weight)) return fail(RBG_EINVAL, "edge_index/edge_weight is NULL");
    if (device < -1) return fail(RBG_EINVAL, "device %d", device);
    if (n_nodes >= (int64_t)INT32_MAX || nnz >= (int64_t)INT32_MAX)
        return fail(RBG_EUNSUPPORTED, "size >= 2^31");
    const int64_t *src = edge_index, *dst = edge_index + nnz;
    // The reference's default branch hands over the bipartite adjacency as a pair (dataset.py:60-66,77-79: users first, items
    // after) without saying where the items start.  If every edge joins a node below some boundary to a node at or above
    // it, that boundary is found here — any value in (largest lower endpoint, smallest upper endpoint] is one; the lowest is
    // taken: the last user normally has interactions, the first item id is the isolated [PAD] — and the handle gets the two
    // row classes and, with them, the column-slab plan (r04).
    int64_t s_lo = 0, s_hi = n_nodes;
    for (int64_t e = 0; e < nnz; ++e) {
        if (src[e] < 0 || src[e] >= n_nodes || dst[e] < 0 || dst[e] >= n_nodes)
            return fail(RBG_EINVAL, "edge %lld (%lld -> %lld) out of [0,%lld)", (long long)e, (long long)src[e],
                        (long long)dst[e], (long long)n_nodes);
        s_lo = std::max(s_lo, std::min(src[e], dst[e]) + 1);
        s_hi = std::min(s_hi, std::max(src[e], dst[e]));
    }
    rbg_graph *g = new (std::nothrow) rbg_graph();
    if (!g) return fail(RBG_ENOMEM, "graph handle allocation failed");
    g->device = device;
    g->flags = flags | (device < 0 ? RBG_GRAPH_KEEP_HOST : 0u);
    g->tuning = current_tuning();
    g->n_rows = g->n_cols = n_nodes;
    g->nnz = nnz;
    if (nnz > 0 && s_lo <= s_hi && s_lo > 0 && s_lo < n_nodes) g->row_split = s_lo;  // bipartite: rows [0, s_lo) / [s_lo, N)
    int rc = RBG_OK;
    try {
        // adj_t: row = target, col = source; sort by (target, source) via counting sort on source then
        // a stable scatter on target (SparseTensor sorts on construction, dataset.py:43-47).
        g->h_rowptr.assign((size_t)n_nodes + 1, 0);
        for (int64_t e = 0; e < nnz; ++e) g->h_rowptr[(size_t)dst[e] + 1]++;
        for (int64_t r = 0; r < n_nodes; ++r) g->h_rowptr[(size_t)r + 1] += g->h_rowptr[(size_t)r];
        std::vector<int64_t> start((size_t)n_nodes + 1, 0);
        for (int64_t e = 0; e < nnz; ++e) start[(size_t)src[e] + 1]++;
        for (int64_t r = 0; r < n_nodes; ++r) start[(size_t)r + 1] += start[(size_t)r];
        std::vector<int64_t> order((size_t)nnz);
        for (int64_t e = 0; e < nnz; ++e) order[(size_t)start[(size_t)src[e]]++] = e;
        std::vector<int32_t> cur(g->h_rowptr.begin(), g->h_rowptr.end() - 1);
        g->h_col.assign((size_t)nnz, 0);
        g->h_val.assign((size_t)nnz, 0.f);
        for (int64_t k = 0; k < nnz; ++k) {
            const int64_t e = order[(size_t)k];
            const int32_t p = cur[(size_t)dst[e]]++;
            g->h_col[(size_t)p] = (int32_t)src[e];
            g->h_val[(size_t)p] = edge_weight[e];
        }
    } catch (const std::bad_alloc &) {
        rc = fail(RBG_ENOMEM, "host allocation failed while building the CSR");
    }
    return finish_create(out, g, rc);
}

int rbg_norm_edges(int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid, const int64_t *iid,
                   int64_t *edge_index, float *edge_weight) {
    clear_error();
    if (n_users < 0 || n_items < 0 || n_inter < 0) return fail(RBG_EINVAL, "negative size");
    if (n_inter > 0 && (!uid || !iid || !edge_index || !edge_weight)) return fail(RBG_EINVAL, "NULL argument");
    const int64_t n = n_users + n_items, m = 2 * n_inter;
    std::vector<float> deg;
    try {
        deg.assign((size_t)std::max<int64_t>(n, 1), 0.f);
    } catch (const std::bad_alloc &) {
        return fail(RBG_ENOMEM, "host allocation failed");
    }
    int64_t *src = edge_index, *dst = edge_index + m;
    for (int64_t e = 0; e < n_inter; ++e) {
        if (uid[e] < 0 || uid[e] >= n_users || iid[e] < 0 || iid[e] >= n_items)
            return fail(RBG_EINVAL, "interaction %lld (%lld,%lld) out of range", (long long)e, (long long)uid[e],
                        (long long)iid[e]);
        const int64_t u = uid[e], i = iid[e] + n_users;
        src[e] = u;
        dst[e] = i;
        src[n_inter + e] = i;
        dst[n_inter + e] = u;
    }
    // deg = scatter_add(w, col) with w = 1 (exact in fp32 below 2^24; beyond, add in the same
    // edge order as torch's sequential CPU scatter_add_)
    for (int64_t e = 0; e < m; ++e) deg[(size_t)dst[e]] += 1.0f;
    for (int64_t r = 0; r < n; ++r) {
        const float s = 1.0f / sqrtf(deg[(size_t)r]);
        deg[(size_t)r] = isinf(s) ? 0.0f : s;
    }
    for (int64_t e = 0; e < m; ++e) edge_weight[e] = (deg[(size_t)src[e]] * 1.0f) * deg[(size_t)dst[e]];
    return RBG_OK;
}


int rbg_graph_create_reweighted(rbg_graph **out, const rbg_graph *src, const float *vals) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (!src) return fail(RBG_EINVAL, "src is NULL");
    if (src->device < 0) return fail(RBG_ENODEV, "re-weighted views exist for device graphs");
    if (src->nnz > 0 && !vals) return fail(RBG_EINVAL, "vals is NULL");
    int rc = set_device_for(src->device);
    if (rc) return rc;
    rbg_graph *g = new (std::nothrow) rbg_graph();
    if (!g) return fail(RBG_ENOMEM, "graph handle allocation failed");
    const rbg_graph *root = src->base ? src->base : src;
    g->n_rows = root->n_rows;
    g->n_cols = root->n_cols;
    g->nnz = root->nnz;
    g->n_users = root->n_users;
    g->row_split = root->row_split;
    g->device = root->device;
    g->flags = root->flags & ~RBG_GRAPH_KEEP_HOST;
    g->d_rowptr = root->d_rowptr;
    g->d_col = root->d_col;
    g->d_val = const_cast<float *>(vals);
    g->tuning = root->tuning;
    g->n_groups = root->n_groups;
    g->xmap = root->xmap;
    for (int q = 0; q < kMaxGroups; ++q) g->groups[q] = root->groups[q];
    g->d_desc = root->d_desc;
    g->n_block_rows = root->n_block_rows;
    g->n_wave = root->n_wave;
    g->n_short = root->n_short;
    g->d_tasks = root->d_tasks;
    g->n_tasks = root->n_tasks;
    g->n_split_rows = root->n_split_rows;
    g->n_partial_slots = root->n_partial_slots;
    g->max_degree = root->max_degree;
    g->base = root;
    // split rows publish partial sums through per-handle scratch: a view gets its own, so launches on the view and on
    // its base may run concurrently
    const size_t pb = std::max<size_t>((size_t)g->n_partial_slots, 1) * kPartialSlotFloats * sizeof(float);
    const size_t cb = 2 * std::max<size_t>((size_t)g->n_split_rows, 1) * sizeof(uint32_t);
    if (hipMalloc((void **)&g->d_partials, pb) != hipSuccess || hipMalloc((void **)&g->d_counters, cb) != hipSuccess ||
        hipMemset(g->d_counters, 0, cb) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(g->d_partials);
        (void)hipFree(g->d_counters);
        delete g;
        return fail(RBG_ENOMEM, "device allocation of the view's scratch failed");
    }
    // the base graph's column-slab plan, with the view's own copy of the valued entries (used from the first
    // rbg_graph_refresh_values on; until then — and without a plan — the view's launches read `vals` directly)
    if (root->sell && sell_make_view(g, root) != RBG_OK) clear_error();
    g->sell_note = g->sell ? "view of a planned graph" : "the base graph has no plan with row-major entries";
    *out = g;
    return RBG_OK;
}

int rbg_graph_info(const rbg_graph *g, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int *device) {
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    if (device) *device = g->device;
    return RBG_OK;
}

int rbg_graph_device_arrays(const rbg_graph *g, const int32_t **rowptr, const int32_t **col, const float **val) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "host graph: no device arrays");
    if (rowptr) *rowptr = g->d_rowptr;
    if (col) *col = g->d_col;
    if (val) *val = g->d_val;
    return RBG_OK;
}

int rbg_graph_export_csr(const rbg_graph *g, int64_t *rowptr, int32_t *col, float *val) {
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (!g->h_rowptr.empty()) {
        if (rowptr)
            for (int64_t r = 0; r <= g->n_rows; ++r) rowptr[r] = g->h_rowptr[(size_t)r];
        if (col && g->nnz) memcpy(col, g->h_col.data(), sizeof(int32_t) * (size_t)g->nnz);
        if (val && g->nnz) memcpy(val, g->h_val.data(), sizeof(float) * (size_t)g->nnz);
        return RBG_OK;
    }
    if (g->device < 0) return fail(RBG_EINVAL, "graph holds no CSR");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    RBG_HIP(hipDeviceSynchronize());
    if (rowptr) {
        std::vector<int32_t> tmp((size_t)g->n_rows + 1);
        RBG_HIP(hipMemcpy(tmp.data(), g->d_rowptr, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r <= g->n_rows; ++r) rowptr[r] = tmp[(size_t)r];
    }
    if (col && g->nnz) RBG_HIP(hipMemcpy(col, g->d_col, sizeof(int32_t) * (size_t)g->nnz, hipMemcpyDeviceToHost));
    if (val && g->nnz) RBG_HIP(hipMemcpy(val, g->d_val, sizeof(float) * (size_t)g->nnz, hipMemcpyDeviceToHost));
    return RBG_OK;
}

void rbg_graph_destroy(rbg_graph *g) {
    if (!g) return;
    // hipFree is an "unsafe" call while ANY stream of the process captures in the global mode (torch.cuda.graph's default): it
    // fails and invalidates that capture, whichever stream this thread is on.  A dying handle's arrays are not part of a capture
    // in progress (a captured graph that launches on the handle needs the handle alive anyway), so the frees run with this
    // thread's capture mode relaxed, and the mode is put back.  (The Python owner additionally parks handles that die while its
    // OWN current stream captures: graph.py GraphHandle.destroy.)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool swapped = g->device >= 0 && hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    free_device(g);
    if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    (void)hipGetLastError();
    delete g;
}

}  // extern "C"
