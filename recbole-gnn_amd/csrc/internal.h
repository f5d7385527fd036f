// internal.h — shared declarations of librbgnn.so (not part of the public ABI; see include/rbgnn.h)
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "rbgnn.h"

namespace rbg {

// ---- error plumbing ------------------------------------------------------------------------
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define RBG_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::rbg::fail(RBG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                            \
    } while (0)

// ---- tuning (rbg_set_tuning) ---------------------------------------------------------------
struct Tuning {
    int short_max;  // degree <= short_max : one row per lane-group
    int wave_max;   // degree <= wave_max  : one row per wavefront; above: workgroup tasks
    int seg_len;    // workgroup rows are cut into segments of this many entries
};
Tuning current_tuning();
int spmm_unroll();
int opt_xcd_split();
int opt_nt_store();
int opt_bignn_dma();     // BiGNN dense layer at d_in = 64, d_out <= 64: LDS-DMA kernel (1) or the general kernel (0)
int opt_shard_single_stream();  // C-ABI sharded layer: pack + exchange on the caller's stream (1) or on the shard's comm stream (0)
int opt_shard_fused();  // rbg_graph_create_sharded: the rank's [interior | halo] block as ONE planned handle, a layer = exchange + one launch (1, default)
int opt_sell_nt();         // sell.hip epilogue: bit 0 = non-temporal stores, bit 1 = non-temporal loads of the mean's addends
int opt_sell_wpb();       // sell.hip: waves per workgroup of the layer launch (1, 2 or 4)
int opt_sell_c16();       // sell.hip: compact launches read 16-bit slab-row numbers where the plan has them (1, default)
int opt_sell_factored();  // sell.hip: factored chains (val_ij = r_i r_j): compact entries, scaled slabs
int opt_sell_rowmajor();  // sell.hip: gather E0 / the incoming gradient row-major where they lie (no conversion to slabs)
int opt_sell();          // rbg_lightgcn_forward_f32: use an attached SELL plan (column-slab propagation, sell.hip)
int opt_sell_auto();     // rbg_graph_create*: plan the column-slab propagation for every device graph with a user / item boundary
// hipMalloc behind the fault-injection hook of the tests (option "fail_alloc_after"): every allocation of the plan code goes
// through it, so that a test can walk the error paths one allocation at a time
hipError_t dev_malloc(void **p, size_t bytes);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc(reinterpret_cast<void **>(p), bytes); }
int opt_slab();          // rbg_lightgcn_forward_f32: keep the layers as two column slabs (column-half kernel over contiguous half rows)
int opt_col_split();    // SpMM: even / odd XCDs own the lower / upper half of the columns
int opt_mfma_split();   // score / top-k: 3 x bf16 split operands on the bf16 matrix cores (1) or the exact-fp32 MFMA (0)
// train.hip: zero `bytes` (4-byte units) at ptr by a fill kernel — NOT hipMemsetAsync, whose node writes garbage on a captured graph's later replays
int zero_async(void *ptr, size_t bytes, hipStream_t s);
int opt_topk_image();  // rbg_full_sort_topk_f32: split the item table once per call into a plane image the passes take by LDS-DMA (1, default)
int opt_topk_screen();  // rbg_full_sort_topk_f32: one bf16 product per pair screens the call, survivors rescored exactly (topk_screen.hip): 0 never, 1 from 256 users (default), 2 always
int opt_topk_short_lists();  // rbg_full_sort_topk_f32: 24-entry LDS lists (three workgroups per CU) at k <= 12, d <= 64: 0 never, 1 from 2048 users (default), 2 always
int opt_deterministic();  // train.hip / lse.hip: row scatters by owner wavefronts in batch order (ordered.h), sums in fixed point: bit-stable
int opt_lse_tr_read();    // lse.hip gradients: the second product's B fragments by ds_read_b64_tr_b16 from the row-major planes (1, default)
int opt_lse_f16();        // rbg_infonce_f32 with gradients, no weights: both products of the gradient passes on the fp16 matrix cores, two terms per operand (3, default; see rbgnn.h)
int opt_lse_image();      // rbg_infonce*_f32 with gradients: the normalised table / batch rows as plane images taken by LDS-DMA (0, default: measured neutral)
int opt_lse_onepass();    // rbg_infonce_f32: denominators and the batch rows' gradient out of one pass over the table (1, default)
int opt_score_uniform();  // rbg_score_f32: workgroups take users of one alignment class and store whole lines without shuffles (1, default)
int opt_score_tiles();  // item tiles one workgroup of rbg_score_f32 walks (0 = auto)
int opt_topk_sample();  // items the fused top-k pre-pass looks at

// One workgroup task: a row (or one segment of a split row).
constexpr int kMaxGroups = 8;

// XCD x runs row group grp[x]; it is XCD number idx[x] of the cnt[x] XCDs serving that group.
struct XcdMap {
    uint8_t grp[8], idx[8], cnt[8];
};

struct BlockTask {
    int32_t row;        // output row
    int32_t beg, end;   // CSR entry range of this segment
    int32_t seg, nseg;  // segment index / number of segments of the row (1 = unsplit)
    int32_t part_base;  // first partial-sum slot of the row (split rows only)
    int32_t ctr;        // arrival counter index of the row (split rows only)
    int32_t pad;
};
static_assert(sizeof(BlockTask) == 32, "BlockTask must stay 32 bytes (loaded as 2 x int4)");

constexpr int kPartialSlotFloats = 256;  // split-row partial slots are sized for d <= 256

// One wavefront / lane-group row: everything the kernel needs in a single 16-byte load.
struct RowDesc {
    int32_t row, beg, end, pad;
};
static_assert(sizeof(RowDesc) == 16, "RowDesc must stay 16 bytes");

// Launch plan of one row group.  Row groups are pinned to XCDs (workgroup b runs on XCD b % 8): by default two
// classes (user rows on XCDs 0-3, item rows on XCDs 4-7, so each XCD's L2 only ever holds ONE embedding table);
// with a caller-supplied node partition up to 8 groups (one community per XCD); or a single group.
struct GroupPlan {
    int32_t n_tasks, task_base;  // workgroup tasks  [task_base, task_base + n_tasks)
    int32_t n_wave, pos_wave;    // wavefront rows   desc[pos_wave  .. +n_wave)
    int32_t n_short, pos_short;  // lane-group rows  desc[pos_short .. +n_short)
};

// SELL-C-sigma plan of the column-slab propagation (sell.hip; planner sell_plan.hip, specification tests/sell_spec.py).  All device.
struct SellDev {
    int W = 0;                       // slab width (the plan serves d = 2 W)
    int32_t unit_base[2] = {0, 0};   // first unit of row class c
    int32_t n_units[2] = {0, 0};
    int32_t n_class[2] = {0, 0};     // rows of class c
    int32_t *ent = nullptr;          // [n_ent + 128][2]: {internal column * W * 4, bits of val}
    int32_t *ent0 = nullptr;         // same, column = original class-local row * 2 W * 4 (a launch that gathers row-major tables); optional
    int32_t *entc = nullptr;         // [n_ent + 256]: the offsets column of ent alone (launches of the factored chain)
    uint16_t *entc16 = nullptr;      // [n_ent + 512]: the same as slab-ROW numbers (0xffff = padding), when both classes have < 65 536 rows
    float *rs = nullptr, *irs = nullptr;  // [n_rows] each (one allocation): r_i with val_ij = r_i r_j, and 1 / r_i; the plan's numbering; optional
    int32_t *head = nullptr;         // [n_units][4]
    int32_t n_wide_units[2] = {0, 0};  // the class's leading units that belong to wide rows (rows of several units, r06)
    float *wide_part = nullptr;      // [n_wide_units total][128]: a wide unit's partial sums, one 128-float slot per unit (NS W <= 128)
    uint32_t *wide_ctr = nullptr;    // [n_wide_units total][4]: arrivals per (row's first unit, slab); zero between launches.
                                     // Scratch of the LAUNCH: a handle's launches must be stream-ordered (as for the binned kernel's
                                     // split rows); every re-weighted view has its own
    int32_t *orig = nullptr;         // [n_rows]: original node id of (class, internal row)
    int32_t *src = nullptr;          // [n_ent]: CSR entry of every slot (-1 = padding): re-weighted views refresh their values through it; optional
    int64_t n_ent = 0;
    int64_t first_ent1 = 0;          // first entry of class 1's units
    int chunk = 0;                   // the planner's chunk (0: an attached plan)
    bool native = false;             // built by rbg_graph_plan_sell (the values are the graph's own)
    bool rect = false;               // a RECTANGULAR block (r06: a shard's [owned | halo] product, sharded.py): the columns index ONE
                                     // row-major table of n_tab rows, ent holds row-major offsets (col * 2 W * 4) and ent0 aliases it;
                                     // no slab chain (entc / rs stay NULL): every launch gathers row-major
    int32_t n_tab = 0;               // rect: rows of the gathered table (= the handle's n_cols)
    const SellDev *borrowed = nullptr;  // a re-weighted view: everything but ent0 / fb0 belongs to the base graph's plan
    bool view_fresh = false;         // a view's values have been refreshed at least once (rbg_graph_refresh_values)
    float *bwd = nullptr;            // [3][n_rows][2 W] slab scratch of the backward chain WITHOUT row-major entries (allocated by the first such backward)
    int64_t bwd_floats = 0;
    std::mutex bwd_mutex;
};
void free_sell(SellDev *sw);
// sell.hip: validate the plan in sw (ent / head / orig filled), build the derived arrays (entc, ent0, first-batch blocks) and
// install it on g (sw is freed on failure)
int sell_adopt(rbg_graph *g, SellDev *sw, bool validate);
int sell_set_factors(rbg_graph *g, const float *r);  // rbg_graph_sell_set_factors without the error reset
// sell_plan.hip
int plan_sell(rbg_graph *g, int W, int chunk, bool rect = false);  // rect: the rectangular form (SellDev::rect)


}  // namespace rbg

// The graph handle.  Arrays named d_* live in HBM (device >= 0); h_* on the host.
struct rbg_graph {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int64_t n_users = -1;  // -1 when built from CSR/COO (unknown split)
    int64_t row_split = -1;  // rows [0, row_split) / [row_split, n_rows) are the two row classes of the XCD split (-1: n_users)
    int device = -1;
    uint32_t flags = 0;

    // host CSR (always present for host graphs; kept for device graphs with RBG_GRAPH_KEEP_HOST)
    std::vector<int32_t> h_rowptr;
    std::vector<int32_t> h_col;
    std::vector<float> h_val;

    // device CSR
    int32_t *d_rowptr = nullptr;
    int32_t *d_col = nullptr;
    float *d_val = nullptr;

    // degree binning (device): per row class, rows sorted by degree descending and cut into workgroup
    // rows (expanded to d_tasks), wavefront rows and lane-group rows (both described by d_desc).
    rbg::Tuning tuning{};
    int n_groups = 1;
    rbg::XcdMap xmap{};
    rbg::GroupPlan groups[rbg::kMaxGroups] = {};
    std::vector<int8_t> h_part;  // optional node -> row group (community) supplied by the caller; consumed by the plan
    int n_parts = 0;
    rbg::RowDesc *d_desc = nullptr;  // [n_wave + n_short over all groups]
    int64_t n_block_rows = 0, n_wave = 0, n_short = 0;
    rbg::BlockTask *d_tasks = nullptr;
    int64_t n_tasks = 0, n_split_rows = 0, n_partial_slots = 0;
    float *d_partials = nullptr;    // [n_partial_slots][kPartialSlotFloats]
    uint32_t *d_counters = nullptr;  // [n_split_rows], zero between launches
    int32_t max_degree = 0;
    const rbg_graph *base = nullptr;  // a re-weighted view (rbg_graph_create_reweighted): structure + plan borrowed from base,
                                      // d_val borrowed from the caller; only partials / counters are its own
    rbg::SellDev *sell = nullptr;         // optional SELL plan of the column-slab propagation (rbg_graph_plan_sell / _attach_sell)
    mutable std::atomic<int> sell_views{0};  // live re-weighted views that BORROW this handle's plan arrays: while > 0 the plan
                                             // is neither detached nor replaced (rbg_graph_detach_sell / _plan_sell / _attach_sell
                                             // return RBG_EUNSUPPORTED)
    std::string sell_note;                // why the handle has no plan (rbg_graph_sell_status)
};

namespace rbg {

// graph_build.cpp
int build_host_csr(rbg_graph *g, int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                   const int64_t *iid, const uint8_t *keep);
struct BinPlan {
    std::vector<RowDesc> desc;
    std::vector<BlockTask> tasks;
    int n_groups = 1;
    XcdMap xmap{};
    GroupPlan groups[kMaxGroups] = {};
    int64_t n_block_rows = 0, n_wave = 0, n_short = 0, n_split = 0, n_slots = 0;
    int32_t max_deg = 0;
};
int plan_bins(const rbg_graph *g, BinPlan &plan);
int upload_graph(rbg_graph *g);  // host CSR + launch plan -> device
int upload_plan(rbg_graph *g);   // launch plan from g->h_rowptr -> device (CSR already resident)
int to_device_raw(void **dst, const void *src, size_t bytes);  // hipMalloc (+ H2D copy when src != NULL)
// graph_build_device.hip
int build_device_csr(rbg_graph *g, int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                     const int64_t *iid, const uint8_t *keep);
int set_device_for(int device);

// spmm.hip — Y[n_rows, d] (row stride ldy) = Â · X (row stride ldx), optionally accumulating.
// sell.hip — the propagation over an attached SELL plan (layers = scratch [K][N][d]; the mean leaves row-major)
bool sell_applicable(const rbg_graph *g, int d);
bool sell_rowmajor_applicable(const rbg_graph *g, int d);  // ... and the plan has its row-major entries (option "sell_rowmajor")
const char *sell_kernel_name(const rbg_graph *g, int d, bool compact);
bool sell_chain_factored(const rbg_graph *g);  // the slab chains read compact entries from the second launch on
// rbg_spmm_f32 / rbg_spmm_noise_f32 (noise != NULL) over the plan
// (X: row stride ldx floats — d, or a column block of a wider buffer when sell_stride_ok; Y contiguous)
bool sell_plain_applicable(const rbg_graph *g, int d, int64_t ldx);  // the plain layer runs on the plan (row-major entries, or the slab scratch)
int sell_spmm(const rbg_graph *g, const float *X, int64_t ldx, float *Y, int d, int accumulate, const float *noise, float eps, hipStream_t s);
// rbg_spmm_mean_f32 over the plan: out = (srcs[0] + ... + srcs[n - 1] + (partial +) A X) / (n + 1), everything row-major [n_rows, d]
int sell_spmm_mean(const rbg_graph *g, const float *X, const float *partial, const float *const *srcs, int n_srcs, float *out_mean, int d, float denom, hipStream_t s);
bool sell_stride_ok(const rbg_graph *g, int d, int64_t ldx);
// every layer row-major (the caller reads `layers`, or one graph per layer)
int sell_forward_rowmajor(const rbg_graph *const *graphs, int n_graphs, const float *user_emb, const float *item_emb, float *out_mean,
                          float *layers, int d, int K, bool keep_last, hipStream_t s);
// RBG_EUNSUPPORTED: run the binned chain.  `work` = the caller's [N, d] scratch of rbg_lightgcn_backward_f32 (K >= 2)
int sell_backward(const rbg_graph *g, const float *grad_out, float *grad_e0, float *work, int d, int K, hipStream_t s);
int sell_make_view(rbg_graph *view, const rbg_graph *base);  // a re-weighted view borrows the base graph's plan (RBG_EUNSUPPORTED: none)
int sell_forward(const rbg_graph *g, const float *user_emb, const float *item_emb, float *out_mean, float *layers, int d, int K,
                 hipStream_t s);

int spmm_strided(const rbg_graph *g, const float *X, int64_t ldx, float *Y, int64_t ldy, int d, int accumulate,
                 hipStream_t s);

}  // namespace rbg
