// topk.hip — full-sort evaluation without the [B, n_items] score matrix.
//
// Replaces, for one evaluation batch: LightGCN.full_sort_predict (lightgcn.py:123-133: u = restore_user_e[user];
// scores = u @ restore_item_e.T) followed by RecBole's Trainer._full_sort_batch_eval [recbole==1.1.1]:
//   scores[:, 0] = -inf; scores[history_index] = -inf; torch.topk(scores, k)
// The history of a user is exactly its row of the training graph (the model was built from train_data.dataset,
// quick_start.py:41), so the mask is read from the graph handle's CSR.
//
// Kernel 1 (score_topk_kernel): same MFMA tiling as score.hip (wave = 32 users x 32 items).  Lane i keeps the threshold tau of
// batch slot i; after a tile ONE more MFMA forms score - tau in a second register block (TauTest), a max3 tree + one compare + one
// ballot says whether anything passes (the common case costs 12 instructions); otherwise every passing lane appends its
// (score, item) to the user's LDS list itself (capacity 24, 48 or 64, raw: no history lookups in the hot loop; an LDS atomic on the
// user's counter hands out the slot).  A full
// list is pruned to its best k valid entries by a 64-lane bitonic sort, its history items dropped by ONE 64-lane-parallel binary
// search in the user's graph row (a chain of dependent loads paid per batch, not per candidate); that raises tau.
// A workgroup covers four 32-user tiles (one per wave) x one chunk of item tiles, which it fetches once, coalesced,
// into LDS (ItemTiles); every wave hands its raw lists to a workspace.
// Kernel 2 (topk_merge_kernel): one wavefront per user compacts all partial lists and folds them 32 at a time.
// Two passes: a pre-pass over the first 8192 items (topk_prepass_kernel: per-lane running maxima, no lists) yields tau0,
// a lower bound of the final k-th best valid score, so the main pass starts selective (~k*n/8192 candidates per user
// instead of warm-up churn).
// K <= 32.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "internal.h"
#include "mfma_common.h"
#include "plane_image.h"
#include "topk_common.h"
#include "topk_screen.h"

namespace rbg {



constexpr int kListStride = 32;  // entries a wave hands over per user (workspace row)

struct TopkParams {
    const float *U;  // [n_users, d] all user embeddings (restore_user_e)
    const float *I;  // [n_items, d]
    const int64_t *users;  // [B] user ids of this batch
    const int32_t *rowptr;  // training graph CSR (may be NULL: no history mask)
    const int32_t *col;
    int64_t n_users, n_items, B;
    int d;
    int k;               // entries kept by a prune (the pre-pass keeps 32 unmasked ones)
    int filter_history;  // prunes drop history items (main pass); the pre-pass defers the mask to its merge
    int tiles_per_chunk, n_chunks;
    int64_t tile_lo, tile_hi;  // item tiles covered by this launch
    const float *tau0;         // [B] lower bound of each user's k-th best valid score (NULL: -inf)
    float *w_val;              // workspace [B][n_chunks*4][kListStride]
    int32_t *w_idx;
    int32_t *w_cnt;            // [B][n_chunks*4]
    const char *image;         // r06: the item table as bf16 planes, tile by tile in the LDS layout (plane_image.h); NULL: fetch + split per workgroup
};

// r06 — the PLANE IMAGE.  Every workgroup of the main pass and of the pre-pass used to fetch each 32-item tile as fp32, split it
// into three bf16 planes and publish them to LDS: per tile and wave ~ 1 000 cycles of fetch issue + ~ 800 of split / LDS writes
// (profiles/r05_topk_clock.jsonl) — 40 % of the loop once the filter had become cheap (r05) — repeated by all 32 workgroup rows
// of a 4 096-user call on the same 10 MB table.  Now one small kernel splits the table ONCE per call into an image that IS the
// LDS layout (plane_image.h), and a workgroup takes a tile by LDS-DMA, one tile ahead.
// Same planes, same products: bit-identical results.  (r02 had tried this when the filter and the list code dominated the loop:
// no gain then.)
// Prune one user's LDS list (n <= 64 raw candidates) to its best k valid entries (sorted; `kept` of them).  Returns the new
// threshold: the k-th best (or -inf while fewer than k exist).
__device__ __forceinline__ float prune_list(const TopkParams &p, int64_t user, float *lv, int *li, int n, int lane, int &kept) {
    __builtin_amdgcn_wave_barrier();  // lists are handed between lanes of ONE wave through LDS: keep program order
    float v = lane < n ? lv[lane] : kNegInf;
    int idx = lane < n ? li[lane] : 0x7fffffff;
    if (p.filter_history && lane < n && in_history(p.rowptr, p.col, p.n_users, user, idx)) {
        v = kNegInf;
        idx = 0x7fffffff;
    }
    wave_sort_desc(v, idx, lane);
    const int valid = __popcll(__ballot(idx != 0x7fffffff));
    if (lane < p.k) {
        lv[lane] = v;
        li[lane] = idx;
    }
    kept = valid < p.k ? valid : p.k;
    __builtin_amdgcn_wave_barrier();
    return __shfl(v, p.k - 1);
}

// compile-time loop over the 16 accumulator rows: tau[] / acc[] must be indexed statically or they
// are demoted to scratch memory (measured: 144 B/lane of scratch and a 1.4 ms kernel)
template <int R>
struct RowLoop {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, R>{});
        RowLoop<R + 1>::run(f);
    }
};
template <>
struct RowLoop<16> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

// Shared by both passes.  A workgroup = 4 waves = 4 tiles of 32 batch users (A fragments in registers); all four walk
// the SAME item tiles, each fetched once per workgroup into a double-buffered LDS tile (mfma_common.h::RowTile).
// SPLIT: the tile is published as three bf16 planes and the product runs on the bf16 matrix cores (mfma_common.h: fp32
// accuracy at 2.3x the rate of the exact-fp32 MFMA); both passes use the same product, so a pair scores identically in both.
template <int NCHUNK, bool VEC, bool SPLIT>
using ItemTiles = std::conditional_t<SPLIT, RowTile3<NCHUNK, (VEC ? RUN_VEC : RUN_ANY)>, RowTile<NCHUNK, (VEC ? RUN_VEC : RUN_ANY)>>;
template <int NCHUNK, bool VEC, bool SPLIT>
using ItemTileMem = std::conditional_t<SPLIT, typename RowTile3<NCHUNK, (VEC ? RUN_VEC : RUN_ANY)>::Planes, float[32][NCHUNK * 64 + 4]>;

// Per-wave phase clock (devtools/microbench/topk_trace.hip builds this file with RBG_TOPK_TRACE; the product does not).
#ifdef RBG_TOPK_TRACE
__device__ unsigned long long *g_topk_trace = nullptr;
__device__ int g_topk_debug = 0;  // what-if switches of the diagnostic build (results are wrong on purpose): 1 no product, 2 no filter, 4 no fetch / publish
#define RBG_TOPK_DBG(bit) ((g_topk_debug & (bit)) != 0)
#define RBG_TOPK_T0() unsigned long long tk_last = clock64(), tk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RBG_TOPK_LAP(k)                              \
    do {                                             \
        const unsigned long long tk_now = clock64(); \
        tk_acc[k] += tk_now - tk_last;               \
        tk_last = tk_now;                            \
    } while (0)
#define RBG_TOPK_DUMP()                                                                                             \
    do {                                                                                                            \
        if (g_topk_trace && lane == 0)                                                                              \
            for (int k = 0; k < 8; ++k)                                                                             \
                g_topk_trace[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + k] = tk_acc[k];      \
    } while (0)
#else
#define RBG_TOPK_DBG(bit) false
#define RBG_TOPK_T0() ((void)0)
#define RBG_TOPK_LAP(k) ((void)0)
#define RBG_TOPK_DUMP() ((void)0)
#endif

// r05: the threshold test as ONE MORE MFMA.  tst = acc + (-tau) x 1 through three k-slots of a seventh / twenty-fifth product: the
// A operand of user row i carries the three bf16 terms of -tau_i (lane half 0, slots 0-2; everything else zero), the B operand a
// one in the same slots, and D is a second register block, so acc keeps the pure score.  The per-tile filter is then a v_max3
// tree over tst (8 instructions) + one compare + one ballot instead of 16 x {compare, select, or} against 16 threshold registers
// — every vector instruction of this kernel issues beside another wave's MFMA stream and costs 4 SIMD cycles, the extra MFMA 32.
// The comparison is made conservative (tau lowered by 8 ulp: the matrix core's summation order is its own), so everything that
// reaches the old test's threshold still passes; an entry that passes from just below it is dropped by the prunes / the merge
// like any other non-winner.  -inf / +inf thresholds become -/+ 3e38.
template <bool SPLIT>
struct TauTest;
template <>
struct TauTest<true> {
    bf16x8 a, one;
    __device__ __forceinline__ void init(int h) {
        const __bf16 z = (__bf16)0.0f, o = (__bf16)1.0f;
        a = bf16x8{z, z, z, z, z, z, z, z};
        one = h == 0 ? bf16x8{o, o, o, z, z, z, z, z} : bf16x8{z, z, z, z, z, z, z, z};
    }
    __device__ __forceinline__ void set(float tau, int h) {
        float t = fminf(fmaxf(tau, -3.0e38f), 3.0e38f);
        t = fabsf(t) * 4.8e-7f - t;  // -(tau - 8 ulp)
        bf16x2 hh, mm, ll;
        split2_bf16(t, 0.f, hh, mm, ll);
        if (h == 0) {
            a[0] = hh[0];
            a[1] = mm[0];
            a[2] = ll[0];
        }
    }
    __device__ __forceinline__ f32x16 run(const f32x16 &acc) const { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, one, acc, 0, 0, 0); }
};
template <>
struct TauTest<false> {  // the exact-fp32 chain: one v_mfma_f32_32x32x2_f32 (lane half h supplies k = h)
    float a, one;
    __device__ __forceinline__ void init(int h) {
        a = 0.f;
        one = h == 0 ? 1.f : 0.f;
    }
    __device__ __forceinline__ void set(float tau, int h) {
        const float t = fminf(fmaxf(tau, -3.0e38f), 3.0e38f);
        if (h == 0) a = fabsf(t) * 4.8e-7f - t;
    }
    __device__ __forceinline__ f32x16 run(const f32x16 &acc) const { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, one, acc, 0, 0, 0); }
};

// CAP = list capacity per user: a prune leaves <= k entries and one tile adds <= 32, so CAP >= k + 32 (48 for k <= 16,
// which lets two workgroups share a CU's LDS; 64 otherwise).
// (r02, tried: taking the item tiles by LDS-DMA from a "plane image" of the item table — the three bf16 planes in the LDS
//  layout, built once per call — instead of fetch + split + publish in every workgroup.  Bit-identical results, but no
//  faster: 231.0 vs 229.5 us at d = 64, 437 vs 457 us at d = 128 (4096 users), +13 us at 128 users for the extra pass: the
//  fetch and the publish were already hidden behind the other resident workgroup's product and filter.  Removed.)
template <int NCHUNK, bool VEC, int CAP, bool SPLIT, bool IMG = false>
__global__ __launch_bounds__(256, (CAP < 48 ? 3 : 1)) void score_topk_kernel(const TopkParams p) {
    static_assert(!IMG || (SPLIT && NCHUNK <= 2), "the plane image serves the split product at d <= 128");
    using Tiles = ItemTiles<NCHUNK, VEC, SPLIT>;
    __shared__ __attribute__((aligned(16))) ItemTileMem<NCHUNK, VEC, SPLIT> s_it[2];
    __shared__ float l_val[4][32][CAP];
    __shared__ int l_idx[4][32][CAP];
    __shared__ int l_cnt[4][32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform, and known to be)
    const int i = lane & 31, h = lane >> 5;
    const int64_t b0 = ((int64_t)blockIdx.y * 4 + wave) * 32;  // first batch slot of this wave's user tile
    const bool wave_live = b0 < p.B;                            // an idle wave still fetches and meets the barriers
    if (lane < 32) l_cnt[wave][lane] = 0;  // entries in the list of batch slot b0 + lane (LDS atomics hand out the slots)
    // A operand: the embedding row of batch slot b0 + i
    const int64_t bi = b0 + i;
    const int64_t my_user = bi < p.B ? p.users[bi] : -1;
    float a[NCHUNK][32];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        load_run32f<(VEC ? RUN_VEC : RUN_ANY)>(p.U + (my_user < 0 ? 0 : my_user) * (int64_t)p.d, my_user >= 0, c * 64 + h * 32, p.d, a[c]);
    std::conditional_t<SPLIT, AFrag3<NCHUNK>, int> a3;
    if constexpr (SPLIT) split_a(a, a3);
    auto tile_product = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (SPLIT) return Tiles::product(s_it[buf], a3, i, h);
        else return Tiles::product(s_it[buf], a, i, h);
    };
    // thresholds start at the pre-pass bound: an item scoring below the k-th best valid score of ANY item subset
    // cannot be in the top k.  Lane i (both halves) keeps the threshold of batch slot b0 + i; a slot past the end never passes.
#ifdef RBG_TOPK_TRACE_NOPASS  // (diagnostic build only: nothing ever passes the threshold)
    float my_tau = __builtin_inff();
#else
    float my_tau = bi < p.B ? (p.tau0 ? p.tau0[bi] : kNegInf) : __builtin_inff();
#endif
    if (RBG_TOPK_DBG(8)) my_tau = __builtin_inff();
    TauTest<SPLIT> tt;
    tt.init(h);
    tt.set(my_tau, h);
    const int64_t t_begin = p.tile_lo + (int64_t)blockIdx.x * p.tiles_per_chunk;
    const int64_t t_end = (t_begin + p.tiles_per_chunk < p.tile_hi) ? t_begin + p.tiles_per_chunk : p.tile_hi;

    const bool opt_prio = !RBG_TOPK_DBG(16);
    RBG_TOPK_T0();
    // filter one finished tile: acc[r] is the score of (user row (r&3)+8(r>>2)+4h, item); the PAD item never qualifies
    auto filter_tile = [&](const f32x16 &acc, const int64_t item) __attribute__((always_inline)) {
        // ONE branch per tile in the common case (r04's form — a compare, a ballot and a branch per row — cost 1 320 cycles per tile
        // with NOTHING passing; r05 first collected per-lane row masks with 48 vector instructions).  Now: tst = score - tau from the
        // matrix core (TauTest), a max3 tree, one compare, one ballot; only a tile with a passing entry builds the row masks and
        // visits the rows that have one (their union comes from the few lanes that do).  Same visiting order as before.
        const bool item_ok = item < p.n_items && item != 0;
        const f32x16 tst = tt.run(acc);
        const float m01 = fmaxf(fmaxf(tst[0], tst[1]), tst[2]), m02 = fmaxf(fmaxf(tst[3], tst[4]), tst[5]);
        const float m03 = fmaxf(fmaxf(tst[6], tst[7]), tst[8]), m04 = fmaxf(fmaxf(tst[9], tst[10]), tst[11]);
        const float m05 = fmaxf(fmaxf(tst[12], tst[13]), tst[14]);
        const float mx = fmaxf(fmaxf(fmaxf(fmaxf(m01, m02), m03), fmaxf(fmaxf(m04, m05), tst[15])), -1.f);  // (NaN rows never pass)
        const unsigned long long any = __builtin_amdgcn_ballot_w64(item_ok && mx >= 0.f);
        RBG_TOPK_LAP(3);
        if (any == 0ull) return;  // the common case once the thresholds have risen
        unsigned bits = 0u;
        RowLoop<0>::run([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            bits |= (tst[r] >= 0.f) ? (1u << r) : 0u;
        });
        if (!item_ok) bits = 0u;
        // r05, the appends: every lane with a passing entry serves ITSELF — the score out of its accumulator block by a select tree,
        // the slot from an LDS atomic on the user's counter, two LDS writes; no ballot, no readlane, no scalar branch per row.  (The
        // per-row form — one ballot + two counter reads + ranks per row — cost ~ 1 000 cycles per passing tile for ~ 100
        // instructions: a chain of vector -> scalar -> vector hops, each waiting for a gap in the other waves' MFMA streams;
        // 237 -> 221 us per call in the diagnostic build.)  A lane that finds its list full stops and keeps its bits.
        unsigned pend = bits;
        while (pend != 0u) {  // (divergent: one or two rounds)
            const int r = __builtin_ctz(pend);
            const bool r0 = r & 1, r1 = r & 2, r2 = r & 4, r3 = r & 8;
            const float t0 = r0 ? acc[1] : acc[0], t1 = r0 ? acc[3] : acc[2], t2 = r0 ? acc[5] : acc[4], t3 = r0 ? acc[7] : acc[6];
            const float t4 = r0 ? acc[9] : acc[8], t5 = r0 ? acc[11] : acc[10], t6 = r0 ? acc[13] : acc[12], t7 = r0 ? acc[15] : acc[14];
            const float u0 = r1 ? t1 : t0, u1 = r1 ? t3 : t2, u2 = r1 ? t5 : t4, u3 = r1 ? t7 : t6;
            const float v0 = r2 ? u1 : u0, v1 = r2 ? u3 : u2;
            const float sv = r3 ? v1 : v0;
            const int lu = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pos = atomicAdd(&l_cnt[wave][lu], 1);
            if (pos >= CAP) break;  // full: this entry and the lane's later ones stay pending
            l_val[wave][lu][pos] = sv;
            l_idx[wave][lu][pos] = (int)item;
            pend &= pend - 1u;
        }
        const unsigned long long left = __builtin_amdgcn_ballot_w64(pend != 0u);
        if (left == 0ull) return;  // the common case: every list had room
        // A list overflowed (rare: the thresholds start at the pre-pass bound; always in the first tiles when they do not).  The failed
        // attempts pushed counters past CAP: clamp them, then the pending entries go row by row through the fill / prune / refill
        // code below (one looped copy: the score leaves the accumulator block by a register-indexed move).
        if (lane < 32 && l_cnt[wave][lane] > CAP) l_cnt[wave][lane] = CAP;
        __builtin_amdgcn_wave_barrier();
        unsigned rows = 0u;
        for (unsigned long long m = left; m; m &= m - 1ull) rows |= (unsigned)__builtin_amdgcn_readlane((int)pend, __builtin_ctzll(m));
#pragma clang loop unroll(disable)
        while (rows != 0u) {
            const int r = __builtin_ctz(rows);
            rows &= rows - 1u;
            const float s = acc[r];
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(((pend >> r) & 1u) != 0u);
            const int lu0 = (r & 3) + 8 * (r >> 2);
#pragma clang loop unroll(disable)
            for (int hh = 0; hh < 2; ++hh) {  // the two lane halves hold two different users: one half at a time
                unsigned rem = hh ? (unsigned)(mask >> 32) : (unsigned)mask;
                if (rem == 0u) continue;
                const int lu = lu0 + 4 * hh;  // user slot inside the tile
                float *lv = l_val[wave][lu];
                int *li = l_idx[wave][lu];
                int base = __builtin_amdgcn_readfirstlane(l_cnt[wave][lu]);
                // fill the list to CAP, prune it to its best k valid entries (that raises the user's threshold), go on with the rest
                while (true) {
                    const int add = __popc(rem);
                    const bool mine = h == hh && ((rem >> i) & 1u);
                    const int rank = __popc(rem & ((1u << i) - 1u));
                    if (base + add <= CAP) {
                        if (mine) {
                            lv[base + rank] = s;
                            li[base + rank] = (int)item;
                        }
                        base += add;
                        break;
                    }
                    const int room = CAP - base;
                    if (mine && rank < room) {
                        lv[base + rank] = s;
                        li[base + rank] = (int)item;
                    }
                    for (int q = 0; q < room; ++q) rem &= rem - 1u;  // those lanes are in
                    const float nt = prune_list(p, __shfl(my_user, lu), lv, li, CAP, lane, base);
                    if (i == lu) {  // (both halves keep the value; lane half 0 carries the operand)
                        my_tau = fmaxf(my_tau, nt);
                        tt.set(my_tau, h);
                    }
                }
                if (lane == 0) l_cnt[wave][lu] = base;
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    Tiles tiles;
    const unsigned lds_it[2] = {(unsigned)(uintptr_t)&s_it[0], (unsigned)(uintptr_t)&s_it[1]};  // (LDS byte addresses: the DMA's destination)
    if (t_begin < t_end) {
        if constexpr (IMG) {
            PlaneImage<NCHUNK>::dma(p.image, t_begin, lds_it[0], tid, wave);
            dma_drain();
        } else {
            tiles.fetch(p.I, p.d, p.n_items, p.d, t_begin, tid);
            tiles.publish(s_it[0], tid);
        }
    }
    __syncthreads();
    RBG_TOPK_LAP(0);
    for (int64_t t = t_begin; t < t_end; ++t) {
        const int buf = (int)(t - t_begin) & 1;
        if (t + 1 < t_end && !RBG_TOPK_DBG(4)) {
            // tile t + 1 into the other buffer (last read before the previous barrier), in flight while this tile feeds the matrix core
            if constexpr (IMG) PlaneImage<NCHUNK>::dma(p.image, t + 1, lds_it[buf ^ 1], tid, wave);
            else tiles.fetch(p.I, p.d, p.n_items, p.d, t + 1, tid);
        }
        RBG_TOPK_LAP(1);
        if (wave_live) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!RBG_TOPK_DBG(1)) acc = tile_product(buf);
            RBG_TOPK_LAP(2);
            // everything between the product and the barrier is on the workgroup's critical path and made of short dependent
            // vector <-> scalar chains: at raised priority its instructions win the issue slots between the other waves' MFMAs
            if (opt_prio) __builtin_amdgcn_s_setprio(2);  // (r05: 241 -> 236 us per call in the diagnostic build; priority 3, or the fetch raised as well: no better / worse)
            if (!(RBG_TOPK_DBG(2) && acc[0] != 12345.678f)) filter_tile(acc, t * 32 + i);
        }
        RBG_TOPK_LAP(4);
        if constexpr (IMG) dma_drain();  // (this wave's requests have landed; the barrier makes that true of all four)
        else if (t + 1 < t_end && !RBG_TOPK_DBG(4)) tiles.publish(s_it[buf ^ 1], tid);  // the other buffer was last read before the previous barrier
        RBG_TOPK_LAP(5);
        __syncthreads();
        if (opt_prio) __builtin_amdgcn_s_setprio(0);
        RBG_TOPK_LAP(6);
    }
    RBG_TOPK_DUMP();
    if (!wave_live) return;
    // hand the lists over: raw candidates (at most kListStride per user; longer lists are pruned first)
    const int lists = p.n_chunks;
    const int my_list = (int)blockIdx.x;
    for (int lu = 0; lu < 32; ++lu) {
        const int64_t b = b0 + lu;
        if (b >= p.B) break;
        int n = __builtin_amdgcn_readfirstlane(l_cnt[wave][lu]);
        if (n > kListStride) prune_list(p, __shfl(my_user, lu), l_val[wave][lu], l_idx[wave][lu], n, lane, n);
        __builtin_amdgcn_wave_barrier();
        const int64_t off = (b * lists + my_list) * kListStride;
        if (lane < n) {
            p.w_val[off + lane] = l_val[wave][lu][lane];
            p.w_idx[off + lane] = l_idx[wave][lu][lane];
        }
        if (lane == 0) p.w_cnt[b * lists + my_list] = n;
    }
}

// Pre-pass: a lower bound tau0 of every user's k-th best VALID score from the first `tile_hi` item tiles, with no
// lists at all.  Each lane keeps, per accumulator row, the running maximum (and its item) over the tiles it sees:
// for one user that is 32 "group maxima" (group = items sharing a lane index), all distinct items.  The k-th largest
// group maximum that is neither PAD nor a history item is <= the k-th best valid score overall, and with k << 32 it is
// nearly as tight as the exact k-th best of the sample.  Same workgroup shape as the main pass (4 user tiles sharing the
// item tiles of one split of the sample); topk_tau_kernel folds the splits and selects.
template <int NCHUNK, bool VEC, bool SPLIT, bool IMG = false>
__global__ __launch_bounds__(256) void topk_prepass_kernel(const TopkParams p, float *__restrict__ g_val,
                                                           int32_t *__restrict__ g_idx) {
    static_assert(!IMG || (SPLIT && NCHUNK <= 2), "the plane image serves the split product at d <= 128");
    using Tiles = ItemTiles<NCHUNK, VEC, SPLIT>;
    __shared__ __attribute__((aligned(16))) ItemTileMem<NCHUNK, VEC, SPLIT> s_it[2];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform, and known to be)
    const int i = lane & 31, h = lane >> 5;
    const int64_t b0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
    const bool wave_live = b0 < p.B;
    const int64_t bi = b0 + i;
    const int64_t my_user = bi < p.B ? p.users[bi] : -1;
    float a[NCHUNK][32];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        load_run32f<(VEC ? RUN_VEC : RUN_ANY)>(p.U + (my_user < 0 ? 0 : my_user) * (int64_t)p.d, my_user >= 0, c * 64 + h * 32, p.d, a[c]);
    std::conditional_t<SPLIT, AFrag3<NCHUNK>, int> a3;
    if constexpr (SPLIT) split_a(a, a3);
    auto tile_product = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (SPLIT) return Tiles::product(s_it[buf], a3, i, h);
        else return Tiles::product(s_it[buf], a, i, h);
    };
    float best_v[16];
    int best_i[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        best_v[r] = kNegInf;
        best_i[r] = 0x7fffffff;
    }
    const int64_t t_begin = p.tile_lo + (int64_t)blockIdx.x * p.tiles_per_chunk;
    const int64_t t_end = (t_begin + p.tiles_per_chunk < p.tile_hi) ? t_begin + p.tiles_per_chunk : p.tile_hi;
    Tiles tiles;
    const unsigned lds_it[2] = {(unsigned)(uintptr_t)&s_it[0], (unsigned)(uintptr_t)&s_it[1]};
    if (t_begin < t_end) {
        if constexpr (IMG) {
            PlaneImage<NCHUNK>::dma(p.image, t_begin, lds_it[0], tid, wave);
            dma_drain();
        } else {
            tiles.fetch(p.I, p.d, p.n_items, p.d, t_begin, tid);
            tiles.publish(s_it[0], tid);
        }
    }
    __syncthreads();
    for (int64_t t = t_begin; t < t_end; ++t) {
        const int buf = (int)(t - t_begin) & 1;
        if (t + 1 < t_end) {
            if constexpr (IMG) PlaneImage<NCHUNK>::dma(p.image, t + 1, lds_it[buf ^ 1], tid, wave);
            else tiles.fetch(p.I, p.d, p.n_items, p.d, t + 1, tid);
        }
        if (wave_live) {
            const f32x16 acc = tile_product(buf);
            if (!RBG_TOPK_DBG(16)) __builtin_amdgcn_s_setprio(2);
            const int64_t item = t * 32 + i;
            const bool item_ok = item < p.n_items && item != 0;
            RowLoop<0>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float sc = acc[r];
                if (item_ok && sc > best_v[r]) {
                    best_v[r] = sc;
                    best_i[r] = (int)item;
                }
            });
        }
        if constexpr (IMG) dma_drain();
        else if (t + 1 < t_end) tiles.publish(s_it[buf ^ 1], tid);
        __syncthreads();
        if (!RBG_TOPK_DBG(16)) __builtin_amdgcn_s_setprio(0);
    }
    if (!wave_live) return;
    // lane (i, h) holds group i of user slot (r&3) + 8(r>>2) + 4h for r = 0..15
    RowLoop<0>::run([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int64_t b = b0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (b < p.B) {
            const int64_t off = (b * p.n_chunks + blockIdx.x) * 32 + i;
            g_val[off] = best_v[r];
            g_idx[off] = best_i[r];
        }
    });
}

// tau0[b] = the k-th largest group maximum of user b that is not a history item (-inf if fewer than k).  The maxima of
// the same lane slot in different splits are different items, but only the slot maximum is needed for a valid bound
// with 32 candidates; one wavefront per user.
__global__ __launch_bounds__(256) void topk_tau_kernel(const float *__restrict__ g_val, const int32_t *__restrict__ g_idx,
                                                       const int64_t *__restrict__ users, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col, int64_t n_users, int64_t B, int splits,
                                                       int k, float *__restrict__ tau_out) {
    __shared__ int s_hist[4][kHistStage];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    HistRow hist;
    hist.stage(rowptr, col, n_users, users[b], s_hist[wave], lane);  // (its loads go out with the maxima's)
    // lane (i, half) folds the splits of its parity, four loads in flight (r05: one at a time it was a chain of `splits` latencies)
    float v = kNegInf;
    int idx = 0x7fffffff;
    const int i = lane & 31, half = lane >> 5;
    for (int sp = half; sp < splits; sp += 8) {
        float ov[4];
        int oi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = sp + 2 * u;
            ov[u] = q < splits ? g_val[(b * splits + q) * 32 + i] : kNegInf;
            oi[u] = q < splits ? g_idx[(b * splits + q) * 32 + i] : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (better(ov[u], oi[u], v, idx)) v = ov[u], idx = oi[u];
    }
    {
        const float pv = __shfl_xor(v, 32);
        const int pi = __shfl_xor(idx, 32);
        if (better(pv, pi, v, idx)) v = pv, idx = pi;
        if (half) v = kNegInf, idx = 0x7fffffff;  // the 32 slot maxima live in the lower half
    }
    wave_sort_desc(v, idx, lane);
    const bool ok = lane < 32 && idx != 0x7fffffff && !hist.has(idx);
    const unsigned long long m = __ballot(ok);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0 && __popcll(m) < k) tau_out[b] = kNegInf;
    if (ok && rank == k - 1) tau_out[b] = v;
}

// One wavefront per batch user.  Stage A compacts the raw candidates of `lists` partial lists into LDS; stage B takes
// them 32 at a time into the upper half-wave, masks history items (one 32-lane-parallel binary search per batch) and
// bitonic-merges them into the running best 32 held by the lower half-wave.
// tau_out != NULL (pre-pass): no mask while merging; afterwards the k-th VALID one of the 32 best becomes the bound.
// r05: 1024 staged entries per wave (it was 2048: 64 KB of LDS per workgroup = 8 resident waves per CU and two rounds of them for
// 4096 users, each wave a chain of dependent loads — 28 us); history items are masked while they are copied (every lane searches
// the user's graph row — its head staged in LDS, HistRow — for its own entries, once), not once per batch of 32 in the merge loop.
constexpr int kStage = 1024;
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ w_val, const int32_t *__restrict__ w_idx,
                                                        const int32_t *__restrict__ w_cnt, const int64_t *__restrict__ users,
                                                        const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        int64_t n_users, int64_t B, int lists, int k,
                                                        float *__restrict__ out_val, int64_t *__restrict__ out_idx,
                                                        float *__restrict__ tau_out) {
    __shared__ float s_val[4][kStage];
    __shared__ int s_idx[4][kStage];
    __shared__ int s_hist[4][kHistStage];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int64_t user = users[b];
    const bool mask_now = (tau_out == nullptr);
    // three dependent round trips instead of five: {user, counts} -> {row bounds, first entries} -> {the row's head}
    const int c_first = (lane >> 1) < lists ? w_cnt[b * lists + (lane >> 1)] : 0;
    HistRow hist;
    hist.begin(rowptr, col, n_users, user, s_hist[wave]);
    const int64_t src_first = (b * lists + (lane >> 1)) * kListStride + (lane & 1);
    const float pre_v = (lane & 1) < c_first ? w_val[src_first] : kNegInf;
    const int pre_i = (lane & 1) < c_first ? w_idx[src_first] : 0x7fffffff;
    hist.finish(s_hist[wave], lane);
    float v = kNegInf;  // lanes 0..31: best so far; lanes 32..63: incoming batch
    int idx = 0x7fffffff;
    int total = 0;
    auto flush = [&]() {
        for (int base = 0; base < total; base += 32) {
            __builtin_amdgcn_wave_barrier();
            if (lane >= 32) {
                const int e = base + lane - 32;
                v = e < total ? s_val[wave][e] : kNegInf;
                idx = e < total ? s_idx[wave][e] : 0x7fffffff;
            }
            wave_sort_desc(v, idx, lane);
            if (lane >= 32) {
                v = kNegInf;
                idx = 0x7fffffff;
            }
        }
        total = 0;
    };
    for (int l0 = 0; l0 < lists; l0 += 32) {
        // 32 lists per round (at most 32 x 32 = kStage entries; the usual 24 lists: ONE round): lane pair (2j, 2j + 1) copies list
        // l0 + j (entries e = lane & 1, + 2, ...)
        const int l = l0 + (lane >> 1);
        const int c = l0 == 0 ? c_first : (l < lists ? w_cnt[b * lists + l] : 0);
        // exclusive prefix of the counts over the 32 lists (each list appears on two lanes: count it once)
        int incl = (lane & 1) ? 0 : c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        const int round_total = __shfl(incl, 63);
        const int my_off = __shfl(incl, lane | 1) - c;  // start of this list inside the round
        if (total + round_total > kStage) flush();
        const int64_t src = (b * lists + l) * kListStride;
        for (int e = lane & 1; e < c; e += 2) {
            const bool first = l0 == 0 && e == (lane & 1);  // (requested above)
            float ev = first ? pre_v : w_val[src + e];
            int ei = first ? pre_i : w_idx[src + e];
            if (mask_now && hist.has(ei)) {
                ev = kNegInf;
                ei = 0x7fffffff;
            }
            s_val[wave][total + my_off + e] = ev;
            s_idx[wave][total + my_off + e] = ei;
        }
        total += round_total;
    }
    flush();
    if (tau_out) {
        // lanes 0..31 hold the 32 best UNMASKED sample scores; the k-th one that is not a history item bounds the
        // user's k-th best valid score from below (-inf if the sample cannot tell)
        const bool ok = lane < 32 && idx != 0x7fffffff && !hist.has(idx);
        const unsigned long long m = __ballot(ok);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0 && __popcll(m) < k) tau_out[b] = kNegInf;
        if (ok && rank == k - 1) tau_out[b] = v;
        return;
    }
    if (lane < k) {
        out_val[b * k + lane] = v;
        out_idx[b * k + lane] = (idx == 0x7fffffff) ? -1 : (int64_t)idx;
    }
}

// Chunk the item tiles so that a launch has about `want_blocks` workgroups (two resident per CU), at least
// `min_tiles` tiles per chunk and at most `max_chunks` chunks (= partial lists per user).
static void topk_geometry(int64_t B, int64_t n_tiles, int64_t want_blocks, int64_t min_tiles, int64_t max_chunks,
                          int &tiles_per_chunk, int &n_chunks) {
    const int64_t user_blocks = std::max<int64_t>(1, (B + 127) / 128);
    int64_t want = std::max<int64_t>(1, want_blocks / user_blocks);
    want = std::min<int64_t>(want, std::max<int64_t>(1, n_tiles / min_tiles));
    want = std::min<int64_t>(want, max_chunks);
    tiles_per_chunk = (int)((n_tiles + want - 1) / want);
    n_chunks = (int)((n_tiles + tiles_per_chunk - 1) / std::max(tiles_per_chunk, 1));
}

template <int NCHUNK, int CAP>
static void launch_topk(const TopkParams &p, bool vec, dim3 grid, hipStream_t s) {
    // three bf16 planes of a d = 256 tile (2 x 50.7 KB) plus 64-entry lists (64 KB) exceed the 160 KB of LDS: that one
    // combination (k > 16 at d > 128) stays on the exact-fp32 MFMA; the pre-pass uses the same product as its main pass
    constexpr bool kSplitFits = !(NCHUNK == 4 && CAP > 48);
    const bool split = kSplitFits && opt_mfma_split() != 0;
    if constexpr (kSplitFits) {
        if constexpr (NCHUNK <= 2) {
            if (split && p.image) {
                hipLaunchKernelGGL((score_topk_kernel<NCHUNK, true, CAP, true, true>), grid, dim3(256), 0, s, p);
                return;
            }
        }
        if (vec && split) {
            hipLaunchKernelGGL((score_topk_kernel<NCHUNK, true, CAP, true>), grid, dim3(256), 0, s, p);
            return;
        }
        if (split) {
            hipLaunchKernelGGL((score_topk_kernel<NCHUNK, false, CAP, true>), grid, dim3(256), 0, s, p);
            return;
        }
    }
    if (vec)
        hipLaunchKernelGGL((score_topk_kernel<NCHUNK, true, CAP, false>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((score_topk_kernel<NCHUNK, false, CAP, false>), grid, dim3(256), 0, s, p);
}

template <int NCHUNK>
static void launch_prepass(const TopkParams &p, bool vec, dim3 grid, float *gv, int32_t *gi, hipStream_t s) {
    const bool split = opt_mfma_split() != 0 && !(NCHUNK == 4 && p.k > 16);  // the same product as the main pass (launch_topk)
    if constexpr (NCHUNK <= 2) {
        if (split && p.image) {
            hipLaunchKernelGGL((topk_prepass_kernel<NCHUNK, true, true, true>), grid, dim3(256), 0, s, p, gv, gi);
            return;
        }
    }
    if (vec && split)
        hipLaunchKernelGGL((topk_prepass_kernel<NCHUNK, true, true>), grid, dim3(256), 0, s, p, gv, gi);
    else if (vec)
        hipLaunchKernelGGL((topk_prepass_kernel<NCHUNK, true, false>), grid, dim3(256), 0, s, p, gv, gi);
    else if (split)
        hipLaunchKernelGGL((topk_prepass_kernel<NCHUNK, false, true>), grid, dim3(256), 0, s, p, gv, gi);
    else
        hipLaunchKernelGGL((topk_prepass_kernel<NCHUNK, false, false>), grid, dim3(256), 0, s, p, gv, gi);
}

static void launch_prepass_d(const TopkParams &p, bool vec, dim3 grid, float *gv, int32_t *gi, hipStream_t s) {
    if (p.d <= 64) launch_prepass<1>(p, vec, grid, gv, gi, s);
    else if (p.d <= 128) launch_prepass<2>(p, vec, grid, gv, gi, s);
    else launch_prepass<4>(p, vec, grid, gv, gi, s);
}

template <int CAP>
static void launch_topk_c(const TopkParams &p, bool vec, dim3 grid, hipStream_t s) {
    if (p.d <= 64) launch_topk<1, CAP>(p, vec, grid, s);
    else if (p.d <= 128) launch_topk<2, CAP>(p, vec, grid, s);
    else launch_topk<4, CAP>(p, vec, grid, s);
}

// 24-entry lists (r04): 52.7 KB of LDS and 168 registers per workgroup = three resident workgroups per CU instead of two.
// Measured (devtools/topk_probe.py short, interleaved, identical results): 4096 users 223.5 -> 212.5 us, 1024 users equal,
// 128 users 77.9 -> 80.4 us: on from 2048 users (option "topk_short_lists": 0 = never, 2 = always).
static bool topk_short_lists(int k, int d, int64_t B) {
    const int o = opt_topk_short_lists();
    return k <= 12 && d <= 64 && opt_mfma_split() != 0 && (o == 2 || (o == 1 && B >= 2048));
}

static void launch_topk_d(const TopkParams &p, bool vec, dim3 grid, hipStream_t s) {
    if (topk_short_lists(p.k, p.d, p.B)) {
        if (p.image) hipLaunchKernelGGL((score_topk_kernel<1, true, 24, true, true>), grid, dim3(256), 0, s, p);
        else if (vec) hipLaunchKernelGGL((score_topk_kernel<1, true, 24, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((score_topk_kernel<1, false, 24, true>), grid, dim3(256), 0, s, p);
        return;
    }
    if (p.k <= 16) launch_topk_c<48>(p, vec, grid, s);
    else launch_topk_c<64>(p, vec, grid, s);
}

// workspace: main lists (val, idx: B*nc*32 each; cnt: B*nc), pre-pass group maxima (val, idx: B*max_splits*32 each),
// tau0 [B].  Independent of the "topk_sample" option.
struct TopkLayout {
    int tpc, nc, max_splits, splits, tpc_s;
    int64_t n_tiles, sample_tiles, main_lists, bytes;
    int64_t image_off, image_bytes;  // the plane image (sized for d <= 128; 0 when it would pass kImageCap: the per-workgroup fetch then)
};
constexpr int64_t kImageCap = 1ll << 30;
static TopkLayout topk_layout(int64_t B, int64_t n_items, int64_t want_blocks) {
    TopkLayout L{};
    L.n_tiles = (n_items + 31) / 32;
    L.sample_tiles = std::min<int64_t>(opt_topk_sample() / 32, L.n_tiles);
    topk_geometry(B, L.n_tiles, want_blocks, 8, 128, L.tpc, L.nc);
    L.main_lists = B * (int64_t)L.nc;
    // the pre-pass splits its sample over up to 64 workgroups per 128 users (aim: ~512 workgroups, >= 2 tiles each)
    const int64_t user_blocks = std::max<int64_t>(1, (B + 127) / 128);
    L.max_splits = (int)std::min<int64_t>(64, std::max<int64_t>(1, 512 / user_blocks));
    int tpc_s = 1, splits = 1;
    topk_geometry(B, std::max<int64_t>(L.sample_tiles, 1), 512, 2, L.max_splits, tpc_s, splits);
    L.tpc_s = tpc_s;
    L.splits = splits;
    L.bytes = L.main_lists * (kListStride * 8 + 4) + B * (int64_t)L.max_splits * 32 * 8 + B * 4 + 256;
    L.image_off = (L.bytes + 255) / 256 * 256;
    L.image_bytes = L.n_tiles * (int64_t)PlaneImage<2>::kTileBytes;
    if (L.image_bytes > kImageCap) L.image_bytes = 0;
    if (L.image_bytes) L.bytes = L.image_off + L.image_bytes;
    return L;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_full_sort_topk_workspace(int64_t B, int64_t n_items, int k, int64_t *bytes) {
    if (!bytes || B < 0 || n_items < 0 || k < 1) return fail(RBG_EINVAL, "bad argument");
    *bytes = std::max(topk_layout(B, n_items, 512).bytes, topk_layout(B, n_items, 768).bytes);  // (either residency of the main pass)
    const ScreenLayout S = screen_layout(B, n_items);  // r06: the screen's image and candidate pool behind the exact passes' part
    if (S.fits) *bytes = (*bytes + 255) / 256 * 256 + S.bytes;
    return RBG_OK;
}

int rbg_full_sort_topk_f32(const rbg_graph *history, const float *user_all, const float *item_all, const int64_t *users,
                           int64_t B, int64_t n_users, int64_t n_items, int d, int k, float *out_val, int64_t *out_idx,
                           void *workspace, void *stream) {
    clear_error();
    if (B < 0 || n_users < 0 || n_items < 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (k < 1 || k > 32) return fail(RBG_EUNSUPPORTED, "k = %d (the fused top-k supports 1..32)", k);
    if (d > 256) return fail(RBG_EUNSUPPORTED, "d = %d > 256", d);
    if (B == 0) return RBG_OK;
    if (!user_all || !item_all || !users || !out_val || !out_idx || !workspace) return fail(RBG_EINVAL, "NULL pointer");
    if (history) {
        if (history->device < 0) return fail(RBG_ENODEV, "history graph is a host graph");
        if (history->n_rows != n_users + n_items) return fail(RBG_ESHAPE, "history graph has %lld nodes, expected %lld",
                                                              (long long)history->n_rows, (long long)(n_users + n_items));
        int rc = set_device_for(history->device);
        if (rc) return rc;
    }
    const int64_t user_tiles = (B + 127) / 128;  // workgroups along the batch
    if (user_tiles > 65535) return fail(RBG_EUNSUPPORTED, "B = %lld too large for one call", (long long)B);
    const TopkLayout L = topk_layout(B, n_items, topk_short_lists(k, d, B) ? 768 : 512);  // three or two resident workgroups per CU
    char *w = reinterpret_cast<char *>(workspace);
    float *main_val = reinterpret_cast<float *>(w);
    int32_t *main_idx = reinterpret_cast<int32_t *>(main_val + L.main_lists * kListStride);
    int32_t *main_cnt = main_idx + L.main_lists * kListStride;
    float *pre_val = reinterpret_cast<float *>(main_cnt + L.main_lists);
    int32_t *pre_idx = reinterpret_cast<int32_t *>(pre_val + B * (int64_t)L.max_splits * 32);
    float *tau0 = reinterpret_cast<float *>(pre_idx + B * (int64_t)L.max_splits * 32);
    const int32_t *rp = history ? history->d_rowptr : nullptr;
    const int32_t *cl = history ? history->d_col : nullptr;
    TopkParams p{};
    p.U = user_all;
    p.I = item_all;
    p.users = users;
    p.rowptr = rp;
    p.col = cl;
    p.n_users = n_users;
    p.n_items = n_items;
    p.B = B;
    p.d = d;
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(user_all) | reinterpret_cast<uintptr_t>(item_all)) & 15u) == 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned merge_blocks = (unsigned)((B + 3) / 4);
    const bool prepass = L.n_tiles > 2 * L.sample_tiles;  // small item sets: one pass
    // r06: ONE bf16 product per pair screens the call, the survivors are rescored exactly (topk_screen.hip; measured:
    // profiles/r06_topk_screen.jsonl); the exact passes below stay for what it does not take (small batches and item sets,
    // d > 128, option "topk_screen" 0)
    if (prepass && L.sample_tiles <= 16 * 256 && screen_applicable(B, n_items, d, k)) {  // (the pre-pass keeps a tile index in 8 bits)
        ScreenCall c{};
        c.U = user_all, c.I = item_all, c.users = users, c.rowptr = rp, c.col = cl;
        c.n_users = n_users, c.n_items = n_items, c.B = B, c.d = d, c.k = k;
        c.w = w + (L.bytes + 255) / 256 * 256;
        // (the screen's threshold kernel folds at most 16 x 32 slot maxima per user)
        c.splits = std::min(L.splits, 16);
        c.tpc_s = (int)((L.sample_tiles + c.splits - 1) / c.splits);
        c.splits = (int)((L.sample_tiles + c.tpc_s - 1) / c.tpc_s);
        c.pre_val = pre_val, c.pre_idx = pre_idx, c.sample_tiles = L.sample_tiles;
        c.tau0 = tau0, c.out_val = out_val, c.out_idx = out_idx;  // (tau0: the screen's own threshold kernel writes it)
        int rc = screen_prepass(c, s);
        if (rc) return rc;
        return screen_main(c, s);
    }
    // r06: the item table as bf16 planes in the LDS layout, once per call (both passes take their tiles from it by LDS-DMA)
    // (measured, profiles/r06_topk_image.jsonl, 4096 users x 40 982 items: d = 128 429.7 -> 393.7 us per call; d = 64 191.5 -> 192.0 —
    // there the fetch and the split were already hidden behind the other resident workgroups' products, as r02 had found: the image
    // is built for 64 < d <= 128 only; option "topk_image" = 2 forces it at d <= 64 as well)
    if (opt_topk_image() && opt_mfma_split() != 0 && d <= 128 && (d > 64 || opt_topk_image() == 2) && L.image_bytes && B >= 1024) {
        char *image = w + L.image_off;
        if (d <= 64) {
            if (vec) hipLaunchKernelGGL((plane_image_kernel<1, true>), dim3((unsigned)L.n_tiles), dim3(256), 0, s, item_all, (int64_t)d, n_items, d, image);
            else hipLaunchKernelGGL((plane_image_kernel<1, false>), dim3((unsigned)L.n_tiles), dim3(256), 0, s, item_all, (int64_t)d, n_items, d, image);
        } else {
            if (vec) hipLaunchKernelGGL((plane_image_kernel<2, true>), dim3((unsigned)L.n_tiles), dim3(256), 0, s, item_all, (int64_t)d, n_items, d, image);
            else hipLaunchKernelGGL((plane_image_kernel<2, false>), dim3((unsigned)L.n_tiles), dim3(256), 0, s, item_all, (int64_t)d, n_items, d, image);
        }
        RBG_HIP(hipGetLastError());
        p.image = image;
    }
    if (prepass) {
        // pass 1: group maxima over the first "topk_sample" (default 8192) items -> tau0[b], a lower bound of the user's k-th best valid score
        p.k = k;
        p.filter_history = 1;
        p.tile_lo = 0;
        p.tile_hi = L.sample_tiles;
        p.tiles_per_chunk = L.tpc_s;
        p.n_chunks = L.splits;
        launch_prepass_d(p, vec, dim3((unsigned)L.splits, (unsigned)user_tiles), pre_val, pre_idx, s);
        RBG_HIP(hipGetLastError());
        hipLaunchKernelGGL(topk_tau_kernel, dim3(merge_blocks), dim3(256), 0, s, pre_val, pre_idx, users, rp, cl, n_users, B,
                           L.splits, k, tau0);
        RBG_HIP(hipGetLastError());
    }
    // pass 2: every item, thresholds seeded with tau0
    p.k = k;
    p.filter_history = 1;
    p.tiles_per_chunk = L.tpc;
    p.n_chunks = L.nc;
    p.tile_lo = 0;
    p.tile_hi = L.n_tiles;
    p.tau0 = prepass ? tau0 : nullptr;
    p.w_val = main_val;
    p.w_idx = main_idx;
    p.w_cnt = main_cnt;
    launch_topk_d(p, vec, dim3((unsigned)L.nc, (unsigned)user_tiles), s);
    RBG_HIP(hipGetLastError());
    hipLaunchKernelGGL(topk_merge_kernel, dim3(merge_blocks), dim3(256), 0, s, main_val, main_idx, main_cnt, users, rp, cl, n_users,
                       B, L.nc, k, out_val, out_idx, (float *)nullptr);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // extern "C"
