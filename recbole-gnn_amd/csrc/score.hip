// score.hip — the dense user x item scoring GEMM of full_sort_predict on the gfx950 matrix cores.
//
// Replaces  scores = torch.matmul(u_embeddings, self.restore_item_e.transpose(0, 1))
//   recbole_gnn/model/general_recommender/lightgcn.py:131 (ngcf.py:147, sgl.py:240).
//
// S[B, n] = U[B, d] · I[n, d]^T with fp32 accuracy: by default both operands are split into three bf16 terms and the six
// products of order >= 2^-16 run on v_mfma_f32_32x32x16_bf16 (option "mfma_split"; 0 = the exact-fp32 chain on
// v_mfma_f32_32x32x2_f32).  At d = 64 the GEMM has 32 flop per output byte and the output is 671 MB at B = 4096 x 40 982:
// the kernel streams item rows once per 128-user block, writes each score exactly once, and is paced by that store stream.
//
// Mapping: a wavefront owns a 32-user x 32-item tile; the A (user) fragments stay in registers for the whole walk, item tiles
// are fetched coalesced by the workgroup into a double-buffered LDS tile (score_kernel below has the details and the
// history of what did not work).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "internal.h"
#include "mfma_common.h"

namespace rbg {



// One 64-wide k chunk of one 32x32 tile: 32 exact-fp32 MFMAs.
__device__ __forceinline__ f32x16 mfma_chunk(const float (&a)[32], const float (&b)[32], f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}

// C/D layout of the 32x32 MFMA: col = lane&31 (item), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (user).
__device__ __forceinline__ void store_tile(float *__restrict__ S, int64_t n, int64_t B, int64_t user0, int64_t item, int h,
                                           const f32x16 &acc) {
    if (item < n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (u < B) S[u * n + item] = acc[r];
        }
    }
}

// Line-aligned stores for the reference's contiguous [B, n] output (lightgcn.py:131 returns scores.view(-1)): a row of
// S starts at byte 4*u*n, so unless n % 32 == 0 the 128-byte piece a tile contributes to a row straddles two cache
// lines and every store instruction costs two L2 write requests (store-only twin of this kernel, r01: 226 us vs 125 us
// with aligned rows, profiles/r01_microbench.md).  A wave walks consecutive item tiles, so it can shift each row's data
// by that row's phase instead: line k of a row = the last 32-s columns of tile k-1 + the first s columns of tile k,
// assembled with one cross-lane read per row.  Only the head of the first tile and the tail of the last one remain
// partial.  `sh[r]` = columns of the first tile that precede the row's first line boundary.
// Everything but `prev` is wave-uniform (scalar registers): addresses are a uniform 64-bit base + a 32-bit lane offset,
// and a row's phase is recomputed from (c0, n mod 32) instead of being kept per row.
// The whole-line stores of the interior tiles are non-temporal: S is written once and read by someone else much later
// (671 MB at B = 4096 x 40 982); with plain stores the lines linger in L2 and their write-back competes with the operand
// tiles — 216 -> 183-191 us at d = 64, 514 -> 421-426 us at 91 600 items, 308 -> 291-301 us at d = 128 (r02,
// profiles/r02_split_probe.jsonl).  The partial lines at the head and the tail of a walk stay plain stores (they merge in
// L2 with the neighbouring walk's part of the line; non-temporal there measured 4-6 % slower).
// Stores go through an explicit global (address space 1) pointer: carried through this struct the pointer is otherwise
// treated as generic, the stores become flat_store, and a pending FLAT access forces every later wait to vmcnt(0).
typedef __attribute__((address_space(1))) float gfloat;

struct AlignedRows {
    float prev[16];
    gfloat *row0;      // &S[user0][0]
    unsigned n;        // row stride (n <= 2^26 so 32 rows of offsets fit 32 bits)
    unsigned c0, nm;   // (address of S[user0][first_col] / 4) mod 32, n mod 32
    int rows_left;     // B - user0, clamped to 32
};

__device__ __forceinline__ void aligned_init(AlignedRows &a, float *S, int64_t n, int64_t B, int64_t user0, int64_t first_col) {
    a.row0 = (gfloat *)(S + user0 * n);
    a.n = (unsigned)n;
    a.c0 = (unsigned)(((reinterpret_cast<uintptr_t>(a.row0) >> 2) + (uint64_t)first_col) & 31u);
    a.nm = (unsigned)(n & 31);
    a.rows_left = (int)((B - user0 < 32) ? B - user0 : 32);
}

// Hand tile `t` (acc: col = lane&31, row = rowmap(reg, h)) to the shifted store stream.
__device__ __forceinline__ void aligned_emit(AlignedRows &a, int64_t n, int64_t t, bool first, int i, int h, const f32x16 &acc) {
    gfloat *base = a.row0 + (first ? t : t - 1) * 32;                      // uniform
    const int64_t left64 = n - (first ? t : t - 1) * 32;
    const int cols_left = (int)(left64 < 64 ? left64 : 64);                // uniform; a line never reaches past +63
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;                      // row inside the tile
        const int s = (int)((32u - ((a.c0 + (unsigned)m * a.nm) & 31u)) & 31u);
        const unsigned row_off = (unsigned)m * a.n;
        if (first) {  // head: the columns before the row's first line boundary
            if (i < s && i < cols_left && m < a.rows_left) base[row_off + (unsigned)i] = acc[r];
        } else {
            const float x = (i >= s) ? a.prev[r] : acc[r];  // what source lane i contributes to the line
            const float v = __shfl(x, ((i + s) & 31) + 32 * h);
            if (s + i < cols_left && m < a.rows_left) base[row_off + (unsigned)(s + i)] = v;
        }
        a.prev[r] = acc[r];
    }
}

// Interior tile (not the first of the walk, line (t-1)*32 + s .. + 31 inside the row for every s, all 32 rows valid):
// 16 cross-lane reads + 16 unconditional whole-line stores, no branches.
__device__ __forceinline__ void aligned_emit_interior(AlignedRows &a, int64_t t, int i, int h, const f32x16 &acc) {
    gfloat *base = a.row0 + (t - 1) * 32;
    // (r02 phase clock, devtools/microbench/score_trace.hip: a wave spends 53 % of its time in this function, 22 % in the
    //  product, 12 % in publish.  hipcc emits read / wait / store per row; issuing the 16 cross-lane reads ahead of the 16
    //  stores does NOT help — 333 vs 308 us at d = 128, and at d = 64 the 16 extra registers force 2 instead of 3 resident
    //  workgroups: 251 vs 216 us — the time here is the store stream's back-pressure, not LDS latency.)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int s = (int)((32u - ((a.c0 + (unsigned)m * a.nm) & 31u)) & 31u);
        const float x = (i >= s) ? a.prev[r] : acc[r];
        __builtin_nontemporal_store(__shfl(x, ((i + s) & 31) + 32 * h), &base[(unsigned)m * a.n + (unsigned)(s + i)]);
        a.prev[r] = acc[r];
    }
}

// After the last tile `t`: the columns from the last line boundary on.
__device__ __forceinline__ void aligned_flush(const AlignedRows &a, int64_t n, int64_t t, int i, int h) {
    gfloat *base = a.row0 + t * 32;
    const int64_t left64 = n - t * 32;
    const int cols_left = (int)(left64 < 32 ? left64 : 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int s = (int)((32u - ((a.c0 + (unsigned)m * a.nm) & 31u)) & 31u);
        const float v = __shfl(a.prev[r], ((i + s) & 31) + 32 * h);
        if (s + i < cols_left && m < a.rows_left) base[(unsigned)m * a.n + (unsigned)(s + i)] = v;
    }
}

// Any d, no LDS, un-pipelined (user fragment re-read per tile): d > 256 or n > 2^26.
template <bool VEC>
__global__ __launch_bounds__(256) void score_generic_kernel(const float *__restrict__ U, int64_t ldu,
                                                            const float *__restrict__ I, int64_t ldi, float *__restrict__ S,
                                                            int64_t B, int64_t n, int d, int tiles_per_wave) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t user_tile = (int64_t)blockIdx.y * 4 + wave;
    const int64_t ur = user_tile * 32 + i;
    if (user_tile * 32 >= B) return;
    const bool u_ok = ur < B;
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wave;
    const int64_t t1 = (t0 + tiles_per_wave < n_tiles) ? t0 + tiles_per_wave : n_tiles;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nchunk = (d + 63) / 64;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t jr = t * 32 + i;
        f32x16 acc = zero;
        for (int c = 0; c < nchunk; ++c) {
            float a[32], b[32];
            load_run32f<(VEC ? RUN_VEC : RUN_ANY)>(U + ur * ldu, u_ok, c * 64 + h * 32, d, a);
            load_run32f<(VEC ? RUN_VEC : RUN_ANY)>(I + jr * ldi, jr < n, c * 64 + h * 32, d, b);
            acc = mfma_chunk(a, b, acc);
        }
        store_tile(S, n, B, user_tile * 32, jr, h, acc);
    }
}

// Per-wave phase clock (devtools/microbench/score_trace.hip builds this file with RBG_SCORE_TRACE; the product does not):
// cycles spent up to each lap point, summed over the walk.
#ifdef RBG_SCORE_TRACE
__device__ unsigned long long *g_score_trace = nullptr;
__device__ int g_score_debug = 0;  // what-if switches of the diagnostic build (results are wrong on purpose): 1 no product, 2 no stores, 4 no fetch / publish
#define RBG_SCORE_DBG(bit) ((g_score_debug & (bit)) != 0)
#define RBG_SCORE_T0() unsigned long long sc_last = clock64(), sc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RBG_SCORE_LAP(k)                         \
    do {                                         \
        const unsigned long long sc_now = clock64(); \
        sc_acc[k] += sc_now - sc_last;           \
        sc_last = sc_now;                        \
    } while (0)
#define RBG_SCORE_DUMP()                                                                                                  \
    do {                                                                                                                  \
        if (g_score_trace && lane == 0)                                                                                   \
            for (int k = 0; k < 8; ++k)                                                                                   \
                g_score_trace[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + k] = sc_acc[k];           \
    } while (0)
#else
#define RBG_SCORE_DBG(bit) false
#define RBG_SCORE_T0() ((void)0)
#define RBG_SCORE_LAP(k) ((void)0)
#define RBG_SCORE_DUMP() ((void)0)
#endif

// d <= 64 NCHUNK <= 256.  A workgroup = 4 waves = 128 users whose A fragments stay in registers; it walks
// `tiles_per_wave` item tiles.  What the earlier versions of this kernel taught (r01, SQ counters + ISA):
//  * operand loads must be COALESCED: "each lane reads its own 128-byte run" is 8 dwordx4 instructions that each touch
//    64 different cache lines — 512 tag lookups per tile and wave, as many cycles as the 32 MFMAs (matrix core busy
//    38 %).  Here the 256 threads fetch an item tile as consecutive float4s (8 lines per instruction) into a
//    double-buffered LDS tile (row stride 64 NCHUNK + 4 floats) and every wave reads its B fragments with b128 LDS loads.
//  * `vmcnt` counts stores, so the fetch of tile t+1 is issued before tile t's stores and consumed (written to LDS)
//    before them as well; stores must be global_store (a pointer that decays to a generic one makes them flat_store and
//    every wait becomes vmcnt(0)) — see AlignedRows.
//  * output rows are not line-aligned (row stride n) — aligned_emit.
//  * SPLIT (default, option "mfma_split"): both operands are split into three bf16 terms and the six products of order
//    >= 2^-16 run on v_mfma_f32_32x32x16_bf16 — the accuracy of the fp32 chain at 2.3x its rate (mfma_common.h).
template <int NCHUNK, bool VEC, bool FAST, bool SPLIT>
__global__ __launch_bounds__(256, (NCHUNK == 1 ? 3 : (NCHUNK == 2 ? 2 : 1))) void score_kernel(const float *__restrict__ U, int64_t ldu,
                                                    const float *__restrict__ I, int64_t ldi, float *__restrict__ S,
                                                    int64_t B, int64_t n, int d, int tiles_per_wave) {
    constexpr int MODE = FAST ? RUN_FAST : (VEC ? RUN_VEC : RUN_ANY);
    using Tile = std::conditional_t<SPLIT, RowTile3<NCHUNK, MODE>, RowTile<NCHUNK, MODE>>;
    using TileMem = std::conditional_t<SPLIT, typename RowTile3<NCHUNK, MODE>::Planes, float[32][NCHUNK * 64 + 4]>;
    __shared__ __attribute__((aligned(16))) TileMem s_it[2];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave-uniform
    const int i = lane & 31, h = lane >> 5;
    const int64_t user_tile = (int64_t)blockIdx.y * 4 + wave;
    const int64_t ur = user_tile * 32 + i;
    const bool wave_live = user_tile * 32 < B;  // an idle wave still fetches and meets the barriers
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wave;
    const int64_t t1 = (t0 + tiles_per_wave < n_tiles) ? t0 + tiles_per_wave : n_tiles;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // clamped rows: a user row >= B / an item row >= n computes values that are never stored
    const float *urow = U + (ur < B ? ur : B - 1) * ldu;
    float a[NCHUNK][32];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(urow, true, c * 64 + h * 32, d, a[c]);

    std::conditional_t<SPLIT, AFrag3<NCHUNK>, int> a3;
    if constexpr (SPLIT) split_a(a, a3);  // the fp32 runs are dead after this
    Tile tile;
    auto fetch = [&](const int64_t t) __attribute__((always_inline)) { tile.fetch(I, ldi, n, d, t, tid); };
    auto publish = [&](const int buf) __attribute__((always_inline)) { tile.publish(s_it[buf], tid); };
    auto tile_product = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (SPLIT) return Tile::product(s_it[buf], a3, i, h);
        else return Tile::product(s_it[buf], a, i, h);
    };
    AlignedRows al;
    aligned_init(al, S, n, B, user_tile * 32, t0 * 32);
    // tiles whose shifted line lies inside the row for every phase, in a wave with 32 valid rows: unconditional stores
    const int64_t t_int_end = (wave_live && al.rows_left >= 32 && n >= 64) ? (n - 64) / 32 + 2 : t0;
    // Pipeline (vmcnt retires in order, stores included): iteration t runs
    //   product(t) from LDS | publish(t+1) LDS <- stage | fetch(t+2) -> stage | stores(t) | barrier
    // so the fetch consumed by publish() was issued one whole iteration earlier and only YOUNGER stores are outstanding
    // when it is awaited (vmcnt(16)); with fetch(t+1) at the top of iteration t the wait also drained tile t-1's stores,
    // whose write acknowledge takes longer than one tile of MFMAs (matrix core busy 59 % -> see DESIGN.md 6.3).
    RBG_SCORE_T0();
    if (t0 < t1) {
        fetch(t0);
        publish(0);
        if (t0 + 1 < t1) fetch(t0 + 1);
    }
    __syncthreads();
    RBG_SCORE_LAP(0);
    for (int64_t t = t0; t < t1; ++t) {
        const int buf = (int)(t - t0) & 1;
        f32x16 acc = zero;
        if (wave_live) acc = tile_product(buf);
        RBG_SCORE_LAP(1);
        if (t + 1 < t1) publish(buf ^ 1);  // the other buffer was last read before the previous barrier
        RBG_SCORE_LAP(2);
        if (t + 2 < t1) fetch(t + 2);
        RBG_SCORE_LAP(3);
        if (wave_live) {
            if (t > t0 && t < t_int_end) aligned_emit_interior(al, t, i, h, acc);
            else aligned_emit(al, n, t, t == t0, i, h, acc);
        }
        RBG_SCORE_LAP(4);
        __syncthreads();
        RBG_SCORE_LAP(5);
    }
    if (wave_live && t1 > t0) aligned_flush(al, n, t1 - 1, i, h);
    RBG_SCORE_LAP(6);
    RBG_SCORE_DUMP();
}

// r04: the same walk without the shifted store stream.  Rows u and u' of the contiguous [B, n] output start at the same
// offset inside a 128-byte line iff (u - u') n = 0 mod 32, i.e. u = u' mod q with q = 32 / gcd(n mod 32, 32).  A workgroup that
// takes its 128 users from ONE residue class r (user = r + q k) has one phase for all its rows, so it can shift its ITEM
// tiles instead of its data: tile t covers items [32 t - delta, 32 t - delta + 32) with delta = (address of S[r][0] / 4) mod 32,
// and every store instruction writes two whole, aligned lines straight from the accumulator — no cross-lane read (16
// ds_bpermute per tile in score_kernel, where a wave spent 53 % of its time), no carried half lines (16 registers), no
// head / tail code beyond a column mask on the first and the last tile.  Used when every class holds >= 64 users
// (B >= 64 q; the reference's own evaluation batches of a few users keep score_kernel).
// (a fourth resident workgroup per CU — the 16 carry registers are gone — measured 161 vs 157 us: three it stays;
//  r05: raised priority between the product and the barrier, which pays in topk.hip, measured 167.4 vs 167.2-168.8 us here: not kept)
template <int NCHUNK, bool VEC, bool FAST, bool SPLIT>
__global__ __launch_bounds__(256, (NCHUNK == 1 ? 3 : (NCHUNK == 2 ? 2 : 1))) void score_uni_kernel(const float *__restrict__ U, int64_t ldu,
                                                    const float *__restrict__ I, int64_t ldi, float *__restrict__ S,
                                                    int64_t B, int64_t n, int d, int tiles_per_wave, int q) {
    constexpr int MODE = FAST ? RUN_FAST : (VEC ? RUN_VEC : RUN_ANY);
    using Tile = std::conditional_t<SPLIT, RowTile3<NCHUNK, MODE>, RowTile<NCHUNK, MODE>>;
    using TileMem = std::conditional_t<SPLIT, typename RowTile3<NCHUNK, MODE>::Planes, float[32][NCHUNK * 64 + 4]>;
    __shared__ __attribute__((aligned(16))) TileMem s_it[2];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave-uniform
    const int i = lane & 31, h = lane >> 5;
    const int r = (int)(blockIdx.y % (unsigned)q);                       // the residue class of this workgroup's users
    const int64_t ut = (int64_t)(blockIdx.y / (unsigned)q) * 4 + wave;   // 32-user tile inside the class
    const int64_t k_first = ut * 32;                                      // user = r + q (k_first + m)
    const bool wave_live = r + (int64_t)q * k_first < B;                  // an idle wave still fetches and meets the barriers
    const int delta = (int)(((reinterpret_cast<uintptr_t>(S) >> 2) + (uint64_t)r * (uint64_t)n) & 31u);
    const int64_t n_tiles = (n + delta + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wave;
    const int64_t t1 = (t0 + tiles_per_wave < n_tiles) ? t0 + tiles_per_wave : n_tiles;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int64_t ur = r + (int64_t)q * (k_first + i);                    // this lane's A row (clamped: never stored beyond B)
    const float *urow = U + (ur < B ? ur : B - 1) * ldu;
    float a[NCHUNK][32];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(urow, true, c * 64 + h * 32, d, a[c]);
    std::conditional_t<SPLIT, AFrag3<NCHUNK>, int> a3;
    if constexpr (SPLIT) split_a(a, a3);  // the fp32 runs are dead after this
    Tile tile;
    auto fetch = [&](const int64_t t) __attribute__((always_inline)) { tile.fetch_rows(I, ldi, n, d, t * 32 - delta, tid); };
    auto publish = [&](const int buf) __attribute__((always_inline)) { tile.publish(s_it[buf], tid); };
    auto tile_product = [&](const int buf) __attribute__((always_inline)) {
        if constexpr (SPLIT) return Tile::product(s_it[buf], a3, i, h);
        else return Tile::product(s_it[buf], a, i, h);
    };
    // output rows of this wave: S + (r + q (k_first + m)) n, m = rowmap(reg, h): a uniform base + a 32-bit offset m q n
    gfloat *const row0 = (gfloat *)(S + (r + (int64_t)q * k_first) * n);
    const unsigned qn = (unsigned)((int64_t)q * n);
    int64_t rows_left = (B - 1 - (r + (int64_t)q * k_first)) / q + 1;    // valid rows of this wave's tile
    rows_left = rows_left < 0 ? 0 : (rows_left > 32 ? 32 : rows_left);
    auto emit = [&](const int64_t t, const f32x16 &acc) __attribute__((always_inline)) {
        const int64_t c0 = t * 32 - delta;                                // first column of the tile (uniform)
        gfloat *base = row0 + c0;
        if (rows_left == 32 && c0 >= 0 && c0 + 32 <= n) {                 // interior: 16 unconditional whole-line stores
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) __builtin_nontemporal_store(acc[rr], &base[(unsigned)mfma_rowmap(rr, h) * qn + (unsigned)i]);
        } else {
            const bool col_ok = c0 + i >= 0 && c0 + i < n;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = mfma_rowmap(rr, h);
                if (col_ok && m < rows_left) base[(int64_t)m * qn + i] = acc[rr];
            }
        }
    };
    // the same pipeline as score_kernel: product(t) | publish(t + 1) | fetch(t + 2) | stores(t) | barrier
    if (t0 < t1) {
        fetch(t0);
        publish(0);
        if (t0 + 1 < t1) fetch(t0 + 1);
    }
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const int buf = (int)(t - t0) & 1;
        f32x16 acc = zero;
        if (wave_live && !RBG_SCORE_DBG(1)) acc = tile_product(buf);
        if (t + 1 < t1 && !RBG_SCORE_DBG(4)) publish(buf ^ 1);  // the other buffer was last read before the previous barrier
        if (t + 2 < t1 && !RBG_SCORE_DBG(4)) fetch(t + 2);
        if (wave_live && !(RBG_SCORE_DBG(2) && acc[0] != 12345.678f)) emit(t, acc);
        __syncthreads();
    }
}

// (Tried, r02: a role-specialised variant — one loader wave publishing the tiles, four compute waves that only multiply
// and store, so that no wave ever waits on `vmcnt` behind its own stores.  Correct, but slower: 237 vs 216 us at d = 64,
// 359 vs 308 us at d = 128 (B = 4096 x 40 982, interleaved timing, profiles/r02_split_probe.jsonl): with 5-wave
// workgroups only 8 instead of 12 compute waves fit a CU, and the store stream, not the operand wait, sets the pace.)

template <int NCHUNK>
static int launch_score(const float *U, int64_t ldu, const float *I, int64_t ldi, float *S, int64_t B, int64_t n, int d,
                        bool vec, hipStream_t s) {
    // Walk length: equal-sized workgroups should fill whole rounds of the resident ones (3 / 2 / 1 per CU for
    // NCHUNK 1 / 2 / 4; lse.hip's block trace), and a walk should be long enough to amortise its masked first tile.
    // Smallest round count that keeps a walk <= 56 tiles (sweep r01: 250 us at exactly one round of 54-tile walks vs
    // 262-294 us otherwise, B = 4096 x 40 982).
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t gy = (B + 127) / 128;
    const int64_t slots = 256 * (NCHUNK == 1 ? 3 : (NCHUNK == 2 ? 2 : 1));
    int64_t tiles_per_wave = n_tiles;
    for (int64_t rounds = 1; rounds <= 64; ++rounds) {
        const int64_t gx_want = std::max<int64_t>(1, rounds * slots / gy);
        tiles_per_wave = (n_tiles + gx_want - 1) / gx_want;
        if (tiles_per_wave <= 56) break;
    }
    tiles_per_wave = std::max<int64_t>(tiles_per_wave, std::min<int64_t>(4, n_tiles));
    if (opt_score_tiles() > 0) tiles_per_wave = std::min<int64_t>(opt_score_tiles(), n_tiles);
    const int64_t gx = (n_tiles + tiles_per_wave - 1) / tiles_per_wave;
    if (gx > INT32_MAX || gy > 65535) return fail(RBG_EUNSUPPORTED, "score grid too large (B = %lld)", (long long)B);
    dim3 grid((unsigned)gx, (unsigned)gy);
    if constexpr (NCHUNK != 0) {
        // uniform-phase form (score_uni_kernel): the users of a workgroup come from one residue class mod q
        int q = 1;
        for (int g = (int)(n & 31); q < 32 && (g * q) % 32 != 0; q <<= 1) {}
        if (opt_score_uniform() && B >= 64 * (int64_t)q && (int64_t)q * n * 32 < ((int64_t)1 << 32)) {
            const int64_t bc = (B + q - 1) / q, wg_per_class = ((bc + 31) / 32 + 3) / 4;
            const int64_t gyu = wg_per_class * q;
            if (gyu <= 65535) {
                dim3 gridu((unsigned)((n_tiles + 1 + tiles_per_wave - 1) / tiles_per_wave), (unsigned)gyu);  // (+1: the shift may add a tile)
                const bool fastu = vec && d == 64 * NCHUNK;
                if (opt_mfma_split()) {
                    if (fastu) hipLaunchKernelGGL((score_uni_kernel<NCHUNK, true, true, true>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                    else if (vec) hipLaunchKernelGGL((score_uni_kernel<NCHUNK, true, false, true>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                    else hipLaunchKernelGGL((score_uni_kernel<NCHUNK, false, false, true>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                } else if (fastu) hipLaunchKernelGGL((score_uni_kernel<NCHUNK, true, true, false>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                else if (vec) hipLaunchKernelGGL((score_uni_kernel<NCHUNK, true, false, false>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                else hipLaunchKernelGGL((score_uni_kernel<NCHUNK, false, false, false>), gridu, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave, q);
                RBG_HIP(hipGetLastError());
                return RBG_OK;
            }
        }
    }
    if constexpr (NCHUNK == 0) {
        if (vec)
            hipLaunchKernelGGL((score_generic_kernel<true>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
        else
            hipLaunchKernelGGL((score_generic_kernel<false>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
    } else {
        const bool fast = vec && d == 64 * NCHUNK;
        if (opt_mfma_split()) {
            if (fast)
                hipLaunchKernelGGL((score_kernel<NCHUNK, true, true, true>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
            else if (vec)
                hipLaunchKernelGGL((score_kernel<NCHUNK, true, false, true>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
            else
                hipLaunchKernelGGL((score_kernel<NCHUNK, false, false, true>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
        } else if (fast)
            hipLaunchKernelGGL((score_kernel<NCHUNK, true, true, false>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
        else if (vec)
            hipLaunchKernelGGL((score_kernel<NCHUNK, true, false, false>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
        else
            hipLaunchKernelGGL((score_kernel<NCHUNK, false, false, false>), grid, dim3(256), 0, s, U, ldu, I, ldi, S, B, n, d, (int)tiles_per_wave);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" int rbg_score_f32(const float *U, int64_t ldu, const float *I, int64_t ldi, float *S, int64_t B, int64_t n,
                             int d, void *stream) {
    clear_error();
    if (B < 0 || n < 0 || d <= 0) return fail(RBG_ESHAPE, "B = %lld, n = %lld, d = %d", (long long)B, (long long)n, d);
    if (ldu < d || ldi < d) return fail(RBG_ESHAPE, "row stride smaller than d");
    if (B == 0 || n == 0) return RBG_OK;
    if (!U || !I || !S) return fail(RBG_EINVAL, "NULL pointer");
    const bool vec = (d % 4 == 0) && (ldu % 4 == 0) && (ldi % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(I)) & 15u) == 0;
    hipStream_t s = (hipStream_t)stream;
    if (n > (int64_t(1) << 26)) return launch_score<0>(U, ldu, I, ldi, S, B, n, d, vec, s);  // 32-bit row offsets in the aligned store stream
    if (d <= 64) return launch_score<1>(U, ldu, I, ldi, S, B, n, d, vec, s);
    if (d <= 128) return launch_score<2>(U, ldu, I, ldi, S, B, n, d, vec, s);
    if (d <= 256) return launch_score<4>(U, ldu, I, ldi, S, B, n, d, vec, s);
    return launch_score<0>(U, ldu, I, ldi, S, B, n, d, vec, s);
}
