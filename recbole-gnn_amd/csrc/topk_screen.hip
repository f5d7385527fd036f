// topk_screen.hip — full-sort top-k as ONE bf16 product per (user, item) + exact rescoring of the few survivors (r06).
//
// The exact passes of topk.hip pay six bf16 MFMA products on 3-way split operands, a split / publish of every item tile and a
// workgroup barrier per 32 items for EVERY (user, item) pair, although > 99.8 % of the pairs are nowhere near a user's top k.
// Here every pair gets one bf16 x bf16 product  s^ = u^ . i^  (u^ = bf16(u), i^ = bf16(i), round to nearest) and a RIGOROUS bound
// of what that lost, from the rounding errors the two rows ACTUALLY have (du = u^ - u, di = i^ - i are known when the fragments are built):
//     u^ . i^ - u . i  =  du . i^ + u . di      =>      |s - s^|  <=  ||du|| ||i^|| + ||u|| ||di||  =:  m_ui        (Cauchy-Schwarz)
// (~ 0.5 x 2^-8 ||u|| ||i|| on average; the worst case of round-to-nearest bf16 — 2^-8 per operand — would be 2^-7 ||u|| ||i||.  The
// fp32 accumulation of the matrix core, the fp32 rescoring and the norms' own rounding, together < 4e-5 ||u|| ||i||, are added to
// ||di||.)  The bound rides on the matrix core like the threshold of topk.hip's TauTest: one more MFMA whose A operand carries
// (-tau_u in three bf16 terms, ||du||, 1, ||u||) and whose B operand carries (1, 1, 1, ||i^||, -3e38 for the PAD item and rows past
// the end, ||di|| + 4e-5 ||i||), every norm rounded UP to bf16, so the accumulator holds
//     tst = s^ + m^_ui - tau_u      (m^ >= m, tau lowered by 1e-5 |tau|)
// and "tst > 0" is the accumulator's sign bit.  No pair with s >= tau_u is ever dropped.
//   launch 1 (screen_image_kernel): the item table AND the batch's user rows as bf16 fragments in the MFMA operand layout
//          ([tile][fragment][lane][16 bytes]: a wave's operand is a coalesced 1 KiB load), norms in a bound fragment per tile.
//   launch 2 (screen_pre_kernel): over the first "topk_sample" items the accumulator holds the LOWER bound s^ - m^; per lane and
//          accumulator row the running maximum and its tile index in one register (slot maxima, as topk.hip's pre-pass keeps them).
//   launch 3 (screen_tau_kernel): tau_u = the k-th largest slot maximum that is not a history item, over all splits x 32 slots, the
//          user's history walked (a history item belongs to exactly one slot) — tau_u <= the k-th best valid exact score.
//   launch 4 (screen_main_kernel): every item; the sign bits of the 16 accumulator registers are shifted into one word per lane
//          (v_alignbit: 16 instructions), one ballot says whether anything passed; a lane whose entry passes appends
//          (item, accumulator row) to its wave's private region of a candidate pool — the slot is a wave-uniform counter + the
//          lane's rank in the ballot (v_mbcnt): no atomic, no LDS, no exact score, no list pruning, no barrier.  The two lane
//          halves (= two sets of 16 users) fill a region from its two ends.  Software-pipelined over the tiles; chunk c = the
//          tiles c, c + n_chunks, ... (popular items spread over the chunks); one residue class of chunks per XCD.
//   launch 5 (screen_merge_kernel): one workgroup per 16 users (one lane half of a tile) gathers its regions, rescoring every
//          candidate exactly in fp32 (16 lanes per candidate, eight candidates in flight, the item row read from the fp32 table),
//          drops what the exact score puts below tau_u and the history items (one thread per candidate searches the user's graph
//          row, staged in an LDS pool), buckets the rest by user and folds them into the user's best 32 by topk.hip's bitonic
//          network (DPP exchanges inside a row of 16) and total order, one wave per user.
// No workgroup shares anything in launches 2 and 4: a wave keeps the bf16 rows of its 32-user tiles as A fragments (16 registers
// per tile at d <= 64 — the split operands of the exact pass take 48) and reads its B fragments straight from the image (L2
// resident: 5 KiB per tile), so there is no LDS tile, no split, no publish and no barrier in the loop.
// What the screen cannot hold is not lost: a candidate region that overflows (scores that are all equal) is replaced in the merge
// by EVERY (user, item) pair of its chunk, a user without a bound (fewer than k valid slots: tau = +inf) by every item — slow and
// exact, no host round trip, no second code path.  Results: the same items as the exact passes wherever scores are not tied to the
// last bit (the rescoring sums in a different order: values agree to ~ 5e-7 relative; tests/test_gpu_parity.py::
// test_full_sort_topk_screen).  Measurements and what was tried: DESIGN.md 2.4a, profiles/r06_topk_screen*.
//
// Replaces (with topk.hip): lightgcn.py:123-133 + Trainer._full_sort_batch_eval [recbole==1.1.1].

#include "topk_screen.h"

#include <algorithm>

#include "internal.h"
#include "mfma_common.h"
#include "topk_common.h"

namespace rbg {

constexpr float kFpSlack = 4.0e-5f;   // x ||u|| ||i||: the matrix core's fp32 accumulation (144 terms, < 1.8e-5 even if it truncated), the fp32 rescoring (< 8e-6), the norms' rounding
constexpr float kPackSlack = 6.2e-5f;  // x ||u|| ||i||: the pre-pass's packed maxima lose 8 mantissa bits (< 2^-15 of the value either way)
constexpr int kRegion = 512;   // entries of one (user block, chunk, user tile) region of the pool
constexpr int kSlab = 4096;    // candidates a merge workgroup holds in LDS at a time
constexpr int kMaxChunks = 64;
constexpr int kHPool = 8192;   // history columns the merge kernel stages for its 16 users together (256 each, the rest to the long rows in turn)
constexpr int kUStride = 132;  // floats per staged user row (d <= 128)
constexpr int kUt64 = 2, kUt128 = 2;  // 32-user tiles a wave of the main pass keeps (d <= 64 / d <= 128)
constexpr int kWantWaves = 4096;      // waves the main pass aims for (user blocks x item chunks)

typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));  // a B fragment as it is loaded (hipcc re-packs copies of __bf16 vectors element by element)
__device__ __forceinline__ bf16x8 as_frag(const i32x4v &v) { return __builtin_bit_cast(bf16x8, v); }

// smallest bf16 >= x (x >= 0, finite, < 3e38)
__device__ __forceinline__ __bf16 bf16_up(float x) {
    const unsigned u = (__float_as_uint(x) + 0xffffu) >> 16;
    return __builtin_bit_cast(__bf16, (unsigned short)u);
}
__device__ __forceinline__ __bf16 bf16_neg(__bf16 x) {
    return __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned short, x) ^ 0x8000u));
}

// elements [k0, k0 + 8) of a row (zero past d / when !ok)
template <bool VEC>
__device__ __forceinline__ f32x8v load8(const float *row, bool ok, int k0, int d) {
    f32x8v v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC) {
        if (ok && k0 < d) {
            const float4 a = *reinterpret_cast<const float4 *>(row + k0);
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
        }
        if (ok && k0 + 4 < d) {
            const float4 b = *reinterpret_cast<const float4 *>(row + k0 + 4);
            v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (ok && k0 + q < d) v[q] = row[k0 + q];
    }
    return v;
}
__device__ __forceinline__ float sumsq8(const f32x8v &v) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s = fmaf(v[q], v[q], s);
    return s;
}

// ---- the image: tile t = S product fragments + 1 bound fragment, each [64 lanes][8 bf16]; lane (i, h) of fragment s holds
// elements 16 s + 8 h + [0, 8) of item 32 t + i — exactly a lane's B operand of v_mfma_f32_32x32x16_bf16 -------------------------
template <int S, bool VEC>
__global__ __launch_bounds__(256) void screen_image_kernel(const float *__restrict__ I, int64_t n_items, int64_t n_tiles, int d,
                                                           char *__restrict__ image, const float *__restrict__ U,
                                                           const int64_t *__restrict__ users, int64_t B, char *__restrict__ uimage) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    const int i = lane & 31, h = lane >> 5;
    const __bf16 z = (__bf16)0.0f, o = (__bf16)1.0f;
    if (t >= n_tiles) {
        // the waves behind the item tiles convert the batch's user rows: the same layout, tile ut = 32 batch slots, the bound
        // fragment (eps ||u|| rounded up, 1); both passes then start with five coalesced loads instead of a gather behind an
        // index load (their prologue was ~ 8 of the pre-pass's 17 us)
        const int64_t ut = t - n_tiles;
        if (ut * 32 >= B) return;
        const int64_t bi = ut * 32 + i;
        const int64_t user = bi < B ? users[bi] : -1;
        const float *row = U + (user < 0 ? 0 : user) * (int64_t)d;
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(uimage + ut * (int64_t)(S + 1) * 1024) + lane;
        float ss = 0.f, sd = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const f32x8v v = load8<VEC>(row, user >= 0, 16 * s + 8 * h, d);
            const bf16x8 r = __builtin_convertvector(v, bf16x8);
            ss += sumsq8(v);
            sd += sumsq8(__builtin_convertvector(r, f32x8v) - v);  // (exact in fp32: the rounding error of every element)
            dst[s * 64] = r;
        }
        ss += __shfl_xor(ss, 32);
        sd += __shfl_xor(sd, 32);
        // slot 3: ||du||, slot 4: 1 (the invalid-item flag's multiplier), slot 5: ||u|| — rounded up; the passes add tau / negate
        bf16x8 f = {z, z, z, z, z, z, z, z};
        if (h == 0) f = bf16x8{z, z, z, bf16_up(fminf(sqrtf(sd) * 1.00001f, 1.0e38f)), o, bf16_up(fminf(sqrtf(ss) * 1.00001f, 1.0e38f)), z, z};
        dst[S * 64] = f;
        return;
    }
    const int64_t item = t * 32 + i;
    const bool ok = item < n_items;
    const float *row = I + (ok ? item : 0) * (int64_t)d;
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(image + t * (int64_t)(S + 1) * 1024) + lane;
    float ss = 0.f, sh = 0.f, sd = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const f32x8v v = load8<VEC>(row, ok, 16 * s + 8 * h, d);
        const bf16x8 r = __builtin_convertvector(v, bf16x8);
        const f32x8v rf = __builtin_convertvector(r, f32x8v);
        ss += sumsq8(v);
        sh += sumsq8(rf);
        sd += sumsq8(rf - v);
        dst[s * 64] = r;
    }
    ss += __shfl_xor(ss, 32);
    sh += __shfl_xor(sh, 32);
    sd += __shfl_xor(sd, 32);
    const float ni = sqrtf(ss) * 1.00001f;
    const float nh = fminf(sqrtf(sh) * 1.00001f, 1.0e38f);                                // ||i^||
    const float nd = fminf(sqrtf(sd) * 1.00001f + kFpSlack * ni, 1.0e38f);                // ||di|| + the fp32 slack
    const float np = fminf(sqrtf(sd) * 1.00001f + (kFpSlack + kPackSlack) * ni, 1.0e38f);  // the pre-pass's: + its packing slack
    const bool valid = ok && item != 0;  // the PAD item and the rows past the end never pass
    // slots 0-2: 1 (x -tau), 3: ||i^|| (x ||du||), 4: the invalid flag (x 1), 5: ||di|| + slack (x ||u||, main pass), 6: the same + the
    // packing slack (x ||u||, pre-pass)
    bf16x8 f = {z, z, z, z, z, z, z, z};
    if (h == 0) f = bf16x8{o, o, o, bf16_up(nh), valid ? z : (__bf16)-3.0e38f, bf16_up(nd), bf16_up(np), z};
    dst[S * 64] = f;
}

// What-if switches of the diagnostic build (devtools/microbench/topk_screen_trace.hip defines RBG_SCREEN_DBG; results are wrong on
// purpose): 1 main pass appends nothing, 2 main pass tests one accumulator row of 16, 4 main pass loads no tile after its first,
// 8 main pass stops after its prologue, 16 merge scores nothing, 32 merge sorts nothing, 64 merge stops after staging, 128 merge tests no history, 256 merge runs no sorting network, 1024 the pre-pass is clocked instead of the main pass.
#ifdef RBG_SCREEN_DBG
__device__ int g_screen_debug = 0;
__device__ unsigned long long *g_screen_trace = nullptr;  // [workgroup][16] phase clock of the merge kernel (thread 0)
__device__ unsigned long long *g_screen_trace_main = nullptr;  // [workgroup][4] clock of the main pass (wave 0): entry, loop entry, loop exit
#define RBG_SCREEN_MAIN_LAP(k)                                                                                            \
    do {                                                                                                                  \
        if (g_screen_trace_main && threadIdx.x == 0) g_screen_trace_main[(int64_t)blockIdx.x * 4 + (k)] = clock64();     \
    } while (0)
#define RBG_SCREEN_LAP(k)                                                                                       \
    do {                                                                                                        \
        if (g_screen_trace && threadIdx.x == 0) g_screen_trace[(int64_t)blockIdx.x * 16 + (k)] = clock64();    \
    } while (0)
#define RBG_SCREEN_DBG_LOAD() const int screen_dbg_bits = g_screen_debug  // (once per kernel: a load per use would stall the loop it is meant to probe)
#define RBG_SCREEN_DBGBIT(bit) ((screen_dbg_bits & (bit)) != 0)
#else
#define RBG_SCREEN_MAIN_LAP(k) ((void)0)
#define RBG_SCREEN_LAP(k) ((void)0)
#define RBG_SCREEN_DBG_LOAD() ((void)0)
#define RBG_SCREEN_DBGBIT(bit) false
#endif

struct ScreenParams {
    int64_t B;
    const char *image;   // item tiles
    const char *uimage;  // the batch's user tiles, same layout
    int tiles_per_chunk, n_chunks;  // pre-pass: chunk c = tiles_per_chunk tiles from tile_lo + c tiles_per_chunk; main pass: see there
    int64_t tile_lo, tile_hi;
    const float *tau0;
    uint32_t *pool;
    int32_t *cnt;
    float *g_val;
    int32_t *g_idx;
};

// ---- pass 1: lower bounds s^ - m^ of the first tile_hi tiles; per lane and accumulator row the running maximum ----------------
// One 32-user tile per wave, two item tiles per iteration.  The maximum and its item travel in ONE register: the low 8 mantissa
// bits of the bound are replaced by the tile's index inside the chunk (< 256), so a row costs v_and_or x 2 + v_max3 per two tiles
// (topk.hip's pre-pass: compare + two selects per tile).  Truncating moves a bound by < 2^-15 |bound| either way; the margin of this
// pass is wider by 6.2e-5 ||u|| ||i|| (> 2^-14: kPackSlack, in the item fragment's slot 6) to stay a lower bound.
template <int S>
__global__ __launch_bounds__(256) void screen_pre_kernel(const ScreenParams p) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    // (XCD-aware like the main pass when the splits allow it: XCD x takes the splits x, x + 8)
    RBG_SCREEN_DBG_LOAD();
    if (RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(0);
    int chunk, yb;
    if ((p.n_chunks & 7) == 0) {
        const int cpx = p.n_chunks >> 3, wg_slot = (int)(blockIdx.x >> 3);
        chunk = (int)(blockIdx.x & 7) + 8 * (wg_slot % cpx);
        yb = wg_slot / cpx;
    } else {
        chunk = (int)(blockIdx.x % p.n_chunks);
        yb = (int)(blockIdx.x / p.n_chunks);
    }
    const int64_t b0 = ((int64_t)yb * 4 + wave) * 32;
    if (b0 >= p.B) return;  // (no barrier in this kernel)
    const __bf16 z = (__bf16)0.0f;
    const bf16x8 *uimg = reinterpret_cast<const bf16x8 *>(p.uimage) + ((b0 >> 5) * (S + 1)) * 64 + lane;
    bf16x8 A[S];
#pragma unroll
    for (int s = 0; s < S; ++s) A[s] = uimg[s * 64];
    bf16x8 At = uimg[S * 64];  // (.., ||du||, 1, ||u||, 0, 0): acc = s^ - m^_pre (+ -3e38 for an invalid item) takes both margins negated,
    At[3] = bf16_neg(At[3]);   // ||u|| against the item fragment's slot 6 (its slack includes the packed maxima's)
    At[6] = bf16_neg(At[5]);
    At[5] = z;
    float best[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = kNegInf;
    const int64_t t_begin = p.tile_lo + (int64_t)chunk * p.tiles_per_chunk;
    const int64_t t_end = (t_begin + p.tiles_per_chunk < p.tile_hi) ? t_begin + p.tiles_per_chunk : p.tile_hi;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(p.image) + lane;
    const int64_t t_last = t_end - 1;
    if (RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(1);
    i32x4v N0[S + 1], N1[S + 1];  // the next pair of tiles, in flight while this pair feeds the matrix core
    if (t_begin < t_end) {
        const int64_t t1 = t_begin + 1 < t_end ? t_begin + 1 : t_last;
#pragma unroll
        for (int s = 0; s <= S; ++s) N0[s] = img[(t_begin * (S + 1) + s) * 64], N1[s] = img[(t1 * (S + 1) + s) * 64];
    }
    for (int64_t t = t_begin; t < t_end; t += 2) {
        const int64_t t1 = t + 1 < t_end ? t + 1 : t_last;  // (an odd tail repeats its tile: the maximum does not change)
        i32x4v B0[S + 1], B1[S + 1];
#pragma unroll
        for (int s = 0; s <= S; ++s) B0[s] = N0[s], B1[s] = N1[s];
        if (t + 2 < t_end) {
            const int64_t u1 = t + 3 < t_end ? t + 3 : t_last;
#pragma unroll
            for (int s = 0; s <= S; ++s) N0[s] = img[((t + 2) * (S + 1) + s) * 64], N1[s] = img[(u1 * (S + 1) + s) * 64];
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, as_frag(B0[S]), zero, 0, 0, 0);
        f32x16 a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, as_frag(B1[S]), zero, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], as_frag(B0[s]), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], as_frag(B1[s]), a1, 0, 0, 0);
        }
        const unsigned tl0 = (unsigned)(t - t_begin), tl1 = (unsigned)(t1 - t_begin);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p0 = __uint_as_float((__float_as_uint(a0[r]) & 0xffffff00u) | tl0);
            const float p1 = __uint_as_float((__float_as_uint(a1[r]) & 0xffffff00u) | tl1);
            best[r] = fmaxf(fmaxf(best[r], p0), p1);
        }
    }
    if (RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(2);
    // lane (i, h) holds group i of user slot rowmap(r, h) (the layout topk.hip's threshold kernel reads)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t b = b0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (b < p.B) {
            const int64_t off = (b * p.n_chunks + chunk) * 32 + i;
            const unsigned u = __float_as_uint(best[r]);
            p.g_val[off] = __uint_as_float(u & 0xffffff00u);
            p.g_idx[off] = t_begin < t_end ? (int)((t_begin + (u & 0xffu)) * 32 + i) : 0x7fffffff;
        }
    }
}

// ---- thresholds: tau_u = the k-th largest slot maximum of user u that is not a history item, over ALL splits x 32 slots -------
// (topk.hip's threshold kernel folds the splits to 32 slot maxima first and searches the user's history for those 32: a user whose
// best sample items ARE its history — every trained model has them — loses slots to history items and ends without a bound; the
// exact passes then start at -inf and prune their way up, the screen cannot.)  Here the history is walked instead: a history item
// inside the sample belongs to exactly one slot (its split and its lane), so one look-up says whether it is that slot's maximum.
// Every lane then folds its eight slots and the k-th largest of the 64 lane maxima is the bound (one sort).  Fewer than k lanes with
// a valid slot: tau = +inf = "no bound" (the merge takes every item for that user).
__global__ __launch_bounds__(256) void screen_tau_kernel(const float *__restrict__ g_val, const int32_t *__restrict__ g_idx,
                                                         const int64_t *__restrict__ users, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, int64_t n_users, int64_t B, int splits,
                                                         int tpc_s, int64_t sample_items, int k, float *__restrict__ tau_out) {
    __shared__ unsigned s_bad[4][16];  // one bit per slot (splits <= 16)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int total = splits * 32;
    if (lane < 16) s_bad[wave][lane] = 0u;
    const int64_t user = users[b];
    int lo = 0, hi = 0;
    if (rowptr && user >= 0) lo = rowptr[user], hi = rowptr[user + 1];
    float val[8];
    int idx[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int e = lane + 64 * t;
        val[t] = e < total ? g_val[b * total + e] : kNegInf;
        idx[t] = e < total ? g_idx[b * total + e] : 0x7fffffff;
    }
    // the row is sorted: its history items inside the sample are a prefix; four loads in flight, stop at the first batch past it
    const float inv_tpc = 1.0f / (float)tpc_s;
    for (int e0 = lo; e0 < hi; e0 += 256) {
        int64_t it[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 64 * u + lane;
            it[u] = e < hi ? (int64_t)col[e] - n_users : sample_items;
        }
        int hit[4], slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = it[u] >= 0 && it[u] < sample_items;
            // split of the item's tile: tile / tpc_s by a float reciprocal (tile < 4096, tpc_s <= 256: exact with the half offset)
            slot[u] = in ? (int)(((float)(int)(it[u] >> 5) + 0.5f) * inv_tpc) * 32 + (int)(it[u] & 31) : total;
            hit[u] = slot[u] < total ? g_idx[b * total + slot[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (hit[u] >= 0 && hit[u] == (int)it[u]) atomicOr(&s_bad[wave][slot[u] >> 5], 1u << (slot[u] & 31));
        if (__ballot(it[3] < sample_items) == 0ull) break;  // (every lane's last column is past the sample)
    }
    __builtin_amdgcn_wave_barrier();
    // every lane folds its (up to) eight valid slots; the k-th largest of the 64 lane maxima — distinct items, a subset of the valid
    // slot maxima — is <= the k-th largest of them all: one 64-lane sort instead of a selection over 512 values
    float m = kNegInf;
    int mi = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int e = lane + 64 * t;
        const bool bad = e >= total || idx[t] == 0x7fffffff || !(val[t] > -1.0e37f) || ((s_bad[wave][(e >> 5) & 15] >> (e & 31)) & 1u);
        if (!bad && better(val[t], idx[t], m, mi)) m = val[t], mi = idx[t];
    }
    wave_sort_desc_dpp(m, mi, lane);
    const float kth = __shfl(m, k - 1);
    const int kth_i = __shfl(mi, k - 1);
    const float tau = kth_i == 0x7fffffff ? __builtin_inff() : kth;  // fewer than k lanes with a valid slot: no bound
    if (lane == 0) tau_out[b] = tau;
}

// ---- pass 2: every item against tau; passing (item, row) pairs go to the wave's regions -----------------------------------------
template <int S, int UT>
__global__ __launch_bounds__(256) void screen_main_kernel(const ScreenParams p) {
    RBG_SCREEN_DBG_LOAD();
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    // workgroup w runs on XCD w & 7 (round-robin dispatch): XCD x takes the chunks x, x + 8, ... for every user block, so the
    // image tiles one XCD's L2 serves are 1/8 of the image (0.8 MB of 6.5 MB at the Gowalla shape: resident, instead of the
    // whole image streaming through every L2 for every user block)
    if (!RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(0);
    const int cpx = p.n_chunks >> 3;  // chunks per XCD (n_chunks is a multiple of 8)
    const int wg_slot = (int)(blockIdx.x >> 3);
    const int chunk = (int)(blockIdx.x & 7) + 8 * (wg_slot % cpx);
    const int64_t ublock = (int64_t)(wg_slot / cpx) * 4 + wave;
    const int64_t b0 = ublock * (UT * 32);
    if (b0 >= p.B) return;  // (no barrier in this kernel)
    const __bf16 z = (__bf16)0.0f;
    bf16x8 A[UT][S], At[UT];
    float tau[UT];
#pragma unroll
    for (int j = 0; j < UT; ++j) {
        const int64_t bi = b0 + 32 * j + i;
        tau[j] = bi < p.B ? p.tau0[bi] : __builtin_inff();
        const bf16x8 *uimg = reinterpret_cast<const bf16x8 *>(p.uimage) + (((b0 >> 5) + j) * (S + 1)) * 64 + lane;
        if (b0 + 32 * j < p.B) {
#pragma unroll
            for (int s = 0; s < S; ++s) A[j][s] = uimg[s * 64];
            At[j] = uimg[S * 64];
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) A[j][s] = bf16x8{z, z, z, z, z, z, z, z};
            At[j] = bf16x8{z, z, z, z, z, z, z, z};
        }
    }
#pragma unroll
    for (int j = 0; j < UT; ++j) {
        float t = fminf(fmaxf(tau[j], -3.0e38f), 3.0e38f);
        t = fabsf(t) * 1.0e-5f + 1.0e-30f - t;  // -(tau lowered): a pair with s >= tau has tst > 0
        bf16x2 hh, mm, ll;
        split2_bf16(t, 0.f, hh, mm, ll);
        if (h == 0) At[j][0] = hh[0], At[j][1] = mm[0], At[j][2] = ll[0];  // (slots 3-5: ||du||, 1, ||u|| from the image)
    }
    int n_lo[UT], n_hi[UT];  // entries of the two lane halves in region j (wave-uniform: scalar registers)
#pragma unroll
    for (int j = 0; j < UT; ++j) n_lo[j] = n_hi[j] = 0;
    // chunk c takes the item tiles c, c + n_chunks, c + 2 n_chunks, ... : item ids are often ordered by popularity, and the popular
    // items are most users' candidates — in contiguous chunks they all met in the first chunk's regions, which overflowed
    // (the bench's trained tables: 12 regions, each a 10 752-pair exhaustive scan in the merge)
    const int64_t t_begin = chunk, t_end = p.tile_hi, t_step = p.n_chunks;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(p.image) + lane;
    if (RBG_SCREEN_DBGBIT(8)) return;
    uint32_t *const reg0 = p.pool + ((ublock * p.n_chunks + chunk) * UT) * (int64_t)kRegion;
    i32x4v Bn[S + 1];
    if (t_begin < t_end) {
#pragma unroll
        for (int s = 0; s <= S; ++s) Bn[s] = img[(t_begin * (S + 1) + s) * 64];
    }
    if (!RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(1);
    // Software pipeline over the tiles: the sign words of tile t - 1 are formed and its candidates appended while the matrix core
    // works on tile t (a wave issues in order: with the filter behind its own products it filtered OR multiplied, and the pipe
    // idled whenever the SIMD's waves filtered at the same time — 2 640 cycles per tile and SIMD against 1 280 of MFMA time).
    auto product = [&](f32x16 (&acc)[UT], const i32x4v (&Bc)[S + 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At[j], as_frag(Bc[S]), zero, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < UT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][s], as_frag(Bc[s]), acc[j], 0, 0, 0);
    };
    auto fetch = [&](int64_t t, i32x4v (&Bx)[S + 1]) __attribute__((always_inline)) {
        if (t < t_end && !RBG_SCREEN_DBGBIT(4)) {
#pragma unroll
            for (int s = 0; s <= S; ++s) Bx[s] = img[(t * (S + 1) + s) * 64];
        }
    };
    auto sift = [&](const f32x16 (&acc)[UT], int64_t t) __attribute__((always_inline)) {
        const uint32_t code = ((uint32_t)(t * 32 + i) << 5) | ((uint32_t)h << 4);
        unsigned bits[UT];
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            // bit (15 - r) of x = sign of acc[r]: one v_alignbit per register shifts it in
            unsigned x = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) x = __builtin_amdgcn_alignbit(x, __float_as_uint(acc[j][RBG_SCREEN_DBGBIT(2) ? 0 : r]), 31);
            bits[j] = ~x & 0xffffu;  // 1 = tst > 0 (a NaN may pass: the merge drops it)
            if (RBG_SCREEN_DBGBIT(1)) bits[j] = 0u;
        }
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            unsigned long long act = __builtin_amdgcn_ballot_w64(bits[j] != 0u);
            uint32_t *reg = reg0 + j * kRegion;
            while (act != 0ull) {  // one round per entry of the lane with the most (usually one)
                const int c_lo = __popc((unsigned)act), c_hi = __popc((unsigned)(act >> 32));
                if (bits[j] != 0u) {
                    const int q = 31 - __builtin_clz(bits[j]);  // the lowest row first
                    bits[j] &= ~(1u << q);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
                    // the lower half fills the region from its start, the upper half from its end
                    const int pos = h ? (kRegion - 1 - n_hi[j] + c_lo) - rank : n_lo[j] + rank;
                    if (n_lo[j] + n_hi[j] + c_lo + c_hi <= kRegion) reg[pos] = code | (uint32_t)(15 - q);
                }
                n_lo[j] += c_lo;
                n_hi[j] += c_hi;
                act = __builtin_amdgcn_ballot_w64(bits[j] != 0u);
            }
        }
    };
    f32x16 accA[UT], accB[UT];
    i32x4v Bm[S + 1];  // the second fragment set: the tiles alternate between Bn and Bm (no register copies)
#pragma unroll
    for (int s = 0; s <= S; ++s) Bm[s] = Bn[s];
    int64_t t = t_begin;
    if (t < t_end) {
        fetch(t + t_step, Bm);
        product(accA, Bn);
        for (t += t_step; t + t_step < t_end; t += 2 * t_step) {  // two tiles per round: accumulators and fragments swap roles without copies
            fetch(t + t_step, Bn);
            product(accB, Bm);
            sift(accA, t - t_step);
            fetch(t + 2 * t_step, Bm);
            product(accA, Bn);
            sift(accB, t);
        }
        if (t < t_end) {  // an even count of tiles: one more product
            product(accB, Bm);
            sift(accA, t - t_step);
            sift(accB, t);
        } else {
            sift(accA, t - t_step);
        }
    }
    if (!RBG_SCREEN_DBGBIT(1024)) RBG_SCREEN_MAIN_LAP(2);
    if (lane < UT * 2) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            const bool over = n_lo[j] + n_hi[j] > kRegion;  // the merge takes the whole chunk instead
            if (lane == 2 * j) c = over ? -1 : n_lo[j];
            if (lane == 2 * j + 1) c = over ? -1 : n_hi[j];
        }
        p.cnt[((ublock * p.n_chunks + chunk) * UT) * 2 + lane] = c;
    }
}

struct MergeParams {
    const float *U, *I;
    const int64_t *users;
    const int32_t *rowptr, *col;
    int64_t n_users, n_items, B;
    int d, k, ut, n_chunks;
    int64_t n_tiles;
    const uint32_t *pool;
    const int32_t *cnt;
    const float *tau0;
    float *out_val;
    int64_t *out_idx;
};

// One workgroup (16 waves) per lane half of a 32-user tile = the 16 users rowmap(r, half), r = 0..15; wave r folds user r.
// D4 = float4 pieces of a row per lane of a 16-lane group (1: d <= 64, 2: d <= 128); E = candidates a lane group has in flight.
constexpr int kMergeThreads = 1024;
template <bool VEC, int D4, int E>
__global__ __launch_bounds__(kMergeThreads) void screen_merge_kernel(const MergeParams q) {
    RBG_SCREEN_DBG_LOAD();
    constexpr int kGroups = kMergeThreads / 16;
    __shared__ uint32_t s_ent[kSlab];  // item << 4 | r   (~0: dropped — a history or PAD item, a row past the end)
    __shared__ float s_val[kSlab];
    __shared__ unsigned short s_perm[kSlab];
    __shared__ __attribute__((aligned(16))) float s_u[16][kUStride];
    __shared__ int s_cnt[kMaxChunks], s_pref[kMaxChunks + 1];
    __shared__ int s_ucnt[16], s_ustart[17], s_ucur[16];
    __shared__ float s_bv[16][32];
    __shared__ int s_bi[16][32];
    __shared__ int s_hpool[kHPool];
    __shared__ int s_hlo[16], s_hhi[16], s_hoff[16], s_hstaged[16];
    __shared__ float s_tau[16];
    __shared__ int s_exh[16], s_nexh;  // users without a bound (tau = +inf from the threshold kernel): every item is their candidate
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int64_t tile = blockIdx.x >> 1;
    const int half = blockIdx.x & 1;
    const int64_t ublock = tile / q.ut;
    const int j = (int)(tile % q.ut);
    const int64_t b0 = tile * 32;
    RBG_SCREEN_LAP(0);
    // candidates per chunk (an overflowed region: every pair of its chunk) -> exclusive prefix (one lane per chunk)
    if (wave == 0) {
        int c = lane < q.n_chunks ? q.cnt[((ublock * q.n_chunks + lane) * q.ut + j) * 2 + half] : 0;
        if (c < 0) {
            const int64_t tiles_c = (q.n_tiles - lane + q.n_chunks - 1) / q.n_chunks;  // tiles lane, lane + n_chunks, ...
            c = -(int)(tiles_c * 32 * 16);  // (kept negative: the loader below synthesises the pairs)
        }
        s_cnt[lane] = c;
        int incl = c < 0 ? -c : c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        s_pref[lane + 1] = incl;
        if (lane == 0) s_pref[0] = 0;
    }
    // the 16 user rows, their history rows' heads, empty result lists
    for (int f = tid; f < 16 * 128; f += kMergeThreads) {  // (zero past d: the row walk below is guarded on the table's side only)
        const int r = f >> 7, k = f & 127;
        const int64_t b = b0 + mfma_rowmap(r, half);
        const int64_t user = b < q.B ? q.users[b] : -1;
        s_u[r][k] = (user >= 0 && k < q.d) ? q.U[user * (int64_t)q.d + k] : 0.f;
    }
    if (tid < 512) s_bv[tid >> 5][tid & 31] = kNegInf, s_bi[tid >> 5][tid & 31] = 0x7fffffff;
    if (wave == 0) {
        // the 16 history rows share one pool: 256 columns each, what is left goes to the long rows in turn (a hub's whole row is
        // searched in LDS; only beyond the pool does a probe of the search leave for global memory)
        int lo = 0, hi = 0;
        if (lane < 16) {
            const int64_t b = b0 + mfma_rowmap(lane, half);
            const int64_t user = b < q.B ? q.users[b] : -1;
            if (q.rowptr && user >= 0) lo = q.rowptr[user], hi = q.rowptr[user + 1];
            s_tau[lane] = b < q.B ? q.tau0[b] : kNegInf;
        }
        const int len = hi - lo;
        int st = len < 256 ? len : 256;
        int left = kHPool - 16 * 256;
        for (int r = 0; r < 16; ++r) {  // (wave-uniform: 16 short rounds)
            const int want = __shfl(len - st, r);
            const int give = want < left ? want : left;
            left -= give;
            if (lane == r) st += give;
        }
        int incl = lane < 16 ? st : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane < 16) s_hlo[lane] = lo, s_hhi[lane] = hi, s_hstaged[lane] = st, s_hoff[lane] = incl - st;
    }
    __syncthreads();
    {
        const int r = wave;  // one wave per user
        const int lo = s_hlo[r], st = s_hstaged[r], off = s_hoff[r];
        for (int e = lane; e < st; e += 64) s_hpool[off + e] = q.col[lo + e];
    }
    __syncthreads();
    if (tid == 0) {
        int ne = 0;
        for (int r = 0; r < 16; ++r)
            if (s_tau[r] == __builtin_inff()) s_exh[ne++] = r, s_tau[r] = kNegInf;
        s_nexh = ne;
    }
    __syncthreads();
    RBG_SCREEN_LAP(1);
    if (RBG_SCREEN_DBGBIT(64)) return;
    const int n_reg = s_pref[kMaxChunks];  // candidates of the regions; behind them: every item for each user without a bound
    const int n_ent = n_reg + s_nexh * (int)q.n_items;
    const int grp = tid >> 4, l16 = tid & 15;
    for (int base = 0; base < n_ent; base += kSlab) {
        const int n = n_ent - base < kSlab ? n_ent - base : kSlab;
        // a. this slab's candidates: sixteen threads per chunk copy (or synthesise) the entries that fall into the slab
        {
            const int c = tid >> 4, sub = tid & 15;
            const int cc = s_cnt[c], cn = cc < 0 ? -cc : cc, pf = s_pref[c];
            const int o_lo = base - pf > 0 ? base - pf : 0, o_hi = base + kSlab - pf < cn ? base + kSlab - pf : cn;
            const uint32_t *reg = q.pool + ((ublock * q.n_chunks + c) * q.ut + j) * (int64_t)kRegion;

            for (int o = o_lo + sub; o < o_hi; o += 16) {
                uint32_t e;
                if (cc >= 0) {
                    const uint32_t raw = reg[half ? kRegion - 1 - o : o];
                    e = ((raw >> 5) << 4) | (raw & 15u);
                } else {
                    const int p = o >> 4;  // pair p of the chunk: tile c + (p >> 5) n_chunks, item p & 31 of it
                    e = ((uint32_t)(((int64_t)c + (int64_t)(p >> 5) * q.n_chunks) * 32 + (p & 31)) << 4) | (uint32_t)(o & 15);
                }
                s_ent[pf + o - base] = e;
            }
        }
        for (int g = (base > n_reg ? base : n_reg) + tid; g < base + n; g += kMergeThreads) {
            const int o = g - n_reg, x = o / (int)q.n_items;
            s_ent[g - base] = ((uint32_t)(o - x * (int)q.n_items) << 4) | (uint32_t)s_exh[x];
        }
        if (tid < 16) s_ucnt[tid] = 0;
        __syncthreads();
        RBG_SCREEN_LAP(2);
        // b. exact scores: 16 lanes per candidate, E candidates per lane group and round — every load of a round is issued before
        //    the first is used (a row walk with a run-time trip count made the rows dependent round trips); the PAD item, rows
        //    past the end and NaN scores never reach the buckets
        for (int e0 = grp * E; e0 < n && !RBG_SCREEN_DBGBIT(16); e0 += kGroups * E) {
            uint32_t ent[E];
#pragma unroll
            for (int x = 0; x < E; ++x) ent[x] = e0 + x < n ? s_ent[e0 + x] : 0u;
            float part[E];
            bool drop[E];
            auto history = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int x = 0; x < E; ++x) {
                    const int item = (int)(ent[x] >> 4);
                    drop[x] = item == 0 || item >= q.n_items;
                }
            };
            if constexpr (VEC) {
                float4 iv[E][D4];
#pragma unroll
                for (int x = 0; x < E; ++x) {
                    const int64_t item = ent[x] >> 4;
                    const float *row = q.I + (item < q.n_items ? item : 0) * (int64_t)q.d;
#pragma unroll
                    for (int m = 0; m < D4; ++m) {
                        const int c4 = l16 * 4 + 64 * m;
                        iv[x][m] = c4 < q.d ? *reinterpret_cast<const float4 *>(row + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                history();
#pragma unroll
                for (int x = 0; x < E; ++x) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < D4; ++m) {
                        const float4 uv = *reinterpret_cast<const float4 *>(&s_u[ent[x] & 15u][l16 * 4 + 64 * m]);  // (zero past d)
                        a = fmaf(iv[x][m].x, uv.x, a);
                        a = fmaf(iv[x][m].y, uv.y, a);
                        a = fmaf(iv[x][m].z, uv.z, a);
                        a = fmaf(iv[x][m].w, uv.w, a);
                    }
                    part[x] = a;
                }
            } else {
                float iv[E][D4 * 4];
#pragma unroll
                for (int x = 0; x < E; ++x) {
                    const int64_t item = ent[x] >> 4;
                    const float *row = q.I + (item < q.n_items ? item : 0) * (int64_t)q.d;
#pragma unroll
                    for (int m = 0; m < D4 * 4; ++m) iv[x][m] = l16 + 16 * m < q.d ? row[l16 + 16 * m] : 0.f;
                }
                history();
#pragma unroll
                for (int x = 0; x < E; ++x) {
                    float a = 0.f;
#pragma unroll
                    for (int m = 0; m < D4 * 4; ++m) a = fmaf(iv[x][m], s_u[ent[x] & 15u][l16 + 16 * m], a);
                    part[x] = a;
                }
            }
#pragma unroll
            for (int x = 0; x < E; ++x) {
                float a = part[x];
                a += __shfl_xor(a, 8);
                a += __shfl_xor(a, 4);
                a += __shfl_xor(a, 2);
                a += __shfl_xor(a, 1);
                const int e = e0 + x;
                if (l16 == 0 && e < n) {
                    // tau <= the user's k-th best valid exact score: what the margins let in below it is not sorted
                    if (drop[x] || !(a >= s_tau[ent[x] & 15u])) s_ent[e] = ~0u;
                    else s_val[e] = a;
                }
            }
        }
        __syncthreads();
        // b2. history: one thread per surviving candidate searches its user's row in the pool (a hub's thousands of history items
        //     — all of them candidates of a trained model — are spread over the whole workgroup, not searched by the one wave that
        //     folds the user)
        for (int e = tid; e < n; e += kMergeThreads) {
            const uint32_t ent = s_ent[e];
            if (ent == ~0u) continue;
            const int r = (int)(ent & 15u);
            HistRow hist;
            hist.col = (const __attribute__((address_space(1))) int32_t *)q.col;
            hist.lds = s_hpool + s_hoff[r];
            hist.lo = s_hlo[r], hist.hi = s_hhi[r];
            hist.staged = s_hstaged[r];
            hist.n_users = q.n_users;
            if (!RBG_SCREEN_DBGBIT(128) && hist.has((int)(ent >> 4))) s_ent[e] = ~0u;
            else atomicAdd(&s_ucnt[r], 1);
        }
        __syncthreads();
        RBG_SCREEN_LAP(3);
        // c. buckets by user
        if (wave == 0) {
            const int c = lane < 16 ? s_ucnt[lane] : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            if (lane < 16) {
                s_ustart[lane + 1] = incl;
                s_ucur[lane] = incl - c;
            }
            if (lane == 0) s_ustart[0] = 0;
        }
        __syncthreads();
        for (int e = tid; e < n; e += kMergeThreads) {
            const uint32_t ent = s_ent[e];
            if (ent != ~0u) {
                const int pos = atomicAdd(&s_ucur[ent & 15u], 1);
                s_perm[pos] = (unsigned short)e;
            }
        }
        __syncthreads();
        RBG_SCREEN_LAP(4);
        // d. wave r folds the bucket of user r into its best 32 (lower half-wave: best so far; upper: arrivals; a user's first
        //    batch fills all 64 lanes)
        {
            const int r = wave;
            const int r_lo = s_ustart[r], r_hi = s_ustart[r + 1];
            if (r_lo != r_hi && !RBG_SCREEN_DBGBIT(32)) {
                float v = lane < 32 ? s_bv[r][lane] : kNegInf;
                int idx = lane < 32 ? s_bi[r][lane] : 0x7fffffff;
                int first = base == 0 ? 0 : 32;  // lanes >= first take arrivals in the first round
                for (int r0 = r_lo; r0 < r_hi; r0 += 64 - first, first = 32) {
                    if (lane >= first) {
                        const int rr = r0 + lane - first;
                        if (rr < r_hi) {
                            const int e = s_perm[rr];
                            v = s_val[e];
                            idx = (int)(s_ent[e] >> 4);
                        }
                    }
                    // a round whose arrivals are all below the k-th best so far changes nothing (a user without a bound brings
                    // every item: ~ k ln(n / k) rounds sort instead of n / 32)
                    const float kth = __shfl(v, q.k - 1);
                    const bool news = lane >= 32 && (v > kth || (v == kth && v > kNegInf));
                    if (first == 0 || __ballot(news) != 0ull) {
                        if (!RBG_SCREEN_DBGBIT(256)) wave_sort_desc_dpp(v, idx, lane);
                    }
                    if (lane >= 32) v = kNegInf, idx = 0x7fffffff;
                }
                if (lane < 32) s_bv[r][lane] = v, s_bi[r][lane] = idx;
            }
        }
        __syncthreads();
        RBG_SCREEN_LAP(5);
    }
    {
        const int r = wave;
        const int64_t b = b0 + mfma_rowmap(r, half);
        if (b < q.B && lane < q.k) {
            const int idx = s_bi[r][lane];
            q.out_val[b * q.k + lane] = idx == 0x7fffffff ? kNegInf : s_bv[r][lane];
            q.out_idx[b * q.k + lane] = idx == 0x7fffffff ? -1 : (int64_t)idx;
        }
    }
    RBG_SCREEN_LAP(6);
    if (threadIdx.x == 0 && RBG_SCREEN_DBGBIT(512)) ((volatile int *)s_cnt)[0] = n_ent;
#ifdef RBG_SCREEN_DBG
    if (g_screen_trace && threadIdx.x == 0) {
        g_screen_trace[(int64_t)blockIdx.x * 16 + 7] = (unsigned long long)n_ent;
        g_screen_trace[(int64_t)blockIdx.x * 16 + 8] = (unsigned long long)n_reg;
        g_screen_trace[(int64_t)blockIdx.x * 16 + 9] = (unsigned long long)s_nexh;
        int over = 0;
        int first_over = -1;
        for (int c = 0; c < kMaxChunks; ++c) {
            over += s_cnt[c] < 0;
            if (s_cnt[c] < 0 && first_over < 0) first_over = c;
        }
        g_screen_trace[(int64_t)blockIdx.x * 16 + 10] = (unsigned long long)over;
        g_screen_trace[(int64_t)blockIdx.x * 16 + 11] = (unsigned long long)(long long)first_over;
        float tmin = __builtin_inff();
        for (int r = 0; r < 16; ++r) tmin = fminf(tmin, s_tau[r]);
        g_screen_trace[(int64_t)blockIdx.x * 16 + 12] = (unsigned long long)__float_as_uint(tmin);
    }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

static int screen_ut(int d) { return d <= 64 ? kUt64 : kUt128; }

static ScreenLayout layout_for(int64_t B, int64_t n_items, int ut) {
    ScreenLayout L{};
    L.n_tiles = (n_items + 31) / 32;
    L.ut = ut;
    L.n_ublocks = (int)((B + ut * 32 - 1) / (ut * 32));
    // chunk c = the item tiles c, c + nc, ...; nc a multiple of 8 (one residue class of chunks per XCD), at most kMaxChunks
    int64_t want = std::max<int64_t>(1, kWantWaves / std::max(L.n_ublocks, 1));
    want = std::min<int64_t>((want + 7) / 8 * 8, kMaxChunks);
    while (want > 8 && want * 4 > L.n_tiles) want -= 8;
    L.nc = (int)want;
    L.tpc = (int)((L.n_tiles + L.nc - 1) / L.nc);
    L.image_off = 0;
    const int64_t image_bytes = L.n_tiles * 9 * 1024;  // (sized for d <= 128: eight product fragments + the bound fragment)
    L.uimage_off = L.image_off + (image_bytes + 255) / 256 * 256;
    const int64_t uimage_bytes = ((B + 31) / 32 + 8) * 9 * 1024;  // (+ the tiles a partial last user block reads past the batch)
    L.pool_off = L.uimage_off + uimage_bytes;
    const int64_t regions = (int64_t)L.n_ublocks * L.nc * ut;
    L.cnt_off = L.pool_off + regions * kRegion * 4;
    L.bytes = (L.cnt_off + regions * 8 + 255) / 256 * 256;
    // (never for item sets the pre-pass cannot sample — its smallest sample is 1024 items and the main pass needs more than two
    //  samples: a million-row nearest-centroid call must not be handed a 4 KB-per-row candidate pool)
    L.fits = L.n_tiles > 64 && n_items < (1ll << 22) && image_bytes <= (1ll << 30) && L.bytes <= (3ll << 30);
    return L;
}

ScreenLayout screen_layout(int64_t B, int64_t n_items) {
    // sized for either d class (the call picks its own by d): the larger of the two
    ScreenLayout a = layout_for(B, n_items, kUt64), b = layout_for(B, n_items, kUt128);
    ScreenLayout L = a.bytes >= b.bytes ? a : b;
    L.fits = a.fits && b.fits;
    return L;
}

bool screen_applicable(int64_t B, int64_t n_items, int d, int k) {
    if (!opt_topk_screen()) return false;
    if (d > 128 || k > 32) return false;
    if (n_items >= (1ll << 22)) return false;  // (the merge counts candidates in 32 bits even when every region overflows: 64 chunks x n_tiles / 64 x 512 pairs + 16 x n_items)
    if (B < 256 && opt_topk_screen() != 2) return false;  // (measured, profiles/r06_topk_screen_small.jsonl: ahead from 256 users, level at 128)
    return screen_layout(B, n_items).fits;
}

static bool rows_vec(const ScreenCall &c) {
    return (c.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(c.U) | reinterpret_cast<uintptr_t>(c.I)) & 15u) == 0;
}

int screen_prepass(const ScreenCall &c, hipStream_t s) {
    const ScreenLayout L = layout_for(c.B, c.n_items, screen_ut(c.d));
    const bool vec = rows_vec(c);
    char *image = c.w + L.image_off, *uimage = c.w + L.uimage_off;
    const unsigned img_blocks = (unsigned)((L.n_tiles + (c.B + 31) / 32 + 3) / 4);  // item tiles, then the batch's user tiles
    if (c.d <= 64) {
        if (vec) hipLaunchKernelGGL((screen_image_kernel<4, true>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image, c.U, c.users, c.B, uimage);
        else hipLaunchKernelGGL((screen_image_kernel<4, false>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image, c.U, c.users, c.B, uimage);
    } else {
        if (vec) hipLaunchKernelGGL((screen_image_kernel<8, true>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image, c.U, c.users, c.B, uimage);
        else hipLaunchKernelGGL((screen_image_kernel<8, false>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image, c.U, c.users, c.B, uimage);
    }
    RBG_HIP(hipGetLastError());
    ScreenParams p{};
    p.B = c.B;
    p.image = image;
    p.uimage = uimage;
    p.tiles_per_chunk = c.tpc_s;
    p.n_chunks = c.splits;
    p.tile_lo = 0;
    p.tile_hi = c.sample_tiles;
    p.g_val = c.pre_val;
    p.g_idx = c.pre_idx;
    const dim3 grid((unsigned)(c.splits * (((c.B + 31) / 32 + 3) / 4)));
    if (c.d <= 64) hipLaunchKernelGGL((screen_pre_kernel<4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((screen_pre_kernel<8>), grid, dim3(256), 0, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int screen_main(const ScreenCall &c, hipStream_t s) {
    const ScreenLayout L = layout_for(c.B, c.n_items, screen_ut(c.d));
    hipLaunchKernelGGL(screen_tau_kernel, dim3((unsigned)((c.B + 3) / 4)), dim3(256), 0, s, c.pre_val, c.pre_idx, c.users, c.rowptr, c.col,
                       c.n_users, c.B, c.splits, c.tpc_s, c.sample_tiles * 32, c.k, c.tau0);
    RBG_HIP(hipGetLastError());
    const bool vec = rows_vec(c);
    ScreenParams p{};
    p.B = c.B;
    p.image = c.w + L.image_off;
    p.uimage = c.w + L.uimage_off;
    p.n_chunks = L.nc;  // (chunk c = the tiles c, c + nc, ... below tile_hi)
    p.tile_hi = L.n_tiles;
    p.tau0 = c.tau0;
    p.pool = reinterpret_cast<uint32_t *>(c.w + L.pool_off);
    p.cnt = reinterpret_cast<int32_t *>(c.w + L.cnt_off);
    const dim3 grid((unsigned)(L.nc * ((L.n_ublocks + 3) / 4)));
    if (c.d <= 64) hipLaunchKernelGGL((screen_main_kernel<4, kUt64>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((screen_main_kernel<8, kUt128>), grid, dim3(256), 0, s, p);
    RBG_HIP(hipGetLastError());
    MergeParams q{};
    q.U = c.U;
    q.I = c.I;
    q.users = c.users;
    q.rowptr = c.rowptr;
    q.col = c.col;
    q.n_users = c.n_users;
    q.n_items = c.n_items;
    q.B = c.B;
    q.d = c.d;
    q.k = c.k;
    q.ut = L.ut;
    q.n_chunks = L.nc;
    q.n_tiles = L.n_tiles;
    q.pool = p.pool;
    q.cnt = p.cnt;
    q.tau0 = c.tau0;
    q.out_val = c.out_val;
    q.out_idx = c.out_idx;
    const unsigned halves = 2u * (unsigned)((c.B + 31) / 32);
    if (c.d <= 64) {
        if (vec) hipLaunchKernelGGL((screen_merge_kernel<true, 1, 8>), dim3(halves), dim3(kMergeThreads), 0, s, q);
        else hipLaunchKernelGGL((screen_merge_kernel<false, 1, 8>), dim3(halves), dim3(kMergeThreads), 0, s, q);
    } else {
        if (vec) hipLaunchKernelGGL((screen_merge_kernel<true, 2, 4>), dim3(halves), dim3(kMergeThreads), 0, s, q);
        else hipLaunchKernelGGL((screen_merge_kernel<false, 2, 4>), dim3(halves), dim3(kMergeThreads), 0, s, q);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // namespace rbg
