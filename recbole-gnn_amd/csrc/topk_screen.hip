// topk_screen.hip — full-sort top-k as ONE bf16 product per (user, item) + exact rescoring of the few survivors (r06).
//
// The exact passes of topk.hip pay six bf16 MFMA products on 3-way split operands, a split / publish of every item tile and a
// workgroup barrier per 32 items for EVERY (user, item) pair, although > 99.8 % of the pairs are nowhere near a user's top k.
// Here every pair gets one bf16 x bf16 product  s^ = bf16(u) . bf16(i)  and a RIGOROUS bound of what that lost:
//     |s - s^|  <=  (2^-8 + 2^-18) sum_k |u_k i_k|  <=  m_ui := eps ||u|| ||i||,  eps = 1.03 x 2^-8
// (round-to-nearest bf16 is off by <= 2^-9 relative per operand; Cauchy-Schwarz; the 3 % slack covers the matrix core's fp32
// accumulation, the norms' rounding and the fp32 rescoring: all < 1e-5 ||u|| ||i||).  The bound rides on the matrix core like the
// threshold of topk.hip's TauTest: one more MFMA whose A operand carries (-tau_u in three bf16 terms, eps ||u|| rounded UP, 1) and
// whose B operand carries (1, 1, 1, ||i|| rounded UP, -3e38 for the PAD item and rows past the end), so the accumulator holds
//     tst = s^ + m^_ui - tau_u      (m^ >= m, tau lowered by 1e-5 |tau|)
// and "tst >= 0" is one v_max3 tree, one compare, one ballot per 32 x 32 tile.  No pair with s >= tau_u is ever dropped.
//   pass 1 (screen_pass_kernel<PRE>): over the first "topk_sample" items the accumulator holds the LOWER bound s^ - m^; per-lane
//          running maxima as in topk.hip's pre-pass; topk.hip's threshold kernel turns them into tau_u <= the k-th best valid
//          exact score.
//   pass 2 (screen_main_kernel): every item; the sign bits of the 16 accumulator registers are shifted into one word per lane
//          (v_alignbit: 16 instructions), one ballot says whether anything passed; a lane whose entry passes appends
//          (item, accumulator row) to its wave's private region of a candidate pool — the slot is a wave-uniform counter + the
//          lane's rank in the ballot (v_mbcnt): no atomic, no LDS, no exact score, no list pruning, no barrier.  The two lane
//          halves (= two sets of 16 users) fill a region from its two ends.
//   pass 3 (screen_merge_kernel): one workgroup per 16 users (one lane half of a tile) gathers its regions, rescoring every
//          candidate exactly in fp32 (16 lanes per candidate, eight candidates in flight, the item row read from the fp32 table),
//          buckets them by user, drops history / PAD items and folds them into the user's best 32 by the same bitonic network
//          and the same total order as topk.hip.
// No workgroup shares anything in passes 1 and 2: a wave keeps the bf16 rows of UT x 32 users as A fragments (16 registers per
// 32 users at d <= 64 — the split operands of the exact pass take 48) and reads its B fragments straight from an image of the
// item table laid out per tile as [fragment][lane][16 bytes] (1 KiB coalesced wave loads, L1 / L2 resident: 5 KiB per tile), so
// there is no LDS tile, no split, no publish and no barrier in the loop.
// A candidate region that overflows (scores that are all equal, a user whose history covers the whole sample: no threshold) is
// not lost: the merge takes EVERY (user, item) pair of that region's chunk as a candidate instead — slow and exact, no host round
// trip, no second code path.  Results: the same items as the exact passes wherever scores are not tied to the last bit (the
// rescoring sums in a different order: values agree to ~ 1e-7 relative; tests/test_gpu_parity.py::test_full_sort_topk_screen).
//
// Replaces (with topk.hip): lightgcn.py:123-133 + Trainer._full_sort_batch_eval [recbole==1.1.1].

#include "topk_screen.h"

#include <algorithm>

#include "internal.h"
#include "mfma_common.h"
#include "topk_common.h"

namespace rbg {

constexpr float kScreenEps = 0.00390625f * 1.03f;
constexpr int kRegion = 512;   // entries of one (user block, chunk, user tile) region of the pool
constexpr int kSlab = 4096;    // candidates a merge workgroup holds in LDS at a time
constexpr int kMaxChunks = 64;
constexpr int kHS = 64;        // history columns staged per user in the merge kernel
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
                                                           char *__restrict__ image) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const int i = lane & 31, h = lane >> 5;
    const int64_t item = t * 32 + i;
    const bool ok = item < n_items;
    const float *row = I + (ok ? item : 0) * (int64_t)d;
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(image + t * (int64_t)(S + 1) * 1024) + lane;
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const f32x8v v = load8<VEC>(row, ok, 16 * s + 8 * h, d);
        ss += sumsq8(v);
        dst[s * 64] = __builtin_convertvector(v, bf16x8);
    }
    ss += __shfl_xor(ss, 32);
    const float nrm = fminf(sqrtf(ss) * 1.00001f, 1.0e38f);
    const __bf16 z = (__bf16)0.0f, o = (__bf16)1.0f;
    const bool valid = ok && item != 0;  // the PAD item and the rows past the end never pass
    bf16x8 f = {z, z, z, z, z, z, z, z};
    if (h == 0) f = bf16x8{o, o, o, bf16_up(nrm), valid ? z : (__bf16)-3.0e38f, z, z, z};
    dst[S * 64] = f;
}

struct ScreenParams {
    const float *U;
    const int64_t *users;
    int64_t B;
    int d;
    const char *image;
    int tiles_per_chunk, n_chunks;
    int64_t tile_lo, tile_hi;
    const float *tau0;
    uint32_t *pool;
    int32_t *cnt;
    float *g_val;
    int32_t *g_idx;
};

// a lane's A operand: the bf16 row of batch slot b0 + i (elements 16 s + 8 h + [0, 8) per fragment); returns ||row||
template <int S, bool VEC>
__device__ __forceinline__ float load_a(const ScreenParams &p, int64_t bi, int h, bf16x8 (&A)[S]) {
    const int64_t user = bi < p.B ? p.users[bi] : -1;
    const float *row = p.U + (user < 0 ? 0 : user) * (int64_t)p.d;
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const f32x8v v = load8<VEC>(row, user >= 0, 16 * s + 8 * h, p.d);
        ss += sumsq8(v);
        A[s] = __builtin_convertvector(v, bf16x8);
    }
    ss += __shfl_xor(ss, 32);
    return sqrtf(ss) * 1.00001f;
}

// ---- pass 1: lower bounds s^ - m^ of the first tile_hi tiles; per lane and accumulator row the running maximum ----------------
// One 32-user tile per wave, two item tiles per iteration.  The maximum and its item travel in ONE register: the low 8 mantissa
// bits of the bound are replaced by the tile's index inside the chunk (< 256), so a row costs v_and_or x 2 + v_max3 per two tiles
// (topk.hip's pre-pass: compare + two selects per tile).  Truncating moves a bound by < 2^-15 |bound| either way; the margin of this
// pass is widened by 2^-14 ||u|| ||i|| to stay a lower bound.
constexpr float kPreEps = kScreenEps + 6.2e-5f;
template <int S, bool VEC>
__global__ __launch_bounds__(256) void screen_pre_kernel(const ScreenParams p) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t b0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
    if (b0 >= p.B) return;  // (no barrier in this kernel)
    const __bf16 z = (__bf16)0.0f, o = (__bf16)1.0f;
    bf16x8 A[S];
    const float nu = load_a<S, VEC>(p, b0 + i, h, A);
    const __bf16 au = bf16_up(fminf(kPreEps * nu, 1.0e38f));
    bf16x8 At = {z, z, z, z, z, z, z, z};
    if (h == 0) At = bf16x8{z, z, z, bf16_neg(au), o, z, z, z};  // acc = s^ - m^ (+ -3e38 for an invalid item)
    float best[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = kNegInf;
    const int chunk = blockIdx.x;
    const int64_t t_begin = p.tile_lo + (int64_t)chunk * p.tiles_per_chunk;
    const int64_t t_end = (t_begin + p.tiles_per_chunk < p.tile_hi) ? t_begin + p.tiles_per_chunk : p.tile_hi;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(p.image) + lane;
    const int64_t t_last = t_end - 1;
    for (int64_t t = t_begin; t < t_end; t += 2) {
        const int64_t t1 = t + 1 < t_end ? t + 1 : t_last;  // (an odd tail repeats its tile: the maximum does not change)
        i32x4v B0[S + 1], B1[S + 1];
#pragma unroll
        for (int s = 0; s <= S; ++s) B0[s] = img[(t * (S + 1) + s) * 64], B1[s] = img[(t1 * (S + 1) + s) * 64];
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

// ---- pass 2: every item against tau; passing (item, row) pairs go to the wave's regions -----------------------------------------
template <int S, int UT, bool VEC>
__global__ __launch_bounds__(256) void screen_main_kernel(const ScreenParams p) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t ublock = (int64_t)blockIdx.y * 4 + wave;
    const int64_t b0 = ublock * (UT * 32);
    if (b0 >= p.B) return;  // (no barrier in this kernel)
    const __bf16 z = (__bf16)0.0f, o = (__bf16)1.0f;
    bf16x8 A[UT][S], At[UT];
#pragma unroll
    for (int j = 0; j < UT; ++j) {
        const int64_t bi = b0 + 32 * j + i;
        const float nu = load_a<S, VEC>(p, bi, h, A[j]);
        const __bf16 au = bf16_up(fminf(kScreenEps * nu, 1.0e38f));
        float t = bi < p.B ? p.tau0[bi] : __builtin_inff();
        t = fminf(fmaxf(t, -3.0e38f), 3.0e38f);
        t = fabsf(t) * 1.0e-5f + 1.0e-30f - t;  // -(tau lowered): a pair with s >= tau has tst > 0
        bf16x2 hh, mm, ll;
        split2_bf16(t, 0.f, hh, mm, ll);
        bf16x8 f = {z, z, z, z, z, z, z, z};
        if (h == 0) f = bf16x8{hh[0], mm[0], ll[0], au, o, z, z, z};
        At[j] = f;
    }
    int n_lo[UT], n_hi[UT];  // entries of the two lane halves in region j (wave-uniform: scalar registers)
#pragma unroll
    for (int j = 0; j < UT; ++j) n_lo[j] = n_hi[j] = 0;
    const int chunk = blockIdx.x;
    const int64_t t_begin = p.tile_lo + (int64_t)chunk * p.tiles_per_chunk;
    const int64_t t_end = (t_begin + p.tiles_per_chunk < p.tile_hi) ? t_begin + p.tiles_per_chunk : p.tile_hi;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(p.image) + lane;
    uint32_t *const reg0 = p.pool + ((ublock * p.n_chunks + chunk) * UT) * (int64_t)kRegion;
    i32x4v Bn[S + 1];
    if (t_begin < t_end) {
#pragma unroll
        for (int s = 0; s <= S; ++s) Bn[s] = img[(t_begin * (S + 1) + s) * 64];
    }
    for (int64_t t = t_begin; t < t_end; ++t) {
        i32x4v Bc[S + 1];
#pragma unroll
        for (int s = 0; s <= S; ++s) Bc[s] = Bn[s];
        if (t + 1 < t_end) {
#pragma unroll
            for (int s = 0; s <= S; ++s) Bn[s] = img[((t + 1) * (S + 1) + s) * 64];
        }
        f32x16 acc[UT];
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At[j], as_frag(Bc[S]), zero, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < UT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][s], as_frag(Bc[s]), acc[j], 0, 0, 0);
        const uint32_t code = ((uint32_t)(t * 32 + i) << 5) | ((uint32_t)h << 4);
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            // bit (15 - r) of x = sign of acc[r]: one v_alignbit per register shifts it in
            unsigned x = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) x = __builtin_amdgcn_alignbit(x, __float_as_uint(acc[j][r]), 31);
            unsigned bits = ~x & 0xffffu;  // 1 = tst > 0 (a NaN may pass: the merge drops it)
            unsigned long long act = __builtin_amdgcn_ballot_w64(bits != 0u);
            uint32_t *reg = reg0 + j * kRegion;
            while (act != 0ull) {  // one round per entry of the lane with the most (usually one)
                const int c_lo = __popc((unsigned)act), c_hi = __popc((unsigned)(act >> 32));
                if (bits != 0u) {
                    const int q = 31 - __builtin_clz(bits);  // the lowest row first
                    bits &= ~(1u << q);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
                    // the lower half fills the region from its start, the upper half from its end
                    const int pos = h ? (kRegion - 1 - n_hi[j] + c_lo) - rank : n_lo[j] + rank;
                    if (n_lo[j] + n_hi[j] + c_lo + c_hi <= kRegion) reg[pos] = code | (uint32_t)(15 - q);
                }
                n_lo[j] += c_lo;
                n_hi[j] += c_hi;
                act = __builtin_amdgcn_ballot_w64(bits != 0u);
            }
        }
    }
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
    int d, k, ut, n_chunks, tiles_per_chunk;
    int64_t n_tiles;
    const uint32_t *pool;
    const int32_t *cnt;
    float *out_val;
    int64_t *out_idx;
};

// One workgroup (8 waves) per lane half of a 32-user tile = the 16 users rowmap(r, half), r = 0..15.
template <bool VEC>
__global__ __launch_bounds__(512) void screen_merge_kernel(const MergeParams q) {
    __shared__ uint32_t s_ent[kSlab];  // item << 4 | r
    __shared__ float s_val[kSlab];
    __shared__ unsigned short s_perm[kSlab];
    __shared__ __attribute__((aligned(16))) float s_u[16][kUStride];
    __shared__ int s_cnt[kMaxChunks], s_pref[kMaxChunks + 1];
    __shared__ int s_ucnt[16], s_ustart[17], s_ucur[16];
    __shared__ float s_bv[16][32];
    __shared__ int s_bi[16][32];
    __shared__ int s_hist[16][kHS];
    __shared__ int s_hlo[16], s_hhi[16];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int64_t tile = blockIdx.x >> 1;
    const int half = blockIdx.x & 1;
    const int64_t ublock = tile / q.ut;
    const int j = (int)(tile % q.ut);
    const int64_t b0 = tile * 32;
    // candidates per chunk (an overflowed region: every pair of its chunk) -> exclusive prefix (one lane per chunk)
    if (wave == 0) {
        int c = lane < q.n_chunks ? q.cnt[((ublock * q.n_chunks + lane) * q.ut + j) * 2 + half] : 0;
        if (c < 0) {
            const int64_t lo = (int64_t)lane * q.tiles_per_chunk, hi = lo + q.tiles_per_chunk < q.n_tiles ? lo + q.tiles_per_chunk : q.n_tiles;
            c = -(int)((hi - lo) * 32 * 16);  // (kept negative: the loader below synthesises the pairs)
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
    // the 16 user rows, their history bounds, empty result lists
    for (int f = tid; f < 16 * q.d; f += 512) {
        const int r = f / q.d, k = f % q.d;
        const int64_t b = b0 + mfma_rowmap(r, half);
        const int64_t user = b < q.B ? q.users[b] : -1;
        s_u[r][k] = user >= 0 ? q.U[user * (int64_t)q.d + k] : 0.f;
    }
    s_bv[tid >> 5][tid & 31] = kNegInf, s_bi[tid >> 5][tid & 31] = 0x7fffffff;
    {
        const int r = tid >> 5, l32 = tid & 31;  // 32 lanes per user
        const int64_t b = b0 + mfma_rowmap(r, half);
        const int64_t user = b < q.B ? q.users[b] : -1;
        int lo = 0, hi = 0;
        if (q.rowptr && user >= 0) lo = q.rowptr[user], hi = q.rowptr[user + 1];
        const int staged = hi - lo < kHS ? hi - lo : kHS;
        for (int e = l32; e < staged; e += 32) s_hist[r][e] = q.col[lo + e];
        if (l32 == 0) s_hlo[r] = lo, s_hhi[r] = hi;
    }
    __syncthreads();
    const int n_ent = s_pref[kMaxChunks];
    const int grp = tid >> 4, l16 = tid & 15;
    for (int base = 0; base < n_ent; base += kSlab) {
        const int n = n_ent - base < kSlab ? n_ent - base : kSlab;
        // a. this slab's candidates: eight threads per chunk copy (or synthesise) the entries that fall into the slab
        {
            const int c = tid >> 3, sub = tid & 7;
            const int cc = s_cnt[c], cn = cc < 0 ? -cc : cc, pf = s_pref[c];
            const int o_lo = base - pf > 0 ? base - pf : 0, o_hi = base + kSlab - pf < cn ? base + kSlab - pf : cn;
            const uint32_t *reg = q.pool + ((ublock * q.n_chunks + c) * q.ut + j) * (int64_t)kRegion;
            const int64_t item_lo = (int64_t)c * q.tiles_per_chunk * 32;
            for (int o = o_lo + sub; o < o_hi; o += 8) {
                uint32_t e;
                if (cc >= 0) {
                    const uint32_t raw = reg[half ? kRegion - 1 - o : o];
                    e = ((raw >> 5) << 4) | (raw & 15u);
                } else {
                    e = ((uint32_t)(item_lo + (o >> 4)) << 4) | (uint32_t)(o & 15);
                }
                s_ent[pf + o - base] = e;
            }
        }
        if (tid < 16) s_ucnt[tid] = 0;
        __syncthreads();
        // b. exact scores: 16 lanes per candidate, eight candidates in flight per lane group
        for (int e0 = grp * 8; e0 < n; e0 += 32 * 8) {
            float part[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int e = e0 + x;
                const uint32_t ent = e < n ? s_ent[e] : 0u;
                const int64_t item = ent >> 4;
                const int r = ent & 15u;
                const float *row = q.I + (item < q.n_items ? item : 0) * (int64_t)q.d;
                float a = 0.f;
                if constexpr (VEC) {
                    for (int c4 = l16 * 4; c4 < q.d; c4 += 64) {
                        const float4 iv = *reinterpret_cast<const float4 *>(row + c4);
                        const float4 uv = *reinterpret_cast<const float4 *>(&s_u[r][c4]);
                        a = fmaf(iv.x, uv.x, a);
                        a = fmaf(iv.y, uv.y, a);
                        a = fmaf(iv.z, uv.z, a);
                        a = fmaf(iv.w, uv.w, a);
                    }
                } else {
                    for (int c = l16; c < q.d; c += 16) a = fmaf(row[c], s_u[r][c], a);
                }
                part[x] = a;
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float a = part[x];
                a += __shfl_xor(a, 8);
                a += __shfl_xor(a, 4);
                a += __shfl_xor(a, 2);
                a += __shfl_xor(a, 1);
                const int e = e0 + x;
                if (l16 == 0 && e < n) {
                    s_val[e] = a;
                    atomicAdd(&s_ucnt[s_ent[e] & 15u], 1);
                }
            }
        }
        __syncthreads();
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
        for (int e = tid; e < n; e += 512) {
            const int pos = atomicAdd(&s_ucur[s_ent[e] & 15u], 1);
            s_perm[pos] = (unsigned short)e;
        }
        __syncthreads();
        // d. every wave folds the buckets of its two users into their best 32 (lower half-wave: best so far; upper: arrivals;
        //    a user's first batch fills all 64 lanes)
        for (int x = 0; x < 2; ++x) {
            const int r = wave * 2 + x;
            const int r_lo = s_ustart[r], r_hi = s_ustart[r + 1];
            if (r_lo == r_hi) continue;
            HistRow hist;
            hist.col = (const __attribute__((address_space(1))) int32_t *)q.col;
            hist.lds = s_hist[r];
            hist.lo = s_hlo[r], hist.hi = s_hhi[r];
            hist.staged = hist.hi - hist.lo < kHS ? hist.hi - hist.lo : kHS;
            hist.n_users = q.n_users;
            float v = lane < 32 ? s_bv[r][lane] : kNegInf;
            int idx = lane < 32 ? s_bi[r][lane] : 0x7fffffff;
            int first = base == 0 ? 0 : 32;  // lanes >= first take arrivals in the first round
            for (int r0 = r_lo; r0 < r_hi; r0 += 64 - first, first = 32) {
                if (lane >= first) {
                    const int rr = r0 + lane - first;
                    if (rr < r_hi) {
                        const int e = s_perm[rr];
                        const int item = (int)(s_ent[e] >> 4);
                        const float sc = s_val[e];
                        const bool drop = item == 0 || item >= q.n_items || !(sc == sc) || hist.has(item);
                        v = drop ? kNegInf : sc;
                        idx = drop ? 0x7fffffff : item;
                    }
                }
                wave_sort_desc(v, idx, lane);
                if (lane >= 32) v = kNegInf, idx = 0x7fffffff;
            }
            if (lane < 32) s_bv[r][lane] = v, s_bi[r][lane] = idx;
        }
        __syncthreads();
    }
    for (int x = 0; x < 2; ++x) {
        const int r = wave * 2 + x;
        const int64_t b = b0 + mfma_rowmap(r, half);
        if (b < q.B && lane < q.k) {
            const int idx = s_bi[r][lane];
            q.out_val[b * q.k + lane] = idx == 0x7fffffff ? kNegInf : s_bv[r][lane];
            q.out_idx[b * q.k + lane] = idx == 0x7fffffff ? -1 : (int64_t)idx;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

static int screen_ut(int d) { return d <= 64 ? kUt64 : kUt128; }

static ScreenLayout layout_for(int64_t B, int64_t n_items, int ut) {
    ScreenLayout L{};
    L.n_tiles = (n_items + 31) / 32;
    L.ut = ut;
    L.n_ublocks = (int)((B + ut * 32 - 1) / (ut * 32));
    int64_t want = std::max<int64_t>(1, kWantWaves / std::max(L.n_ublocks, 1));
    want = std::min<int64_t>(want, std::min<int64_t>(kMaxChunks, std::max<int64_t>(1, L.n_tiles / 4)));
    L.tpc = (int)((L.n_tiles + want - 1) / want);
    L.nc = (int)((L.n_tiles + L.tpc - 1) / std::max(L.tpc, 1));
    L.image_off = 0;
    const int64_t image_bytes = L.n_tiles * 9 * 1024;  // (sized for d <= 128: eight product fragments + the bound fragment)
    L.pool_off = L.image_off + (image_bytes + 255) / 256 * 256;
    const int64_t regions = (int64_t)L.n_ublocks * L.nc * ut;
    L.cnt_off = L.pool_off + regions * kRegion * 4;
    L.bytes = (L.cnt_off + regions * 8 + 255) / 256 * 256;
    L.fits = image_bytes <= (1ll << 30) && L.bytes <= (3ll << 30);
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
    if (n_items >= (1ll << 26)) return false;  // (item << 5 | accumulator row) in 32 bits
    if (B < 1024 && opt_topk_screen() != 2) return false;
    return screen_layout(B, n_items).fits;
}

static bool rows_vec(const ScreenCall &c) {
    return (c.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(c.U) | reinterpret_cast<uintptr_t>(c.I)) & 15u) == 0;
}

int screen_prepass(const ScreenCall &c, hipStream_t s) {
    const ScreenLayout L = layout_for(c.B, c.n_items, screen_ut(c.d));
    const bool vec = rows_vec(c);
    char *image = c.w + L.image_off;
    const unsigned img_blocks = (unsigned)((L.n_tiles + 3) / 4);
    if (c.d <= 64) {
        if (vec) hipLaunchKernelGGL((screen_image_kernel<4, true>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image);
        else hipLaunchKernelGGL((screen_image_kernel<4, false>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image);
    } else {
        if (vec) hipLaunchKernelGGL((screen_image_kernel<8, true>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image);
        else hipLaunchKernelGGL((screen_image_kernel<8, false>), dim3(img_blocks), dim3(256), 0, s, c.I, c.n_items, L.n_tiles, c.d, image);
    }
    RBG_HIP(hipGetLastError());
    ScreenParams p{};
    p.U = c.U;
    p.users = c.users;
    p.B = c.B;
    p.d = c.d;
    p.image = image;
    p.tiles_per_chunk = c.tpc_s;
    p.n_chunks = c.splits;
    p.tile_lo = 0;
    p.tile_hi = c.sample_tiles;
    p.g_val = c.pre_val;
    p.g_idx = c.pre_idx;
    const dim3 grid((unsigned)c.splits, (unsigned)(((c.B + 31) / 32 + 3) / 4));
    if (c.d <= 64) {
        if (vec) hipLaunchKernelGGL((screen_pre_kernel<4, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((screen_pre_kernel<4, false>), grid, dim3(256), 0, s, p);
    } else {
        if (vec) hipLaunchKernelGGL((screen_pre_kernel<8, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((screen_pre_kernel<8, false>), grid, dim3(256), 0, s, p);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int screen_main(const ScreenCall &c, hipStream_t s) {
    const ScreenLayout L = layout_for(c.B, c.n_items, screen_ut(c.d));
    const bool vec = rows_vec(c);
    ScreenParams p{};
    p.U = c.U;
    p.users = c.users;
    p.B = c.B;
    p.d = c.d;
    p.image = c.w + L.image_off;
    p.tiles_per_chunk = L.tpc;
    p.n_chunks = L.nc;
    p.tile_lo = 0;
    p.tile_hi = L.n_tiles;
    p.tau0 = c.tau0;
    p.pool = reinterpret_cast<uint32_t *>(c.w + L.pool_off);
    p.cnt = reinterpret_cast<int32_t *>(c.w + L.cnt_off);
    const dim3 grid((unsigned)L.nc, (unsigned)((L.n_ublocks + 3) / 4));
    if (c.d <= 64) {
        if (vec) hipLaunchKernelGGL((screen_main_kernel<4, kUt64, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((screen_main_kernel<4, kUt64, false>), grid, dim3(256), 0, s, p);
    } else {
        if (vec) hipLaunchKernelGGL((screen_main_kernel<8, kUt128, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((screen_main_kernel<8, kUt128, false>), grid, dim3(256), 0, s, p);
    }
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
    q.tiles_per_chunk = L.tpc;
    q.n_tiles = L.n_tiles;
    q.pool = p.pool;
    q.cnt = p.cnt;
    q.out_val = c.out_val;
    q.out_idx = c.out_idx;
    const unsigned halves = 2u * (unsigned)((c.B + 31) / 32);
    if (vec) hipLaunchKernelGGL((screen_merge_kernel<true>), dim3(halves), dim3(512), 0, s, q);
    else hipLaunchKernelGGL((screen_merge_kernel<false>), dim3(halves), dim3(512), 0, s, q);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // namespace rbg
