// lse.hip — row-wise log-sum-exp of a scaled dot-product matrix that is never written:
//   lse[b] = log sum_j exp(scale * <Q_b, C_j>)          and its backward (dQ, dC).
//
// Replaces the contrastive denominator of SGL's InfoNCE:
//   v2 = u_emd1.matmul(all_user2.T);  v2 = torch.sum(torch.exp(v2 / self.ssl_tau), dim=1)
//   recbole_gnn/model/general_recommender/sgl.py:195-198 (users), :204-207 (items)
// which materialises [B, n] three times forward (matmul, /tau, exp) and again in autograd — 336 MB each at
// B = 2048 x 40 982 items, 82 GB at BASELINE config #5 (10 M users).  Here a 32x32 tile of the product lives only in
// the accumulator registers of one wavefront (fp32-accurate: bf16 matrix cores on 3-way split operands, or the exact-fp32
// v_mfma_f32_32x32x2_f32 chain with "mfma_split" = 0, as score.hip).
//
// One kernel serves the forward and both gradients.  A wave OWNS 32 rows of one operand ("own", in the MFMA B slot, so
// the tile's column = lane&31 = own row) and LOOPS over 32-row tiles of the other ("oth", A slot, tile row =
// rowmap(reg, lane>>5)):
//   forward : own = Q, oth = C chunked over blockIdx.x;  z[b] += sum_tile exp2(x*s2 - shift2)      -> partial sums
//   dQ      : own = Q, oth = C;   dQ[b]  = sum_j W[b][j] C[j]      W = exp2(..) * coef[b],  coef = g*scale*exp(shift-lse)
//   dC      : own = C, oth = Q;   dC[j]  = sum_b W[b][j] Q[b]
// For the gradients the weight tile is consumed where it is: lane (own = l&31, h) holds W for 16 oth rows rowmap(s,h),
// which is exactly the A-slot layout of a second 32x32x2 MFMA whose k index runs over oth rows; its B slot (oth rows x
// 32 feature columns) is read from a wave-private LDS copy of the oth tile (row stride d+1: conflict-free both ways).
// Nothing is accumulated with atomics: chunk partials are summed in a fixed order by small reduce kernels.
//
// r06: inside rbg_infonce_f32 (unit rows, weights that are probabilities) the two gradient launches run in the fp16 two-term form
// (template flag F16, option "lse_f16"): operands as TWO fp16 terms after a power-of-two scale, three products on
// v_mfma_f32_32x32x16_f16 instead of six bf16 ones, tiles by LDS-DMA from fp16 plane images the row kernels write, the tile loop
// software-pipelined inside a wave (PIPE).  rbg_lse_rows*_f32 and the weighted InfoNCE (the caller's numbers) keep the bf16 / fp32 forms.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>

#include "internal.h"
#include "mfma_common.h"
#include "ordered.h"
#include "plane_image.h"

namespace rbg {


struct LseParams {
    const float *own;
    int64_t ld_own, n_own;
    const float *oth;
    int64_t ld_oth, n_oth;
    int d;
    float s2, shift2;       // weight = exp2(x * s2 - shift2)
    const float *coef_own;  // [n_own] or NULL (gradients)
    const float *coef_oth;  // [n_oth] or NULL
    float *out;             // forward: [n_chunks][n_own]; gradients: [n_chunks][n_own][d]
    float *den_out;         // gradients only, or NULL: [n_chunks][n_own] row sums of the weights (the forward's output) from the same pass
    int tiles_per_chunk, n_chunks;
    int64_t own_blocks, total_blocks, blocks_per_xcd;  // workgroup j = chunk * own_blocks + own_block, j < total_blocks
    const char *oth_image;  // r06: `oth` as bf16 planes in the LDS layout (plane_image.h), taken by LDS-DMA; NULL: fetch + split per workgroup
    // r06, the fp16 form (F16; 0 = not that form): the weights enter the second product as w * w_scale (<= 2^14 by the caller's
    // choice of w_scale), den_out receives zsum * inv_w_scale, out receives g * out_scale
    float w_scale, inv_w_scale, out_scale, w_shift2;
};

// diagnostic builds only (devtools/r06_lse_whatif.sh; results are wrong, timings tell what a phase costs): bit 0 = no second product,
// bit 1 = second product without its transpose reads, bit 2 = weights without exp / split (w = x), bit 3 = no barrier in the tile loop,
// bit 4 = no fetch / publish after the first tile, bit 5 = no first product
#ifndef RBG_LSE_WHATIF
#define RBG_LSE_WHATIF 0
#endif
constexpr int kWhatIf = RBG_LSE_WHATIF;
#ifndef RBG_LSE_STAGGER
#define RBG_LSE_STAGGER 0
#endif
constexpr int kStagger = RBG_LSE_STAGGER;  // experiment: the co-resident workgroups of a CU start 64 x this many cycles apart
constexpr float kF16RowScale = 256.f;  // unit rows as fp16 pairs: x * 2^8 (mfma_common.h, split2_f16)

template <int R>
struct LseRows {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, R>{});
        LseRows<R + 1>::run(f);
    }
};
template <>
struct LseRows<16> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

constexpr int lse_rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// NC = ceil(d / 64) in {1, 2}.  GRAD = false: forward partial sums; true: gradient of the own side.
// A workgroup owns 128 own rows (32 per wave, in registers as the MFMA B fragments) and walks the oth tiles of one
// chunk; every oth tile is fetched ONCE per workgroup into a double-buffered LDS tile (row stride 64 NC + 4 floats:
// 16-byte aligned rows, full-rate b128 reads of the A fragments, conflict-free b32 reads of the second product's B
// fragments).  Workgroups are numbered so that the ones an XCD runs back to back share a chunk: the chunk's oth rows
// stay in that XCD's L2 and cross the fabric once.
// SPLIT (option "mfma_split", default): the score product x = <oth, own> runs on the bf16 matrix cores with both operands
// split into three bf16 terms (mfma_common.h: the accuracy of the fp32 chain, 24 x 32 instead of 32 x 64 cycles per
// 64 k); the oth tile is then published as three bf16 planes, and — for the gradients, whose second product reads the
// tile as fp32 columns — as the fp32 tile too.
// F16 (r06, option "lse_f16"; rbg_infonce_f32's unit rows and weights in [0, 1] only): both products on v_mfma_f32_32x32x16_f16 with
// every operand split into TWO fp16 terms after a power-of-two scale — three products (h h, h l, l h) instead of six, two planes
// instead of three, 32 instead of 48 own-fragment registers: three workgroups per CU at d <= 64, two at d = 128 (one before).
// PASS (F16 only) says at compile time what the two launches of the one-pass InfoNCE differ in — 1: denominators, no row weights;
// 2: the batch rows' coefficients in s_coef, no denominators — so the weights' code has no (uniform) branches between the products.
// PIPE (F16 + IMG only): the tile loop software-pipelined inside a wave — the first product of tile t is issued in front of the exp2 /
// split of tile t - 1 (independent: the matrix core works on t while the vector unit finishes t - 1), then the second product of
// t - 1; three LDS tiles instead of two.
template <int NC, bool GRAD, bool VEC, bool SPLIT, bool TRR = false, bool IMG = false, bool F16 = false, int PASS = 0, bool PIPE = false>
__global__ __launch_bounds__(256, (GRAD ? (NC == 1 ? (SPLIT ? (IMG || F16 ? 3 : 2) : 3) : (F16 ? 2 : 1)) : (NC == 1 ? (SPLIT ? 3 : 4) : 2))) void lse_tile_kernel(const LseParams p) {
    static_assert(!IMG || (SPLIT && TRR), "the plane image serves the split products with transpose reads (no fp32 tile, no transposed copy)");
    static_assert(!F16 || (SPLIT && TRR && GRAD), "the fp16 form: gradient passes, transpose reads");
    static_assert(F16 == (PASS != 0), "PASS belongs to the fp16 form");
    static_assert(!PIPE || (F16 && IMG), "the pipelined loop is written for the fp16 form on plane images");
    constexpr int NBUF = PIPE ? 3 : 2;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int LD = NC * 64 + 4;
    constexpr int LDH = NC * 64 + 8;
    // who reads s_oth: the exact-fp32 products.  (r03: with split operands the gradients' second product takes its B
    // fragments from the SAME bf16 planes the first product reads — the split is elementwise, so the planes already hold
    // the three terms of every tile element; the per-lane re-split of 16 element pairs per tile is gone, and the fp32 tile with it.)
    constexpr bool kFp32Tile = !SPLIT;
    __shared__ __attribute__((aligned(16))) float s_oth[kFp32Tile ? 2 : 1][kFp32Tile ? 32 : 1][kFp32Tile ? LD : 4];
    __shared__ __attribute__((aligned(16))) __bf16 s_pl[SPLIT ? NBUF : 1][NPL][SPLIT ? 32 : 1][SPLIT ? LDH : 8];  // (F16: two planes of fp16 bit patterns)
    // (r03) the gradients' second product reads the tile by COLUMN (8 oth rows of one feature column per lane): a transposed
    // copy of the planes [plane][column][row], row stride 36 bf16 (8-byte aligned, 18 dwords: b64 reads of 32 columns land in
    // 32 different bank pairs), makes a B fragment two 8-byte reads instead of eight 2-byte reads plus their packing
    // r06: the transposed copy is gone — gfx950's LDS transpose read (ds_read_b64_tr_b16) takes the second product's B fragments
    // straight out of the row-major planes: a 16-lane group reads a [4 rows][16 columns] block (each lane supplies the address of
    // one 8-byte piece of it) and every lane receives its column's four rows.  Saves the 24 two-byte, 4-way bank-conflicted LDS
    // stores per thread and tile of the transposed publish and 27.6 KB of LDS per workgroup (option "lse_tr_read" = 0 keeps the copy).
    constexpr int LDT = 36;
    constexpr bool kPlT = SPLIT && GRAD && !TRR;
    __shared__ __attribute__((aligned(16))) __bf16 s_plt[kPlT ? 2 : 1][3][kPlT ? NC * 64 : 1][kPlT ? LDT : 4];
    __shared__ float s_coef[NBUF][32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform, and known to be)
    const int i = lane & 31, h = lane >> 5;
    const int64_t j = (int64_t)(blockIdx.x & 7) * p.blocks_per_xcd + (blockIdx.x >> 3);
    if (j >= p.total_blocks) return;  // whole workgroup
    const int64_t chunk = j / p.own_blocks, ob = j - chunk * p.own_blocks;
    if constexpr (F16 && kStagger > 0) {  // an XCD's 32 CUs take its workgroups 32 at a time: slot = (index in the XCD / 32) % 3
        const int color = (int)((blockIdx.x >> 8) % 3);
        for (int k = 0; k < color * kStagger; ++k) __builtin_amdgcn_s_sleep(1);
    }
    const int64_t own0 = (ob * 4 + wave) * 32;
    const bool wave_live = own0 < p.n_own;  // an idle wave still takes part in the tile loads and barriers
    const int64_t own_row = own0 + i;
    const bool own_ok = own_row < p.n_own;
    float bo[NC][32];
#pragma unroll
    for (int c = 0; c < NC; ++c) load_run32f<(VEC ? RUN_VEC : RUN_ANY)>(p.own + own_row * p.ld_own, own_ok, c * 64 + h * 32, p.d, bo[c]);
    std::conditional_t<SPLIT && !F16, AFrag3<NC>, int> bo3;
    std::conditional_t<F16, AFrag2<NC>, int> bo2;
    if constexpr (F16) split_a_f16(bo, kF16RowScale, bo2);
    else if constexpr (SPLIT) split_a(bo, bo3);
    // (F16: the product of two scaled rows is x * 2^16, and the weights carry w_scale into the second product)
    const float s2 = F16 ? p.s2 * (1.0f / (kF16RowScale * kF16RowScale)) : p.s2;
    // pass 1 folds its power-of-two w_scale into the exponent (w_shift2 = log2 w_scale, exact), pass 2 multiplies the 32 coefficients of a tile
    const float shift2 = F16 ? p.shift2 - p.w_shift2 : p.shift2;
    const float c_own = (!F16 && GRAD && p.coef_own) ? (own_ok ? p.coef_own[own_row] : 0.f) : 1.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float zsum = 0.f;
    f32x16 g[GRAD ? NC * 2 : 1];
#pragma unroll
    for (int q = 0; q < (GRAD ? NC * 2 : 1); ++q) g[q] = zero;

    const int64_t n_tiles = (p.n_oth + 31) / 32;
    const int64_t t0 = chunk * p.tiles_per_chunk;
    const int64_t t1 = (t0 + p.tiles_per_chunk < n_tiles) ? t0 + p.tiles_per_chunk : n_tiles;

    // cooperative tile fetch: 32 rows x NC*16 float4 slots, NC*2 slots per thread
    float4 stage[NC * 2];
    float stage_coef = 0.f;
    auto fetch = [&](const int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NC * 2; ++k) {
            const int f = tid + 256 * k, row = f / (NC * 16), c4 = (f % (NC * 16)) * 4;
            const int64_t r = t * 32 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < p.n_oth) {
                const float *src = p.oth + r * p.ld_oth + c4;
                if (VEC) {
                    if (c4 < p.d) v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (c4 + 0 < p.d) v.x = src[0];
                    if (c4 + 1 < p.d) v.y = src[1];
                    if (c4 + 2 < p.d) v.z = src[2];
                    if (c4 + 3 < p.d) v.w = src[3];
                }
            }
            stage[k] = v;
        }
        if constexpr (F16) {
            if (PASS == 2 && tid < 32) stage_coef = (t * 32 + tid < p.n_oth) ? p.coef_oth[t * 32 + tid] * p.w_scale : 0.f;
        } else {
            if (GRAD && p.coef_oth && tid < 32) stage_coef = (t * 32 + tid < p.n_oth) ? p.coef_oth[t * 32 + tid] : 0.f;
        }
    };
    auto publish = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NC * 2; ++k) {
            const int f = tid + 256 * k, row = f / (NC * 16), c4 = (f % (NC * 16)) * 4;
            if constexpr (kFp32Tile) *reinterpret_cast<float4 *>(&s_oth[buf][row][c4]) = stage[k];
            if constexpr (F16) {
                f16x2 h0, l0, h1, l1;
                split2_f16(stage[k].x, stage[k].y, kF16RowScale, h0, l0);
                split2_f16(stage[k].z, stage[k].w, kF16RowScale, h1, l1);
                *reinterpret_cast<f16x4 *>(&s_pl[buf][0][row][c4]) = (f16x4){h0[0], h0[1], h1[0], h1[1]};
                *reinterpret_cast<f16x4 *>(&s_pl[buf][1][row][c4]) = (f16x4){l0[0], l0[1], l1[0], l1[1]};
            } else if constexpr (SPLIT) {
                bf16x2 h0, m0, l0, h1, m1, l1;
                split2_bf16(stage[k].x, stage[k].y, h0, m0, l0);
                split2_bf16(stage[k].z, stage[k].w, h1, m1, l1);
                *reinterpret_cast<bf16x4 *>(&s_pl[buf][0][row][c4]) = (bf16x4){h0[0], h0[1], h1[0], h1[1]};
                *reinterpret_cast<bf16x4 *>(&s_pl[buf][1][row][c4]) = (bf16x4){m0[0], m0[1], m1[0], m1[1]};
                *reinterpret_cast<bf16x4 *>(&s_pl[buf][2][row][c4]) = (bf16x4){l0[0], l0[1], l1[0], l1[1]};
                if constexpr (kPlT) {
                    s_plt[buf][0][c4 + 0][row] = h0[0], s_plt[buf][0][c4 + 1][row] = h0[1], s_plt[buf][0][c4 + 2][row] = h1[0], s_plt[buf][0][c4 + 3][row] = h1[1];
                    s_plt[buf][1][c4 + 0][row] = m0[0], s_plt[buf][1][c4 + 1][row] = m0[1], s_plt[buf][1][c4 + 2][row] = m1[0], s_plt[buf][1][c4 + 3][row] = m1[1];
                    s_plt[buf][2][c4 + 0][row] = l0[0], s_plt[buf][2][c4 + 1][row] = l0[1], s_plt[buf][2][c4 + 2][row] = l1[0], s_plt[buf][2][c4 + 3][row] = l1[1];
                }
            }
        }
        if constexpr (F16) {
            if (PASS == 2 && tid < 32) s_coef[buf][tid] = stage_coef;
        } else {
            if (GRAD && p.coef_oth && tid < 32) s_coef[buf][tid] = stage_coef;
        }
    };
    auto compute = [&](const int buf, const int64_t t) __attribute__((always_inline)) {
        f32x16 x = zero;
        if constexpr (F16) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = c * 64 + h * 32 + q * 8;
                    const f16x8 ah = *reinterpret_cast<const f16x8 *>(&s_pl[buf][0][i][off]);
                    const f16x8 al = *reinterpret_cast<const f16x8 *>(&s_pl[buf][1][i][off]);
                    const int sidx = c * 4 + q;
                    if constexpr (kWhatIf & 32) {
                        x[q] += (float)ah[0] + (float)al[1];
                        continue;
                    }
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bo2.h[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bo2.l[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bo2.h[sidx], x, 0, 0, 0);
                }
        } else if constexpr (SPLIT) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // oth in the A slot (tile row i, 8 consecutive k), own in the B slot; small terms first
                    const int off = c * 64 + h * 32 + q * 8;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&s_pl[buf][0][i][off]);
                    const bf16x8 am = *reinterpret_cast<const bf16x8 *>(&s_pl[buf][1][i][off]);
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&s_pl[buf][2][i][off]);
                    const int sidx = c * 4 + q;
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bo3.h[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bo3.l[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bo3.m[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bo3.h[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bo3.m[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bo3.h[sidx], x, 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {  // lane (i, h) walks k = 64c + 32h + s of oth row i, as the own fragment does
                    const float4 a = *reinterpret_cast<const float4 *>(&s_oth[buf][i][c * 64 + h * 32 + s4 * 4]);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bo[c][s4 * 4 + 0], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bo[c][s4 * 4 + 1], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bo[c][s4 * 4 + 2], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bo[c][s4 * 4 + 3], x, 0, 0, 0);
                }
        }
        // x[r] = <oth row t*32 + rowmap(r,h), own row own0 + i>
        const int64_t left = p.n_oth - t * 32;  // oth rows of this tile that exist
        if constexpr (!GRAD) {
            LseRows<0>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float e = __builtin_amdgcn_exp2f(x[r] * s2 - p.shift2);
                zsum += (lse_rowmap(r, h) < left) ? e : 0.f;
            });
        } else {
            float w[16];
            if constexpr (F16) {
                LseRows<0>::run([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float e = (kWhatIf & 4) ? x[r] : __builtin_amdgcn_exp2f(x[r] * s2 - shift2);
                    w[r] = PASS == 2 ? e * s_coef[buf][lse_rowmap(r, h)] : e;  // (rows past the end have coefficient 0)
                });
                if constexpr (PASS == 1) {
                    if (left < 32) {  // (uniform: the table's last tile) rows past the end are zero rows, their weight is not
                        const int left32 = (int)left;
                        LseRows<0>::run([&](auto rc) {
                            constexpr int r = decltype(rc)::value;
                            w[r] = (lse_rowmap(r, h) < left32) ? w[r] : 0.f;
                        });
                    }
                    LseRows<0>::run([&](auto rc) { zsum += w[decltype(rc)::value]; });
                }
            } else {
            LseRows<0>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int o = lse_rowmap(r, h);
                float e = __builtin_amdgcn_exp2f(x[r] * s2 - p.shift2) * c_own;
                if (p.coef_oth) e *= s_coef[buf][o];
                w[r] = (o < left) ? e : 0.f;
            });
            if (p.den_out) {  // (uniform) forward and own-side gradient in one pass: the weights' row sums are the denominators
                LseRows<0>::run([&](auto rc) { zsum += w[decltype(rc)::value]; });
            }
            }
            if constexpr (F16) {
                // w <= 2^14 (the caller's w_scale), two fp16 terms; the tile's columns come out of the two planes by transpose reads
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f16x2 hh, ll;
                        if constexpr (kWhatIf & 4) {
                            hh = __builtin_bit_cast(f16x2, w[8 * u + 2 * j]), ll = __builtin_bit_cast(f16x2, w[8 * u + 2 * j + 1]);
                        } else
                        split2_f16(w[8 * u + 2 * j], w[8 * u + 2 * j + 1], 1.f, hh, ll);
                        wh[u][2 * j] = hh[0], wh[u][2 * j + 1] = hh[1];
                        wl[u][2 * j] = ll[0], wl[u][2 * j + 1] = ll[1];
                    }
#pragma unroll
                for (int q = 0; q < NC * 2; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        auto col8 = [&](const int pl) __attribute__((always_inline)) {
                            typedef short s16x4 __attribute__((ext_vector_type(4)));
                            const int t = lane & 15;
                            const __bf16 *src = &s_pl[buf][pl][16 * u + 4 * h + (t >> 2)][q * 32 + (i & 16) + 4 * (t & 3)];
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(src + 8 * LDH));
                            const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
                            return (f16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                        };
                        if constexpr (kWhatIf & 1) {
                            g[q][u] += (float)wh[u][0] + (float)wl[u][1];
                            continue;
                        }
                        const f16x8 bh = (kWhatIf & 2) ? wh[u ^ 1] : col8(0), bl = (kWhatIf & 2) ? wl[u ^ 1] : col8(1);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u], bh, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], bl, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], bh, g[q], 0, 0, 0);
                    }
            } else if constexpr (SPLIT) {
                // second product on the bf16 matrix cores too: MFMA u of a pair covers the oth rows rowmap(8 u + j, h),
                // j = 0..7 — the weights this lane already holds (A slot) against a column of the fp32 tile (B slot), both
                // split here, six products of order >= 2^-16
                bf16x8 wh[2], wm[2], wl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x2 hh, mm, ll;
                        split2_bf16(w[8 * u + 2 * j], w[8 * u + 2 * j + 1], hh, mm, ll);
                        wh[u][2 * j] = hh[0], wh[u][2 * j + 1] = hh[1];
                        wm[u][2 * j] = mm[0], wm[u][2 * j + 1] = mm[1];
                        wl[u][2 * j] = ll[0], wl[u][2 * j + 1] = ll[1];
                    }
#pragma unroll
                for (int q = 0; q < NC * 2; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        // rows rowmap(8 u + j, h), j = 0..7 = 16 u + 4 h + {0..3} and 16 u + 8 + 4 h + {0..3}: two runs of the transposed planes
                        bf16x8 bh, bm, bl;
                        auto col8 = [&](const int pl) __attribute__((always_inline)) {
                            if constexpr (TRR) {
                                // lane t of a 16-lane group addresses piece (row t / 4, columns 4 (t % 4) ..) of the block
                                // [rows 16 u + 4 h (+ 8) .. + 3][columns q 32 + 16 (i / 16) .. + 15]; it receives column q 32 + i, rows .. + 3
                                typedef short s16x4 __attribute__((ext_vector_type(4)));
                                const int t = lane & 15;
                                const __bf16 *src = &s_pl[buf][pl][16 * u + 4 * h + (t >> 2)][q * 32 + (i & 16) + 4 * (t & 3)];
                                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(src + 8 * LDH));
                                const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
                                return (bf16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                            } else {
                                const __bf16 *src = &s_plt[buf][pl][q * 32 + i][16 * u + 4 * h];
                                const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(src), hi = *reinterpret_cast<const bf16x4 *>(src + 8);
                                return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            }
                        };
                        bh = col8(0), bm = col8(1), bl = col8(2);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[u], bh, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[u], bl, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[u], bm, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[u], bh, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[u], bm, g[q], 0, 0, 0);
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[u], bh, g[q], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int q = 0; q < NC * 2; ++q) {
                    LseRows<0>::run([&](auto rc) {
                        constexpr int s = decltype(rc)::value;
                        const float bv = s_oth[buf][lse_rowmap(s, h)][q * 32 + i];
                        g[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], bv, g[q], 0, 0, 0);
                    });
                }
            }
        }
    };
    // Pipeline (r04): iteration t runs  compute(t) from LDS | publish(t + 1) LDS <- stage | fetch(t + 2) -> stage | barrier,
    // so the fetch a publish consumes was issued one whole iteration earlier.  (r03 fetched tile t + 1 at the top of
    // iteration t and published it at the bottom: with the products knocked out the loop still ran at one memory latency
    // per tile.  Measured neutral (399 vs 397 us): the gradient kernel is not bound by the fetch but by the missing overlap of its
    // phases at two workgroups per CU — profiles/r04_lse_phase_probe.jsonl.  r05: raised priority between the first product and
    // the barrier, which pays 3 % in topk.hip: 390-395 vs 394-401 us per InfoNCE with gradients at 2048 x 40 982 — inside the noise, not kept.
    // Three workgroups per CU for the split gradient kernel (transposed planes at row stride 34: 54.0 KB of LDS; launch bounds 3 force
    // 192 -> 168 registers with 20-27 spilled, whatever the order of the second product's loops): 434 / 300 us against 395 / 262 at
    // 40 982 / 29 858 table rows, 479 / 351 with the chunking re-tuned for three — the scratch traffic costs more than the third wave hides.
    // The transposed planes at stride 34 alone (two workgroups per CU): SQ_LDS_BANK_CONFLICT 11.8 M -> 3.9 M cycles per launch (the 2-byte
    // transposed stores of publish() are 4-way conflicted at stride 36), launch time unchanged (398 vs 395 us): not on the critical path.)
    if constexpr (PIPE) {
        auto p1 = [&](const int buf) __attribute__((always_inline)) {
            f32x16 x = zero;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = c * 64 + h * 32 + q * 8;
                    const f16x8 ah = *reinterpret_cast<const f16x8 *>(&s_pl[buf][0][i][off]);
                    const f16x8 al = *reinterpret_cast<const f16x8 *>(&s_pl[buf][1][i][off]);
                    const int sidx = c * 4 + q;
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bo2.h[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bo2.l[sidx], x, 0, 0, 0);
                    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bo2.h[sidx], x, 0, 0, 0);
                }
            return x;
        };
        // exp2, coefficients, end-of-table mask, denominators, fp16 split of tile t (whose coefficients sit in s_coef[buf])
        auto weights = [&](const f32x16 &x, const int buf, const int64_t t, f16x8 (&wh)[2], f16x8 (&wl)[2]) __attribute__((always_inline)) {
            float w[16];
            LseRows<0>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const float e = __builtin_amdgcn_exp2f(x[r] * s2 - shift2);
                w[r] = PASS == 2 ? e * s_coef[buf][lse_rowmap(r, h)] : e;
            });
            if constexpr (PASS == 1) {
                const int64_t left = p.n_oth - t * 32;
                if (left < 32) {
                    const int left32 = (int)left;
                    LseRows<0>::run([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        w[r] = (lse_rowmap(r, h) < left32) ? w[r] : 0.f;
                    });
                }
                LseRows<0>::run([&](auto rc) { zsum += w[decltype(rc)::value]; });
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f16x2 hh, ll;
                    split2_f16(w[8 * u + 2 * jj], w[8 * u + 2 * jj + 1], 1.f, hh, ll);
                    wh[u][2 * jj] = hh[0], wh[u][2 * jj + 1] = hh[1];
                    wl[u][2 * jj] = ll[0], wl[u][2 * jj + 1] = ll[1];
                }
        };
        auto p2 = [&](const int buf, const f16x8 (&wh)[2], const f16x8 (&wl)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NC * 2; ++q)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    auto col8 = [&](const int pl) __attribute__((always_inline)) {
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        const int tt = lane & 15;
                        const __bf16 *src = &s_pl[buf][pl][16 * u + 4 * h + (tt >> 2)][q * 32 + (i & 16) + 4 * (tt & 3)];
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(src + 8 * LDH));
                        const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
                        return (f16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                    };
                    const f16x8 bh = col8(0), bl = col8(1);
                    g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u], bh, g[q], 0, 0, 0);
                    g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], bl, g[q], 0, 0, 0);
                    g[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], bh, g[q], 0, 0, 0);
                }
        };
        const unsigned lds_pl[3] = {(unsigned)(uintptr_t)&s_pl[0], (unsigned)(uintptr_t)&s_pl[1], (unsigned)(uintptr_t)&s_pl[2]};
        constexpr bool has_coef = PASS == 2;
        auto coef_of = [&](const int64_t t) __attribute__((always_inline)) {
            return (has_coef && tid < 32 && t * 32 + tid < p.n_oth) ? p.coef_oth[t * 32 + tid] * p.w_scale : 0.f;
        };
        const int T = (int)(t1 - t0);
        f32x16 xc = zero;
        if (T > 0) {
            PlaneImage<NC, NPL>::dma(p.oth_image, t0, lds_pl[0], tid, wave);
            if (has_coef && tid < 32) s_coef[0][tid] = coef_of(t0);
            dma_drain();
        }
        __syncthreads();
        if (T > 0) {
            if (T > 1) {
                PlaneImage<NC, NPL>::dma(p.oth_image, t0 + 1, lds_pl[1], tid, wave);
                stage_coef = coef_of(t0 + 1);
            }
            if (wave_live) xc = p1(0);
            if (T > 1 && has_coef && tid < 32) s_coef[1][tid] = stage_coef;
            dma_drain();
            __syncthreads();
        }
        int cur = 1, prev = 0;  // LDS tiles of tile k and k - 1; the third one receives tile k + 1
        for (int k = 1; k < T; ++k) {
            const int nxt = 3 - cur - prev;
            if (k + 1 < T) {
                PlaneImage<NC, NPL>::dma(p.oth_image, t0 + k + 1, lds_pl[nxt], tid, wave);  // (last read by tile k - 2, before the previous barrier)
                stage_coef = coef_of(t0 + k + 1);
            }
            if (wave_live) {
                const f32x16 xp = xc;
                xc = p1(cur);
                f16x8 wh[2], wl[2];
                weights(xp, prev, t0 + k - 1, wh, wl);
                p2(prev, wh, wl);
            }
            if (k + 1 < T && has_coef && tid < 32) s_coef[nxt][tid] = stage_coef;
            dma_drain();
            __syncthreads();
            prev = cur, cur = nxt;
        }
        if (T > 0 && wave_live) {
            f16x8 wh[2], wl[2];
            weights(xc, prev, t1 - 1, wh, wl);
            p2(prev, wh, wl);
        }
    } else
    if constexpr (IMG) {
        // r06: the oth tiles come from the plane image by LDS-DMA, one tile ahead; only the tile's 32 weights (gradient of the table
        // side) still travel through registers
        const unsigned lds_pl[2] = {(unsigned)(uintptr_t)&s_pl[0], (unsigned)(uintptr_t)&s_pl[1]};
        const bool has_coef = F16 ? PASS == 2 : (GRAD && p.coef_oth != nullptr);
        auto coef_of = [&](const int64_t t) __attribute__((always_inline)) {
            return (has_coef && tid < 32 && t * 32 + tid < p.n_oth) ? p.coef_oth[t * 32 + tid] * (F16 ? p.w_scale : 1.f) : 0.f;
        };
        if (t0 < t1) {
            PlaneImage<NC, NPL>::dma(p.oth_image, t0, lds_pl[0], tid, wave);
            if (has_coef && tid < 32) s_coef[0][tid] = coef_of(t0);
            dma_drain();
        }
        __syncthreads();
        for (int64_t t = t0; t < t1; ++t) {
            const int buf = (int)(t - t0) & 1;
            if (t + 1 < t1) {
                PlaneImage<NC, NPL>::dma(p.oth_image, t + 1, lds_pl[buf ^ 1], tid, wave);  // (the other buffer was last read before the previous barrier)
                stage_coef = coef_of(t + 1);
            }
            if (wave_live) compute(buf, t);
            if (has_coef && tid < 32 && t + 1 < t1) s_coef[buf ^ 1][tid] = stage_coef;
            dma_drain();
            __syncthreads();
        }
    } else {
    if (t0 < t1) {
        fetch(t0);
        publish(0);
        if (t0 + 1 < t1) fetch(t0 + 1);
    }
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const int buf = (int)(t - t0) & 1;
        if (wave_live) compute(buf, t);
        if constexpr (!(kWhatIf & 16)) {
            if (t + 1 < t1) publish(buf ^ 1);  // the other buffer was last read before the previous barrier
            if (t + 2 < t1) fetch(t + 2);
        }
        if constexpr (!(kWhatIf & 8)) __syncthreads();
    }
    }
    if (!wave_live) return;
    if constexpr (!GRAD) {
        zsum += __shfl_xor(zsum, 32);
        if (h == 0 && own_ok) p.out[chunk * p.n_own + own_row] = zsum;
    } else {
        if (F16 ? PASS == 1 : p.den_out != nullptr) {
            zsum += __shfl_xor(zsum, 32);
            if (h == 0 && own_ok) p.den_out[chunk * p.n_own + own_row] = F16 ? zsum * p.inv_w_scale : zsum;
        }
        // g[q][r]: own row own0 + rowmap(r,h), feature column q*32 + i
        float *base = p.out + chunk * p.n_own * p.d;
#pragma unroll
        for (int q = 0; q < NC * 2; ++q) {
            const int col = q * 32 + i;
            LseRows<0>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int64_t row = own0 + lse_rowmap(r, h);
                if (row < p.n_own && col < p.d) base[row * p.d + col] = F16 ? g[q][r] * p.out_scale : g[q][r];
            });
        }
    }
}

// lse[b] = log(sum_c part[c][b]) + shift
__global__ void lse_finish_kernel(const float *__restrict__ part, int n_chunks, int64_t B, float shift, float *__restrict__ lse) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float z = 0.f;
    for (int c = 0; c < n_chunks; ++c) z += part[(int64_t)c * B + b];
    lse[b] = logf(z) + shift;
}

// coef[b] = g[b] * scale * exp(shift - lse[b])
__global__ void lse_coef_kernel(const float *__restrict__ g, const float *__restrict__ lse, int64_t B, float scale, float shift,
                                float *__restrict__ coef) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) coef[b] = g[b] * scale * expf(shift - lse[b]);
}

// out[x] = sum_c part[c * len + x], fixed order
__global__ void lse_reduce_kernel(const float *__restrict__ part, int n_chunks, int64_t len, float *__restrict__ out) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= len) return;
    float a = 0.f;
    for (int c = 0; c < n_chunks; ++c) a += part[(int64_t)c * len + x];
    out[x] = a;
}

// Chunking.  Measured (per-workgroup clock trace, r01): the kernel is matrix-core bound per CU, all workgroups start at
// once and the dispatcher spreads them evenly, so the launch lasts as long as the fullest CU: ceil(blocks / 256)
// workgroups x tiles per chunk.  832 workgroups (3.25 per CU) ran 130 us where 768 run 89 us.  Pick the chunk count
// minimising that product, discounted when fewer than 3 waves per SIMD are left to hide the LDS/exp phases, plus the
// cost of one partial array per chunk (`chunk_cost`, in tile units).
static void lse_geometry(int64_t n_own, int64_t n_oth, int resident, double chunk_cost, int &tiles_per_chunk, int &n_chunks) {
    const int64_t own_blocks = std::max<int64_t>(1, (n_own + 127) / 128), oth_tiles = std::max<int64_t>(1, (n_oth + 31) / 32);
    const int64_t max_chunks = std::max<int64_t>(1, std::min<int64_t>(oth_tiles / 4, 64));
    int64_t best = 1;
    double best_cost = 1e300;
    for (int64_t c = 1; c <= max_chunks; ++c) {
        const int64_t per_cu = (own_blocks * c + 255) / 256, tiles = (oth_tiles + c - 1) / c;
        const int64_t conc = std::min<int64_t>(per_cu, resident);
        double eff = conc >= 3 ? 1.0 : (conc == 2 ? 0.85 : 0.6);
        if (per_cu > resident && per_cu < 3 * resident) eff *= 0.8;  // a short second round runs with idle SIMDs
        const double cost = (double)(per_cu * tiles) / eff + chunk_cost * (double)c;
        if (cost < best_cost - 1e-9) {
            best = c;
            best_cost = cost;
        }
    }
    tiles_per_chunk = (int)((oth_tiles + best - 1) / best);
    n_chunks = (int)((oth_tiles + tiles_per_chunk - 1) / tiles_per_chunk);
}

struct LseLayout {
    int tpc_f, nc_f;  // forward (own = Q)
    int tpc_q, nc_q;  // dQ (own = Q)
    int tpc_c, nc_c;  // dC (own = C)
    int64_t off_coef, off_q, off_c, off_den, bytes;
};
// the fp16 form of the gradient passes (rbg_infonce_f32 without weights; F16 in lse_tile_kernel) is on
// (1 = its tiles from fp16 plane images by LDS-DMA, 2 = fetched and split per workgroup, 3 = 1 with the tile loop software-pipelined: default)
static int lse_f16_mode(int d) { return (opt_mfma_split() != 0 && opt_lse_tr_read() && !opt_lse_image() && d % 4 == 0) ? opt_lse_f16() : 0; }
static bool lse_f16_on(int d) { return lse_f16_mode(d) != 0; }

static LseLayout lse_layout_of(int64_t B, int64_t n, int d, int f16) {  // f16: 0 = the bf16 / fp32 kernels, 1 / 2 = lse_f16_mode
    LseLayout L{};
    const bool split = opt_mfma_split() != 0;  // (the split kernels hold 48 instead of 32 own registers per chunk)
    // resident workgroups per CU (register-limited; r06: the gradient kernel that takes its tiles from plane images and its column
    // fragments by transpose reads fits three at d <= 64: 168 registers, 28 KB of LDS)
    const bool img3 = split && opt_lse_image() && opt_lse_tr_read();
    const int res_f = d <= 64 ? (split ? 3 : 4) : 2, res_g = f16 ? (d <= 64 ? 3 : 2) :  // (the image form's 124 registers would allow four: chunking for four measured 214-218 vs 211 us)
                    d <= 64 ? (split ? (img3 ? 3 : 2) : 3) : 1;
    // one partial array per chunk: own rows x d x 8 bytes (write + read) at ~5 TB/s, in units of a ~1.2 us tile
    auto partial_cost = [&](int64_t rows) { return (double)rows * d * 8.0 / 5e6 / 1.2; };
    lse_geometry(B, n, res_f, 0.02, L.tpc_f, L.nc_f);
    lse_geometry(B, n, res_g, partial_cost(B), L.tpc_q, L.nc_q);
    lse_geometry(n, B, res_g, partial_cost(n), L.tpc_c, L.nc_c);
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    L.off_coef = 0;
    L.off_q = up(B * 4);  // forward partial sums [nc_q][B] live here too (before the gradients need the space)
    const int64_t q_bytes = std::max<int64_t>((int64_t)L.nc_q * B * d * 4, (int64_t)L.nc_f * B * 4);
    L.off_c = L.off_q + up(q_bytes);
    const int64_t c_bytes = L.nc_c > 1 ? (int64_t)L.nc_c * n * d * 4 : 0;
    L.off_den = L.off_c + up(c_bytes);  // [nc_q][B]: the denominators' partial sums of the one-pass form
    L.bytes = L.off_den + up((int64_t)L.nc_q * B * 4) + 256;
    return L;
}
// (a workspace serves either form: the masked InfoNCE keeps the bf16 kernels while the option is on)
static LseLayout lse_layout(int64_t B, int64_t n, int d, int f16 = 0) {
    LseLayout L = lse_layout_of(B, n, d, f16);
    if (lse_f16_on(d))
        for (int m = 0; m <= 2; ++m) L.bytes = std::max(L.bytes, lse_layout_of(B, n, d, m).bytes);
    return L;
}

template <int NC, bool GRAD>
static void lse_launch(LseParams p, bool vec, hipStream_t s) {
    p.own_blocks = (p.n_own + 127) / 128;
    p.total_blocks = p.own_blocks * p.n_chunks;
    p.blocks_per_xcd = (p.total_blocks + 7) / 8;
    dim3 grid((unsigned)(p.blocks_per_xcd * 8));
    const bool split = opt_mfma_split() != 0;
    if constexpr (GRAD) {  // r06: the gradients' second product reads its B fragments by LDS transpose reads (option "lse_tr_read", default 1)
        if (p.w_scale != 0.f) {  // the fp16 form (the caller checked lse_f16_mode and the alignment)
            if (p.oth_image && opt_lse_f16() == 3) {  // the software-pipelined loop (default): 212.5 -> 209 us per call, 691 -> 686 at d = 128
                if (p.den_out) hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, true, true, 1, true>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, true, true, 2, true>), grid, dim3(256), 0, s, p);
            } else if (p.oth_image) {
                if (p.den_out) hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, true, true, 1>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, true, true, 2>), grid, dim3(256), 0, s, p);
            } else {
                if (p.den_out) hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, false, true, 1>), grid, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, false, true, 2>), grid, dim3(256), 0, s, p);
            }
            return;
        }
        if (split && opt_lse_tr_read()) {
            if (p.oth_image) hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true, true>), grid, dim3(256), 0, s, p);
            else if (vec) hipLaunchKernelGGL((lse_tile_kernel<NC, true, true, true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((lse_tile_kernel<NC, true, false, true, true>), grid, dim3(256), 0, s, p);
            return;
        }
    }
    if (vec && split)
        hipLaunchKernelGGL((lse_tile_kernel<NC, GRAD, true, true>), grid, dim3(256), 0, s, p);
    else if (vec)
        hipLaunchKernelGGL((lse_tile_kernel<NC, GRAD, true, false>), grid, dim3(256), 0, s, p);
    else if (split)
        hipLaunchKernelGGL((lse_tile_kernel<NC, GRAD, false, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((lse_tile_kernel<NC, GRAD, false, false>), grid, dim3(256), 0, s, p);
}

template <bool GRAD>
static void lse_launch_d(const LseParams &p, bool vec, hipStream_t s) {
    if (p.d <= 64) lse_launch<1, GRAD>(p, vec, s);
    else lse_launch<2, GRAD>(p, vec, s);
}

static int lse_check(const float *Q, int64_t ldq, int64_t B, const float *C, int64_t ldc, int64_t n, int d) {
    if (B < 0 || n < 0 || d <= 0) return fail(RBG_ESHAPE, "B = %lld, n = %lld, d = %d", (long long)B, (long long)n, d);
    if (d > 128) return fail(RBG_EUNSUPPORTED, "lse_rows: d = %d > 128", d);
    if (ldq < d || ldc < d) return fail(RBG_ESHAPE, "row stride smaller than d");
    if ((B && !Q) || (n && !C)) return fail(RBG_EINVAL, "NULL pointer");
    return RBG_OK;
}

static bool lse_vec(const float *Q, int64_t ldq, const float *C, int64_t ldc, int d) {
    return (d % 4 == 0) && (ldq % 4 == 0) && (ldc % 4 == 0) &&
           ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(C)) & 15u) == 0;
}

constexpr float kLog2e = 1.4426950408889634f;

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_lse_rows_workspace(int64_t B, int64_t n, int d, int64_t *bytes) {
    if (!bytes || B < 0 || n < 0 || d <= 0) return fail(RBG_EINVAL, "bad argument");
    *bytes = lse_layout(B, n, d).bytes;
    return RBG_OK;
}

int rbg_lse_rows_f32(const float *Q, int64_t ldq, int64_t B, const float *C, int64_t ldc, int64_t n, int d, float scale,
                     float shift, float *lse, void *workspace, void *stream) {
    clear_error();
    int rc = lse_check(Q, ldq, B, C, ldc, n, d);
    if (rc) return rc;
    if (B == 0) return RBG_OK;
    if (!lse || !workspace) return fail(RBG_EINVAL, "NULL pointer");
    if (n == 0) return fail(RBG_ESHAPE, "lse_rows over an empty candidate set");
    const LseLayout L = lse_layout(B, n, d);
    hipStream_t s = (hipStream_t)stream;
    float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + L.off_q);
    LseParams p{};
    p.own = Q, p.ld_own = ldq, p.n_own = B;
    p.oth = C, p.ld_oth = ldc, p.n_oth = n;
    p.d = d;
    p.s2 = scale * kLog2e;
    p.shift2 = shift * kLog2e;
    p.out = part;
    p.tiles_per_chunk = L.tpc_f;
    p.n_chunks = L.nc_f;
    lse_launch_d<false>(p, lse_vec(Q, ldq, C, ldc, d), s);
    RBG_HIP(hipGetLastError());
    hipLaunchKernelGGL(lse_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, part, L.nc_f, B, shift, lse);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_lse_rows_backward_f32(const float *Q, int64_t ldq, int64_t B, const float *C, int64_t ldc, int64_t n, int d,
                              float scale, float shift, const float *lse, const float *grad_lse, float *grad_Q,
                              float *grad_C, void *workspace, void *stream) {
    clear_error();
    int rc = lse_check(Q, ldq, B, C, ldc, n, d);
    if (rc) return rc;
    if (!workspace || (B && (!lse || !grad_lse))) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0 || n == 0) {
        if (grad_Q && B && (rc = zero_async(grad_Q, (size_t)B * d * 4, s))) return rc;
        if (grad_C && n && (rc = zero_async(grad_C, (size_t)n * d * 4, s))) return rc;
        return RBG_OK;
    }
    const LseLayout L = lse_layout(B, n, d);
    char *w = reinterpret_cast<char *>(workspace);
    float *coef = reinterpret_cast<float *>(w + L.off_coef);
    float *part_q = reinterpret_cast<float *>(w + L.off_q);
    float *part_c = reinterpret_cast<float *>(w + L.off_c);
    const bool vec = lse_vec(Q, ldq, C, ldc, d);
    hipLaunchKernelGGL(lse_coef_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, grad_lse, lse, B, scale, shift, coef);
    RBG_HIP(hipGetLastError());
    LseParams p{};
    p.d = d;
    p.s2 = scale * kLog2e;
    p.shift2 = shift * kLog2e;
    if (grad_Q) {
        p.own = Q, p.ld_own = ldq, p.n_own = B;
        p.oth = C, p.ld_oth = ldc, p.n_oth = n;
        p.coef_own = coef, p.coef_oth = nullptr;
        p.tiles_per_chunk = L.tpc_q, p.n_chunks = L.nc_q;
        p.out = L.nc_q > 1 ? part_q : grad_Q;
        lse_launch_d<true>(p, vec, s);
        RBG_HIP(hipGetLastError());
        if (L.nc_q > 1) {
            const int64_t len = B * d;
            hipLaunchKernelGGL(lse_reduce_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, part_q, L.nc_q, len, grad_Q);
            RBG_HIP(hipGetLastError());
        }
    }
    if (grad_C) {
        p.own = C, p.ld_own = ldc, p.n_own = n;
        p.oth = Q, p.ld_oth = ldq, p.n_oth = B;
        p.coef_own = nullptr, p.coef_oth = coef;
        p.tiles_per_chunk = L.tpc_c, p.n_chunks = L.nc_c;
        p.out = L.nc_c > 1 ? part_c : grad_C;
        lse_launch_d<true>(p, vec, s);
        RBG_HIP(hipGetLastError());
        if (L.nc_c > 1) {
            const int64_t len = n * d;
            hipLaunchKernelGGL(lse_reduce_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, part_c, L.nc_c, len, grad_C);
            RBG_HIP(hipGetLastError());
        }
    }
    return RBG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// InfoNCE between two views of one embedding table (one half of SGL.calc_ssl_loss, sgl.py:191-199 / :201-208), value
// and gradients in one call: everything around the lse kernels — F.normalize of the table and of the batch rows, the
// gathers, the positive term, the backward of the normalisations and the row scatters — is a handful of row kernels
// here instead of ~75 elementwise / reduction / index launches of 4-5 us each in torch autograd.
// ---------------------------------------------------------------------------------------------------------------------
namespace rbg {

constexpr float kNormEps = 1e-12f;  // F.normalize default eps

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// r06: row j of a unit-row table as the two fp16 planes of lse_tile_kernel's fp16 form, in the tile image PlaneImage<NC, 2>
// ([tile][plane][32 rows][64 NC + 8] fp16 bit patterns); lane c holds columns c and c + 64; rows past the end are written as zeros by
// their would-be waves (the second product multiplies them by a zero weight: they must be finite)
__device__ __forceinline__ void nce_image_row(char *img, int64_t j, int d, int lane, float v0, float v1) {
    const int ldh = d <= 64 ? 72 : 136;
    _Float16 *tile = reinterpret_cast<_Float16 *>(img + (j >> 5) * (int64_t)(2 * 32 * ldh * 2));
    const int row = (int)(j & 31);
    f16x2 h, l;
    split2_f16(v0, v1, kF16RowScale, h, l);
    if (lane < ldh - 8) tile[row * ldh + lane] = h[0], tile[(32 + row) * ldh + lane] = l[0];
    if (d > 64) tile[row * ldh + lane + 64] = h[1], tile[(32 + row) * ldh + lane + 64] = l[1];
}

// C[j] = T[j] / max(||T[j]||, eps), inv[j] = 1 / max(||T[j]||, eps).  One wave per row.  gidx != NULL (r06, the batch form:
// rbg_infonce_batch_f32): row j of the candidate table is row gidx[j] of T.
__global__ __launch_bounds__(256) void nce_norm_table_kernel(const float *__restrict__ T, int64_t n, int d, float *__restrict__ C,
                                                             float *__restrict__ inv, char *__restrict__ img16,
                                                             const int64_t *__restrict__ gidx) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (j >= n) {
        if (img16 && j < (n + 31) / 32 * 32) nce_image_row(img16, j, d, lane, 0.f, 0.f);
        return;
    }
    const float *row = T + (gidx ? gidx[j] : j) * d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss = fmaf(row[c], row[c], ss);
    ss = wave_sum(ss);
    const float iv = 1.0f / fmaxf(sqrtf(ss), kNormEps);
    for (int c = lane; c < d; c += 64) C[j * d + c] = row[c] * iv;
    if (lane == 0) inv[j] = iv;
    if (img16) nce_image_row(img16, j, d, lane, lane < d ? row[lane] * iv : 0.f, lane + 64 < d ? row[lane + 64] * iv : 0.f);
}

// A[b] = normalize(T1[idx[b]]), inv1[b], pos[b] = <A[b], C[idx[b]]>.  One wave per batch row.  c_by_pos (the batch form): the
// candidate table IS the batch, its row of position b is C[b].  pos_map (rbg_infonce_map_f32): the positive of row r is C[pos_map[r]].
__global__ __launch_bounds__(256) void nce_batch_prep_kernel(const float *__restrict__ T1, const float *__restrict__ C,
                                                             const int64_t *__restrict__ idx, int64_t B, int d,
                                                             float *__restrict__ A, float *__restrict__ inv1, float *__restrict__ pos,
                                                             char *__restrict__ img16, int c_by_pos, const int64_t *__restrict__ pos_map) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= B) {
        if (img16 && b < (B + 31) / 32 * 32) nce_image_row(img16, b, d, lane, 0.f, 0.f);
        return;
    }
    const int64_t r = idx[b];
    const float *row = T1 + r * d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss = fmaf(row[c], row[c], ss);
    ss = wave_sum(ss);
    const float iv = 1.0f / fmaxf(sqrtf(ss), kNormEps);
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float a = row[c] * iv;
        A[b * d + c] = a;
        dot = fmaf(a, C[(pos_map ? pos_map[r] : (c_by_pos ? b : r)) * d + c], dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        inv1[b] = iv;
        pos[b] = dot;
    }
    if (img16) nce_image_row(img16, b, d, lane, lane < d ? row[lane] * iv : 0.f, lane + 64 < d ? row[lane + 64] * iv : 0.f);
}

// *loss += weight * sum_b (lse[b] - scale * pos[b]);  gl[b] = weight (the upstream gradient of every lse[b]).  One block.
__global__ __launch_bounds__(256) void nce_loss_kernel(const float *__restrict__ lse, const float *__restrict__ pos, int64_t B,
                                                       float scale, float weight, float *__restrict__ loss, float *__restrict__ gl) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 256) {  // fixed order per thread, fixed tree below: reproducible
        acc += lse[b] - scale * pos[b];
        gl[b] = weight;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss += weight * (((part[0] + part[1]) + part[2]) + part[3]);
}

// Batch rows, backward: gA = dA[b] - weight*scale*C[r]  (lse term + positive term), pushed through normalize into
// grad_T1[r]; the positive term's gradient on the table side, -weight*scale*A[b], is added to dC[r].  r = idx[b] may
// repeat inside a batch: float atomics, like torch's GPU index_add.
struct NceBackArgs {
    const float *dA, *A, *C, *inv1;
    const int64_t *idx;
    int64_t B;
    int d;
    float ws;
    float *dC, *grad_T1;
};

template <bool ORDERED>
__device__ __forceinline__ void nce_batch_back_elem(const NceBackArgs &a, int64_t b, int lane) {
    const int d = a.d;
    const int64_t r = a.idx[b];
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float g = a.dA[b * d + c] - a.ws * a.C[r * d + c];
        dot = fmaf(g, a.A[b * d + c], dot);
    }
    dot = wave_sum(dot);
    const float iv = a.inv1[b];
    const bool clamped = iv >= 1.0f / kNormEps;  // ||x|| < eps: normalize is x / eps, a plain scaling
    for (int c = lane; c < d; c += 64) {
        const float av = a.A[b * d + c];
        const float g = a.dA[b * d + c] - a.ws * a.C[r * d + c];
        if (a.grad_T1) row_add<ORDERED>(a.grad_T1 + r * d + c, (clamped ? g : g - av * dot) * iv);
        row_add<ORDERED>(a.dC + r * d + c, -a.ws * av);
    }
}

__global__ __launch_bounds__(256) void nce_batch_back_kernel(const NceBackArgs a) {
    const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= a.B) return;
    nce_batch_back_elem<false>(a, b, threadIdx.x & 63);
}

// option "deterministic" (ordered.h): the rows of a repeated id are written by the wavefront of its first occurrence, in batch order.
// NOTE: an occurrence reads C[r] and writes dC[r] / grad_T1[r] — different buffers — so owners and non-owners never race.
struct NceBackRows {
    NceBackArgs a;
    __device__ __forceinline__ int segments(int64_t w, KeySeg (&seg)[2], int64_t &mine) const {
        mine = a.idx[w];
        seg[0] = KeySeg{a.idx, a.B, 0};
        return 1;
    }
    __device__ __forceinline__ void apply(int64_t m, int lane) const { nce_batch_back_elem<true>(a, m, lane); }
};

// Table rows, backward of C = normalize(T2): grad_T2[j] += (g - C[j] <g, C[j]>) * inv[j],  g = dC[j].
__global__ __launch_bounds__(256) void nce_table_back_kernel(const float *__restrict__ dC, const float *__restrict__ C,
                                                             const float *__restrict__ inv, int64_t n, int d,
                                                             float *__restrict__ grad_T2) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (j >= n) return;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) dot = fmaf(dC[j * d + c], C[j * d + c], dot);
    dot = wave_sum(dot);
    const float iv = inv[j];
    const bool clamped = iv >= 1.0f / kNormEps;
    for (int c = lane; c < d; c += 64) {
        const float g = dC[j * d + c];
        grad_T2[j * d + c] += (clamped ? g : g - C[j * d + c] * dot) * iv;
    }
}

// ---- the one-pass form's small kernels, fused (r04): 11 launches per InfoNCE half -> 7 ----------------------------------
// den[b] = sum_c den_part[c][b];  lse = log(den) + shift;  term[b] = lse - scale pos[b];  coef[b] = weight * scale / den[b]
// (= the upstream gradient weight times scale * exp(shift - lse)).  One thread per batch row; the loss is the fixed-order sum of
// term[] by nce_sum_kernel (one block: reproducible).
__global__ __launch_bounds__(256) void nce_finish_kernel(const float *__restrict__ den_part, int n_chunks, const float *__restrict__ pos,
                                                         const float *__restrict__ row_w, int64_t B, float scale, float shift, float weight,
                                                         float *__restrict__ term, float *__restrict__ coef) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float den = 0.f;
#pragma unroll 8
    for (int c = 0; c < n_chunks; ++c) den += den_part[(int64_t)c * B + b];
    const float lse = logf(den) + shift;
    const float r = row_w ? row_w[b] : 1.f;  // (rbg_infonce_masked_f32: a row's weight in the sum)
    term[b] = r * (lse - scale * pos[b]);
    coef[b] = r * weight * scale * expf(shift - lse);
}

__global__ __launch_bounds__(256) void nce_sum_kernel(const float *__restrict__ term, int64_t B, float weight, float *__restrict__ loss) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 256) acc += term[b];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss += weight * (((part[0] + part[1]) + part[2]) + part[3]);
}

// nce_batch_back_kernel with dA[b] = coef[b] * sum_c part_q[c][b] formed here (the chunk reduction of the batch-side gradient)
struct NceBackPartsArgs {
    const float *part_q;
    int n_chunks;
    const float *coef, *A, *C, *inv1;
    const int64_t *idx;
    const float *row_w;
    int64_t B;
    int d;
    float ws;
    float *dC, *grad_T1;
    int c_by_pos;  // the batch form: C and dC are indexed by the position b, grad_T1 by idx[b]
    const int64_t *pos_map;  // rbg_infonce_map_f32: C and dC are indexed by pos_map[idx[b]]
};

template <bool ORDERED>
__device__ __forceinline__ void nce_batch_back_parts_elem(const NceBackPartsArgs &a, int64_t b, int lane) {
    const int d = a.d;
    const int64_t B = a.B;
    const int64_t r = a.idx[b], rc = a.pos_map ? a.pos_map[r] : (a.c_by_pos ? b : r);
    const float cb = a.coef[b];
    const float ws = a.row_w ? a.ws * a.row_w[b] : a.ws;  // the positive term carries the row's weight too
    float g0 = 0.f, g1 = 0.f;  // d <= 128: columns lane and lane + 64
    const bool in0 = lane < d, in1 = lane + 64 < d;
#pragma unroll 8
    for (int c = 0; c < a.n_chunks; ++c) {  // (unrolled: the chunks' loads go out together instead of one latency each)
        const float *src = a.part_q + ((int64_t)c * B + b) * d;
        g0 += in0 ? src[lane] : 0.f;
        g1 += in1 ? src[lane + 64] : 0.f;
    }
    g0 = lane < d ? g0 * cb - ws * a.C[rc * d + lane] : 0.f;
    g1 = lane + 64 < d ? g1 * cb - ws * a.C[rc * d + lane + 64] : 0.f;
    const float a0 = lane < d ? a.A[b * d + lane] : 0.f, a1 = lane + 64 < d ? a.A[b * d + lane + 64] : 0.f;
    const float dot = wave_sum(fmaf(g0, a0, g1 * a1));
    const float iv = a.inv1[b];
    const bool clamped = iv >= 1.0f / kNormEps;
    if (lane < d) {
        if (a.grad_T1) row_add<ORDERED>(a.grad_T1 + r * d + lane, (clamped ? g0 : g0 - a0 * dot) * iv);
        row_add<ORDERED>(a.dC + rc * d + lane, -ws * a0);
    }
    if (lane + 64 < d) {
        if (a.grad_T1) row_add<ORDERED>(a.grad_T1 + r * d + lane + 64, (clamped ? g1 : g1 - a1 * dot) * iv);
        row_add<ORDERED>(a.dC + rc * d + lane + 64, -ws * a1);
    }
}

__global__ __launch_bounds__(256) void nce_batch_back_parts_kernel(const NceBackPartsArgs a) {
    const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= a.B) return;
    nce_batch_back_parts_elem<false>(a, b, threadIdx.x & 63);
}

struct NceBackPartsRows {
    NceBackPartsArgs a;
    __device__ __forceinline__ int segments(int64_t w, KeySeg (&seg)[2], int64_t &mine) const {
        mine = a.idx[w];
        seg[0] = KeySeg{a.idx, a.B, 0};
        return 1;
    }
    __device__ __forceinline__ void apply(int64_t m, int lane) const { nce_batch_back_parts_elem<true>(a, m, lane); }
};

// nce_table_back_kernel with g = sum_c part_c[c][j] formed here (the chunk reduction of the table-side gradient; the batch rows'
// positive-term contributions were added onto chunk 0 by the kernel above)
// sidx != NULL (the batch form): candidate j is row sidx[j] of the gradient table; ids repeat, so the rows are added with float atomics
// (a repeated id's other positions carry zero weight: they add exact zeros — the sum does not depend on the order)
__global__ __launch_bounds__(256) void nce_table_back_parts_kernel(const float *__restrict__ part_c, int n_chunks, const float *__restrict__ C,
                                                                   const float *__restrict__ inv, int64_t n, int d, float *__restrict__ grad_T2,
                                                                   const int64_t *__restrict__ sidx) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (j >= n) return;
    float g0 = 0.f, g1 = 0.f;
    const bool in0 = lane < d, in1 = lane + 64 < d;
#pragma unroll 8
    for (int c = 0; c < n_chunks; ++c) {
        const float *src = part_c + ((int64_t)c * n + j) * d;
        g0 += in0 ? src[lane] : 0.f;
        g1 += in1 ? src[lane + 64] : 0.f;
    }
    const float c0 = lane < d ? C[j * d + lane] : 0.f, c1 = lane + 64 < d ? C[j * d + lane + 64] : 0.f;
    const float dot = wave_sum(fmaf(g0, c0, g1 * c1));
    const float iv = inv[j];
    const bool clamped = iv >= 1.0f / kNormEps;
    if (sidx) {
        float *dst = grad_T2 + sidx[j] * d;
        if (lane < d) atomicAdd(dst + lane, (clamped ? g0 : g0 - c0 * dot) * iv);
        if (lane + 64 < d) atomicAdd(dst + lane + 64, (clamped ? g1 : g1 - c1 * dot) * iv);
        return;
    }
    if (lane < d) grad_T2[j * d + lane] += (clamped ? g0 : g0 - c0 * dot) * iv;
    if (lane + 64 < d) grad_T2[j * d + lane + 64] += (clamped ? g1 : g1 - c1 * dot) * iv;
}

struct NceLayout {
    int64_t off_C, off_inv2, off_A, off_inv1, off_pos, off_lse, off_gl, off_dA, off_dC, off_lse_ws, bytes;
    int64_t off_imgC, off_imgA;  // r06: plane images of the normalised table and batch rows (d <= 128)
};
static int64_t nce_image_bytes(int64_t rows, int d) {
    return (rows + 31) / 32 * (int64_t)(d <= 64 ? PlaneImage<1>::kTileBytes : PlaneImage<2>::kTileBytes);
}
static NceLayout nce_layout(int64_t B, int64_t n, int d) {
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    NceLayout L{};
    int64_t o = 0;
    L.off_C = o, o += up(n * d * 4);
    L.off_inv2 = o, o += up(n * 4);
    L.off_A = o, o += up(B * d * 4);
    L.off_inv1 = o, o += up(B * 4);
    L.off_pos = o, o += up(B * 4);
    L.off_lse = o, o += up(B * 4);
    L.off_gl = o, o += up(B * 4);
    L.off_dA = o, o += up(B * d * 4);
    L.off_dC = o, o += up(n * d * 4);
    L.off_lse_ws = o, o += up(lse_layout(B, n, d).bytes);
    L.off_imgC = o, o += up(nce_image_bytes(n, d));
    L.off_imgA = o, o += up(nce_image_bytes(B, d));
    L.bytes = o + 256;
    return L;
}

// rbg_infonce_f32 with gradients, one-pass form (r04, option "lse_onepass").  With a FIXED shift (unit rows: 1 / tau; no running
// maximum) the unnormalised batch-side gradient sum_j w[b][j] C[j] accumulates beside the denominator sum_j w[b][j] in ONE pass over
// the table, so the separate forward launch (57 us of a 385 us half at 2048 x 40 982, d = 64) is gone; the normalisation moves
// into the consumers of the chunk partials, which also absorb the chunk reductions: 8 launches per half instead of 13.
static int infonce_onepass(const float *A, const float *C, const float *inv1, const float *inv2, const float *pos, float *term, float *dC,
                           const int64_t *idx, const float *row_w, const float *col_w, int64_t n, int d, int64_t B, float scale, float weight,
                           float *loss, float *grad_T1, float *grad_T2, void *lse_ws, hipStream_t s, char *imgC = nullptr, char *imgA = nullptr,
                           int f16_mode = 0, const int64_t *batch_ids = nullptr, const int64_t *pos_map = nullptr) {
    const bool vec = lse_vec(A, d, C, d, d);
    // r06, the fp16 form: unit rows and weights in [0, 1] — the plain InfoNCE only (row / candidate weights are the caller's numbers);
    // f16_mode 1: imgC / imgA hold the fp16 plane images the row kernels wrote
    const bool f16 = f16_mode != 0;
    const LseLayout L = lse_layout(B, n, d, f16_mode);
    char *w = reinterpret_cast<char *>(lse_ws);
    float *coef = reinterpret_cast<float *>(w + L.off_coef), *part_q = reinterpret_cast<float *>(w + L.off_q);
    float *part_c = L.nc_c > 1 ? reinterpret_cast<float *>(w + L.off_c) : dC, *den = reinterpret_cast<float *>(w + L.off_den);
    // r06: the normalised table and batch rows as plane images (plane_image.h), built once per call: both gradient passes take their
    // oth tiles by LDS-DMA instead of fetching, splitting and publishing them in every workgroup
    const bool img = !f16 && imgC && imgA && d <= 128 && opt_mfma_split() != 0 && opt_lse_tr_read() && opt_lse_image();
    if (img) {
        const unsigned tc = (unsigned)((n + 31) / 32), ta = (unsigned)((B + 31) / 32);
        if (d <= 64) {
            hipLaunchKernelGGL((plane_image_kernel<1, false>), dim3(tc), dim3(256), 0, s, C, (int64_t)d, n, d, imgC);
            hipLaunchKernelGGL((plane_image_kernel<1, false>), dim3(ta), dim3(256), 0, s, A, (int64_t)d, B, d, imgA);
        } else {
            hipLaunchKernelGGL((plane_image_kernel<2, false>), dim3(tc), dim3(256), 0, s, C, (int64_t)d, n, d, imgC);
            hipLaunchKernelGGL((plane_image_kernel<2, false>), dim3(ta), dim3(256), 0, s, A, (int64_t)d, B, d, imgA);
        }
        RBG_HIP(hipGetLastError());
    }
    LseParams p{};
    p.d = d;
    p.s2 = scale * kLog2e;
    p.shift2 = scale * kLog2e;
    // pass 1 (own = the batch rows): denominators' and gradient's partials per chunk
    p.own = A, p.ld_own = d, p.n_own = B;
    p.oth = C, p.ld_oth = d, p.n_oth = n;
    p.oth_image = (img || (f16_mode & 1)) ? imgC : nullptr;
    p.tiles_per_chunk = L.tpc_q, p.n_chunks = L.nc_q;
    p.out = part_q;
    p.den_out = den;
    p.coef_oth = col_w;  // (masked form: a candidate's weight in every denominator; NULL = 1)
    if (f16) {  // pass 1: w = exp(..) <= 1 enters as w 2^14; the partials leave as g 2^-14 2^-8, the denominators as zsum 2^-14 (all exact)
        p.w_scale = 16384.f, p.inv_w_scale = 1.f / 16384.f, p.w_shift2 = 14.f;
        p.out_scale = p.inv_w_scale / kF16RowScale;
    }
    lse_launch_d<true>(p, vec, s);
    RBG_HIP(hipGetLastError());
    // (r06: both in one single-block launch measured SLOWER — 215 vs 211 us per call, 132 vs 101 at 5 000 rows: 2 048 rows x 48 chunk
    //  partials in one workgroup is a latency chain)
    hipLaunchKernelGGL(nce_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, den, L.nc_q, pos, row_w, B, scale, scale, weight, term, coef);
    hipLaunchKernelGGL(nce_sum_kernel, dim3(1), dim3(256), 0, s, term, B, weight, loss);
    RBG_HIP(hipGetLastError());
    if (!grad_T1 && !grad_T2) return RBG_OK;  // (value only; the masked form has no other forward)
    // pass 2 (own = the table rows): needs the finished denominators (coef); only for the table's gradient (r06: rbg_infonce_map_f32
    // against constant prototypes asks for the batch rows' alone — its positive term still lands in part_c, a scratch nobody reads)
    if (grad_T2) {
    p.own = C, p.n_own = n;
    p.oth = A, p.n_oth = B;
    p.oth_image = (img || (f16_mode & 1)) ? imgA : nullptr;
    p.coef_own = col_w, p.coef_oth = coef;
    p.tiles_per_chunk = L.tpc_c, p.n_chunks = L.nc_c;
    p.out = part_c;
    p.den_out = nullptr;
    if (f16) {  // pass 2: w = exp(..) coef[b] = (weight scale) x a probability: enters as p 2^14, leaves multiplied back
        const float ws = weight * scale, k = 16384.f / ws;
        const bool ok = ws != 0.f && std::isfinite(k) && std::isfinite(ws / 16384.f);
        p.w_scale = ok ? k : 16384.f, p.inv_w_scale = 1.f / p.w_scale, p.w_shift2 = 0.f;
        p.out_scale = (ok ? ws / 16384.f : 1.f / 16384.f) / kF16RowScale;
    }
    lse_launch_d<true>(p, vec, s);
    RBG_HIP(hipGetLastError());
    }
    const unsigned nb = (unsigned)((n + 3) / 4), bb = (unsigned)((B + 3) / 4);
    {
        const NceBackPartsArgs ba{part_q, L.nc_q, coef, A, C, inv1, idx, row_w, B, d, weight * scale, part_c, grad_T1, batch_ids ? 1 : 0, pos_map};
        if (opt_deterministic()) launch_ordered_scatter(NceBackPartsRows{ba}, B, s);
        else hipLaunchKernelGGL(nce_batch_back_parts_kernel, dim3(bb), dim3(256), 0, s, ba);
    }
    RBG_HIP(hipGetLastError());
    if (grad_T2) {
        hipLaunchKernelGGL(nce_table_back_parts_kernel, dim3(nb), dim3(256), 0, s, part_c, L.nc_c, C, inv2, n, d, grad_T2, batch_ids);
        RBG_HIP(hipGetLastError());
    }
    return RBG_OK;
}

}  // namespace rbg

extern "C" {

int rbg_infonce_workspace(int64_t B, int64_t n, int d, int64_t *bytes) {
    if (!bytes || B < 0 || n < 0 || d <= 0) return fail(RBG_EINVAL, "bad argument");
    *bytes = nce_layout(B, n, d).bytes;
    return RBG_OK;
}

// batch_form (rbg_infonce_batch_f32): the candidates are the batch's own rows T2[idx[b]] (n = B), the gradients go to rows idx[b] of both tables
static int infonce_impl(const float *T1, const float *T2, int64_t n, int d, const int64_t *idx, int64_t B, float tau, float weight,
                        const float *row_w, const float *col_w, float *loss, float *grad_T1, float *grad_T2, void *workspace, void *stream,
                        bool batch_form = false, const int64_t *pos_map = nullptr) {
    clear_error();
    if (B < 0 || n <= 0 || d <= 0) return fail(RBG_ESHAPE, "B = %lld, n = %lld, d = %d", (long long)B, (long long)n, d);
    if (d > 128) return fail(RBG_EUNSUPPORTED, "infonce: d = %d > 128", d);
    if (!(tau > 0.f)) return fail(RBG_EINVAL, "tau must be positive");
    if (B == 0) return RBG_OK;
    if (!T1 || !T2 || !idx || !loss || !workspace) return fail(RBG_EINVAL, "NULL pointer");
    const NceLayout L = nce_layout(B, n, d);
    char *w = reinterpret_cast<char *>(workspace);
    auto f = [&](int64_t off) { return reinterpret_cast<float *>(w + off); };
    float *C = f(L.off_C), *inv2 = f(L.off_inv2), *A = f(L.off_A), *inv1 = f(L.off_inv1), *pos = f(L.off_pos);
    float *lse = f(L.off_lse), *gl = f(L.off_gl), *dA = f(L.off_dA), *dC = f(L.off_dC);
    void *lse_ws = w + L.off_lse_ws;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / tau;
    const unsigned nb = (unsigned)((n + 3) / 4), bb = (unsigned)((B + 3) / 4);
    const bool grads = grad_T1 || grad_T2;
    const bool masked = row_w || col_w;
    const bool onepass = masked || batch_form || pos_map || (grads && opt_lse_onepass());  // denominators and dA out of one pass over the table
    const int64_t *bids = batch_form ? idx : nullptr;
    // r06: the fp16 form of the gradient launches (the plain InfoNCE with gradients); mode 1: the row kernels write the fp16 plane images too
    const int f16_mode = (onepass && grads && !masked) ? lse_f16_mode(d) : 0;
    char *imgC = grads ? w + L.off_imgC : nullptr, *imgA = grads ? w + L.off_imgA : nullptr;
    const unsigned nb_img = (unsigned)((n + 31) / 32 * 8), bb_img = (unsigned)((B + 31) / 32 * 8);  // (whole tiles: the rows past the end are zeroed)
    hipLaunchKernelGGL(nce_norm_table_kernel, dim3((f16_mode & 1) ? nb_img : nb), dim3(256), 0, s, T2, n, d, C, inv2, (f16_mode & 1) ? imgC : nullptr, bids);
    RBG_HIP(hipGetLastError());
    hipLaunchKernelGGL(nce_batch_prep_kernel, dim3((f16_mode & 1) ? bb_img : bb), dim3(256), 0, s, T1, C, idx, B, d, A, inv1, pos, (f16_mode & 1) ? imgA : nullptr,
                       batch_form ? 1 : 0, pos_map);
    RBG_HIP(hipGetLastError());
    if (onepass) return infonce_onepass(A, C, inv1, inv2, pos, gl, dC, idx, row_w, col_w, n, d, B, scale, weight, loss, grad_T1, grad_T2, lse_ws, s,
                                        imgC, imgA, f16_mode, bids, pos_map);
    int rc = rbg_lse_rows_f32(A, d, B, C, d, n, d, scale, scale, lse, lse_ws, stream);  // unit rows: shift = 1/tau
    if (rc) return rc;
    hipLaunchKernelGGL(nce_loss_kernel, dim3(1), dim3(256), 0, s, lse, pos, B, scale, weight, loss, gl);
    RBG_HIP(hipGetLastError());
    if (!grads) return RBG_OK;
    rc = rbg_lse_rows_backward_f32(A, d, B, C, d, n, d, scale, scale, lse, gl, dA, dC, lse_ws, stream);
    if (rc) return rc;
    {
        const NceBackArgs ba{dA, A, C, inv1, idx, B, d, weight * scale, dC, grad_T1};
        if (opt_deterministic()) launch_ordered_scatter(NceBackRows{ba}, B, s);
        else hipLaunchKernelGGL(nce_batch_back_kernel, dim3(bb), dim3(256), 0, s, ba);
    }
    RBG_HIP(hipGetLastError());
    if (grad_T2) {
        hipLaunchKernelGGL(nce_table_back_kernel, dim3(nb), dim3(256), 0, s, dC, C, inv2, n, d, grad_T2);
        RBG_HIP(hipGetLastError());
    }
    return RBG_OK;
}

int rbg_infonce_f32(const float *T1, const float *T2, int64_t n, int d, const int64_t *idx, int64_t B, float tau,
                    float weight, float *loss, float *grad_T1, float *grad_T2, void *workspace, void *stream) {
    return infonce_impl(T1, T2, n, d, idx, B, tau, weight, nullptr, nullptr, loss, grad_T1, grad_T2, workspace, stream);
}

int rbg_infonce_masked_f32(const float *T1, const float *T2, int64_t n, int d, const int64_t *idx, int64_t B, float tau, float weight,
                           const float *row_w, const float *col_w, float *loss, float *grad_T1, float *grad_T2, void *workspace,
                           void *stream) {
    return infonce_impl(T1, T2, n, d, idx, B, tau, weight, row_w, col_w, loss, grad_T1, grad_T2, workspace, stream);
}

int rbg_infonce_map_f32(const float *T1, const float *T2, int64_t n2, int d, const int64_t *idx, const int64_t *pos_map, int64_t B, float tau,
                        float weight, float *loss, float *grad_T1, float *grad_T2, void *workspace, void *stream) {
    if (B > 0 && !pos_map) {
        clear_error();
        return fail(RBG_EINVAL, "infonce_map: pos_map is NULL");
    }
    return infonce_impl(T1, T2, n2, d, idx, B, tau, weight, nullptr, nullptr, loss, grad_T1, grad_T2, workspace, stream, false, pos_map);
}

int rbg_infonce_batch_f32(const float *TA, const float *TB, int d, const int64_t *ids, int64_t B, float tau, float weight,
                          const float *row_w, const float *col_w, float *loss, float *grad_TA, float *grad_TB, void *workspace,
                          void *stream) {
    if (B > 0 && ((grad_TA != nullptr) != (grad_TB != nullptr)))
        return fail(RBG_EINVAL, "infonce_batch: both gradient tables or none");
    return infonce_impl(TA, TB, B > 0 ? B : 1, d, ids, B, tau, weight, row_w, col_w, loss, grad_TA, grad_TB, workspace, stream, true);
}

}  // extern "C"
