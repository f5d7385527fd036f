// bignn.hip — NGCF's bi-interaction layer on gfx950.
//
// Replaces BiGNNConv.forward  (recbole_gnn/model/layers.py:54-58)
//     x_prop  = propagate(x)                 -> P = ÂX           (one SpMM, spmm.hip)
//     x_trans = lin1(x_prop + x)
//     x_inter = lin2(x_prop * x)             -> the "Hadamard term": (ÂX) ⊙ X, an epilogue, not a 2nd SpMM
//     return x_trans + x_inter
// and, with RBG_BIGNN_LEAKY_NORM, the per-layer tail of NGCF.forward (ngcf.py:96,98):
//     LeakyReLU(0.2) -> F.normalize(p=2, dim=1)   (message dropout, ngcf.py:97, is not applied)
//
// The dense part is ONE GEMM with the two linears concatenated along k:
//     Y = [P+X | P⊙X] (N x 2d_in) · [W1^T ; W2^T] (2d_in x d_out) + (b1 + b2)
// run on v_mfma_f32_32x32x2_f32 (exact fp32).  A wavefront owns 32 rows and all d_out columns, so the
// row L2 norm is a 32-lane reduction inside the wave (DPP).  The A operand is built in registers straight
// from the lane's contiguous 128-byte runs of P and X (same k-walk trick as score.hip); the
// concatenated weight matrix is staged once per workgroup in LDS as Wl[part][j][k] (coalesced loads, row stride
// d_in + 4) and read back as b128 B fragments.
// Per-wave clock trace (r01, Gowalla shape, one 32-row tile per wave, all 2 216 waves resident at once): the kernel
// runs as three lock-stepped phases — staging + barrier ~10 us, MFMA ~5 us, epilogue + stores ~8 us — so the matrix
// core is a minor term.  The staging phase is dominated by the row loads themselves (each lane's own 128-byte run =
// 64 cache lines per instruction); fetching rows coalesced and transposing them through LDS cut that phase to 6.8 us
// but cost 4.5 us of LDS passes and a resident workgroup (measured, not kept).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "internal.h"
#include "lds_dma.h"
#include "mfma_common.h"

namespace rbg {


struct BignnParams {
    const float *P;  // [N, d_in] contiguous
    const float *X;
    int64_t ldx;
    const float *W1, *b1, *W2, *b2;
    float *Y;
    int64_t ldy;
    int64_t n_rows;
    int d_in, d_out;
    int leaky_norm;
    float slope;
    float *inv_norm;  // optional [N]: 1 / max(||LeakyReLU(z)||, eps) per row, what the tail's backward needs
    const float *drop_mask;  // optional [N, d_out] contiguous: 0 or 1/(1-p), applied between LeakyReLU and normalize (ngcf.py:97)
};

// Sum over the 32 lanes of a wave half, result in every lane: four DPP adds (xor 1, xor 2 as quad permutes, then
// half-mirror and mirror inside the 16-lane row) and ONE cross-row exchange, instead of five dependent ds_bpermute.
__device__ __forceinline__ float row32_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
    return x + __shfl_xor(x, 16);
}

// NT = number of 32-column output tiles held by a wave (d_out <= 32*NT).
// FAST = d_in is a multiple of 64 and P / X / W rows are 16-byte aligned (the NGCF configuration).
// LDS holds W as Wl[part][j][k] (part 0: W1 against P+X, 1: W2 against P⊙X), DP = 32*NT rows of KP = 64 nch + 4 floats.
// (A variant that feeds B straight from global memory with no LDS and no barrier measured slower:
//  42.8 vs 34.3 us at the Gowalla shape.)
template <int NT, bool FAST>
__global__ __launch_bounds__(256, (NT <= 2 ? 3 : 1)) void bignn_dense_kernel(const BignnParams p) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    constexpr int DP = 32 * NT;
    const int nch = (p.d_in + 63) / 64;  // k chunks per part
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (p.n_rows + 31) / 32;
    const int64_t tile_first = (int64_t)blockIdx.x * 4 + wave;
    // Row operands of this wave's first tile go out first: their latency hides under the weight staging.
    // (rows past the end are clamped for the loads and masked at the store)
    float a1[32], a2[32];
    {
        const int64_t r = min(tile_first * 32 + i, p.n_rows - 1);
        load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(p.P + r * p.d_in, true, 32 * h, p.d_in, a1);
        load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(p.X + r * p.ldx, true, 32 * h, p.d_in, a2);
    }
    // Stage [W1 ; W2] as Wl[part][j][k] (row stride KP = 64 nch + 4 floats, zero-padded to DP rows and 64 nch columns).
    // Coalesced: consecutive threads fetch consecutive float4s of a weight row, and a whole batch of loads is in flight
    // before the first LDS write (the first version took two dependent round trips of lane-strided loads).
    {
        const int KP = nch * 64 + 4;
        const int slots_per_row = nch * 16, per_part = DP * slots_per_row, total = 2 * per_part;
        constexpr int BATCH = 8;
        for (int f0 = threadIdx.x; f0 < total; f0 += 256 * BATCH) {
            float4 w[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + 256 * u;
                w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < total) {
                    const int part = f >= per_part, g = f - part * per_part;
                    const int j = g / slots_per_row, k = 4 * (g % slots_per_row);
                    if (j < p.d_out) {
                        const float *src = (part ? p.W2 : p.W1) + (int64_t)j * p.d_in + k;
                        if (FAST) {
                            w[u] = *reinterpret_cast<const float4 *>(src);
                        } else {
                            if (k + 0 < p.d_in) w[u].x = src[0];
                            if (k + 1 < p.d_in) w[u].y = src[1];
                            if (k + 2 < p.d_in) w[u].z = src[2];
                            if (k + 3 < p.d_in) w[u].w = src[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + 256 * u;
                if (f < total) {
                    const int part = f >= per_part, g = f - part * per_part;
                    const int j = g / slots_per_row, k = 4 * (g % slots_per_row);
                    *reinterpret_cast<float4 *>(Wl + (part * DP + j) * KP + k) = w[u];
                }
            }
        }
    }
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = t * 32 + i;
        bias[t] = c < p.d_out ? p.b1[c] + p.b2[c] : 0.f;
    }
    __syncthreads();

    for (int64_t tile = tile_first; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t r = min(tile * 32 + i, p.n_rows - 1);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            if (c > 0 || tile != tile_first) {  // (chunk 0 of the first tile was fetched before the staging)
                load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(p.P + r * p.d_in, true, 64 * c + 32 * h, p.d_in, a1);
                load_run32f<(FAST ? RUN_FAST : RUN_ANY)>(p.X + r * p.ldx, true, 64 * c + 32 * h, p.d_in, a2);
            }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float pv = a1[s], xv = a2[s];
                a1[s] = pv + xv;  // lin1 operand (layers.py:56)
                a2[s] = pv * xv;  // lin2 operand (layers.py:57)
            }
            const int KP = nch * 64 + 4;
            const float *w1 = Wl + (0 * DP + i) * KP + c * 64 + 32 * h;
            const float *w2 = Wl + (1 * DP + i) * KP + c * 64 + 32 * h;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 w = *reinterpret_cast<const float4 *>(w1 + t * 32 * KP + 4 * q);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[4 * q + 0], w.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[4 * q + 1], w.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[4 * q + 2], w.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[4 * q + 3], w.w, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 w = *reinterpret_cast<const float4 *>(w2 + t * 32 * KP + 4 * q);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 0], w.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 1], w.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 2], w.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 3], w.w, acc[t], 0, 0, 0);
                }
            }
        }
        // epilogue on the C layout: col = 32t + (lane&31), row = (reg&3) + 8*(reg>>2) + 4*h
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            float v[NT];
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float x = acc[t][reg] + bias[t];
                if (p.leaky_norm) x = x > 0.f ? x : x * p.slope;
                if (p.drop_mask) {
                    const int64_t mrow = tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const int mc = t * 32 + i;
                    x *= (mrow < p.n_rows && mc < p.d_out) ? p.drop_mask[mrow * p.d_out + mc] : 0.f;
                }
                v[t] = x;
                ss = fmaf(x, x, ss);  // padded columns hold exact zeros
            }
            if (p.leaky_norm) {
                ss = row32_sum(ss);
                const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||, eps)
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] *= inv;
                const int64_t nrow = tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (p.inv_norm && i == 0 && nrow < p.n_rows) p.inv_norm[nrow] = inv;
            }
            const int64_t row = tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < p.n_rows) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int c = t * 32 + i;
                    if (c < p.d_out) p.Y[row * p.ldy + c] = v[t];
                }
            }
        }
    }
}

template <int NT>
static int launch_dense(const BignnParams &p, int fast, hipStream_t s) {
    const int nch = (p.d_in + 63) / 64;
    const size_t lds = (size_t)2 * 32 * NT * (nch * 64 + 4) * sizeof(float);
    if (lds > 160 * 1024) return fail(RBG_EUNSUPPORTED, "BiGNNConv %d x %d needs %zu bytes of LDS", p.d_in, p.d_out, lds);
    static std::atomic<int> lds_attr_device{-2};  // per template instantiation: set once per device, not per launch
    int cur_dev = -1;
    (void)hipGetDevice(&cur_dev);
    if (lds > 64 * 1024 && lds_attr_device.load() != cur_dev) {
        lds_attr_device = cur_dev;
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_dense_kernel<NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_dense_kernel<NT, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int64_t n_tiles = (p.n_rows + 31) / 32;
    // one 32-row tile per wavefront (the grid-stride loop in the kernel only matters beyond 2^20 tiles)
    // (capping the grid so that a workgroup stages the weights once for several tiles changes nothing: 28.7 us at every cap
    //  from 768 workgroups up, slower below — r02, Gowalla shape; the kernel is bound by its row loads, not by the staging)
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, 1 << 18));
    if (fast)
        hipLaunchKernelGGL((bignn_dense_kernel<NT, true>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((bignn_dense_kernel<NT, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}


// ---- the NGCF configuration (d_in = 64, d_out in {16, 32, 48, 64}): rows by LDS-DMA, weights in registers ------------
//
// The kernel above spends most of its time fetching rows: a lane needs ITS row's contiguous k-run (the MFMA operand
// layout ties lanes to rows), so every load instruction touches 64 cache lines for 16 bytes each, the lines are revisited by
// the next seven instructions, and with 12 waves per CU their 196 KB of in-flight rows do not survive in the 32 KB L1.
// bignn_dense_pipe_kernel instead has a wave fetch a 16-row tile of P and of X as 8 fully coalesced 1 KiB
// `global_load_lds_dwordx4` requests straight into LDS (no VGPR round trip, nothing for the L1 to keep) and read its k-runs
// back with 8 conflict-free ds_read_b128: LDS-DMA writes lane-linear, so the 16-byte chunk c of tile row r is SOURCED into
// slot c ^ r and read from slot c ^ r (same involution on both sides).
// Matrix-core shape: v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand and the data rows as B, i.e. the tile of
// Y^T: lane (n = lane & 15, g = lane >> 4) then holds output columns 16 t + 4 g .. + 3 of data row n — a float4 store per
// column tile, one row norm per lane (in-lane sum + two cross-group shuffles), one rsqrt per tile instead of 16.
// The concatenated weights are 128 registers per lane (W1 / W2 rows 16 t + n, k-run 16 g .. + 15), filled once per
// persistent wave from a per-workgroup LDS copy (itself one DMA pass, biases included): the MFMA loop reads registers
// only, and no load is visible to hipcc, whose own vmcnt waits would otherwise drain the DMA queue it does not count.
// Per-wave clock trace (devtools/microbench/bignn_trace.hip builds this file with RBG_BIGNN_TRACE; the product does not):
// slot k of wave w receives s_memtime at stamp k.
#ifdef RBG_BIGNN_TRACE
__device__ unsigned long long *g_bignn_trace = nullptr;
#define RBG_STAMP(k)                                                                                         \
    do {                                                                                                     \
        if (g_bignn_trace && lane == 0 && (k) < 32) g_bignn_trace[((int64_t)blockIdx.x * 8 + wave) * 32 + (k)] = clock64(); \
    } while (0)
#else
#define RBG_STAMP(k) ((void)0)
#endif

// ONE wave per SIMD runs a rotated loop (a first version with two waves per SIMD and the epilogue after the MFMAs ran
// 17.7 us at the Gowalla shape against 16.8 us: its clock trace showed a wave's own chain per tile — MFMA 4.1 k cycles,
// wait / fetch / DMA issue 1-2 k, epilogue 2.5-3.5 k — and the three-tile waves setting the kernel time with the matrix
// core 40 % busy):
//     iteration k:  MFMA(tile k), first 3/4, with epilogue(tile k-1) in the same basic block (sched_group_barrier pattern)
//                   counted vmcnt wait (tile k+1 was issued a tile ago), ds_read it
//                   MFMA(tile k), last 1/4, with the DMA of tile k+2 spread between the k-steps
// so the in-order instruction stream never waits on anything that is not long done.  Tiles are double-buffered in LDS
// (the DMA for k+2 overwrites the buffer tile k was read from an iteration ago).  Rows past the end are CLAMPED copies of
// the last row at the loads and at the stores (identical inputs give identical outputs, so the duplicate stores are
// benign): no predicated store, no branch in the epilogue.
// Steady state is 2.06 us per tile round against 1.73 us of pure MFMA time (devtools/microbench/mfma_rate.hip:
// 13.5 ns per 16x16x4 instruction); the rest of the 16.8 us at the Gowalla shape is fixed: weights + first tiles ~3.8 us,
// the fifth round that only a third of the waves have (4 428 tiles on 1 024 waves) 1.4 us, launch and drain.
// TAIL = LeakyReLU + L2-normalize fused, INV = the row's 1 / norm is stored too, MASK = dropout mask between the two: compile
// time, so that the epilogue is straight-line code the scheduler can spread between the MFMAs.
template <int NT, bool MASK, bool TAIL, bool INV>
__global__ __launch_bounds__(256) void bignn_dense_pipe_kernel(const BignnParams p) {
    constexpr int WAVES = 4;
    __shared__ __attribute__((aligned(1024))) float tiles[WAVES * 4096];  // per wave: 2 x {P tile [16][64], X tile [16][64]}
    __shared__ __attribute__((aligned(1024))) float wstage[2 * 64 * 64];
    __shared__ __attribute__((aligned(1024))) float bstage[256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int64_t n_tiles = (p.n_rows + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t tile = blockIdx.x + (int64_t)gridDim.x * wave;
    float *buf = tiles + wave * 4096;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)buf);

    auto issue_row = [&](int64_t t, int par, int j) __attribute__((always_inline)) {  // rows 4 j .. 4 j + 3 of tile t
        const int rr = 4 * j + g;
        const int64_t r = min(t * 16 + rr, p.n_rows - 1);
        lds_dma16(p.P + r * 64 + 4 * (n ^ rr), buf_lds + par * 8192 + j * 1024);
        lds_dma16(p.X + r * p.ldx + 4 * (n ^ rr), buf_lds + par * 8192 + 4096 + j * 1024);
    };
    auto issue = [&](int64_t t, int par) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_row(t, par, j);
    };
    auto fetch = [&](int par, float(&pv)[16], float(&xv)[16]) __attribute__((always_inline)) {
        const float *b = buf + par * 2048;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = (4 * g + q) ^ n;
            const float4 a = *reinterpret_cast<const float4 *>(b + n * 64 + slot * 4);
            const float4 c = *reinterpret_cast<const float4 *>(b + 1024 + n * 64 + slot * 4);
            pv[4 * q + 0] = a.x, pv[4 * q + 1] = a.y, pv[4 * q + 2] = a.z, pv[4 * q + 3] = a.w;
            xv[4 * q + 0] = c.x, xv[4 * q + 1] = c.y, xv[4 * q + 2] = c.z, xv[4 * q + 3] = c.w;
        }
    };

    RBG_STAMP(0);
    const bool has_tile = tile < n_tiles;
    {
        const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)wstage);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 8 * wave + u, part = i >> 4, rr = 4 * (i & 15) + g;
            const float *w = part ? p.W2 : p.W1;
            lds_dma16(w + min(rr, p.d_out - 1) * 64 + 4 * (n ^ (rr & 15)), w_lds + i * 1024);
        }
        if (wave == 0) {
            const float *b = (g & 1) ? p.b2 : p.b1;
            lds_dma16(b + min(4 * n, p.d_out - 4), __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)bstage));
        }
    }
    if (has_tile) {
        issue(tile, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RBG_STAMP(1);
    __syncthreads();
    RBG_STAMP(2);
    if (!has_tile) return;
    float w1[NT][16], w2[NT][16], bias[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = 16 * t + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = (4 * g + q) ^ n;
            const float4 a = *reinterpret_cast<const float4 *>(wstage + j * 64 + slot * 4);
            const float4 b = *reinterpret_cast<const float4 *>(wstage + 4096 + j * 64 + slot * 4);
            w1[t][4 * q + 0] = a.x, w1[t][4 * q + 1] = a.y, w1[t][4 * q + 2] = a.z, w1[t][4 * q + 3] = a.w;
            w2[t][4 * q + 0] = b.x, w2[t][4 * q + 1] = b.y, w2[t][4 * q + 2] = b.z, w2[t][4 * q + 3] = b.w;
        }
        const int c = 16 * t + 4 * g;
        const float4 ba = *reinterpret_cast<const float4 *>(bstage + c), bb = *reinterpret_cast<const float4 *>(bstage + 64 + c);
        bias[t][0] = ba.x + bb.x, bias[t][1] = ba.y + bb.y, bias[t][2] = ba.z + bb.z, bias[t][3] = ba.w + bb.w;
    }
    float pv[16], xv[16], pn[16], xn[16];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fetch(0, pv, xv);
    if (tile + stride < n_tiles) issue(tile + stride, 1);
    RBG_STAMP(3);

    f32x4 acc[NT], old[NT];
    int64_t old_tile = tile;
    int par = 0;

    // finishes a tile: lane (n, g) holds columns 16 t + 4 g + r of data row n
    auto epilogue = [&](const f32x4(&a)[NT], int64_t t_id) __attribute__((always_inline)) {
        const int64_t row = min(t_id * 16 + n, p.n_rows - 1);
        float v[NT][4];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (MASK) m = *reinterpret_cast<const float4 *>(p.drop_mask + row * p.d_out + 16 * t + 4 * g);
            const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = a[t][r] + bias[t][r];
                if constexpr (TAIL) x = fmaxf(x, 0.f) + p.slope * fminf(x, 0.f);
                if constexpr (MASK) x *= mm[r];
                v[t][r] = x;
                ss = fmaf(x, x, ss);
            }
        }
        if constexpr (TAIL) {
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||, eps)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t][r] *= inv;
            if constexpr (INV) p.inv_norm[row] = inv;  // the four lane groups of a row hold the same value: no predicate
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<float4 *>(p.Y + row * p.ldy + 16 * t + 4 * g) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
    };
    // the MFMAs of k-steps [s0, s1) of the concatenated K = 128 (steps 0-15: lin1 on P + X, 16-31: lin2 on P * X)
    auto mfma_steps = [&](int s0, int s1) __attribute__((always_inline)) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            const int k = s & 15;
            const float a = s < 16 ? pv[k] + xv[k] : pv[k] * xv[k];  // layers.py:56 / :57
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(s < 16 ? w1[t][k] : w2[t][k], a, acc[t], 0, 0, 0);
        }
    };
    auto body = [&](auto with_old) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_steps(0, 24);
        if constexpr (decltype(with_old)::value) {
            epilogue(old, old_tile);
            // one MFMA, then the epilogue's VALU work in its 32-cycle shadow
#pragma unroll
            for (int i = 0; i < 24 * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
            }
        }
    };

    int stamp = 4;
    (void)stamp;
    // One iteration = the tile in (pv, xv).  Returns false after the wave's last tile.
    auto iteration = [&](auto with_old) __attribute__((always_inline)) -> bool {
        body(with_old);
        const int64_t next = tile + stride;
        const bool more = next < n_tiles;
        RBG_STAMP(stamp);
        if (more) {
            // Tile `next` was issued a whole tile ago.  Requests retire in order, and the only ones younger than that
            // DMA are the epilogue's NT + INV stores, so a counted wait does not sit on their acknowledgements.
            if constexpr (decltype(with_old)::value)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT + (INV ? 1 : 0)) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fetch(par ^ 1, pn, xn);
        }
        RBG_STAMP(stamp + 1);
        // the last 8 k-steps, with the DMA of tile k + 2 (into the buffer tile k was read from an iteration ago) spread
        // between them so that its issue cost sits in MFMA shadows
        const bool more2 = more && next + stride < n_tiles;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (more2) issue_row(next + stride, par, j);
            mfma_steps(24 + 2 * j, 26 + 2 * j);
        }
        RBG_STAMP(stamp + 2);
#pragma unroll
        for (int t = 0; t < NT; ++t) old[t] = acc[t];
        old_tile = tile;
        stamp += 4;
        if (!more) return false;
#pragma unroll
        for (int s = 0; s < 16; ++s) pv[s] = pn[s], xv[s] = xn[s];
        tile = next;
        par ^= 1;
        return true;
    };
    if (iteration(std::false_type{}))
        while (iteration(std::true_type{})) {
        }
    epilogue(old, old_tile);
    RBG_STAMP(stamp);
}

static int device_cu_count() {  // of the CURRENT device (cached per device: a process may drive several)
    static std::atomic<int> cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const bool slot = dev >= 0 && dev < 16;
    int v = slot ? cached[dev].load() : 0;
    if (v > 0) return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    if (slot) cached[dev] = v;
    return v;
}

template <int NT>
static int launch_dense_pipe(const BignnParams &p, hipStream_t s) {
    const int64_t n_tiles = (p.n_rows + 15) / 16;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, device_cu_count()));
    const dim3 gr((unsigned)grid), bl(256);
    if (p.drop_mask && p.leaky_norm && p.inv_norm)
        hipLaunchKernelGGL((bignn_dense_pipe_kernel<NT, true, true, true>), gr, bl, 0, s, p);
    else if (p.drop_mask)
        return fail(RBG_EINVAL, "a dropout mask needs the fused tail and the inv_norm buffer");
    else if (p.leaky_norm && p.inv_norm)
        hipLaunchKernelGGL((bignn_dense_pipe_kernel<NT, false, true, true>), gr, bl, 0, s, p);
    else if (p.leaky_norm)
        hipLaunchKernelGGL((bignn_dense_pipe_kernel<NT, false, true, false>), gr, bl, 0, s, p);
    else
        hipLaunchKernelGGL((bignn_dense_pipe_kernel<NT, false, false, false>), gr, bl, 0, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

// g != NULL: P_save = ÂX first (n_rows from the graph); g == NULL: P_save already holds the product for n_rows rows.
static int bignn_conv_impl(const rbg_graph *g, int64_t n_rows, const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2,
                           const float *b2, float *Y, int64_t ldy, float *P_save, float *inv_norm, const float *drop_mask,
                           int d_in, int d_out, uint32_t flags, float slope, void *stream) {
    clear_error();
    if (g) {
        if (g->device < 0) return fail(RBG_ENODEV, "operator called on a host graph (create it with device >= 0)");
        if (g->n_rows != g->n_cols) return fail(RBG_ESHAPE, "graph is not square");
        n_rows = g->n_rows;
    }
    if (n_rows < 0) return fail(RBG_ESHAPE, "n_rows = %lld", (long long)n_rows);
    if (d_in <= 0 || d_out <= 0 || ldx < d_in || ldy < d_out)
        return fail(RBG_ESHAPE, "d_in = %d, d_out = %d, ldx = %lld, ldy = %lld", d_in, d_out, (long long)ldx, (long long)ldy);
    if (d_out > 256) return fail(RBG_EUNSUPPORTED, "d_out = %d > 256", d_out);
    if (n_rows == 0) return RBG_OK;
    if (!X || !W1 || !b1 || !W2 || !b2 || !Y) return fail(RBG_EINVAL, "NULL pointer");
    if (!P_save) return fail(RBG_EINVAL, "P_save is NULL: the caller provides the [N, d_in] buffer that receives / holds ÂX");
    int rc;
    hipStream_t s = (hipStream_t)stream;
    if (g) {
        if ((rc = set_device_for(g->device))) return rc;
        if ((rc = spmm_strided(g, X, ldx, P_save, d_in, d_in, 0, s))) return rc;
    }
    BignnParams p{};
    p.P = P_save;
    p.X = X;
    p.ldx = ldx;
    p.W1 = W1;
    p.b1 = b1;
    p.W2 = W2;
    p.b2 = b2;
    p.Y = Y;
    p.ldy = ldy;
    p.n_rows = n_rows;
    p.d_in = d_in;
    p.d_out = d_out;
    p.leaky_norm = (flags & RBG_BIGNN_LEAKY_NORM) ? 1 : 0;
    p.slope = slope;
    p.inv_norm = p.leaky_norm ? inv_norm : nullptr;
    p.drop_mask = p.leaky_norm ? drop_mask : nullptr;
    const int fast = (d_in % 64 == 0) && (ldx % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(P_save) | reinterpret_cast<uintptr_t>(W1) |
                       reinterpret_cast<uintptr_t>(W2)) & 15u) == 0;
    // the NGCF configuration takes the LDS-DMA kernel ("bignn_dma" option, default on)
    const bool vec_out = d_out % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0 &&
                         (!p.drop_mask || (reinterpret_cast<uintptr_t>(p.drop_mask) & 15u) == 0) &&
                         ((reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2)) & 15u) == 0;
    if (fast && d_in == 64 && d_out <= 64 && d_out % 16 == 0 && vec_out && opt_bignn_dma()) {
        switch (d_out / 16) {
            case 1: return launch_dense_pipe<1>(p, s);
            case 2: return launch_dense_pipe<2>(p, s);
            case 3: return launch_dense_pipe<3>(p, s);
            default: return launch_dense_pipe<4>(p, s);
        }
    }
    if (d_out <= 32) return launch_dense<1>(p, fast, s);
    if (d_out <= 64) return launch_dense<2>(p, fast, s);
    if (d_out <= 128) return launch_dense<4>(p, fast, s);
    return launch_dense<8>(p, fast, s);
}

extern "C" int rbg_bignn_conv_f32(const rbg_graph *g, const float *X, int64_t ldx, const float *W1, const float *b1,
                                  const float *W2, const float *b2, float *Y, int64_t ldy, float *P_save, int d_in,
                                  int d_out, uint32_t flags, float slope, void *stream) {
    if (!g) return (clear_error(), fail(RBG_EINVAL, "graph is NULL"));
    return bignn_conv_impl(g, 0, X, ldx, W1, b1, W2, b2, Y, ldy, P_save, nullptr, nullptr, d_in, d_out, flags, slope, stream);
}

extern "C" int rbg_bignn_layer_f32(const rbg_graph *g, const float *X, int64_t ldx, const float *W1, const float *b1,
                                   const float *W2, const float *b2, float *Y, int64_t ldy, float *P_save, float *inv_norm,
                                   const float *drop_mask, int d_in, int d_out, float slope, void *stream) {
    if (!inv_norm) return fail(RBG_EINVAL, "inv_norm is NULL: the caller provides the [N] buffer the backward needs");
    if (!g) return (clear_error(), fail(RBG_EINVAL, "graph is NULL"));
    return bignn_conv_impl(g, 0, X, ldx, W1, b1, W2, b2, Y, ldy, P_save, inv_norm, drop_mask, d_in, d_out, RBG_BIGNN_LEAKY_NORM, slope,
                           stream);
}

// The dense half of the layer alone — lin1(P + X) + lin2(P ⊙ X) [+ LeakyReLU + L2-normalize] from a product P = ÂX the
// caller already holds (layers.py:55-57 after :55's propagate): the node-range sharded path forms P with
// rbg_spmm_sharded_f32 (halo exchange) and finishes the layer on its own rows here.
extern "C" int rbg_bignn_dense_f32(const float *P, const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2,
                                   const float *b2, float *Y, int64_t ldy, int64_t n_rows, int d_in, int d_out, uint32_t flags,
                                   float slope, void *stream) {
    return bignn_conv_impl(nullptr, n_rows, X, ldx, W1, b1, W2, b2, Y, ldy, const_cast<float *>(P), nullptr, nullptr, d_in, d_out, flags,
                           slope, stream);
}
