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

#include "internal.h"
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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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


// ---- the NGCF configuration (d_in = 64, d_out <= 64): rows arrive by LDS-DMA, the weights live in registers ----------
//
// The kernel above spends most of its time fetching rows: a lane needs ITS row's contiguous k-run (the MFMA operand
// layout ties lanes to rows), so every load instruction touches 64 cache lines for 16 bytes each, the lines are revisited by
// the next seven instructions, and with 12 waves per CU their 196 KB of in-flight rows do not survive in the 32 KB L1.
// Here a wave fetches a 16-row tile of P and of X as 8 fully coalesced 1 KiB `global_load_lds_dwordx4` requests straight
// into its private 8 KB of LDS (no VGPR round trip, nothing for the L1 to keep), and then reads its k-runs back with
// 8 conflict-free ds_read_b128: LDS-DMA writes lane-linear, so the 16-byte chunk c of tile row r is SOURCED into slot
// c ^ r and read from slot c ^ r (same involution on both sides).
// Matrix-core shape: v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand and the data rows as B, i.e. the tile of
// Y^T: lane (n = lane & 15, g = lane >> 4) then holds output columns 16 t + 4 g .. + 3 of data row n — a float4 store per
// column tile, one row norm per lane (in-lane sum + two cross-group shuffles), one rsqrt per tile instead of 16.
// The concatenated weights are 128 registers per lane (W1 / W2 rows 16 t + n, k-run 16 g .. + 15), filled once per
// persistent wave from a per-workgroup LDS copy (itself one DMA pass): the MFMA loop reads registers only.
// Pipeline per wave (8 waves per CU, 2 per SIMD so one wave's epilogue runs under the other's MFMAs):
//   MFMA(tile i) -> vmcnt(0) [tile i+1 landed long ago] -> ds_read tile i+1 -> lgkmcnt(0) -> DMA(tile i+2) -> epilogue(i)
// so the stores of tile i are never waited on and a DMA has a whole tile of MFMAs to land.  (With a dropout mask the DMA
// goes out after the epilogue: the mask loads are ordinary loads, and the wait hipcc places for them would drain it.)
__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_dst) {  // lds_dst: wave-uniform LDS byte address
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

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

__device__ __forceinline__ void pin8(float *a) {  // the compiler must have these eight registers' loads complete here
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

// NT = 16-column output tiles held by a wave (d_out <= 16 NT); MASK = a dropout mask is applied (its own instantiation:
// the wait hipcc places for the mask loads must not exist in the kernel that overlaps the DMA with the epilogue)
template <int NT, bool MASK>
__global__ __launch_bounds__(512) void bignn_dense_dma_kernel(const BignnParams p) {
    constexpr int WAVES = 8;
    __shared__ __attribute__((aligned(1024))) float tiles[WAVES * 2048];  // per wave: P tile [16][64], X tile [16][64]
    __shared__ __attribute__((aligned(1024))) float wstage[2 * 64 * 64];   // W1, W2 [64][64], chunk-swizzled like the tiles
    __shared__ __attribute__((aligned(1024))) float bstage[256];           // b1 [64], b2 [64] (+ the rest of the DMA's 1 KiB)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int64_t n_tiles = (p.n_rows + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t tile = blockIdx.x + (int64_t)gridDim.x * wave;  // consecutive tiles go to different CUs
    float *buf = tiles + wave * 2048;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)buf);

    auto issue = [&](int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = 4 * j + g;  // this lane's destination is slot n of tile row rr; it sources chunk n ^ rr
            const int64_t r = min(t * 16 + rr, p.n_rows - 1);
            lds_dma16(p.P + r * 64 + 4 * (n ^ rr), buf_lds + j * 1024);
            lds_dma16(p.X + r * p.ldx + 4 * (n ^ rr), buf_lds + 4096 + j * 1024);
        }
    };
    auto fetch = [&](float(&pv)[16], float(&xv)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = (4 * g + q) ^ n;
            const float4 a = *reinterpret_cast<const float4 *>(buf + n * 64 + slot * 4);
            const float4 b = *reinterpret_cast<const float4 *>(buf + 1024 + n * 64 + slot * 4);
            pv[4 * q + 0] = a.x, pv[4 * q + 1] = a.y, pv[4 * q + 2] = a.z, pv[4 * q + 3] = a.w;
            xv[4 * q + 0] = b.x, xv[4 * q + 1] = b.y, xv[4 * q + 2] = b.z, xv[4 * q + 3] = b.w;
        }
    };

    RBG_STAMP(0);
    // Two waves share a SIMD (w and w + 4).  Started together they would run their MFMA phases together and their epilogues
    // together, leaving the matrix core idle through every epilogue; the older four get issue priority, so their MFMAs run
    // at full rate first and from then on one wave's epilogue sits under the other's MFMAs.
    if (wave < 4) __builtin_amdgcn_s_setprio(2);
    const bool has_tile = tile < n_tiles;  // (wave-uniform) a wave without work still stages its share of the weights
    // Weights and biases: staged ONCE per workgroup by DMA (each wave brings 8 rows of W1 / W2, coalesced; wave 0 also the
    // two bias vectors), then every lane reads its A operands — rows 16 t + n of W1 / W2, k-run 16 g .. 16 g + 15 — into
    // registers.  (Per-lane global loads of the same runs cost 64 cache lines per instruction and 32 KB per WAVE out of
    // 256 L2 lines the whole grid shares: ~15 us.)  They go out BEFORE the first tile so that vmcnt(8) — requests retire
    // in order — means "weights landed" while the tile is still in flight, and no load is visible to hipcc, whose waits
    // would otherwise drain the DMA queue.
    {
        const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)wstage);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * wave + u, part = i >> 4, rr = 4 * (i & 15) + g;  // destination: slot n of weight row rr
            const float *w = part ? p.W2 : p.W1;
            lds_dma16(w + min(rr, p.d_out - 1) * 64 + 4 * (n ^ (rr & 15)), w_lds + i * 1024);
        }
        if (wave == 0) {  // lanes 0-15: b1[4 n ..], lanes 16-31: b2[4 n ..] (clamped inside the vectors; padded columns are zeroed below)
            const float *b = (g & 1) ? p.b2 : p.b1;
            lds_dma16(b + min(4 * n, p.d_out - 4), __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)bstage));
        }
    }
    if (has_tile) {
        issue(tile);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RBG_STAMP(1);
    __syncthreads();  // every wave's share of the weights is in LDS (the only barrier of the kernel)
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
        const int c = 16 * t + 4 * g;  // this lane's four output columns of tile t (d_out is a multiple of 4)
        const int cc = min(c, p.d_out - 4);
        const float4 ba = *reinterpret_cast<const float4 *>(bstage + cc), bb = *reinterpret_cast<const float4 *>(bstage + 64 + cc);
        const bool in = c < p.d_out;
        bias[t][0] = in ? ba.x + bb.x : 0.f, bias[t][1] = in ? ba.y + bb.y : 0.f, bias[t][2] = in ? ba.z + bb.z : 0.f,
        bias[t][3] = in ? ba.w + bb.w : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // weight rows past d_out (the DMA clamped them to the last row) are zero
        if (16 * t + n >= p.d_out) {
#pragma unroll
            for (int s = 0; s < 16; ++s) w1[t][s] = 0.f, w2[t][s] = 0.f;
        }
    }
    float pv[16], xv[16];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fetch(pv, xv);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next DMA overwrites the tile
    if (tile + stride < n_tiles) issue(tile + stride);
    RBG_STAMP(3);
    int stamp = 4;
    (void)stamp;

    for (;;) {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a1 = pv[s] + xv[s];  // lin1 operand (layers.py:56)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t][s], a1, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a2 = pv[s] * xv[s];  // lin2 operand (layers.py:57)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[t][s], a2, acc[t], 0, 0, 0);
        }
        const int64_t next = tile + stride;
        const bool more = next < n_tiles;
        RBG_STAMP(stamp);  // (the last MFMA is issued, not necessarily complete)
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile `next` (issued a whole tile ago) is in LDS
            RBG_STAMP(stamp + 1);
            fetch(pv, xv);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // with a dropout mask the epilogue has compiler-visible loads, whose wait must not see a DMA behind them
            if (!MASK && next + stride < n_tiles) issue(next + stride);
        }
        RBG_STAMP(stamp + 2);
        // epilogue: lane (n, g) holds columns 16 t + 4 g + r of data row n
        const int64_t row = tile * 16 + n;
        const bool live = row < p.n_rows;
        float v[NT][4];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = 16 * t + 4 * g;
            float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (MASK)
                if (live && c < p.d_out) m = *reinterpret_cast<const float4 *>(p.drop_mask + row * p.d_out + c);
            const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[t][r] + bias[t][r];
                if (p.leaky_norm) x = x > 0.f ? x : x * p.slope;
                if constexpr (MASK) x *= mm[r];
                v[t][r] = x;
                ss = fmaf(x, x, ss);  // padded columns hold exact zeros
            }
        }
        if (p.leaky_norm) {
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||, eps)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t][r] *= inv;
            if (p.inv_norm && g == 0 && live) p.inv_norm[row] = inv;
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = 16 * t + 4 * g;
                if (c < p.d_out) *reinterpret_cast<float4 *>(p.Y + row * p.ldy + c) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
            }
        }
        RBG_STAMP(stamp + 3);
        stamp += 4;
        if (!more) break;
        if (MASK && next + stride < n_tiles) issue(next + stride);
        tile = next;
    }
}

static int device_cu_count() {
    static std::atomic<int> cached{0};
    int v = cached.load();
    if (v > 0) return v;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
        v = 256;
    cached = v;
    return v;
}

template <int NT>
static int launch_dense_dma(const BignnParams &p, hipStream_t s) {
    const int64_t n_tiles = (p.n_rows + 15) / 16;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((n_tiles + 7) / 8, device_cu_count()));
    if (p.drop_mask)
        hipLaunchKernelGGL((bignn_dense_dma_kernel<NT, true>), dim3((unsigned)grid), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL((bignn_dense_dma_kernel<NT, false>), dim3((unsigned)grid), dim3(512), 0, s, p);
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
    if (fast && d_in == 64 && d_out <= 64 && vec_out && opt_bignn_dma())
        return d_out <= 32 ? launch_dense_dma<2>(p, s) : launch_dense_dma<4>(p, s);
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
