// bignn_grad.hip — the weight gradients of NGCF's bi-interaction layer.
//
// Autograd of BiGNNConv.forward (recbole_gnn/model/layers.py:54-58) with G = dL/dY, P = ÂX:
//     dW1 = G^T (P + X)      dW2 = G^T (P ⊙ X)      db1 = db2 = sum_rows G
// are [d_out, d_in] results reduced over ALL N rows (N = 70 841 at the Gowalla shape, d = 64): "tall-skinny A^T B".
// torch hands each to rocBLAS as a 64 x 64 x 70 841 GEMM, which runs 203 us — two per layer, 41 % of the whole NGCF
// training step (r01 kernel stats).  Here G, P and X are read ONCE for all three results: a workgroup streams 32-row
// tiles (coalesced float4 loads, double-buffered in LDS as G | P+X | P⊙X), every wave owns output tiles and feeds
// v_mfma_f32_32x32x2_f32 with both operands read as LDS *columns* (k = row index), and the per-workgroup partial
// results are summed in a fixed order by a second small kernel (split-K without atomics: bit-reproducible).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "internal.h"
#include "lds_dma.h"
#include "mfma_common.h"

namespace rbg {


struct WgradParams {
    const float *G;
    int64_t ldg;
    const float *P;  // [N, d_in] contiguous
    const float *X;
    int64_t ldx;
    int64_t n_rows;
    int d_in, d_out;
    float *part;  // [grid][2 * d_out * d_in + d_out]
};

// TO / TI = 32-column tiles of d_out / d_in (1, 2 or 4).  FAST: d_out, d_in multiples of 4 and 16-byte aligned rows.
template <int TO, int TI, bool FAST>
__global__ __launch_bounds__(256) void bignn_wgrad_kernel(const WgradParams p) {
    constexpr int DO = 32 * TO, DI = 32 * TI;
    constexpr int KG = (32 * DO / 4 + 255) / 256, KS = (32 * DI / 4 + 255) / 256;  // float4 slots per thread and tile
    constexpr int PAIRS = 2 * TO * TI, MAXQ = (PAIRS + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // layout per buffer: g[32][DO] | s[32][DI] | h[32][DI]
    constexpr int BUF = 32 * (DO + 2 * DI);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (p.n_rows + 31) / 32;

    float4 sg[KG], sp[KS], sx[KS];
    float4 gsum[KG];
#pragma unroll
    for (int k = 0; k < KG; ++k) gsum[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load4 = [&](const float *row, const int c4, const int width, const bool row_ok) __attribute__((always_inline)) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok) {
            if (FAST) {
                if (c4 < width) v = *reinterpret_cast<const float4 *>(row + c4);
            } else {
                if (c4 + 0 < width) v.x = row[c4 + 0];
                if (c4 + 1 < width) v.y = row[c4 + 1];
                if (c4 + 2 < width) v.z = row[c4 + 2];
                if (c4 + 3 < width) v.w = row[c4 + 3];
            }
        }
        return v;
    };
    auto fetch = [&](const int64_t tile) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int f = tid + 256 * k, row = f / (DO / 4), c4 = (f % (DO / 4)) * 4;
            const int64_t r = tile * 32 + row;
            sg[k] = load4(p.G + r * p.ldg, c4, p.d_out, f < 32 * DO / 4 && r < p.n_rows);  // rows past the end: zeros
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int f = tid + 256 * k, row = f / (DI / 4), c4 = (f % (DI / 4)) * 4;
            const int64_t r = tile * 32 + row;
            const bool ok = f < 32 * DI / 4 && r < p.n_rows;
            sp[k] = load4(p.P + r * p.d_in, c4, p.d_in, ok);
            sx[k] = load4(p.X + r * p.ldx, c4, p.d_in, ok);
        }
    };
    auto publish = [&](const int buf) __attribute__((always_inline)) {
        float *g = lds + buf * BUF, *s = g + 32 * DO, *hh = s + 32 * DI;
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int f = tid + 256 * k;
            if (f < 32 * DO / 4) {
                *reinterpret_cast<float4 *>(g + 4 * f) = sg[k];
                gsum[k].x += sg[k].x, gsum[k].y += sg[k].y, gsum[k].z += sg[k].z, gsum[k].w += sg[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int f = tid + 256 * k;
            if (f < 32 * DI / 4) {
                const float4 a = sp[k], b = sx[k];
                *reinterpret_cast<float4 *>(s + 4 * f) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);   // layers.py:56
                *reinterpret_cast<float4 *>(hh + 4 * f) = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);  // layers.py:57
            }
        }
    };

    f32x16 acc[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) acc[q] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    int64_t tile = blockIdx.x;
    if (tile < n_tiles) {
        fetch(tile);
        publish(0);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        const int64_t next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);  // in flight while this tile feeds the matrix core
        const float *g = lds + buf * BUF, *s = g + 32 * DO, *hh = s + 32 * DI;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int pair = wave + 4 * q;  // wave-uniform
            if (pair < PAIRS) {
                const int part = pair / (TO * TI), rem = pair % (TO * TI), jt = rem / TI, ct = rem % TI;
                const float *a = g + jt * 32 + i, *b = (part ? hh : s) + ct * 32 + i;
#pragma unroll
                for (int st = 0; st < 16; ++st)  // k = tile row 2 st + h: dW[j][c] += G[row][j] * operand[row][c]
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(2 * st + h) * DO], b[(2 * st + h) * DI], acc[q], 0, 0, 0);
            }
        }
        if (next < n_tiles) publish(buf ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // partial results of this workgroup: acc[q][r] = dW_part[j = jt*32 + rowmap(r, h)][c = ct*32 + i]
    float *out = p.part + (int64_t)blockIdx.x * (2 * p.d_out * p.d_in + p.d_out);
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int pair = wave + 4 * q;
        if (pair < PAIRS) {
            const int part = pair / (TO * TI), rem = pair % (TO * TI), jt = rem / TI, ct = rem % TI;
            const int c = ct * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j < p.d_out && c < p.d_in) out[(part * p.d_out + j) * p.d_in + c] = acc[q][r];
            }
        }
    }
    // column sums of G: every thread's slots keep their column group across tiles; combine through LDS in a fixed order
    __syncthreads();
    float4 *red = reinterpret_cast<float4 *>(lds);
#pragma unroll
    for (int k = 0; k < KG; ++k) red[k * 256 + tid] = gsum[k];
    __syncthreads();
    if (tid < p.d_out) {
        const int grp = tid >> 2, comp = tid & 3;
        float ssum = 0.f;
        for (int k = 0; k < KG; ++k)
            for (int t = grp; t < 256; t += DO / 4) {  // slots f = t + 256 k with f % (DO/4) == grp  (DO/4 divides 256)
                if (t + 256 * k < 32 * DO / 4) ssum += reinterpret_cast<const float *>(&red[k * 256 + t])[comp];
            }
        out[2 * p.d_out * p.d_in + tid] = ssum;
    }
}

// out[x] = sum_g part[g * len + x].  A workgroup = 32 outputs x 8 slices of the partials (coalesced 128-byte reads,
// 8 independent chains per output); slice sums are combined in a fixed order through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int n_parts, int64_t len, int d_out,
                                                           int d_in, float *__restrict__ dW1, float *__restrict__ dW2,
                                                           float *__restrict__ db) {
    __shared__ float red[8][32];
    const int xo = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t x = (int64_t)blockIdx.x * 32 + xo;
    float a0 = 0.f, a1 = 0.f;
    if (x < len) {
        int g = slice;
        for (; g + 8 < n_parts; g += 16) {  // two loads in flight per thread
            a0 += part[(int64_t)g * len + x];
            a1 += part[(int64_t)(g + 8) * len + x];
        }
        if (g < n_parts) a0 += part[(int64_t)g * len + x];
    }
    red[slice][xo] = a0 + a1;
    __syncthreads();
    if (slice == 0 && x < len) {
        float a = red[0][xo];
#pragma unroll
        for (int q = 1; q < 8; ++q) a += red[q][xo];
        const int64_t w = (int64_t)d_out * d_in;
        if (x < w) dW1[x] = a;
        else if (x < 2 * w) dW2[x - w] = a;
        else if (db) db[x - 2 * w] = a;
    }
}

static int64_t wgrad_grid(int64_t n_rows) { return std::max<int64_t>(1, std::min<int64_t>((n_rows + 31) / 32, 256)); }

template <int TO, int TI>
static int launch_wgrad(const WgradParams &p, bool fast, int64_t grid, hipStream_t s) {
    const size_t lds = std::max<size_t>((size_t)2 * 32 * (32 * TO + 2 * 32 * TI) * sizeof(float),
                                        (size_t)((32 * 32 * TO / 4 + 255) / 256) * 256 * sizeof(float4));
    static std::atomic<int> lds_attr_device{-2};  // per template instantiation: set once per device, not per launch
    int cur_dev = -1;
    (void)hipGetDevice(&cur_dev);
    if (lds > 64 * 1024 && lds_attr_device.load() != cur_dev) {
        lds_attr_device = cur_dev;
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_wgrad_kernel<TO, TI, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_wgrad_kernel<TO, TI, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (fast)
        hipLaunchKernelGGL((bignn_wgrad_kernel<TO, TI, true>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((bignn_wgrad_kernel<TO, TI, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

template <int TO>
static int launch_wgrad_o(const WgradParams &p, bool fast, int64_t grid, hipStream_t s) {
    if (p.d_in <= 32) return launch_wgrad<TO, 1>(p, fast, grid, s);
    if (p.d_in <= 64) return launch_wgrad<TO, 2>(p, fast, grid, s);
    return launch_wgrad<TO, 4>(p, fast, grid, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Input-side gradients of one NGCF layer (autograd of layers.py:54-58 followed by ngcf.py:96,98), per 32-row tile:
//   G  = dL/dz:  tail backward of y = normalize(LeakyReLU(z)) from the saved output y and 1/||a|| (or G = dL/dy)
//   gt = G W1,  gi = G W2                               (MFMA, k = d_out)
//   GP = gt + gi ⊙ X   (to be propagated: dX += Â^T GP),   GX = gt + gi ⊙ P   (the direct part of dX)
// G is also written out: it is the operand of the weight-gradient kernel above.
// ---------------------------------------------------------------------------------------------------------------------
struct DgradParams {
    const float *GY;
    int64_t ldgy;
    const float *Y;  // saved layer output (tail only)
    int64_t ldy;
    const float *inv_norm;  // [N] or NULL (no tail)
    const float *drop_mask;  // [N, d_out] or NULL: the forward's scaled dropout mask (tail only)
    const float *X;
    int64_t ldx;
    const float *P;          // [N, d_in]
    const float *Wt1, *Wt2;  // TRANSPOSED weights [d_in, d_out] (wt_transpose_kernel)
    float *G;                // [N, d_out] out
    float *GP, *GX;          // [N, d_in] out
    int64_t n_rows;
    int d_in, d_out;
    float slope;
};

// Wt[c][j] = W[j][c]
__global__ void wt_transpose_kernel(const float *__restrict__ W1, const float *__restrict__ W2, int d_out, int d_in,
                                    float *__restrict__ Wt1, float *__restrict__ Wt2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= d_out * d_in) return;
    const int c = x / d_out, j = x % d_out;
    Wt1[x] = W1[j * d_in + c];
    Wt2[x] = W2[j * d_in + c];
}

// TI = 32-column tiles of d_in held by a wave; NCO = 64-wide k chunks of d_out.  FAST: d_out a multiple of 64, 16-byte
// aligned rows.  LDS: Wl[2][32 TI][KP], KP = 64 NCO + 4 (rows = input column c, k = output feature j).
template <int TI, int NCO, bool FAST>
__global__ __launch_bounds__(256, (TI <= 2 && NCO == 1 ? 2 : 1)) void bignn_dgrad_kernel(const DgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    constexpr int DP = 32 * TI, KP = 64 * NCO + 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (p.n_rows + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const bool live = tile < n_tiles;
    const int64_t row_i = live ? min(tile * 32 + i, p.n_rows - 1) : 0;  // clamped: rows past the end are masked at the stores

    // this lane's runs of dL/dy and y go out first (their latency hides under the weight staging)
    float gy[NCO][32], yv[NCO][32];
    auto load_run = [&](const float *row, const int k0, float (&r)[32]) __attribute__((always_inline)) {
        if (FAST) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(row + k0 + 4 * q);
                r[4 * q + 0] = v.x, r[4 * q + 1] = v.y, r[4 * q + 2] = v.z, r[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) r[s] = (k0 + s < p.d_out) ? row[k0 + s] : 0.f;
        }
    };
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        load_run(p.GY + row_i * p.ldgy, 64 * c + 32 * h, gy[c]);
        if (p.inv_norm) load_run(p.Y + row_i * p.ldy, 64 * c + 32 * h, yv[c]);
    }
    // stage the transposed weights, coalesced, a batch of loads in flight at a time (zero-padded)
    {
        const int slots_per_row = NCO * 16, per_part = DP * slots_per_row, total = 2 * per_part;
        constexpr int BATCH = 8;
        for (int f0 = threadIdx.x; f0 < total; f0 += 256 * BATCH) {
            float4 w[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + 256 * u;
                w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < total) {
                    const int part = f >= per_part, g = f - part * per_part;
                    const int c = g / slots_per_row, k = 4 * (g % slots_per_row);
                    if (c < p.d_in) {
                        const float *src = (part ? p.Wt2 : p.Wt1) + (int64_t)c * p.d_out + k;
                        if (FAST) {
                            w[u] = *reinterpret_cast<const float4 *>(src);
                        } else {
                            if (k + 0 < p.d_out) w[u].x = src[0];
                            if (k + 1 < p.d_out) w[u].y = src[1];
                            if (k + 2 < p.d_out) w[u].z = src[2];
                            if (k + 3 < p.d_out) w[u].w = src[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int f = f0 + 256 * u;
                if (f < total) {
                    const int part = f >= per_part, g = f - part * per_part;
                    *reinterpret_cast<float4 *>(Wl + (part * DP + g / slots_per_row) * KP + 4 * (g % slots_per_row)) = w[u];
                }
            }
        }
    }
    __syncthreads();
    if (!live) return;

    // G = dL/dz in the fragment layout (lane = row, registers = a 32-wide k run)
    if (p.inv_norm) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int s = 0; s < 32; ++s) dot = fmaf(gy[c][s], yv[c][s], dot);
        dot += __shfl_xor(dot, 32);  // the other half of the row
        const float inv = p.inv_norm[row_i];
        const bool clamped = inv >= 1e12f;  // ||a|| < eps: normalize is a / eps, a plain scaling
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            float mk[32];
            if (p.drop_mask) {
                load_run(p.drop_mask + row_i * (int64_t)p.d_out, 64 * c + 32 * h, mk);
            }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const float y = yv[c][s];
                float da = (clamped ? gy[c][s] : gy[c][s] - y * dot) * inv;
                if (p.drop_mask) da *= mk[s];            // dropout sits between LeakyReLU and normalize (ngcf.py:96-98)
                gy[c][s] = y > 0.f ? da : da * p.slope;  // sign(z) = sign(y) where kept; LeakyReLU'(0) = slope as in torch
            }
        }
    }
    if (tile * 32 + i < p.n_rows) {
        float *grow = p.G + (tile * 32 + i) * (int64_t)p.d_out;
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            const int k0 = 64 * c + 32 * h;
            if (FAST) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *reinterpret_cast<float4 *>(grow + k0 + 4 * q) =
                        make_float4(gy[c][4 * q + 0], gy[c][4 * q + 1], gy[c][4 * q + 2], gy[c][4 * q + 3]);
            } else {
#pragma unroll
                for (int s = 0; s < 32; ++s)
                    if (k0 + s < p.d_out) grow[k0 + s] = gy[c][s];
            }
        }
    }
    f32x16 at[TI], ai[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        at[t] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        ai[t] = at[t];
    }
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
        const float *w1 = Wl + (0 * DP + i) * KP + c * 64 + 32 * h;
        const float *w2 = Wl + (1 * DP + i) * KP + c * 64 + 32 * h;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const float4 a = *reinterpret_cast<const float4 *>(w1 + t * 32 * KP + 4 * q);
                const float4 b = *reinterpret_cast<const float4 *>(w2 + t * 32 * KP + 4 * q);
                at[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 0], a.x, at[t], 0, 0, 0);
                ai[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 0], b.x, ai[t], 0, 0, 0);
                at[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 1], a.y, at[t], 0, 0, 0);
                ai[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 1], b.y, ai[t], 0, 0, 0);
                at[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 2], a.z, at[t], 0, 0, 0);
                ai[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 2], b.z, ai[t], 0, 0, 0);
                at[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 3], a.w, at[t], 0, 0, 0);
                ai[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gy[c][4 * q + 3], b.w, ai[t], 0, 0, 0);
            }
        }
    }
    // C layout: col = 32 t + (lane & 31) = input column, row = (reg & 3) + 8 (reg >> 2) + 4 h
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t row = tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (row < p.n_rows) {
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const int c = t * 32 + i;
                if (c < p.d_in) {
                    const float gt = at[t][reg], gi = ai[t][reg];
                    const float x = p.X[row * p.ldx + c], pv = p.P[row * (int64_t)p.d_in + c];
                    p.GP[row * (int64_t)p.d_in + c] = fmaf(gi, x, gt);   // d/dP
                    p.GX[row * (int64_t)p.d_in + c] = fmaf(gi, pv, gt);  // direct d/dX
                }
            }
        }
    }
}

template <int TI, int NCO>
static int launch_dgrad(const DgradParams &p, bool fast, hipStream_t s) {
    const size_t lds = (size_t)2 * 32 * TI * (64 * NCO + 4) * sizeof(float);
    static std::atomic<int> lds_attr_device{-2};  // per template instantiation: set once per device, not per launch
    int cur_dev = -1;
    (void)hipGetDevice(&cur_dev);
    if (lds > 64 * 1024 && lds_attr_device.load() != cur_dev) {
        lds_attr_device = cur_dev;
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_dgrad_kernel<TI, NCO, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        RBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bignn_dgrad_kernel<TI, NCO, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int64_t grid = std::max<int64_t>(1, ((p.n_rows + 31) / 32 + 3) / 4);
    if (grid > INT32_MAX) return fail(RBG_EUNSUPPORTED, "grid too large");
    if (fast)
        hipLaunchKernelGGL((bignn_dgrad_kernel<TI, NCO, true>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((bignn_dgrad_kernel<TI, NCO, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// ---- the NGCF configuration (d_in = d_out = 64): the whole layer backward but the propagation in ONE kernel ----------
//
// The same recipe as bignn_dense_pipe_kernel (bignn.hip, lds_dma.h).  Per row the input-gradient part reads GY, Y, the
// dropout mask, X, P (5 x 256 B) and writes GP, GX against 2 x 64 x 64 MACs — bound by memory, with the matrix core mostly
// idle (as a kernel of its own it ran 26.5 us at the Gowalla shape, 50.7 us before LDS-DMA) — and the weight gradients
//     dW1 = G^T (P + X),  dW2 = G^T (P * X),  db = column sums of G          (layers.py:56-57 backwards)
// need exactly the tiles that kernel holds (a separate kernel re-read G, P, X: 54 MB, 21.5 us).  So one wave per SIMD, per
// 16-row tile:
//     counted vmcnt wait -> ds_read everything the tile needs into registers (k-runs of GY / Y / mask, the output-layout
//     columns of X and P, and COLUMNS of X and P — rows 4 s + g, column 16 t + n — for the weight gradients)
//     -> tail backward in the k-run layout (lane = row; row dot = in-lane + two shuffles); G goes to a private LDS tile
//        (it is never written to memory) and comes back as columns, the A operand of the weight gradients
//     -> 128 x v_mfma_f32_16x16x4_f32 for gt = G W1, gi = G W2 (A = Wt rows read from the workgroup's LDS copy four k-steps
//        at a time, B = G rows), the next tile's DMA issued a quarter per k-group in their shadows
//     -> 128 more for dW1 / dW2 into 32 f32x4 accumulators (AGPRs); rows past the end enter with G = 0
//     -> GP = gt + gi * X, GX = gt + gi * P as float4 stores (rows past the end are clamped copies: unconditional stores,
//        which is what makes the wait countable — 8 stores per tile are the only requests younger than the next DMA).
// Nothing compiler-visible is loaded from memory inside the loop (the row's 1 / norm comes by a 4-byte-per-lane DMA).
// At the end (w0 + w2) + (w1 + w3) is formed through LDS and the workgroup writes ONE partial result (<= 256 per launch);
// wgrad_reduce_kernel sums the partials in a fixed order, so the result is bit-reproducible like the two-kernel path.
__device__ __forceinline__ float row16_sum(float x) {  // sum over the 16 lanes of a DPP row, result in every lane
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
    return x;
}

template <bool TAIL, bool MASK>
__global__ __launch_bounds__(256) void bignn_backward_fused_kernel(const DgradParams p, float *__restrict__ part) {
    constexpr int WAVES = 4, NT = 4;
    constexpr int kGY = 0, kX = 1024, kP = 2048, kG = 3072, kY = 4096, kM = 5120, kInv = 6144;  // float offsets in a wave's buffer
    constexpr int kPerWave = TAIL ? 6400 : 4096;
    constexpr int kDmaOps = 12 + (TAIL ? 5 : 0) + (MASK ? 4 : 0);
    __shared__ __attribute__((aligned(1024))) float tiles[WAVES * kPerWave];
    __shared__ __attribute__((aligned(1024))) float wstage[2 * 64 * 64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int64_t n_tiles = (p.n_rows + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t tile = blockIdx.x + (int64_t)gridDim.x * wave;
    float *buf = tiles + wave * kPerWave;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)buf);

    auto issue = [&](int64_t t) __attribute__((always_inline)) {
        tile_dma(p.GY, p.ldgy, t, p.n_rows, buf_lds + kGY * 4, n, g);
        if constexpr (TAIL) {
            tile_dma(p.Y, p.ldy, t, p.n_rows, buf_lds + kY * 4, n, g);
            if constexpr (MASK) tile_dma(p.drop_mask, 64, t, p.n_rows, buf_lds + kM * 4, n, g);
            lds_dma4(p.inv_norm + min(t * 16 + n, p.n_rows - 1), buf_lds + kInv * 4);
        }
        tile_dma(p.X, p.ldx, t, p.n_rows, buf_lds + kX * 4, n, g);
        tile_dma(p.P, 64, t, p.n_rows, buf_lds + kP * 4, n, g);
    };
    auto issue_part = [&](int64_t t, int q) __attribute__((always_inline)) {  // the same requests in four instalments
        if (q == 0) {
            tile_dma(p.GY, p.ldgy, t, p.n_rows, buf_lds + kGY * 4, n, g);
            if constexpr (TAIL) lds_dma4(p.inv_norm + min(t * 16 + n, p.n_rows - 1), buf_lds + kInv * 4);
        } else if (q == 1) {
            if constexpr (TAIL) tile_dma(p.Y, p.ldy, t, p.n_rows, buf_lds + kY * 4, n, g);
            if constexpr (MASK) tile_dma(p.drop_mask, 64, t, p.n_rows, buf_lds + kM * 4, n, g);
        } else if (q == 2) {
            tile_dma(p.X, p.ldx, t, p.n_rows, buf_lds + kX * 4, n, g);
        } else {
            tile_dma(p.P, 64, t, p.n_rows, buf_lds + kP * 4, n, g);
        }
    };
    // element (row, 16 tc + n) of a swizzled tile: this lane's column of tile row `row`
    auto col_at = [&](const float *t, int row, int tc) __attribute__((always_inline)) {
        return t[row * 64 + (((4 * tc + (n >> 2)) ^ row) << 2) + (n & 3)];
    };

    {
        const unsigned w_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)wstage);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 8 * wave + u, part_i = i >> 4, rr = 4 * (i & 15) + g;
            const float *w = part_i ? p.Wt2 : p.Wt1;
            lds_dma16(w + rr * 64 + 4 * (n ^ (rr & 15)), w_lds + i * 1024);
        }
    }
    const bool has_tile = tile < n_tiles;
    if (has_tile) {
        issue(tile);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaOps) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    f32x4 dw1[NT][NT], dw2[NT][NT];  // [tj][tc]: lane (n, g) holds dW[16 tj + 4 g + r][16 tc + n]
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) dw1[a][b] = dw2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float colsum[16];  // sum over this lane's rows of G[row][16 g + s]
#pragma unroll
    for (int s = 0; s < 16; ++s) colsum[s] = 0.f;

    if (has_tile) {
        // (the transposed weights stay in LDS here: 128 more registers next to the 128 of the weight-gradient accumulators
        //  spill; the A operands are read four k-steps at a time, 32 conflict-free ds_read_b128 per tile)
        bool first = true;
        for (;;) {
            if (first)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // only the previous tile's 8 stores are younger than this DMA
            first = false;
            float gy[16], xv[NT][4], pv[NT][4], sb[NT][4], hb[NT][4];
            tile_run(buf + kGY, n, g, gy);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 a = tile_cols(buf + kX, n, g, t), b = tile_cols(buf + kP, n, g, t);
                xv[t][0] = a.x, xv[t][1] = a.y, xv[t][2] = a.z, xv[t][3] = a.w;
                pv[t][0] = b.x, pv[t][1] = b.y, pv[t][2] = b.z, pv[t][3] = b.w;
#pragma unroll
                for (int s = 0; s < 4; ++s) {  // B operands of the weight gradients: rows 4 s + g, column 16 t + n
                    const float xc = col_at(buf + kX, 4 * s + g, t), pc = col_at(buf + kP, 4 * s + g, t);
                    sb[t][s] = pc + xc;  // layers.py:56
                    hb[t][s] = pc * xc;  // layers.py:57
                }
            }
            float yv[16], mk[16], inv = 0.f;
            if constexpr (TAIL) {
                tile_run(buf + kY, n, g, yv);
                if constexpr (MASK) tile_run(buf + kM, n, g, mk);
                inv = buf[kInv + n];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every DMA buffer has been read: the next tile may land
            const int64_t next = tile + stride;
            const bool more = next < n_tiles;
            if constexpr (TAIL) {
                float dot = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) dot = fmaf(gy[s], yv[s], dot);
                dot += __shfl_xor(dot, 16);
                dot += __shfl_xor(dot, 32);
                const bool clamped = inv >= 1e12f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float y = yv[s];
                    float da = (clamped ? gy[s] : gy[s] - y * dot) * inv;
                    if constexpr (MASK) da *= mk[s];
                    gy[s] = y > 0.f ? da : da * p.slope;
                }
            }
            // G into this wave's LDS tile (swizzled like the others), column sums for db
            const bool row_ok = tile * 16 + n < p.n_rows;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<float4 *>(buf + kG + n * 64 + (((4 * g + q) ^ n) << 2)) =
                    make_float4(gy[4 * q + 0], gy[4 * q + 1], gy[4 * q + 2], gy[4 * q + 3]);
            }
            if (row_ok) {
#pragma unroll
                for (int s = 0; s < 16; ++s) colsum[s] += gy[s];
            }
            f32x4 at[NT], ai[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) at[t] = ai[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (more) issue_part(next, q);  // the next tile's DMA, a quarter per k-group: its issue cost sits in MFMA shadows
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 wa = *reinterpret_cast<const float4 *>(wstage + (16 * t + n) * 64 + (((4 * g + q) ^ n) << 2));
                    const float4 wb = *reinterpret_cast<const float4 *>(wstage + 4096 + (16 * t + n) * 64 + (((4 * g + q) ^ n) << 2));
                    at[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, gy[4 * q + 0], at[t], 0, 0, 0);  // gt = G W1
                    ai[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.x, gy[4 * q + 0], ai[t], 0, 0, 0);  // gi = G W2
                    at[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, gy[4 * q + 1], at[t], 0, 0, 0);
                    ai[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.y, gy[4 * q + 1], ai[t], 0, 0, 0);
                    at[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, gy[4 * q + 2], at[t], 0, 0, 0);
                    ai[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.z, gy[4 * q + 2], ai[t], 0, 0, 0);
                    at[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, gy[4 * q + 3], at[t], 0, 0, 0);
                    ai[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb.w, gy[4 * q + 3], ai[t], 0, 0, 0);
                }
            }
            // weight gradients: A = columns of G (rows past the end contribute nothing), B = columns of P + X / P * X
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool k_ok = tile * 16 + 4 * s + g < p.n_rows;
                float ga[NT];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const float v = col_at(buf + kG, 4 * s + g, tj);
                    ga[tj] = k_ok ? v : 0.f;
                }
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int tc = 0; tc < NT; ++tc) {
                        dw1[tj][tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[tj], sb[tc][s], dw1[tj][tc], 0, 0, 0);
                        dw2[tj][tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[tj], hb[tc][s], dw2[tj][tc], 0, 0, 0);
                    }
            }
            const int64_t row = min(tile * 16 + n, p.n_rows - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float4 gp, gx;
                gp.x = fmaf(ai[t][0], xv[t][0], at[t][0]), gx.x = fmaf(ai[t][0], pv[t][0], at[t][0]);
                gp.y = fmaf(ai[t][1], xv[t][1], at[t][1]), gx.y = fmaf(ai[t][1], pv[t][1], at[t][1]);
                gp.z = fmaf(ai[t][2], xv[t][2], at[t][2]), gx.z = fmaf(ai[t][2], pv[t][2], at[t][2]);
                gp.w = fmaf(ai[t][3], xv[t][3], at[t][3]), gx.w = fmaf(ai[t][3], pv[t][3], at[t][3]);
                *reinterpret_cast<float4 *>(p.GP + row * 64 + 16 * t + 4 * g) = gp;
                *reinterpret_cast<float4 *>(p.GX + row * 64 + 16 * t + 4 * g) = gx;
            }
            tile += stride;
            if (tile >= n_tiles) break;
        }
    }
    // Workgroup partial = (w0 + w2) + (w1 + w3), a fixed order (bit-reproducible).  The accumulators travel through LDS in
    // the lane-linear layout they have in registers (slot i of lane l at [i][l], b128, conflict-free): waves 2 and 3 hand
    // theirs to waves 0 and 1, then wave 1 hands the sum to wave 0, which writes the partial result.
#pragma unroll
    for (int s = 0; s < 16; ++s) colsum[s] = row16_sum(colsum[s]);  // over the 16 rows a lane group holds
    f32x4 *xa = reinterpret_cast<f32x4 *>(wstage);  // 32 slots x 64 lanes x 16 B = 32 KB (the weights are dead: every loop is done)
    f32x4 *xb = reinterpret_cast<f32x4 *>(tiles);   // the tiles are dead too
    f32x4 *ca = reinterpret_cast<f32x4 *>(tiles + 8192), *cb = ca + 4 * 64;  // column sums: 4 slots x 64 lanes
    auto put = [&](f32x4 *x, f32x4 *c) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                x[(a * NT + b) * 64 + lane] = dw1[a][b];
                x[(16 + a * NT + b) * 64 + lane] = dw2[a][b];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q * 64 + lane] = (f32x4){colsum[4 * q], colsum[4 * q + 1], colsum[4 * q + 2], colsum[4 * q + 3]};
    };
    auto take = [&](const f32x4 *x, const f32x4 *c) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                dw1[a][b] += x[(a * NT + b) * 64 + lane];
                dw2[a][b] += x[(16 + a * NT + b) * 64 + lane];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = c[q * 64 + lane];
            colsum[4 * q] += v[0], colsum[4 * q + 1] += v[1], colsum[4 * q + 2] += v[2], colsum[4 * q + 3] += v[3];
        }
    };
    __syncthreads();
    if (wave == 2) put(xa, ca);
    if (wave == 3) put(xb, cb);
    __syncthreads();
    if (wave == 0) take(xa, ca);
    if (wave == 1) take(xb, cb);
    __syncthreads();
    if (wave == 1) put(xa, ca);
    __syncthreads();
    if (wave == 0) {
        take(xa, ca);
        float *out = part + (int64_t)blockIdx.x * (2 * 4096 + 64);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int tc = 0; tc < NT; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = (16 * tj + 4 * g + r) * 64 + 16 * tc + n;  // dW[j][c]
                    out[idx] = dw1[tj][tc][r];
                    out[4096 + idx] = dw2[tj][tc][r];
                }
        if (n == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) out[8192 + 16 * g + s] = colsum[s];
        }
    }
}

static int dgrad_cu_count() {
    static std::atomic<int> cached{0};
    int v = cached.load();
    if (v > 0) return v;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
        v = 256;
    cached = v;
    return v;
}

// grid = number of partial results written to `part`
static int launch_backward_fused(const DgradParams &p, float *part, int64_t *n_parts, hipStream_t s) {
    const int64_t n_tiles = (p.n_rows + 15) / 16;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>({(n_tiles + 3) / 4, (int64_t)dgrad_cu_count(), (int64_t)256}));
    const dim3 gr((unsigned)grid), bl(256);
    if (p.inv_norm && p.drop_mask)
        hipLaunchKernelGGL((bignn_backward_fused_kernel<true, true>), gr, bl, 0, s, p, part);
    else if (p.inv_norm)
        hipLaunchKernelGGL((bignn_backward_fused_kernel<true, false>), gr, bl, 0, s, p, part);
    else
        hipLaunchKernelGGL((bignn_backward_fused_kernel<false, false>), gr, bl, 0, s, p, part);
    RBG_HIP(hipGetLastError());
    *n_parts = grid;
    return RBG_OK;
}

template <int TI>
static int launch_dgrad_i(const DgradParams &p, bool fast, hipStream_t s) {
    if (p.d_out <= 64) return launch_dgrad<TI, 1>(p, fast, s);
    return launch_dgrad<TI, 2>(p, fast, s);
}

static int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_bignn_wgrad_workspace(int64_t n_rows, int d_in, int d_out, int64_t *bytes) {
    if (!bytes || n_rows < 0 || d_in <= 0 || d_out <= 0) return fail(RBG_EINVAL, "bad argument");
    *bytes = wgrad_grid(n_rows) * (2 * (int64_t)d_out * d_in + d_out) * 4 + 256;
    return RBG_OK;
}

int rbg_bignn_wgrad_f32(const float *G, int64_t ldg, const float *P, const float *X, int64_t ldx, int64_t n_rows, int d_in,
                        int d_out, float *dW1, float *dW2, float *db, void *workspace, void *stream) {
    clear_error();
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || ldg < d_out || ldx < d_in)
        return fail(RBG_ESHAPE, "n_rows = %lld, d_in = %d, d_out = %d, ldg = %lld, ldx = %lld", (long long)n_rows, d_in, d_out,
                    (long long)ldg, (long long)ldx);
    if (d_in > 128 || d_out > 128) return fail(RBG_EUNSUPPORTED, "bignn_wgrad: d_in = %d, d_out = %d (both <= 128)", d_in, d_out);
    if (!dW1 || !dW2 || !workspace) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t w = (int64_t)d_out * d_in;
    if (n_rows == 0) {
        int zrc = zero_async(dW1, (size_t)w * 4, s);
        if (!zrc) zrc = zero_async(dW2, (size_t)w * 4, s);
        if (!zrc && db) zrc = zero_async(db, (size_t)d_out * 4, s);
        return zrc;
    }
    if (!G || !P || !X) return fail(RBG_EINVAL, "NULL pointer");
    WgradParams p{};
    p.G = G, p.ldg = ldg, p.P = P, p.X = X, p.ldx = ldx, p.n_rows = n_rows, p.d_in = d_in, p.d_out = d_out;
    p.part = reinterpret_cast<float *>(workspace);
    const bool fast = (d_in % 4 == 0) && (d_out % 4 == 0) && (ldg % 4 == 0) && (ldx % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(X)) & 15u) == 0;
    const int64_t grid = wgrad_grid(n_rows);
    int rc;
    if (d_out <= 32) rc = launch_wgrad_o<1>(p, fast, grid, s);
    else if (d_out <= 64) rc = launch_wgrad_o<2>(p, fast, grid, s);
    else rc = launch_wgrad_o<4>(p, fast, grid, s);
    if (rc) return rc;
    const int64_t len = 2 * w + d_out;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, s, p.part, (int)grid, len, d_out, d_in,
                       dW1, dW2, db);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_bignn_backward_workspace(int64_t n_rows, int d_in, int d_out, int64_t *bytes) {
    if (!bytes || n_rows < 0 || d_in <= 0 || d_out <= 0) return fail(RBG_EINVAL, "bad argument");
    int64_t wg = 0;
    rbg_bignn_wgrad_workspace(n_rows, d_in, d_out, &wg);
    *bytes = up256(2 * (int64_t)d_in * d_out * 4) + up256(n_rows * d_out * 4) + up256(n_rows * d_in * 4) + wg + 256;
    return RBG_OK;
}

int rbg_bignn_backward_f32(const rbg_graph *g_t, const float *GY, int64_t ldgy, const float *Y, int64_t ldy,
                           const float *inv_norm, const float *drop_mask, const float *X, int64_t ldx, const float *P,
                           const float *W1, const float *W2, int d_in, int d_out, float slope, float *GX, float *dW1,
                           float *dW2, float *db, void *workspace, void *stream) {
    clear_error();
    if (!g_t) return fail(RBG_EINVAL, "graph is NULL");
    if (g_t->device < 0) return fail(RBG_ENODEV, "operator called on a host graph (create it with device >= 0)");
    if (g_t->n_rows != g_t->n_cols) return fail(RBG_ESHAPE, "graph is not square");
    const int64_t n = g_t->n_rows;
    if (d_in <= 0 || d_out <= 0 || ldgy < d_out || ldx < d_in || (inv_norm && ldy < d_out))
        return fail(RBG_ESHAPE, "d_in = %d, d_out = %d, ldgy = %lld, ldy = %lld, ldx = %lld", d_in, d_out, (long long)ldgy,
                    (long long)ldy, (long long)ldx);
    if (d_in > 128 || d_out > 128) return fail(RBG_EUNSUPPORTED, "bignn_backward: d_in = %d, d_out = %d (both <= 128)", d_in, d_out);
    if (n == 0) return RBG_OK;
    if (!GY || !X || !P || !W1 || !W2 || !GX || !dW1 || !dW2 || !workspace || (inv_norm && !Y))
        return fail(RBG_EINVAL, "NULL pointer");
    int rc = set_device_for(g_t->device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    char *w = reinterpret_cast<char *>(workspace);
    float *Wt1 = reinterpret_cast<float *>(w), *Wt2 = Wt1 + (int64_t)d_in * d_out;
    w += up256(2 * (int64_t)d_in * d_out * 4);
    float *G = reinterpret_cast<float *>(w);
    w += up256(n * d_out * 4);
    float *GP = reinterpret_cast<float *>(w);
    w += up256(n * d_in * 4);
    const int wn = d_in * d_out;
    hipLaunchKernelGGL(wt_transpose_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, s, W1, W2, d_out, d_in, Wt1, Wt2);
    RBG_HIP(hipGetLastError());
    DgradParams p{};
    p.GY = GY, p.ldgy = ldgy, p.Y = Y, p.ldy = ldy, p.inv_norm = inv_norm, p.X = X, p.ldx = ldx, p.P = P;
    p.drop_mask = inv_norm ? drop_mask : nullptr;
    p.Wt1 = Wt1, p.Wt2 = Wt2, p.G = G, p.GP = GP, p.GX = GX, p.n_rows = n, p.d_in = d_in, p.d_out = d_out, p.slope = slope;
    const bool fast = (d_out % 64 == 0) && (ldgy % 4 == 0) && (!inv_norm || ldy % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(GY) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(drop_mask)) & 15u) == 0;
    const bool dma = fast && opt_bignn_dma() && d_in == 64 && d_out == 64 && ldx % 4 == 0 &&
                     ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(GX)) & 15u) == 0;
    if (dma) {
        // input and weight gradients in one kernel; its <= 256 partial results go through the same fixed-order reduction
        int64_t n_parts = 0;
        float *part = reinterpret_cast<float *>(w);
        if ((rc = launch_backward_fused(p, part, &n_parts, s))) return rc;
        const int64_t len = 2 * (int64_t)d_out * d_in + d_out;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, s, part, (int)n_parts, len, d_out, d_in,
                           dW1, dW2, db);
        RBG_HIP(hipGetLastError());
        return spmm_strided(g_t, GP, d_in, GX, d_in, d_in, 1, s);  // dX = GX + Â^T GP
    }
    if (d_in <= 32) rc = launch_dgrad_i<1>(p, fast, s);
    else if (d_in <= 64) rc = launch_dgrad_i<2>(p, fast, s);
    else rc = launch_dgrad_i<4>(p, fast, s);
    if (rc) return rc;
    if ((rc = rbg_bignn_wgrad_f32(G, d_out, P, X, ldx, n, d_in, d_out, dW1, dW2, db, w, stream))) return rc;
    return spmm_strided(g_t, GP, d_in, GX, d_in, d_in, 1, s);  // dX = GX + Â^T GP
}

}  // extern "C"
