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

#include "internal.h"

namespace rbg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
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
    if (lds > 64 * 1024) {
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
        RBG_HIP(hipMemsetAsync(dW1, 0, (size_t)w * 4, s));
        RBG_HIP(hipMemsetAsync(dW2, 0, (size_t)w * 4, s));
        if (db) RBG_HIP(hipMemsetAsync(db, 0, (size_t)d_out * 4, s));
        return RBG_OK;
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

}  // extern "C"
