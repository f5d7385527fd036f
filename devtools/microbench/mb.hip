// Microbenchmarks that calibrate the speed of light for the SpMM's access pattern on MI355X:
//   mb_gather : random 256-byte row gathers (16 lanes x float4) from a table of R rows, idx pre-generated
//   mb_stream : streaming float4 read of a buffer (L2 / Infinity Cache / HBM depending on size)
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int U>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ tab, const int* __restrict__ idx,
                                                     int64_t m, float* __restrict__ out, int per_group) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const int64_t group = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
    int64_t beg = group * per_group, end = beg + per_group;
    if (end > m) end = m;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t e = beg; e < end; e += 16) {
        int c = (e + sl < end) ? idx[e + sl] : 0;
        int cnt = (int)((end - e) < 16 ? (end - e) : 16);
        for (int j = 0; j + U <= cnt; j += U) {
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int cj = __shfl(c, j + u, 16);
                x[u] = *reinterpret_cast<const float4*>(tab + (int64_t)cj * 64 + sl * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        }
    }
    if (acc.x == 12345.678f) out[group] = acc.x + acc.y + acc.z + acc.w;  // keep the loads alive
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, int64_t n4, int reps, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            float4 v = buf[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

extern "C" int mb_gather(const float* tab, const int* idx, int64_t m, float* out, int per_group, int unroll, void* stream) {
    int64_t groups = (m + per_group - 1) / per_group;
    int64_t blocks = (groups + 15) / 16;
    dim3 g((unsigned)blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    switch (unroll) {
        case 2: hipLaunchKernelGGL(gather_kernel<2>, g, b, 0, s, tab, idx, m, out, per_group); break;
        case 4: hipLaunchKernelGGL(gather_kernel<4>, g, b, 0, s, tab, idx, m, out, per_group); break;
        case 8: hipLaunchKernelGGL(gather_kernel<8>, g, b, 0, s, tab, idx, m, out, per_group); break;
        default: hipLaunchKernelGGL(gather_kernel<16>, g, b, 0, s, tab, idx, m, out, per_group); break;
    }
    return (int)hipGetLastError();
}

extern "C" int mb_stream(const float* buf, int64_t n_floats, int reps, int blocks, float* out, void* stream) {
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)buf, n_floats / 4, reps, out);
    return (int)hipGetLastError();
}

// Per-XCD private working set: blocks with (blockIdx & 7) == x sweep region x (bytes_per_xcd each) `reps` times.
__global__ __launch_bounds__(256) void xcd_private_kernel(const float4* __restrict__ buf, int64_t n4_per_xcd, int reps, float* out) {
    const int x = blockIdx.x & 7;
    const int64_t lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const float4* base = buf + (int64_t)x * n4_per_xcd;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r)
        for (int64_t i = lb * 256 + threadIdx.x; i < n4_per_xcd; i += nb * 256) {
            float4 v = base[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}
extern "C" int mb_xcd_private(const float* buf, int64_t floats_per_xcd, int reps, int blocks, float* out, void* stream) {
    hipLaunchKernelGGL(xcd_private_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)buf, floats_per_xcd / 4, reps, out);
    return (int)hipGetLastError();
}

// Gather with a hot/cold split: rows with index >= hot_rows are fetched with non-temporal loads.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, int POL>
__global__ __launch_bounds__(256) void gather_nt_kernel(const float* __restrict__ tab, const int* __restrict__ idx,
                                                        int64_t m, float* __restrict__ out, int per_group, int hot_rows) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const int64_t group = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
    int64_t beg = group * per_group, end = beg + per_group;
    if (end > m) end = m;
    v4f acc = {0, 0, 0, 0};
    for (int64_t e = beg; e < end; e += 16) {
        int c = (e + sl < end) ? idx[e + sl] : 0;
        int cnt = (int)((end - e) < 16 ? (end - e) : 16);
        for (int j = 0; j + U <= cnt; j += U) {
            v4f x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int cj = __shfl(c, j + u, 16);
                const v4f* p = reinterpret_cast<const v4f*>(tab + (int64_t)cj * 64 + sl * 4);
                x[u] = (v4f){0, 0, 0, 0};
                // explicit cache-policy bit: the two flavours run under complementary exec masks
                if (cj >= hot_rows) {
                    if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(x[u]) : "v"(p) : "memory");
                    if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "+v"(x[u]) : "v"(p) : "memory");
                    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(x[u]) : "v"(p) : "memory");
                    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "+v"(x[u]) : "v"(p) : "memory");
                    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "+v"(x[u]) : "v"(p) : "memory");
                    if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "+v"(x[u]) : "v"(p) : "memory");
                } else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(x[u]) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc += x[u];
        }
    }
    if (acc.x == 12345.678f) out[group] = acc.x + acc.y + acc.z + acc.w;
}
extern "C" int mb_gather_nt(const float* tab, const int* idx, int64_t m, float* out, int per_group, int hot_rows, int pol, void* stream) {
    int64_t groups = (m + per_group - 1) / per_group;
    dim3 g((unsigned)((groups + 15) / 16)), b(256);
    hipStream_t s = (hipStream_t)stream;
    switch (pol) {
        case 0: hipLaunchKernelGGL((gather_nt_kernel<8, 0>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
        case 1: hipLaunchKernelGGL((gather_nt_kernel<8, 1>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
        case 2: hipLaunchKernelGGL((gather_nt_kernel<8, 2>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
        case 3: hipLaunchKernelGGL((gather_nt_kernel<8, 3>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
        case 4: hipLaunchKernelGGL((gather_nt_kernel<8, 4>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
        default: hipLaunchKernelGGL((gather_nt_kernel<8, 5>), g, b, 0, s, tab, idx, m, out, per_group, hot_rows); break;
    }
    return (int)hipGetLastError();
}

// XCD census: which XCC does workgroup b land on?  (the SpMM pins row classes to XCDs by blockIdx % 8)
__global__ void xcc_census_kernel(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
extern "C" int mb_xcc_census(int* out, int blocks, void* stream) {
    hipLaunchKernelGGL(xcc_census_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}

// fp32 matrix-core issue rate: every wave runs `iters` x 32 v_mfma_f32_32x32x2_f32 on CHAINS independent accumulators
// (CHAINS = 1: each MFMA depends on the previous one, as in a k-ordered dot-product tile).
typedef float mb_f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters, float seed) {
    mb_f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32 / CHAINS; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        a += 1e-6f;
    }
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[c][r];
    if (t == 12345.678f) out[0] = t;
}
extern "C" int mb_mfma_rate(float* out, int blocks, int threads, int iters, int chains, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (chains) {
        case 1: hipLaunchKernelGGL((mfma_rate_kernel<1>), dim3(blocks), dim3(threads), 0, s, out, iters, 1.0f); break;
        case 2: hipLaunchKernelGGL((mfma_rate_kernel<2>), dim3(blocks), dim3(threads), 0, s, out, iters, 1.0f); break;
        default: hipLaunchKernelGGL((mfma_rate_kernel<4>), dim3(blocks), dim3(threads), 0, s, out, iters, 1.0f); break;
    }
    return (int)hipGetLastError();
}

// Store-only twin of score_kernel: same grid, same 32x32 tile store pattern (lane = item column, 16 user rows per lane),
// no loads, no MFMA.  ld = row stride of S in floats (n, or n rounded up to 32 for line-aligned rows).
__global__ __launch_bounds__(256) void store_tiles_kernel(float* __restrict__ S, int64_t B, int64_t n, int64_t ld, int tiles_per_wave) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t user0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
    if (user0 >= B) return;
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wave;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t item = t * 32 + i;
        if (item < n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (u < B) S[u * ld + item] = (float)(r + lane);
            }
        }
    }
}
extern "C" int mb_store_tiles(float* S, int64_t B, int64_t n, int64_t ld, int tiles_per_wave, void* stream) {
    const int64_t n_tiles = (n + 31) / 32;
    dim3 grid((unsigned)((n_tiles + tiles_per_wave - 1) / tiles_per_wave), (unsigned)((B + 127) / 128));
    hipLaunchKernelGGL(store_tiles_kernel, grid, dim3(256), 0, (hipStream_t)stream, S, B, n, ld, tiles_per_wave);
    return (int)hipGetLastError();
}
