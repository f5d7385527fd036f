// What bounds the screened top-k's main loop: MFMA chains of 5 on a fresh accumulator per tile, with / without the tile loads and the
// sign filter.  Standalone: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 screen_loop.hip -o screen_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 as_frag(const i32x4v &v) { return __builtin_bit_cast(bf16x8, v); }

template <int UT, int MODE>  // MODE bit 0: load tiles, bit 1: filter, bit 2: zero-init by the first MFMA (C = 0) instead of a chain on one accumulator
__global__ __launch_bounds__(256) void loop_kernel(const char *image, int tiles, int n_tiles, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(image) + lane;
    i32x4v A[UT][5];
    for (int j = 0; j < UT; ++j)
        for (int s = 0; s < 5; ++s) A[j][s] = i32x4v{lane + j, s, wave, 0x3f803f80};
    const int64_t t0 = ((int64_t)blockIdx.x * tiles) % n_tiles;
    i32x4v Bn[5];
    for (int s = 0; s < 5; ++s) Bn[s] = img[(t0 * 5 + s) * 64];
    unsigned found = 0;
    for (int t = 0; t < tiles; ++t) {
        i32x4v Bc[5];
        for (int s = 0; s < 5; ++s) Bc[s] = Bn[s];
        if ((MODE & 1) && t + 1 < tiles) {
            const int64_t tt = (t0 + t + 1) % n_tiles;
#pragma unroll
            for (int s = 0; s < 5; ++s) Bn[s] = img[(tt * 5 + s) * 64];
        }
        f32x16 acc[UT];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < UT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[j][4]), as_frag(Bc[4]), zero, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < UT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[j][s]), as_frag(Bc[s]), acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < UT; ++j) {
            if (MODE & 2) {
                unsigned x = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) x = __builtin_amdgcn_alignbit(x, __float_as_uint(acc[j][r]), 31);
                if (__builtin_amdgcn_ballot_w64((~x & 0xffffu) != 0u) != 0ull) found += x;
            } else {
                if (__builtin_amdgcn_ballot_w64(acc[j][0] == 123.f) != 0ull) found += 1;
            }
        }
    }
    if (found == 0xdeadbeefu) out[0] = found;
}

template <int UT, int MODE>
static void run(const char *name, const char *image, int n_tiles, unsigned *out, int blocks, int tiles) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<UT, MODE>), dim3(blocks), dim3(256), 0, 0, image, tiles, n_tiles, out);
    hipEventRecord(a);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((loop_kernel<UT, MODE>), dim3(blocks), dim3(256), 0, 0, image, tiles, n_tiles, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 100.0;
    const double mfma = (double)blocks * 4 * tiles * UT * 5;
    printf("{\"what\": \"%s\", \"ut\": %d, \"mode\": %d, \"blocks\": %d, \"tiles\": %d, \"us\": %.1f, \"cycles_per_mfma_per_simd_at_2.1GHz\": %.1f}\n", name, UT, MODE,
           blocks, tiles, us, us * 1e-6 * 2.1e9 * 1024 / mfma);
}

// UT user tiles per wave, processed two chains at a time (four accumulators live at most); DIST = tiles the loads run ahead
template <int UT, int DIST>
__global__ __launch_bounds__(256) void loop2_kernel(const char *image, int tiles, int n_tiles, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i32x4v *img = reinterpret_cast<const i32x4v *>(image) + lane;
    i32x4v A[UT][5];
    for (int j = 0; j < UT; ++j)
        for (int s = 0; s < 5; ++s) A[j][s] = i32x4v{lane + j, s, wave, 0x3f803f80};
    const int64_t t0 = ((int64_t)blockIdx.x * tiles) % n_tiles;
    i32x4v Bq[DIST + 1][5];
#pragma unroll
    for (int q = 0; q < DIST; ++q)
#pragma unroll
        for (int s = 0; s < 5; ++s) Bq[q][s] = img[(((t0 + q) % n_tiles) * 5 + s) * 64];
    unsigned found = 0;
    for (int t = 0; t < tiles; t += DIST + 1) {
#pragma unroll
        for (int u = 0; u <= DIST; ++u) {  // (unrolled ring: slot (u + DIST) % (DIST + 1) receives tile t + u + DIST)
            const int64_t tt = (t0 + t + u + DIST) % n_tiles;
#pragma unroll
            for (int s = 0; s < 5; ++s) Bq[(u + DIST) % (DIST + 1)][s] = img[(tt * 5 + s) * 64];
            const i32x4v *Bc = Bq[u];
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jp = 0; jp < UT; jp += 2) {
                f32x16 a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[jp][4]), as_frag(Bc[4]), zero, 0, 0, 0);
                f32x16 a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[jp + 1][4]), as_frag(Bc[4]), zero, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[jp][s]), as_frag(Bc[s]), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[jp + 1][s]), as_frag(Bc[s]), a1, 0, 0, 0);
                }
                unsigned x = 0, y = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) x = __builtin_amdgcn_alignbit(x, __float_as_uint(a0[r]), 31), y = __builtin_amdgcn_alignbit(y, __float_as_uint(a1[r]), 31);
                if (__builtin_amdgcn_ballot_w64((~x & 0xffffu) != 0u) != 0ull) found += x;
                if (__builtin_amdgcn_ballot_w64((~y & 0xffffu) != 0u) != 0ull) found += y;
            }
        }
    }
    if (found == 0xdeadbeefu) out[0] = found;
}

template <int UT, int DIST>
static void run2(const char *image, int n_tiles, unsigned *out, int blocks, int tiles) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop2_kernel<UT, DIST>), dim3(blocks), dim3(256), 0, 0, image, tiles, n_tiles, out);
    hipEventRecord(a);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((loop2_kernel<UT, DIST>), dim3(blocks), dim3(256), 0, 0, image, tiles, n_tiles, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 100.0;
    const double mfma = (double)blocks * 4 * tiles * UT * 5;
    printf("{\"what\": \"loop2\", \"ut\": %d, \"dist\": %d, \"blocks\": %d, \"tiles\": %d, \"us\": %.1f, \"cycles_per_mfma_per_simd_at_2.1GHz\": %.1f}\n", UT, DIST,
           blocks, tiles, us, us * 1e-6 * 2.1e9 * 1024 / mfma);
}

int main() {
    const int n_tiles = 1281;
    char *image;
    unsigned *out;
    hipMalloc(&image, (size_t)n_tiles * 5 * 1024);
    hipMemset(image, 0x3c, (size_t)n_tiles * 5 * 1024);
    hipMalloc(&out, 64);
    run<2, 0>("mfma only", image, n_tiles, out, 1024, 21);
    run<2, 1>("mfma + loads", image, n_tiles, out, 1024, 21);
    run<2, 2>("mfma + filter", image, n_tiles, out, 1024, 21);
    run<2, 3>("all", image, n_tiles, out, 1024, 21);
    run<4, 0>("mfma only", image, n_tiles, out, 512, 21);
    run<4, 3>("all", image, n_tiles, out, 512, 21);
    run<1, 0>("mfma only", image, n_tiles, out, 2048, 21);
    run<1, 3>("all", image, n_tiles, out, 2048, 21);
    run<2, 0>("mfma only, long", image, n_tiles, out, 1024, 210);
    run<2, 3>("all, long", image, n_tiles, out, 1024, 210);
    // the same total work (4096 users x 1281 tiles... here 128 user tiles x 21 x 64 chunks) in the loop2 forms
    run2<2, 1>(image, n_tiles, out, 1024, 22);
    run2<2, 2>(image, n_tiles, out, 1024, 21);
    run2<4, 1>(image, n_tiles, out, 512, 22);
    run2<4, 2>(image, n_tiles, out, 512, 21);
    run2<8, 1>(image, n_tiles, out, 256, 22);
    run2<8, 2>(image, n_tiles, out, 256, 21);
    run2<8, 1>(image, n_tiles, out, 512, 11);
    run2<8, 2>(image, n_tiles, out, 512, 12);
    return 0;
}
