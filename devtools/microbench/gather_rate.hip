// gather_rate: what the vector memory path of a CU sustains for RANDOM ROW GATHERS — the access pattern of the column-slab
// propagation without anything else of it (no entry stream, no headers, no epilogue).  r05, DESIGN 6.12.
//
// Every lane-group (RB / 16 lanes) of a wave gathers one RB-byte row per load (buffer_load_dwordx4), NB loads in flight per wave,
// `iters` rounds; row indices come from a hash of (wave, round, slot, lane-group) — no index loads.  Table sizes from L1-resident
// to beyond the L2 / the Infinity Cache, cache-policy bits of the load (aux: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1), residency
// (waves per SIMD by a dummy LDS allocation).  Prints one JSON line per configuration: us per launch, bytes / clk / CU at the
// measured duration, wave-loads (1 KiB each) per us.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip      run: ./gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// RB = bytes per gathered row (64, 128, 256); NB = loads in flight per wave; AUX = cache policy immediate
template <int RB, int NB, int AUX>
__global__ __launch_bounds__(256) void gather_kernel(const float *table, unsigned row_mask, int iters, float *out, int lds_pad, int skew_shift) {
    extern __shared__ float pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0.f;  // (keeps the allocation)
    constexpr int G = RB / 16;                 // lanes per row
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, (row_mask + 1) * RB, 0x00020000);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    unsigned seed = mix(wid * 64u + lg + 0x9e3779b9u);
    for (int it = 0; it < iters; ++it) {
        v4f x[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            seed = seed * 1664525u + 1013904223u;
            unsigned r = mix(seed);
            if (skew_shift) r = r >> (mix(seed ^ 0x5bd1e995u) % skew_shift);  // (power-law-ish: small indices are hot)
            const int off = (int)((r & row_mask) * RB + sl * 16);
            x[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) acc += x[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[wid] = acc.x;  // (never true: keeps the loads)
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

template <int RB, int NB, int AUX>
static void run(const float *table, size_t table_bytes, int wgs_per_cu, int iters, int lds_bytes, int skew, float *out, int cus, double clk_ghz) {
    const unsigned rows = (unsigned)(table_bytes / RB);
    const unsigned grid = (unsigned)(wgs_per_cu * cus);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gather_kernel<RB, NB, AUX>), dim3(grid), dim3(256), lds_bytes, 0, table, rows - 1, iters, out, 0, skew);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gather_kernel<RB, NB, AUX>), dim3(grid), dim3(256), lds_bytes, 0, table, rows - 1, iters, out, 0, skew);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    const double loads = (double)grid * 4 * iters * NB;  // wave-loads of 1 KiB
    printf("{\"kind\": \"gather_rate\", \"row_bytes\": %d, \"in_flight\": %d, \"aux\": %d, \"table_mb\": %.3f, \"wgs_per_cu\": %d, \"iters\": %d, \"lds_bytes\": %d, "
           "\"skew\": %d, \"us\": %.2f, \"wave_loads\": %.0f, \"wave_loads_per_us\": %.0f, \"GBps\": %.0f, \"bytes_per_clk_per_cu_at_%.1fGHz\": %.1f, "
           "\"clk_per_wave_load_per_cu\": %.1f}\n",
           RB, NB, AUX, table_bytes / 1048576.0, wgs_per_cu, iters, lds_bytes, skew, us, loads, loads / us, loads * 1024 / us / 1e3, clk_ghz,
           loads * 1024 / (us * 1e-6) / (clk_ghz * 1e9) / cus, (us * 1e-6) * (clk_ghz * 1e9) * cus / loads);
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char **argv) {
    int dev = 0, cus = 0, mhz = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev));
    const double ghz = 2.1;  // (what the chip holds under this load; the attribute's %d kHz is the ceiling)
    fprintf(stderr, "CUs %d, clock attribute %d kHz\n", cus, mhz);
    const size_t max_bytes = (size_t)1 << 30;
    float *table = nullptr, *out = nullptr;
    CK(hipMalloc(&table, max_bytes));
    CK(hipMalloc(&out, 1 << 22));
    {
        std::vector<float> h(1 << 22);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i & 1023) * 1e-3f;
        for (size_t o = 0; o < max_bytes; o += h.size() * 4) CK(hipMemcpy((char *)table + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    const size_t KB = 1024, MB = 1024 * 1024;
    // 1. the table's place in the hierarchy (128-byte rows, 8 in flight, 8 workgroups per CU, 64 rounds: long enough to drown the launch)
    for (size_t tb : {16 * KB, 256 * KB, 1 * MB, 2 * MB, 4 * MB, 6 * MB, 16 * MB, 128 * MB, 1024 * MB}) run<128, 8, 0>(table, tb, 8, 64, 0, 0, out, cus, ghz);
    // 2. row width at an L2-resident and at a 6 MB table
    for (size_t tb : {2 * MB, 6 * MB}) {
        run<64, 8, 0>(table, tb, 8, 64, 0, 0, out, cus, ghz);
        run<256, 8, 0>(table, tb, 8, 64, 0, 0, out, cus, ghz);
    }
    // 3. cache policy of the gather
    for (size_t tb : {16 * KB, 2 * MB, 6 * MB, 128 * MB}) {
        run<128, 8, 1>(table, tb, 8, 64, 0, 0, out, cus, ghz);
        run<128, 8, 2>(table, tb, 8, 64, 0, 0, out, cus, ghz);
        run<128, 8, 16>(table, tb, 8, 64, 0, 0, out, cus, ghz);
        run<128, 8, 17>(table, tb, 8, 64, 0, 0, out, cus, ghz);
    }
    // 4. loads in flight per wave x residency (LDS allocation limits workgroups per CU: 160 KB / n)
    for (size_t tb : {2 * MB, 6 * MB}) {
        run<128, 4, 0>(table, tb, 8, 128, 0, 0, out, cus, ghz);
        run<128, 16, 0>(table, tb, 8, 32, 0, 0, out, cus, ghz);
        run<128, 8, 0>(table, tb, 4, 128, 40000, 0, out, cus, ghz);
        run<128, 16, 0>(table, tb, 4, 64, 40000, 0, out, cus, ghz);
        run<128, 8, 0>(table, tb, 2, 256, 80000, 0, out, cus, ghz);
        run<128, 16, 0>(table, tb, 2, 128, 80000, 0, out, cus, ghz);
    }
    // 5. the propagation's own size: 8 rounds of 8 per wave (512 k wave-loads over 8 192 waves): what a launch adds
    for (size_t tb : {2 * MB, 6 * MB}) {
        run<128, 8, 0>(table, tb, 8, 8, 0, 0, out, cus, ghz);
        run<128, 8, 0>(table, tb, 8, 4, 0, 0, out, cus, ghz);
        run<128, 8, 0>(table, tb, 8, 1, 0, 0, out, cus, ghz);
    }
    // 6. skewed indices (hot rows): what the L1 does with them
    for (size_t tb : {6 * MB}) {
        run<128, 8, 0>(table, tb, 8, 64, 0, 8, out, cus, ghz);
        run<128, 8, 0>(table, tb, 8, 64, 0, 16, out, cus, ghz);
    }
    return 0;
}
