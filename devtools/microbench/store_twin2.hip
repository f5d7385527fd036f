// Store-only twins of the scoring kernel, second set (r05): does the [B, n] write stream approach fill_'s rate when the lines a wave
// writes into ONE row go out back to back (k tiles of a row pair per iteration), or when the four waves of a workgroup write
// adjacent pieces of the same rows?  Whole aligned lines, dword per lane, two rows x 128 bytes per instruction (what
// score_uni_kernel does).
//   k      : item tiles a wave covers per iteration
//   order  : 0 = row pair outer, tile inner (k lines of a row back to back)   1 = tile outer, row pair inner (the kernel's order)
//   layout : 0 = the four waves take four 32-user tiles and the same item tiles (the kernel)   1 = the four waves take the same
//            32 users and adjacent groups of k item tiles (a workgroup iteration covers 4 k tiles of 32 rows)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) float gfloat;

template <bool NT>
__global__ __launch_bounds__(256) void store_twin2_kernel(float *__restrict__ S, int64_t B, int64_t n, int64_t ld, int tiles_per_walk, int k,
                                                          int order, int layout) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int64_t user0 = layout == 0 ? ((int64_t)blockIdx.y * 4 + wave) * 32 : (int64_t)blockIdx.y * 32;
    if (user0 >= B) return;
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_walk;
    const int64_t t1 = t0 + tiles_per_walk < n_tiles ? t0 + tiles_per_walk : n_tiles;
    gfloat *base = (gfloat *)S;
    const int step = layout == 0 ? k : 4 * k;
    for (int64_t t = t0 + (layout == 0 ? 0 : wave * k); t < t1; t += step) {
        const int kk = (int)((t1 - t) < k ? (t1 - t) : k);
        if (order == 0) {
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                for (int j = 0; j < kk; ++j) {
                    const int64_t a = (u * ld + (t + j) * 32) & ~(int64_t)31;
                    if (u < B && a + 32 <= B * ld) {
                        if (NT) __builtin_nontemporal_store((float)(r + lane), &base[a + i]);
                        else base[a + i] = (float)(r + lane);
                    }
                }
            }
        } else {
            for (int j = 0; j < kk; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int64_t a = (u * ld + (t + j) * 32) & ~(int64_t)31;
                    if (u < B && a + 32 <= B * ld) {
                        if (NT) __builtin_nontemporal_store((float)(r + lane), &base[a + i]);
                        else base[a + i] = (float)(r + lane);
                    }
                }
            }
        }
    }
}

extern "C" int mb_store_twin2(float *S, int64_t B, int64_t n, int64_t ld, int tiles_per_walk, int k, int order, int layout, int nt, void *stream) {
    const int64_t n_tiles = (n + 31) / 32;
    dim3 grid((unsigned)((n_tiles + tiles_per_walk - 1) / tiles_per_walk), (unsigned)(layout == 0 ? (B + 127) / 128 : (B + 31) / 32));
    hipStream_t s = (hipStream_t)stream;
    if (nt) hipLaunchKernelGGL((store_twin2_kernel<true>), grid, dim3(256), 0, s, S, B, n, ld, tiles_per_walk, k, order, layout);
    else hipLaunchKernelGGL((store_twin2_kernel<false>), grid, dim3(256), 0, s, S, B, n, ld, tiles_per_walk, k, order, layout);
    return (int)hipGetLastError();
}
