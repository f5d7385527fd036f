// Store-only twins of score_kernel with WIDER stores (r03): does the [B, n] write get closer to fill_'s rate when a wave
// store carries 16 bytes per lane (8 or 16 lanes per row piece) instead of 4 (32 lanes per 128-byte row piece)?
// Same grid as score_kernel: blockIdx.y = 128 user rows (4 waves x 32), blockIdx.x = tiles_per_wave item tiles of 32 columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) float gfloat;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4f gv4f;

// MODE 0: dword per lane, 2 rows x 128 B per instruction (what score_kernel does), 16 instructions per 32 x 32 tile
// MODE 1: dwordx4 per lane, 8 rows x 128 B per instruction, 4 instructions per tile
// MODE 2: dwordx4 per lane, 4 rows x 256 B per instruction (a 32 x 64 tile pair), 8 instructions per 2 tiles
// Rows are written at their line-aligned position: with ld % 32 != 0 the piece is shifted down to the row's line boundary
// (what aligned_emit does with the previous tile's tail), so every store is whole lines.
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void store_twin_kernel(float *__restrict__ S, int64_t B, int64_t n, int64_t ld, int tiles_per_wave) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t user0 = ((int64_t)blockIdx.y * 4 + wave) * 32;
    if (user0 >= B) return;
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wave;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    gfloat *base = (gfloat *)S;
    constexpr int STEP = MODE == 2 ? 2 : 1;
    for (int64_t t = t0; t + STEP <= t1; t += STEP) {
        if (MODE == 0) {
            const int i = lane & 31, h = lane >> 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t u = user0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int64_t a = (u * ld + t * 32) & ~(int64_t)31;  // line-aligned
                if (u < B && a + 32 <= B * ld) {
                    if (NT) __builtin_nontemporal_store((float)(r + lane), &base[a + i]);
                    else base[a + i] = (float)(r + lane);
                }
            }
        } else if (MODE == 1) {
            const int j = lane & 7, rr = lane >> 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t u = user0 + r * 8 + rr;
                const int64_t a = (u * ld + t * 32) & ~(int64_t)31;
                if (u < B && a + 32 <= B * ld) {
                    const v4f v = {(float)r, (float)lane, 1.f, 2.f};
                    if (NT) __builtin_nontemporal_store(v, (gv4f *)&base[a + j * 4]);
                    else *(gv4f *)&base[a + j * 4] = v;
                }
            }
        } else {
            const int j = lane & 15, rr = lane >> 4;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int64_t u = user0 + r * 4 + rr;
                const int64_t a = (u * ld + t * 32) & ~(int64_t)31;
                if (u < B && a + 64 <= B * ld) {
                    const v4f v = {(float)r, (float)lane, 1.f, 2.f};
                    if (NT) __builtin_nontemporal_store(v, (gv4f *)&base[a + j * 4]);
                    else *(gv4f *)&base[a + j * 4] = v;
                }
            }
        }
    }
}

extern "C" int mb_store_twin(float *S, int64_t B, int64_t n, int64_t ld, int tiles_per_wave, int mode, int nt, void *stream) {
    const int64_t n_tiles = (n + 31) / 32;
    dim3 grid((unsigned)((n_tiles + tiles_per_wave - 1) / tiles_per_wave), (unsigned)((B + 127) / 128));
    hipStream_t s = (hipStream_t)stream;
#define L(M, N) hipLaunchKernelGGL((store_twin_kernel<M, N>), grid, dim3(256), 0, s, S, B, n, ld, tiles_per_wave)
    switch (mode * 2 + (nt ? 1 : 0)) {
        case 0: L(0, false); break;
        case 1: L(0, true); break;
        case 2: L(1, false); break;
        case 3: L(1, true); break;
        case 4: L(2, false); break;
        default: L(2, true); break;
    }
#undef L
    return (int)hipGetLastError();
}
