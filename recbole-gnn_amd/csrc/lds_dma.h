// lds_dma.h — LDS-DMA (global_load_lds) helpers shared by the BiGNN forward and backward kernels.
//
// A request moves 16 (or 4) bytes per lane from a per-lane global address to LDS at  M0 + 16 (4) x lane : the destination
// is lane-linear, so any LDS layout other than "as the lanes are numbered" is made on the SOURCE side.  The kernels here
// keep 16-row x 64-float tiles whose 16-byte chunk c of tile row r sits in slot c ^ r (an involution): four requests per
// tile, each covering four rows, and a lane reads its 16-float k-run or its 4 output columns back with conflict-free
// ds_read_b128.
// hipcc does not count these requests in its vmcnt bookkeeping: the caller waits (s_waitcnt vmcnt(N); requests retire in
// order) before reading the tile, and keeps compiler-visible global loads out of the loop — a wait hipcc places for one of
// its own loads after a DMA issue would drain the DMA queue as well.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbg {

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

__device__ __forceinline__ void lds_dma4(const void *gsrc, unsigned lds_dst) {  // one dword per lane, 256 bytes per request
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// Rows 16 t .. 16 t + 15 of a row-major [n_rows, >= 64] array (row stride ld floats) into the 4 KB tile at lds_dst; rows
// past the end are copies of the last row.  n = lane & 15, g = lane >> 4.
__device__ __forceinline__ void tile_dma(const float *base, int64_t ld, int64_t t, int64_t n_rows, unsigned lds_dst, int n, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = 4 * j + g;  // this lane's destination is slot n of tile row rr; it sources chunk n ^ rr
        const int64_t r = min(t * 16 + rr, n_rows - 1);
        lds_dma16(base + r * ld + 4 * (n ^ rr), lds_dst + j * 1024);
    }
}

// this lane's k-run [16 g, 16 g + 16) of tile row n (the MFMA operand layout: lane = row, registers = k)
__device__ __forceinline__ void tile_run(const float *tile, int n, int g, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4 *>(tile + n * 64 + (((4 * g + q) ^ n) << 2));
        v[4 * q + 0] = a.x, v[4 * q + 1] = a.y, v[4 * q + 2] = a.z, v[4 * q + 3] = a.w;
    }
}

// this lane's columns 16 t + 4 g .. + 3 of tile row n (the 16x16x4 accumulator layout of the transposed product)
__device__ __forceinline__ float4 tile_cols(const float *tile, int n, int g, int t) {
    return *reinterpret_cast<const float4 *>(tile + n * 64 + (((4 * t + g) ^ n) << 2));
}

}  // namespace rbg
