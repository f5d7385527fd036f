// plane_image.h — an fp32 row table split ONCE into three bf16 planes stored tile by tile in the LDS layout of RowTile3::Planes
// (tile t = 3 planes x 32 rows x (64 NCHUNK + 8) bf16, contiguous), and the LDS-DMA that takes a tile (r06).
//
// The MFMA kernels on split operands (top-k, InfoNCE gradients) walk 32-row tiles of a table; every workgroup used to fetch a tile
// as fp32, split it (3-level bf16 split: ~ 45 VALU per 8 floats) and publish three planes with LDS stores — the same work in every
// workgroup that walks the table.  With the image a workgroup issues global_load_lds_dwordx4 (16 bytes per lane, lane-linear: the
// image IS the LDS layout) one tile ahead: no registers, no VALU, no LDS-write instructions in the loop.  Same planes, same
// products: results are bit-identical.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lds_dma.h"
#include "mfma_common.h"

namespace rbg {

// NPL = 3: the bf16 planes of RowTile3; NPL = 2 (r06): the two fp16 planes of lse.hip's fp16 form (written by its row kernels)
template <int NCHUNK, int NPL = 3>
struct PlaneImage {
    static constexpr int LDH = NCHUNK * 64 + 8;
    static constexpr int kTileBytes = NPL * 32 * LDH * 2;  // NPL = 3: 13 824 (d <= 64), 26 112 (d <= 128); NPL = 2: 9 216, 17 408: multiples of 16
    static constexpr int kRounds = (kTileBytes + 4095) / 4096;
    // all 256 threads: tile t of the image -> the LDS tile at lds_dst (byte address), 16 bytes per lane per round
    static __device__ __forceinline__ void dma(const char *image, int64_t t, unsigned lds_dst, int tid, int wave) {
        const char *src = image + t * (int64_t)kTileBytes;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int chunk = tid + 256 * k;
            if (chunk * 16 < kTileBytes) lds_dma16(src + chunk * 16, lds_dst + (unsigned)(k * 4096 + wave * 1024));
        }
    }
};
// this wave's DMA requests have landed (hipcc does not count them: lds_dma.h)
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// rows [32 t, 32 t + 32) of the row-major table T [n_rows, >= d] (row stride ld floats; rows past the end repeat the last row) -> tile t
template <int NCHUNK, bool VEC>
__global__ __launch_bounds__(256) void plane_image_kernel(const float *__restrict__ T, int64_t ld, int64_t n_rows, int d, char *__restrict__ image) {
    using Tiles = RowTile3<NCHUNK, (VEC ? RUN_VEC : RUN_ANY)>;
    Tiles tiles;
    const int64_t t = blockIdx.x;
    tiles.fetch(T, ld, n_rows, d, t, threadIdx.x);
    tiles.publish(*reinterpret_cast<typename Tiles::Planes *>(image + t * (int64_t)PlaneImage<NCHUNK>::kTileBytes), threadIdx.x);
}

}  // namespace rbg
