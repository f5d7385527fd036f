// topk_screen.h — the bf16 screen + exact rescoring form of the fused top-k (topk_screen.hip), as topk.hip drives it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbg {

struct ScreenLayout {
    bool fits;                       // the screen's workspace exists for this (B, n_items)
    int64_t n_tiles;                 // 32-item tiles
    int ut, n_ublocks, tpc, nc;      // main pass: user tiles per wave, user blocks (ut x 32 users), item tiles per chunk, chunks
    int64_t image_off, uimage_off, pool_off, cnt_off, bytes;  // byte offsets inside the screen's part of the workspace
};
ScreenLayout screen_layout(int64_t B, int64_t n_items);
bool screen_applicable(int64_t B, int64_t n_items, int d, int k);

struct ScreenCall {
    const float *U, *I;
    const int64_t *users;
    const int32_t *rowptr, *col;  // history (may be NULL)
    int64_t n_users, n_items, B;
    int d, k;
    char *w;              // the screen's part of the workspace (ScreenLayout offsets)
    float *pre_val;       // [B][splits][32] group maxima of the pre-pass (lower bounds) ...
    int32_t *pre_idx;     // ... and their items
    int splits, tpc_s;    // pre-pass geometry: `splits` chunks of `tpc_s` tiles over the first sample_tiles tiles
    int64_t sample_tiles;
    float *tau0;          // [B] the bound the main pass screens against (workspace: screen_main's threshold kernel writes it)
    float *out_val;
    int64_t *out_idx;
};
// image + pre-pass (fills pre_val / pre_idx)
int screen_prepass(const ScreenCall &c, hipStream_t s);
// thresholds + main pass + rescoring merge (fills out_val / out_idx)
int screen_main(const ScreenCall &c, hipStream_t s);

}  // namespace rbg
