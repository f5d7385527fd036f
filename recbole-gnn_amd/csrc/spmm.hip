// spmm.hip — the hot path: Y = Â·X on gfx950 (MI355X), CSR in HBM, fp32.
//
// Replaces LightGCNConv.message_and_aggregate / message+aggregate
//   (recbole_gnn/model/layers.py:13-20 -> torch_sparse.matmul / PyG gather-scatter)
// and the layer loop + stack/mean of LightGCN.forward / SGL.forward
//   (recbole_gnn/model/general_recommender/lightgcn.py:70-81, sgl.py:128-145).
//
// Design: HBM/L2-bound gather.  A dense row of d floats is covered by a
// lane-group of d/4 lanes holding one float4 each, so a 64-wide wavefront has 64/(d/4) lane-groups
// (4 at d = 64) and every gathered neighbour row is one fully coalesced d*4-byte read.  Rows are
// degree-binned at graph build (graph_build.cpp):
//   short rows   (deg <= short_max): one row per lane-group, neighbours visited in column order
//   wave rows    (deg <= wave_max) : the lane-groups of one wavefront split the row, ds_bpermute butterfly
//   block rows   (deg  > wave_max) : a 256-thread workgroup per row segment, LDS reduce; rows longer
//                                    than seg_len are split into segments whose partial sums are added
//                                    in fixed segment order by the last segment to finish
// For graphs built from interactions the rows are additionally split by class: workgroups on XCDs 0-3 take
// user rows (they gather item embeddings only), XCDs 4-7 item rows — each XCD's private 4 MB L2 then
// caches one embedding table, not two (measured: the kernel is bound by L2 misses).
// A caller-supplied node partition (communities; rbg_graph_create_partitioned) pins each community's rows to its
// own XCD(s) instead, so an XCD's L2 only sees that community's embeddings plus the cut.
// (col,val) are fetched lane-parallel (one coalesced non-temporal read per lane-group chunk) and
// broadcast with ds_bpermute; U (8) neighbour rows are in flight per lane-group before the FMAs.
// No float atomics anywhere: the result is bit-stable run to run.
// The layer-mean of LightGCN.forward is fused into the last layer's epilogue.

#include <hip/hip_runtime.h>
#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <type_traits>

#include "internal.h"

namespace rbg {

struct RowSrc {  // where dense rows live: row r is p0 + r*ld (r < split) or p1 + r*ld (r >= split);
    const float *p0;  // p1 is PRE-OFFSET by the host (second table base minus split*ld), see make_src()
    const float *p1;
    int32_t split;
    int32_t pad;
    int64_t ld;
};

enum { MODE_STORE = 0, MODE_ACCUM = 1, MODE_MEAN = 2, MODE_HORNER = 3, MODE_NOISE = 4 };

struct SpmmParams {
    const int32_t *rowptr;
    const int32_t *col;
    const float *val;
    const RowDesc *desc;
    const BlockTask *tasks;
    float *partials;
    uint32_t *counters;   // [2][n_split]: second set for the upper column half in column-half mode
    int32_t n_split;
    RowSrc x;
    float *y;  // may be NULL in MODE_MEAN
    int64_t ldy;
    int32_t n_rows;
    int32_t n_groups;  // > 1: workgroup b (XCD b & 7) runs row group xmap.grp[b & 7]
    XcdMap xmap;
    GroupPlan grp[kMaxGroups];
    int32_t mode;
    int32_t nt_store;
    int32_t slab;  // column-half launch over operands in slab layout [2][rows][D] (SLAB instantiation): X, Y, e0, prev, addend
    int32_t pad2;
    // MODE_HORNER: y[row] = (addend[row] + acc) / denom   (one step of the backward chain)
    // MODE_NOISE:  y[row] = acc + sign(acc) * addend[row] / max(||addend[row]||, 1e-12) * denom   (addend = noise, denom = eps)
    const float *addend;
    // MODE_MEAN: mean_out[row] = (e0[row] + sum_i prev[i][row] + acc) / denom, acc += partial[row] first when partial != NULL
    // (the sharded last layer: partial = the interior product, this launch = the halo product)
    float *mean_out;
    const float *partial;
    RowSrc e0;
    const float *prev[RBG_MAX_FUSED_LAYERS];
    int32_t n_prev;
    float denom;
};

__device__ __forceinline__ const float *src_row(const RowSrc &s, int r) {
    return (r < s.split ? s.p0 : s.p1) + (int64_t)r * s.ld;
}
// same with a compile-time row stride (the common contiguous case): one select + one shift-add
template <int LD>
__device__ __forceinline__ const float *src_row_c(const RowSrc &s, int r) {
    return (r < s.split ? s.p0 : s.p1) + (int64_t)r * LD;
}

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 ld4_stream(const float *p, int nt) {
    if (nt) {
        const v4f w = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return make_float4(w.x, w.y, w.z, w.w);
    }
    return ld4(p);
}
// streaming store: the row is not read again by this launch, so it should not displace gathered rows in L2
__device__ __forceinline__ void st4_stream(float *p, float4 v, int nt) {
    if (nt) {
        const v4f w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p));
    } else {
        st4(p, v);
    }
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 fma4(float s, float4 x, float4 a) {
    return make_float4(fmaf(s, x.x, a.x), fmaf(s, x.y, a.y), fmaf(s, x.z, a.z), fmaf(s, x.w, a.w));
}

// Sum of val[e] * X[col[e], 4*sl .. 4*sl+3] over the chunks of [beg,end) owned by lane-group g of G.
// A chunk is LPR consecutive entries; lane sl of the group fetches entry sl of the chunk.
// XS = compile-time row stride of a contiguous source (D, or 2 D in column-half mode); coff = first column of the slice.
template <int D, int U, bool CONTIG, int XS = D>
__device__ __forceinline__ float4 gather_range(const SpmmParams &p, int beg, int end, int g, int G, int sl, int64_t coff = 0) {
    constexpr int LPR = D / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = beg + g * LPR; e < end; e += G * LPR) {
        const int idx = e + sl;
        int c = 0;
        float v = 0.f;
        if (idx < end) {
            c = p.col[idx];
            v = p.val[idx];
        }
        const int cnt = min(LPR, end - e);
        int j = 0;
        for (; j + U <= cnt; j += U) {
            float4 xv[U];
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, j + u, LPR);
                vv[u] = __shfl(v, j + u, LPR);
                xv[u] = ld4((CONTIG ? src_row_c<XS>(p.x, cj) : src_row(p.x, cj)) + coff + sl * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = fma4(vv[u], xv[u], acc);
        }
        if (j < cnt) {  // fewer than U entries left in this chunk: issue them together, predicated
            float4 xv[U - 1];
            float vv[U - 1];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                const int src = min(j + u, LPR - 1);
                const int cj = __shfl(c, src, LPR);
                const float vj = __shfl(v, src, LPR);
                const bool on = (j + u) < cnt;
                vv[u] = on ? vj : 0.f;
                xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) xv[u] = ld4((CONTIG ? src_row_c<XS>(p.x, cj) : src_row(p.x, cj)) + coff + sl * 4);
            }
#pragma unroll
            for (int u = 0; u < U - 1; ++u) acc = fma4(vv[u], xv[u], acc);
        }
    }
    return acc;
}

// Add the lane-groups of a wavefront together (every lane ends with the total of its column slice).
template <int D>
__device__ __forceinline__ float4 reduce_groups(float4 a) {
    constexpr int LPR = D / 4;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off);
        a.y += __shfl_xor(a.y, off);
        a.z += __shfl_xor(a.z, off);
        a.w += __shfl_xor(a.w, off);
    }
    return a;
}

// Row epilogue, executed by the LPR lanes that hold the finished row.
// D = width of the contiguous [N, D] side arrays (prev, mean_out, addend); c0 = first column of this lane's float4.
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }  // torch.sign

__device__ __forceinline__ void finish_row(const SpmmParams &p, int row, float4 acc, int sl, int D, int64_t coff = 0) {
    const int64_t c0 = coff + sl * 4;
    if (p.mode == MODE_NOISE) {  // simgcl.py:32-33: all_embs + sign(all_embs) * F.normalize(random_noise, dim=-1) * eps
        const float4 nz = ld4(p.addend + (int64_t)row * D + c0);
        float ss = nz.x * nz.x + nz.y * nz.y + nz.z * nz.z + nz.w * nz.w;
        for (int off = 1; off < D / 4; off <<= 1) ss += __shfl_xor(ss, off);  // the D/4 lanes holding this row
        const float sc = p.denom / fmaxf(sqrtf(ss), 1e-12f);
        const float4 y = make_float4(fmaf(sgn(acc.x) * nz.x, sc, acc.x), fmaf(sgn(acc.y) * nz.y, sc, acc.y),
                                     fmaf(sgn(acc.z) * nz.z, sc, acc.z), fmaf(sgn(acc.w) * nz.w, sc, acc.w));
        st4_stream(p.y + (int64_t)row * p.ldy + c0, y, p.nt_store);
        return;
    }
    if (p.mode == MODE_MEAN) {
        if (p.partial) acc = add4(acc, ld4(p.partial + (int64_t)row * D + c0));  // same order as the accumulate epilogue
        // the layer outputs are read once here, in row order: streaming loads (opt "nt_store"), so that 2-3 x 18 MB of them do
        // not displace the gathered rows of X in L2 — except the last one when it IS the gathered table
        float4 s = ld4_stream(src_row(p.e0, row) + c0, p.nt_store);
        for (int i = 0; i < p.n_prev; ++i) {
            const float *pr = p.prev[i] + (int64_t)row * D + c0;
            s = add4(s, (p.prev[i] == p.x.p0) ? ld4(pr) : ld4_stream(pr, p.nt_store));
        }
        s = add4(s, acc);
        s = make_float4(s.x / p.denom, s.y / p.denom, s.z / p.denom, s.w / p.denom);
        // slab mode: the layers live as column slabs [2][N][D]; the mean leaves in the caller's row-major [N, 2 D]
        float *mo = p.slab ? p.mean_out + (int64_t)row * (2 * D) + (coff ? D : 0) + sl * 4 : p.mean_out + (int64_t)row * D + c0;
        st4_stream(mo, s, p.nt_store);
        if (p.y) st4_stream(p.y + (int64_t)row * p.ldy + c0, acc, p.nt_store);
    } else if (p.mode == MODE_HORNER) {
        float4 s = add4(ld4(p.addend + (int64_t)row * D + c0), acc);
        s = make_float4(s.x / p.denom, s.y / p.denom, s.z / p.denom, s.w / p.denom);
        st4_stream(p.y + (int64_t)row * p.ldy + c0, s, p.nt_store);
    } else {
        float *dst = p.y + (int64_t)row * p.ldy + c0;
        if (p.mode == MODE_ACCUM) {  // the row's old value is read once and rewritten: both streaming
            st4_stream(dst, add4(acc, ld4_stream(dst, p.nt_store)), p.nt_store);
        } else {
            st4_stream(dst, acc, p.nt_store);
        }
    }
}

// HALF (column-half mode, "col_split" option): the launch covers rows of width W = 2 D; workgroups on even XCDs own
// columns [0, D), on odd XCDs [D, 2 D), so an XCD's L2 holds half-width rows of one table (the per-XCD working set of
// the gather halves; the CSR is read twice).  Only for graphs without split rows (the host checks).
// (The DPP-broadcast + buffer-load gather of the r02 sweep kernel (devtools/experiments/sweep) was also tried here at D = 64: parity holds
// but the layer runs 43.0 us vs 39.5 us at the Gowalla shape — profiles/r02_binned_lean_gather.jsonl — because with one
// row per 16 lanes the plain gather already has its 8 loads in flight and the OOB-padded slots cost issue cycles.)
// SLAB (with HALF, option "slab"): the operands are stored as two column slabs [2][rows][D] instead of row-major [rows][2 D],
// so the half rows an XCD gathers are CONTIGUOUS 4 D-byte lines (r03: with row-major operands the half rows sit at a
// 8 D-byte stride and use every second L2 line slot only — the mode then reads 190 MB per layer at the Gowalla shape; as
// slabs 134 MB).  rbg_lightgcn_forward_f32 converts E0 once and keeps the layers as slabs; the mean leaves row-major.
template <int D, int U, bool CONTIG, bool HALF = false, bool SLAB = false>
__global__ __launch_bounds__(256) void spmm_binned_kernel(const SpmmParams p) {
    constexpr int LPR = D / 4;
    constexpr int SUBS = 64 / LPR;
    constexpr int W = (HALF && !SLAB) ? 2 * D : D;
    const int hoff = HALF ? (int)(blockIdx.x & 1) * D : 0;  // column offset of this half inside a full-width row
    const int64_t coff = SLAB ? (int64_t)(blockIdx.x & 1) * p.n_rows * D : hoff;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int sl = lane % LPR;
    // Row class of this workgroup.  Workgroup b is dispatched to XCD b % 8 (observed, used for speed only):
    // XCDs 0-3 take class 0 (user rows, gather item embeddings), XCDs 4-7 class 1, so an XCD's 4 MB L2 is
    // shared by ONE embedding table instead of two.
    int grp = 0, vb = blockIdx.x;
    if (p.n_groups > 1) {
        const int x = blockIdx.x & 7;
        grp = p.xmap.grp[x];
        vb = (blockIdx.x >> 3) * p.xmap.cnt[x] + p.xmap.idx[x];
    }
    const GroupPlan gp = p.grp[grp];
    const int blocks_wave = (gp.n_wave + 3) >> 2;

    if (vb < gp.n_tasks) {
        // ---- one workgroup per row segment ------------------------------------------------------
        __shared__ float red[4][D];
        __shared__ int last_flag;
        const int4 t0 = *reinterpret_cast<const int4 *>(&p.tasks[gp.task_base + vb]);
        const int4 t1 = *(reinterpret_cast<const int4 *>(&p.tasks[gp.task_base + vb]) + 1);
        const int row = t0.x, beg = t0.y, end = t0.z, seg = t0.w;
        const int nseg = t1.x, part_base = t1.y, ctr = t1.z;
        float4 acc = gather_range<D, U, CONTIG, W>(p, beg, end, wave * SUBS + sub, 4 * SUBS, sl, coff);
        acc = reduce_groups<D>(acc);
        if (sub == 0) st4(&red[wave][sl * 4], acc);
        __syncthreads();
        const bool owner = (wave == 0 && sub == 0);
        if (owner)
            acc = add4(add4(ld4(&red[0][sl * 4]), ld4(&red[1][sl * 4])),
                       add4(ld4(&red[2][sl * 4]), ld4(&red[3][sl * 4])));
        if (nseg == 1) {
            if (owner) finish_row(p, row, acc, sl, W, coff);
            return;
        }
        // Split row.  Publish this segment's partial sum WRITE-THROUGH (agent-scope relaxed stores = sc1, no
        // L2-flushing release fence), drain, then one lane bumps the row's arrival counter; the last segment
        // to arrive re-reads all partials with agent-scope loads (served past L1) and adds them in segment
        // order, so the result does not depend on arrival order (MI355X guide §6 G16, form R1).
        if (owner) {
            float *dst = p.partials + (int64_t)(part_base + seg) * kPartialSlotFloats + hoff + sl * 4;
            __hip_atomic_store(dst + 0, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 3, acc.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t *arrivals = p.counters + ctr + (hoff ? p.n_split : 0);
            const unsigned old = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == (unsigned)(nseg - 1));
            if (last)  // self-cleaning: the next (stream-ordered) launch finds the counter at zero
                __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = last;
        }
        __syncthreads();
        if (last_flag && owner) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = 0; i < nseg; ++i) {
                const float *src = p.partials + (int64_t)(part_base + i) * kPartialSlotFloats + hoff + sl * 4;
                float4 q;
                q.x = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.w = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s = add4(s, q);
            }
            finish_row(p, row, s, sl, W, coff);
        }
        return;
    }

    if (vb < gp.n_tasks + blocks_wave) {
        // ---- one wavefront per row --------------------------------------------------------------
        const int slot = (vb - gp.n_tasks) * 4 + wave;
        if (slot >= gp.n_wave) return;
        const int4 dsc = *reinterpret_cast<const int4 *>(&p.desc[gp.pos_wave + slot]);
        float4 acc = gather_range<D, U, CONTIG, W>(p, dsc.y, dsc.z, sub, SUBS, sl, coff);
        acc = reduce_groups<D>(acc);
        if (sub == 0) finish_row(p, dsc.x, acc, sl, W, coff);
        return;
    }

    // ---- one lane-group per row (degree-sorted, so the groups of a wave have similar lengths) ----
    const int slot = ((vb - gp.n_tasks - blocks_wave) * 4 + wave) * SUBS + sub;
    if (slot >= gp.n_short) return;
    const int4 dsc = *reinterpret_cast<const int4 *>(&p.desc[gp.pos_short + slot]);
    const float4 acc = gather_range<D, U, CONTIG, W>(p, dsc.y, dsc.z, 0, 1, sl, coff);
    finish_row(p, dsc.x, acc, sl, W, coff);
}

// (r02: a column-sweep kernel — persistent grid, LDS accumulators, entries walked column range by column range — lived here;
//  measured slower than this kernel on every shape (DESIGN results log 6.5) and moved out of the product in r05:
//  devtools/experiments/sweep/)


// Any d / any alignment: one wavefront per row, lanes stride the feature dimension.
__global__ __launch_bounds__(256) void spmm_generic_kernel(const SpmmParams p, int d) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.n_rows) return;
    const int beg = p.rowptr[row], end = p.rowptr[row + 1];
    float noise_scale = 0.f;
    if (p.mode == MODE_NOISE) {
        float ss = 0.f;
        for (int k = lane; k < d; k += 64) ss = fmaf(p.addend[(int64_t)row * d + k], p.addend[(int64_t)row * d + k], ss);
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        noise_scale = p.denom / fmaxf(sqrtf(ss), 1e-12f);
    }
    for (int k0 = 0; k0 < d; k0 += 64) {
        const int k = k0 + lane;
        float acc = 0.f;
        for (int e = beg; e < end; e += 64) {
            const int idx = e + lane;
            int c = 0;
            float v = 0.f;
            if (idx < end) {
                c = p.col[idx];
                v = p.val[idx];
            }
            const int cnt = min(64, end - e);
            for (int j = 0; j < cnt; ++j) {
                const int cj = __shfl(c, j, 64);
                const float vj = __shfl(v, j, 64);
                if (k < d) acc = fmaf(vj, src_row(p.x, cj)[k], acc);
            }
        }
        if (k < d) {
            if (p.mode == MODE_MEAN) {
                if (p.partial) acc += p.partial[(int64_t)row * d + k];
                float s = src_row(p.e0, row)[k];
                for (int i = 0; i < p.n_prev; ++i) s += p.prev[i][(int64_t)row * d + k];
                s += acc;
                p.mean_out[(int64_t)row * d + k] = s / p.denom;
                if (p.y) p.y[(int64_t)row * p.ldy + k] = acc;
            } else if (p.mode == MODE_NOISE) {
                p.y[(int64_t)row * p.ldy + k] = fmaf(sgn(acc) * p.addend[(int64_t)row * d + k], noise_scale, acc);
            } else if (p.mode == MODE_HORNER) {
                p.y[(int64_t)row * p.ldy + k] = (p.addend[(int64_t)row * d + k] + acc) / p.denom;
            } else {
                float *dst = p.y + (int64_t)row * p.ldy + k;
                *dst = (p.mode == MODE_ACCUM) ? (*dst + acc) : acc;
            }
        }
    }
}

// out = (e0 + sum_k layers[k]) / (K+1) — only for K beyond the fused range.
__global__ void layer_mean_kernel(RowSrc e0, const float *layers, int64_t n_rows, int d, int K, float *out) {
    const int64_t nd = n_rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / d), k = (int)(i % d);
        float s = src_row(e0, r)[k];
        for (int l = 0; l < K; ++l) s += layers[(int64_t)l * nd + i];
        out[i] = s / (float)(K + 1);
    }
}

// out = scale * (s[0] + s[1] + ... ), left to right (the layer mean of the sharded propagation: one launch instead of
// clone + K adds + one divide)
struct MeanSrcs {
    const float *s[RBG_MAX_FUSED_LAYERS + 1];
    int n;
};
__global__ void mean_kernel(const MeanSrcs m, int64_t len, float scale, float *__restrict__ out, int vec) {
    if (vec) {
        const int64_t q = len / 4;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < q; t += (int64_t)gridDim.x * blockDim.x) {
            float4 a = ld4(m.s[0] + 4 * t);
            for (int i = 1; i < m.n; ++i) a = add4(a, ld4(m.s[i] + 4 * t));
            st4(out + 4 * t, make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale));
        }
    } else {
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < len; t += (int64_t)gridDim.x * blockDim.x) {
            float a = m.s[0][t];
            for (int i = 1; i < m.n; ++i) a += m.s[i][t];
            out[t] = a * scale;
        }
    }
}

__global__ void gather_rows_kernel(const float *src, int64_t lds, const int64_t *idx, float *dst, int64_t n_idx,
                                   int d, int vec) {
    if (vec) {
        const int q = d / 4;
        const int64_t total = n_idx * q;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / q;
            const int k = (int)(t % q) * 4;
            st4(dst + i * d + k, ld4(src + idx[i] * lds + k));
        }
    } else {
        const int64_t total = n_idx * d;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / d;
            const int k = (int)(t % d);
            dst[i * d + k] = src[idx[i] * lds + k];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------

// Row source over one table (split = 0) or two tables (rows [0,split) in a, the rest in b).
static RowSrc make_src(const float *a, const float *b, int64_t split, int64_t ld) {
    RowSrc s;
    s.p0 = a ? a : b;
    s.p1 = (b ? b : a) - split * ld;  // pre-offset: row r >= split is p1 + r*ld
    s.split = (int32_t)split;
    s.pad = 0;
    s.ld = ld;
    return s;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static bool vec_ok(const RowSrc &s) { return aligned16(s.p0) && aligned16(s.p1) && (s.ld % 4) == 0; }

static void fill_graph(const rbg_graph *g, SpmmParams &p) {
    p.rowptr = g->d_rowptr;
    p.col = g->d_col;
    p.val = g->d_val;
    p.desc = g->d_desc;
    p.tasks = g->d_tasks;
    p.partials = g->d_partials;
    p.counters = g->d_counters;
    p.n_split = (int32_t)g->n_split_rows;
    p.n_rows = (int32_t)g->n_rows;
    p.n_groups = g->n_groups;
    p.xmap = g->xmap;
    for (int q = 0; q < kMaxGroups; ++q) p.grp[q] = g->groups[q];
    p.nt_store = opt_nt_store();
}

static int64_t group_blocks(const GroupPlan &gp, int subs) {
    return (int64_t)gp.n_tasks + (gp.n_wave + 3) / 4 + (gp.n_short + 4 * subs - 1) / (4 * subs);
}

template <int D>
static int64_t grid_for(const rbg_graph *g) {
    constexpr int SUBS = 64 / (D / 4);
    if (g->n_groups > 1) {
        int64_t m = 0;  // rounds of 8 workgroups: the slowest group decides
        for (int x = 0; x < 8; ++x) {
            const int64_t c = g->xmap.cnt[x];
            m = std::max(m, (group_blocks(g->groups[g->xmap.grp[x]], SUBS) + c - 1) / c);
        }
        return 8 * m;
    }
    return group_blocks(g->groups[0], SUBS);
}

// Column-half launch of a width-2H problem: XCD x serves group x / 4, column half x & 1, as XCD (x % 4) / 2 of 2.
template <int H>
static int launch_binned_half(const rbg_graph *g, SpmmParams &p, hipStream_t s, bool slab = false) {
    constexpr int SUBS = 64 / (H / 4);
    int64_t m = 0;
    for (int x = 0; x < 8; ++x) {
        p.xmap.grp[x] = (uint8_t)(x / 4);
        p.xmap.idx[x] = (uint8_t)((x % 4) / 2);
        p.xmap.cnt[x] = 2;
        m = std::max(m, (group_blocks(g->groups[x / 4], SUBS) + 1) / 2);
    }
    const int64_t grid = 8 * m;
    if (grid == 0) return RBG_OK;
    if (grid > INT32_MAX) return fail(RBG_EUNSUPPORTED, "grid too large");
    const dim3 gr((unsigned)grid), bl(256);
    if (slab) {
        if (spmm_unroll() == 8) hipLaunchKernelGGL((spmm_binned_kernel<H, 8, true, true, true>), gr, bl, 0, s, p);
        else hipLaunchKernelGGL((spmm_binned_kernel<H, 4, true, true, true>), gr, bl, 0, s, p);
    } else {
        if (spmm_unroll() == 8) hipLaunchKernelGGL((spmm_binned_kernel<H, 8, true, true>), gr, bl, 0, s, p);
        else hipLaunchKernelGGL((spmm_binned_kernel<H, 4, true, true>), gr, bl, 0, s, p);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// the default class split (user rows on XCDs 0-3, item rows on 4-7): what the column-half launches re-map
static bool half_structure_ok(const rbg_graph *g) {
    return g->n_groups == 2 && g->xmap.cnt[0] == 4 && g->xmap.cnt[7] == 4 && g->xmap.grp[0] == 0 && g->xmap.grp[3] == 0 &&
           g->xmap.grp[4] == 1 && g->xmap.grp[7] == 1;
}

// slab propagation (rbg_lightgcn_forward_f32 with scratch layers, option "slab"): width 64 or 128, square graph
static bool slab_eligible(const rbg_graph *g, int d) {
    return (d == 64 || d == 128) && half_structure_ok(g) && g->n_rows == g->n_cols &&
           (int64_t)g->n_rows * (d / 2) < ((int64_t)1 << 40);
}

// column-half eligibility of a launch of width D on this graph (see launch_binned)
static bool use_col_half(const rbg_graph *g, int D, int mode, int64_t ldx) {
    if (D != 64 && D != 128) return false;
    const int cs = opt_col_split();
    const bool cache_scale = (int64_t)g->n_rows * D * 4 <= ((int64_t)512 << 20);
    return mode != MODE_NOISE && (cs == 1 || (cs < 0 && D == 128 && cache_scale)) && g->n_groups == 2 && ldx == D &&
           g->xmap.cnt[0] == 4 && g->xmap.cnt[7] == 4 && g->xmap.grp[0] == 0 && g->xmap.grp[3] == 0 && g->xmap.grp[4] == 1 &&
           g->xmap.grp[7] == 1;
}

template <int D>
static int launch_binned(const rbg_graph *g, SpmmParams &p, hipStream_t s) {
    if constexpr (D == 64 || D == 128) {
        // default class split (users on XCDs 0-3, items on 4-7), contiguous rows: eligible for column halves
        // auto: only at d = 128 (measured r01, Gowalla shape: 86.4 -> 77.8 us; at d = 64 the doubled CSR / index work costs
        // more than the better L2 hit rate returns: 42.0 -> 49.8 us) and only while the table is cache-scale (<= 512 MB):
        // at the config-#5 shape (15 M rows, 7.7 GB) L2 residency is out of reach and reading the CSR twice loses 5 %
        // (31.9 -> 33.7 ms)
        // (MODE_NOISE is excluded: the noise row norm spans both halves)
        if (use_col_half(g, D, p.mode, p.x.ld)) return launch_binned_half<D / 2>(g, p, s);
    }
    const int64_t grid = grid_for<D>(g);
    if (grid == 0) return RBG_OK;
    if (grid > INT32_MAX) return fail(RBG_EUNSUPPORTED, "grid too large");
    const bool contig = (p.x.ld == D);
    const dim3 gr((unsigned)grid), bl(256);
    if (spmm_unroll() == 8) {
        if (contig) hipLaunchKernelGGL((spmm_binned_kernel<D, 8, true>), gr, bl, 0, s, p);
        else hipLaunchKernelGGL((spmm_binned_kernel<D, 8, false>), gr, bl, 0, s, p);
    } else {
        if (contig) hipLaunchKernelGGL((spmm_binned_kernel<D, 4, true>), gr, bl, 0, s, p);
        else hipLaunchKernelGGL((spmm_binned_kernel<D, 4, false>), gr, bl, 0, s, p);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// One SpMM launch with the epilogue described by p (graph fields are filled here).
static int launch_spmm(const rbg_graph *g, SpmmParams &p, int d, hipStream_t s) {
    fill_graph(g, p);
    if (g->n_rows == 0) return RBG_OK;
    bool vec = (d % 4 == 0) && vec_ok(p.x) && (p.ldy % 4 == 0) && (!p.y || aligned16(p.y));
    if (p.mode == MODE_HORNER || p.mode == MODE_NOISE) vec = vec && aligned16(p.addend);
    if (p.mode == MODE_MEAN) {
        vec = vec && vec_ok(p.e0) && aligned16(p.mean_out) && (!p.partial || aligned16(p.partial));
        for (int i = 0; i < p.n_prev; ++i) vec = vec && aligned16(p.prev[i]);
    }
    if (vec) {
        switch (d) {
            case 32: return launch_binned<32>(g, p, s);
            case 64: return launch_binned<64>(g, p, s);
            case 128: return launch_binned<128>(g, p, s);
            case 256: return launch_binned<256>(g, p, s);
            default: break;
        }
    }
    const int64_t grid = (g->n_rows + 3) / 4;
    hipLaunchKernelGGL(spmm_generic_kernel, dim3((unsigned)grid), dim3(256), 0, s, p, d);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// row-major rows of width 2 H (one array or the two embedding tables) -> column slabs [2][n][H]
__global__ __launch_bounds__(256) void to_slab_kernel(const RowSrc src, float *dst, int64_t n, int H) {
    const int q = (2 * H) / 4;
    const int64_t total = n * q;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / q;
        const int c4 = (int)(t % q) * 4;
        const int half = c4 >= H;
        st4(dst + (int64_t)half * n * H + row * H + (c4 - half * H), ld4(src_row(src, (int)row) + c4));
    }
}

// One layer over slab operands (p.x / p.y / p.e0 / p.prev in slab layout, row stride d / 2; p.mean_out row-major).
static int launch_slab(const rbg_graph *g, SpmmParams &p, int d, hipStream_t s) {
    fill_graph(g, p);
    p.slab = 1;
    if (d == 64) return launch_binned_half<32>(g, p, s, true);
    return launch_binned_half<64>(g, p, s, true);
}

int spmm_strided(const rbg_graph *g, const float *X, int64_t ldx, float *Y, int64_t ldy, int d, int accumulate,
                 hipStream_t s) {
    // contiguous operands on a planned handle: the column-slab kernel over row-major tables (every caller of the plain product
    // comes through here: rbg_spmm_f32, the NGCF layer and its backward, the sharded interior / halo blocks)
    if (ldy == d && sell_plain_applicable(g, d, ldx) && aligned16(X) && aligned16(Y) && g->n_rows > 0) {
        const int rc = sell_spmm(g, X, ldx, Y, d, accumulate, nullptr, 0.f, s);
        if (rc != RBG_EUNSUPPORTED) return rc;  // (no slab scratch available now: the binned kernel below)
    }
    SpmmParams p{};
    p.x = make_src(X, X, 0, ldx);
    p.y = Y;
    p.ldy = ldy;
    p.mode = accumulate ? MODE_ACCUM : MODE_STORE;
    return launch_spmm(g, p, d, s);
}

static int check_device_graph(const rbg_graph *g) {
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "operator called on a host graph (create it with device >= 0)");
    return RBG_OK;
}

}  // namespace rbg

using namespace rbg;

extern "C" {

static int csr_kernel_name(const rbg_graph *g, int d, char *buf, int len);

int rbg_spmm_kernel_name(const rbg_graph *g, int d, char *buf, int len) {
    clear_error();
    if (!g || !buf || len <= 0) return fail(RBG_EINVAL, "NULL argument");
    // mirrors rbg_spmm_f32 / launch_spmm for contiguous, 16-byte aligned fp32 operands in store mode
    if (sell_plain_applicable(g, d, d)) {
        snprintf(buf, (size_t)len, "%s", sell_kernel_name(g, d, false));
        return RBG_OK;
    }
    return csr_kernel_name(g, d, buf, len);
}

// the kernel over the CSR arrays (no plan, or a launch the plan does not serve)
static int csr_kernel_name(const rbg_graph *g, int d, char *buf, int len) {
    if (d != 32 && d != 64 && d != 128 && d != 256) {
        snprintf(buf, (size_t)len, "spmm_generic_kernel");
        return RBG_OK;
    }
    const bool half = use_col_half(g, d, MODE_STORE, d);
    snprintf(buf, (size_t)len, "spmm_binned_kernel<%d, %d, true, %s>", half ? d / 2 : d, spmm_unroll(), half ? "true" : "false");
    return RBG_OK;
}

int rbg_lightgcn_forward_kernel_name(const rbg_graph *g, int d, uint32_t flags, char *buf, int len) {
    clear_error();
    if (!g || !buf || len <= 0) return fail(RBG_EINVAL, "NULL argument");
    // mirrors rbg_lightgcn_forward_f32 for one graph and 16-byte aligned operands
    // (the slab chain of a factored plan: its launches after the first — the majority — read compact entries)
    if (sell_applicable(g, d) && (flags & RBG_FWD_LAYERS_SCRATCH) && !(flags & RBG_FWD_KEEP_LAST_LAYER)) {
        snprintf(buf, (size_t)len, "%s", sell_kernel_name(g, d, sell_chain_factored(g)));
        return RBG_OK;
    }
    if (sell_rowmajor_applicable(g, d) && !g->sell->rect) {
        snprintf(buf, (size_t)len, "%s", sell_kernel_name(g, d, false));
        return RBG_OK;
    }
    return csr_kernel_name(g, d, buf, len);  // (per-layer outputs without row-major entries: the CSR kernels)
}

int rbg_graph_bins(const rbg_graph *g, int d, int64_t *n_short, int64_t *n_wave, int64_t *n_block_tasks,
                   int64_t *n_split_rows, int64_t *grid_blocks) {
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (n_short) *n_short = g->n_short;
    if (n_wave) *n_wave = g->n_wave;
    if (n_block_tasks) *n_block_tasks = g->n_tasks;
    if (n_split_rows) *n_split_rows = g->n_split_rows;
    if (grid_blocks) {
        int64_t grid = (g->n_rows + 3) / 4;
        switch (d) {
            case 32: grid = grid_for<32>(g); break;
            case 64: grid = grid_for<64>(g); break;
            case 128: grid = grid_for<128>(g); break;
            case 256: grid = grid_for<256>(g); break;
            default: break;
        }
        *grid_blocks = grid;
    }
    return RBG_OK;
}

int rbg_spmm_f32(const rbg_graph *g, const float *X, float *Y, int d, int accumulate, void *stream) {
    clear_error();
    int rc = check_device_graph(g);
    if (rc) return rc;
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    if (g->n_rows == 0) return RBG_OK;
    if (!X || !Y) return fail(RBG_EINVAL, "X or Y is NULL");
    if (X == Y) return fail(RBG_EINVAL, "X and Y alias");
    if ((rc = set_device_for(g->device))) return rc;
    return spmm_strided(g, X, d, Y, d, d, accumulate, (hipStream_t)stream);
}

int rbg_mean_f32(const float *const *srcs, int n_srcs, int64_t n_floats, float scale, float *out, void *stream) {
    clear_error();
    if (!srcs || !out || n_srcs < 1 || n_srcs > RBG_MAX_FUSED_LAYERS + 1 || n_floats < 0)
        return fail(RBG_EINVAL, "rbg_mean_f32: 1 <= n_srcs <= %d", RBG_MAX_FUSED_LAYERS + 1);
    if (n_floats == 0) return RBG_OK;
    MeanSrcs m{};
    m.n = n_srcs;
    bool vec = (n_floats % 4 == 0) && aligned16(out);
    for (int i = 0; i < n_srcs; ++i) {
        if (!srcs[i]) return fail(RBG_EINVAL, "srcs[%d] is NULL", i);
        m.s[i] = srcs[i];
        vec = vec && aligned16(srcs[i]);
    }
    const int64_t work = vec ? n_floats / 4 : n_floats;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((work + 255) / 256, 1 << 16));
    hipLaunchKernelGGL(mean_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, m, n_floats, scale, out, vec ? 1 : 0);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_spmm_mean_f32(const rbg_graph *g, const float *X, const float *partial, const float *const *srcs, int n_srcs,
                      float *out_mean, int d, void *stream) {
    clear_error();
    int rc = check_device_graph(g);
    if (rc) return rc;
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    if (n_srcs < 1 || n_srcs > RBG_MAX_FUSED_LAYERS + 1) return fail(RBG_EINVAL, "1 <= n_srcs <= %d", RBG_MAX_FUSED_LAYERS + 1);
    if (g->n_rows == 0) return RBG_OK;
    if (!X || !srcs || !out_mean) return fail(RBG_EINVAL, "NULL pointer");
    for (int i = 0; i < n_srcs; ++i)
        if (!srcs[i]) return fail(RBG_EINVAL, "srcs[%d] is NULL", i);
    if (out_mean == X || out_mean == partial) return fail(RBG_EINVAL, "out_mean aliases an input");
    if ((rc = set_device_for(g->device))) return rc;
    if (sell_plain_applicable(g, d, d) && aligned16(X) && aligned16(out_mean) && (!partial || aligned16(partial))) {
        bool al = true;
        for (int i = 0; i < n_srcs; ++i) al = al && aligned16(srcs[i]);
        if (al) {
            rc = sell_spmm_mean(g, X, partial, srcs, n_srcs, out_mean, d, (float)(n_srcs + 1), (hipStream_t)stream);
            if (rc != RBG_EUNSUPPORTED) return rc;
        }
    }
    SpmmParams p{};
    p.x = make_src(X, X, 0, d);
    p.y = nullptr;
    p.ldy = d;
    p.mode = MODE_MEAN;
    p.mean_out = out_mean;
    p.partial = partial;
    p.e0 = make_src(srcs[0], srcs[0], 0, d);
    p.n_prev = n_srcs - 1;
    for (int i = 1; i < n_srcs; ++i) p.prev[i - 1] = srcs[i];
    p.denom = (float)(n_srcs + 1);
    return launch_spmm(g, p, d, (hipStream_t)stream);
}

int rbg_spmm_add_f32(const rbg_graph *g, const float *X, const float *Z, float *Y, int d, void *stream) {
    clear_error();
    int rc = check_device_graph(g);
    if (rc) return rc;
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    if (g->n_rows == 0) return RBG_OK;
    if (!X || !Z || !Y) return fail(RBG_EINVAL, "NULL pointer");
    if (Y == X || Y == Z) return fail(RBG_EINVAL, "Y aliases an input");
    if ((rc = set_device_for(g->device))) return rc;
    // the mean epilogue with one addend and no division: (Z + A X) / 1
    if (sell_plain_applicable(g, d, d) && aligned16(X) && aligned16(Y) && aligned16(Z)) {
        const float *srcs[1] = {Z};
        rc = sell_spmm_mean(g, X, nullptr, srcs, 1, Y, d, 1.0f, (hipStream_t)stream);
        if (rc != RBG_EUNSUPPORTED) return rc;
    }
    SpmmParams p{};
    p.x = make_src(X, X, 0, d);
    p.y = nullptr;
    p.ldy = d;
    p.mode = MODE_MEAN;
    p.mean_out = Y;
    p.partial = nullptr;
    p.e0 = make_src(Z, Z, 0, d);
    p.n_prev = 0;
    p.denom = 1.0f;
    return launch_spmm(g, p, d, (hipStream_t)stream);
}

int rbg_spmm_noise_f32(const rbg_graph *g, const float *X, float *Y, const float *noise, int d, float eps, void *stream) {
    clear_error();
    int rc = check_device_graph(g);
    if (rc) return rc;
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    if (g->n_rows == 0) return RBG_OK;
    if (!X || !Y || !noise) return fail(RBG_EINVAL, "NULL pointer");
    if (X == Y) return fail(RBG_EINVAL, "X and Y alias");
    if ((rc = set_device_for(g->device))) return rc;
    if (sell_plain_applicable(g, d, d) && aligned16(X) && aligned16(Y) && aligned16(noise)) {
        rc = sell_spmm(g, X, d, Y, d, 0, noise, eps, (hipStream_t)stream);
        if (rc != RBG_EUNSUPPORTED) return rc;
    }
    SpmmParams p{};
    p.x = make_src(X, X, 0, d);
    p.y = Y;
    p.ldy = d;
    p.mode = MODE_NOISE;
    p.addend = noise;
    p.denom = eps;
    return launch_spmm(g, p, d, (hipStream_t)stream);
}

int rbg_lightgcn_forward_f32(const rbg_graph *const *graphs, int n_graphs, int64_t n_users, const float *user_emb,
                             const float *item_emb, float *out_mean, float *layers, int d, int K, uint32_t flags,
                             void *stream) {
    clear_error();
    if (!graphs || n_graphs < 1) return fail(RBG_EINVAL, "graphs is NULL or empty");
    if (K < 0) return fail(RBG_EINVAL, "K = %d", K);
    if (n_graphs != 1 && n_graphs != K) return fail(RBG_ESHAPE, "n_graphs = %d but K = %d", n_graphs, K);
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    int rc;
    for (int i = 0; i < n_graphs; ++i) {
        if ((rc = check_device_graph(graphs[i]))) return rc;
        if (graphs[i]->n_rows != graphs[0]->n_rows || graphs[i]->n_cols != graphs[0]->n_rows ||
            graphs[i]->device != graphs[0]->device)
            return fail(RBG_ESHAPE, "graph %d is not a square graph of the same size/device as graph 0", i);
    }
    const rbg_graph *g0 = graphs[0];
    const int64_t n = g0->n_rows;
    if (g0->n_cols != n) return fail(RBG_ESHAPE, "graph is not square (%lld x %lld)", (long long)n, (long long)g0->n_cols);
    if (g0->n_users >= 0 && n_users != g0->n_users)
        return fail(RBG_ESHAPE, "n_users = %lld but the graph was built with %lld", (long long)n_users,
                    (long long)g0->n_users);
    if (n_users < 0 || n_users > n) return fail(RBG_ESHAPE, "n_users = %lld out of [0,%lld]", (long long)n_users, (long long)n);
    if (n == 0) return RBG_OK;
    if (!out_mean || (n_users > 0 && !user_emb) || (n_users < n && !item_emb)) return fail(RBG_EINVAL, "NULL embedding pointer");
    if (K > 1 && !layers) return fail(RBG_EINVAL, "layers buffer is NULL (needed for K > 1)");
    if ((flags & RBG_FWD_KEEP_LAST_LAYER) && K > 0 && !layers) return fail(RBG_EINVAL, "layers buffer is NULL");
    if ((rc = set_device_for(g0->device))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nd = n * d;
    const RowSrc e0 = make_src(user_emb, item_emb, n_users, d);
    if (K == 0) {  // mean of the single layer E0
        if (n_users) RBG_HIP(hipMemcpyAsync(out_mean, user_emb, sizeof(float) * n_users * d, hipMemcpyDeviceToDevice, s));
        if (n_users < n)
            RBG_HIP(hipMemcpyAsync(out_mean + n_users * d, item_emb, sizeof(float) * (n - n_users) * d,
                                   hipMemcpyDeviceToDevice, s));
        return RBG_OK;
    }
    const bool fused = (K - 1) <= RBG_MAX_FUSED_LAYERS;
    if (!fused && !layers) return fail(RBG_EINVAL, "layers buffer is NULL");
    // Slab propagation (option "slab"; the caller must not look at `layers`: RBG_FWD_LAYERS_SCRATCH): E0 is re-laid out as
    // two column slabs once, every layer gathers from and writes slabs (column-half kernel, SLAB instantiation), the
    // last layer's epilogue writes the mean row-major.  layers[K-1] holds E0's slabs, layers[k] layer k + 1's.
    // Column-slab propagation over an attached SELL plan (sell.hip): one graph for every layer (a plan carries its own row
    // numbering), the caller does not read `layers` (RBG_FWD_LAYERS_SCRATCH), the mean leaves row-major as always.
    // (a plan splits the rows at ITS boundary: a CSR-built handle does not pin n_users, so the caller's must agree)
    if (n_graphs == 1 && sell_applicable(g0, d) && g0->sell->n_class[0] == n_users && fused && layers && (flags & RBG_FWD_LAYERS_SCRATCH) &&
        !(flags & RBG_FWD_KEEP_LAST_LAYER) && aligned16(user_emb) && aligned16(item_emb) && aligned16(layers) && aligned16(out_mean))
        return sell_forward(g0, user_emb, item_emb, out_mean, layers, d, K, s);
    // the same plan with every layer row-major where the caller reads them (NCL, keep_layers)
    // ... and with one graph per layer (SGL's RW views, sgl.py:89-91: every plan has its own row numbering)
    bool all_rm = fused && aligned16(user_emb) && aligned16(item_emb) && aligned16(layers) && aligned16(out_mean);
    for (int i = 0; i < n_graphs && all_rm; ++i) all_rm = sell_rowmajor_applicable(graphs[i], d) && !graphs[i]->sell->rect && graphs[i]->sell->n_class[0] == n_users;
    if (all_rm) return sell_forward_rowmajor(graphs, n_graphs, user_emb, item_emb, out_mean, layers, d, K, (flags & RBG_FWD_KEEP_LAST_LAYER) != 0, s);
    bool slab = opt_slab() && fused && layers && (flags & RBG_FWD_LAYERS_SCRATCH) && !(flags & RBG_FWD_KEEP_LAST_LAYER) &&
                aligned16(user_emb) && aligned16(item_emb) && aligned16(layers) && aligned16(out_mean);
    for (int i = 0; i < n_graphs && slab; ++i) slab = slab_eligible(graphs[i], d);
    if (slab) {
        const int H = d / 2;
        float *e0s = layers + (int64_t)(K - 1) * nd;
        const int64_t work = n * (d / 4);
        hipLaunchKernelGGL(to_slab_kernel, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 8192)), dim3(256), 0, s, e0, e0s, n, H);
        RBG_HIP(hipGetLastError());
        for (int k = 0; k < K; ++k) {
            const rbg_graph *g = graphs[n_graphs == 1 ? 0 : k];
            SpmmParams p{};
            const float *x = (k == 0) ? e0s : layers + (int64_t)(k - 1) * nd;
            p.x = make_src(x, x, 0, H);
            p.ldy = H;
            if (k == K - 1) {
                p.mode = MODE_MEAN;
                p.y = nullptr;
                p.mean_out = out_mean;
                p.e0 = make_src(e0s, e0s, 0, H);
                p.n_prev = K - 1;
                for (int i = 0; i < K - 1; ++i) p.prev[i] = layers + (int64_t)i * nd;
                p.denom = (float)(K + 1);
            } else {
                p.mode = MODE_STORE;
                p.y = layers + (int64_t)k * nd;
            }
            if ((rc = launch_slab(g, p, d, s))) return rc;
        }
        return RBG_OK;
    }
    for (int k = 0; k < K; ++k) {
        const rbg_graph *g = graphs[n_graphs == 1 ? 0 : k];
        SpmmParams p{};
        if (k == 0)
            p.x = e0;
        else
            p.x = make_src(layers + (int64_t)(k - 1) * nd, layers + (int64_t)(k - 1) * nd, 0, d);
        p.ldy = d;
        const bool last = (k == K - 1);
        if (last && fused) {
            p.mode = MODE_MEAN;
            p.y = (flags & RBG_FWD_KEEP_LAST_LAYER) ? layers + (int64_t)k * nd : nullptr;
            p.mean_out = out_mean;
            p.e0 = e0;
            p.n_prev = K - 1;
            for (int i = 0; i < K - 1; ++i) p.prev[i] = layers + (int64_t)i * nd;
            p.denom = (float)(K + 1);
        } else {
            p.mode = MODE_STORE;
            p.y = layers + (int64_t)k * nd;
        }
        if ((rc = launch_spmm(g, p, d, s))) return rc;
    }
    if (!fused) {
        hipLaunchKernelGGL(layer_mean_kernel, dim3(2048), dim3(256), 0, s, e0, layers, n, d, K, out_mean);
        RBG_HIP(hipGetLastError());
    }
    return RBG_OK;
}

int rbg_lightgcn_backward_f32(const rbg_graph *const *graphs, int n_graphs, const float *grad_out, float *grad_e0,
                              float *work, int d, int K, void *stream) {
    clear_error();
    if (!graphs || n_graphs < 1) return fail(RBG_EINVAL, "graphs is NULL or empty");
    if (K < 0) return fail(RBG_EINVAL, "K = %d", K);
    if (n_graphs != 1 && n_graphs != K) return fail(RBG_ESHAPE, "n_graphs = %d but K = %d", n_graphs, K);
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    int rc;
    for (int i = 0; i < n_graphs; ++i) {
        if ((rc = check_device_graph(graphs[i]))) return rc;
        if (graphs[i]->n_rows != graphs[0]->n_rows || graphs[i]->n_cols != graphs[0]->n_rows ||
            graphs[i]->device != graphs[0]->device)
            return fail(RBG_ESHAPE, "graph %d is not a square graph of the same size/device as graph 0", i);
    }
    const int64_t n = graphs[0]->n_rows;
    if (n == 0) return RBG_OK;
    if (!grad_out || !grad_e0) return fail(RBG_EINVAL, "NULL gradient pointer");
    if (grad_out == grad_e0 || work == grad_out || (work && work == grad_e0)) return fail(RBG_EINVAL, "buffers alias");
    if (K >= 2 && !work) return fail(RBG_EINVAL, "work buffer is NULL (needed for K >= 2)");
    if ((rc = set_device_for(graphs[0]->device))) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (K == 0) {
        RBG_HIP(hipMemcpyAsync(grad_e0, grad_out, sizeof(float) * n * d, hipMemcpyDeviceToDevice, s));
        return RBG_OK;
    }
    // the column-slab chain (sell.hip) when the handle carries a plan: one graph, symmetric (the caller passes the transposed
    // handles — the handle itself for a graph built from interactions)
    if (n_graphs == 1 && sell_applicable(graphs[0], d) && aligned16(grad_out) && aligned16(grad_e0)) {
        rc = sell_backward(graphs[0], grad_out, grad_e0, work, d, K, s);
        if (rc != RBG_EUNSUPPORTED) return rc;
        clear_error();
    }
    // dE0 = (g + Â_0 (g + Â_1 (... (g + Â_{K-1} g)))) / (K+1): step i uses graph K-1-i; outputs ping-pong so that the
    // last one lands in grad_e0.
    const float *x = grad_out;
    for (int i = 0; i < K; ++i) {
        const rbg_graph *g = graphs[n_graphs == 1 ? 0 : K - 1 - i];
        float *y = ((K - 1 - i) % 2 == 0) ? grad_e0 : work;
        SpmmParams p{};
        p.x = make_src(x, x, 0, d);
        p.y = y;
        p.ldy = d;
        p.mode = MODE_HORNER;
        p.addend = grad_out;
        p.denom = (i == K - 1) ? (float)(K + 1) : 1.0f;
        if ((rc = launch_spmm(g, p, d, s))) return rc;
        x = y;
    }
    return RBG_OK;
}

int rbg_gather_rows_f32(const float *src, int64_t lds, const int64_t *idx, float *dst, int64_t n_idx, int d,
                        void *stream) {
    clear_error();
    if (n_idx < 0 || d <= 0 || lds < d) return fail(RBG_ESHAPE, "n_idx = %lld, d = %d, lds = %lld", (long long)n_idx, d, (long long)lds);
    if (n_idx == 0) return RBG_OK;
    if (!src || !idx || !dst) return fail(RBG_EINVAL, "NULL pointer");
    const int vec = (d % 4 == 0) && (lds % 4 == 0) && aligned16(src) && aligned16(dst);
    const int64_t work = n_idx * (vec ? d / 4 : d);
    const int64_t grid = std::min<int64_t>((work + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, lds, idx, dst,
                       n_idx, d, vec);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

// ---- one layer of the node-range sharded propagation, host side in two calls ------------------------------------------
// The N > 1 path is host-bound (a propagation is ~190 us of GPU work but cost ~320 us of Python: every layer made ~10
// torch / ctypes calls, r01 world-size-1 RCCL probe).  begin() and end() take BOTH raw stream handles and do the
// cross-stream ordering with two events of their own, so that a layer is: begin, the caller's all_to_all on the comm
// stream, end.
struct rbg_shard_ctx {
    int device;
    hipEvent_t x_ready, halo_ready;
};

int rbg_shard_ctx_create(rbg_shard_ctx **out, int device) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    int rc = set_device_for(device);
    if (rc) return rc;
    rbg_shard_ctx *c = new (std::nothrow) rbg_shard_ctx{device, nullptr, nullptr};
    if (!c) return fail(RBG_ENOMEM, "out of host memory");
    if (hipEventCreateWithFlags(&c->x_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail(RBG_EHIP, "hipEventCreate failed");
    }
    *out = c;
    return RBG_OK;
}

void rbg_shard_ctx_destroy(rbg_shard_ctx *c) {
    if (!c) return;
    (void)hipEventDestroy(c->x_ready);
    (void)hipEventDestroy(c->halo_ready);
    delete c;
}

// comm stream: wait until X is ready on the main stream, then pack send_buf[i] = X[send_idx[i]]  (n_send may be 0);
// main stream: Y = A_interior X.  The caller then enqueues its collective on the comm stream.
int rbg_shard_layer_begin(rbg_shard_ctx *c, const rbg_graph *g_int, const float *X, float *Y, const int64_t *send_idx,
                          int64_t n_send, float *send_buf, int d, void *main_stream, void *comm_stream) {
    clear_error();
    if (!c) return fail(RBG_EINVAL, "ctx is NULL");
    int rc = check_device_graph(g_int);
    if (rc) return rc;
    if (d <= 0 || n_send < 0) return fail(RBG_ESHAPE, "d = %d, n_send = %lld", d, (long long)n_send);
    if (!X || !Y || (n_send && (!send_idx || !send_buf))) return fail(RBG_EINVAL, "NULL pointer");
    if ((rc = set_device_for(g_int->device))) return rc;
    hipStream_t ms = (hipStream_t)main_stream, cs = (hipStream_t)comm_stream;
    RBG_HIP(hipEventRecord(c->x_ready, ms));
    RBG_HIP(hipStreamWaitEvent(cs, c->x_ready, 0));
    if (n_send) {
        const int vec = (d % 4 == 0) && aligned16(X) && aligned16(send_buf);
        const int64_t work = n_send * (vec ? d / 4 : d);
        const int64_t grid = std::min<int64_t>((work + 255) / 256, 4096);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)grid), dim3(256), 0, cs, X, (int64_t)d, send_idx, send_buf, n_send, d, vec);
        RBG_HIP(hipGetLastError());
    }
    if (g_int->n_rows == 0) return RBG_OK;
    return spmm_strided(g_int, X, d, Y, d, d, 0, ms);
}

// main stream: wait for the comm stream (the collective filled `halo`), then Y += A_halo halo  (g_halo may be NULL).
int rbg_shard_layer_end(rbg_shard_ctx *c, const rbg_graph *g_halo, const float *halo, float *Y, int d, void *main_stream,
                        void *comm_stream) {
    clear_error();
    if (!c) return fail(RBG_EINVAL, "ctx is NULL");
    hipStream_t ms = (hipStream_t)main_stream, cs = (hipStream_t)comm_stream;
    RBG_HIP(hipEventRecord(c->halo_ready, cs));
    RBG_HIP(hipStreamWaitEvent(ms, c->halo_ready, 0));
    if (!g_halo) return RBG_OK;
    int rc = check_device_graph(g_halo);
    if (rc) return rc;
    if (d <= 0) return fail(RBG_ESHAPE, "d = %d", d);
    if (!halo || !Y) return fail(RBG_EINVAL, "NULL pointer");
    if (g_halo->n_rows == 0) return RBG_OK;
    return spmm_strided(g_halo, halo, d, Y, d, d, 1, ms);
}

}  // extern "C"
