// slab.hip — prototype of the column-slab SpMM (r03).  Stand-alone: built and driven by devtools/slab/run.py.
//
// Idea: the binned kernel is bound by L2 misses because an XCD's 4 MB L2 cannot hold the embedding table its rows
// gather from (10.5 MB of item rows at the Gowalla shape).  Here every XCD owns ONE column slab of ONE table: with W = 16
// columns per slab the item slab is 2.6 MB, so after the first touch every gather is an L2 hit, and the hottest rows of the
// slab can sit in the CU's LDS.  Nodes are renumbered per class in processing order (hot rows first), the dense operand is
// kept in slab layout [slab][row][W] between the layers, the CSR is read once per slab.
//
// The sparse operand is SELL-C-sigma over lane-groups (devtools/slab/plan.py): a unit = the LGW = 64 / (W/4) lane-groups of
// one wave; its rows are consecutive (or the parts of a split row in adjacent lane-groups, added with a butterfly in fixed
// order: bit-stable, no float atomics); its entries are stored unit-major, padded to the unit's longest piece, so a batch
// of 8 slots per lane-group is ONE wave-wide 16-byte load and no lane masks anything.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

enum { SLAB_STORE = 0, SLAB_MEAN = 1 };

struct SlabParams {
    const v4i *ent;         // pairs of entries {col * W * 4, bits of val}
    const int4 *head;       // unit headers {first entry, first row, nh | nc << 16, log2(parts) | rows << 8}
    int32_t unit_base[2];   // first unit of class c
    int32_t n_units[2];
    const float *xs;        // gathered operand, slab layout
    float *ys;              // result, slab layout (SLAB_STORE)
    int64_t slab_off[2][4]; // float offset of (class, slab)
    int32_t n_class[2];     // rows of class c
    int32_t hot_rows;       // rows of a slab kept in LDS
    int32_t mode;
    // SLAB_MEAN: out[orig[row]][slab cols] = (sum_i prev[i] + acc) / denom, prev in slab layout, out row-major [N, 64]
    const float *prev[4];
    int32_t n_prev;
    float denom;
    float *out;
    const int32_t *orig;    // [n_class[0] + n_class[1]] original node id of (class, internal row)
    int32_t nt_ent;         // entries with non-temporal loads
    int32_t pad;
    unsigned long long *trace;  // optional [n_waves][6]: start, header, staged, end (s_memtime), batches, xcd | unit << 8
};

template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // lane K of every quad, in all its lanes
    return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);
}
// entry J (0..7) of the 8 a quad holds: lane J / 2, components (J & 1) * 2 + {0, 1}
template <int J>
__device__ __forceinline__ int ent_col(const v4i &w) { return quad_bcast<J / 2>((J & 1) ? w.z : w.x); }
template <int J>
__device__ __forceinline__ float ent_val(const v4i &w) { return __int_as_float(quad_bcast<J / 2>((J & 1) ? w.w : w.y)); }

template <int J, int N>
struct StaticFor {
    template <class F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, J>{});
        StaticFor<J + 1, N>::run(f);
    }
};
template <int N>
struct StaticFor<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F &&) {}
};

struct Acc {
    v2f lo, hi;
};
__device__ __forceinline__ void fma_row(Acc &a, float v, v4f x) {
    const v2f vv = {v, v};
    a.lo = __builtin_elementwise_fma(vv, __builtin_shufflevector(x, x, 0, 1), a.lo);
    a.hi = __builtin_elementwise_fma(vv, __builtin_shufflevector(x, x, 2, 3), a.hi);
}

template <int W, bool HOT>
__global__ __launch_bounds__(1024) void slab_spmm_kernel(const SlabParams p) {
    extern __shared__ __align__(16) float lds[];
    constexpr int G = W / 4;        // lanes per lane-group
    constexpr int LGW = 64 / G;     // lane-groups per wave = pieces per unit
    constexpr int NS = 64 / W;      // slabs per table
    constexpr int XPR = 4 / NS;     // XCDs per role (class, slab)
    const int x = blockIdx.x & 7, cls = x >> 2, s = (x & 3) % NS;
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G, q4 = lane & 3;
    const float *xtab = p.xs + p.slab_off[1 - cls][s];
    const int n_tab = p.n_class[1 - cls];
    // wave w of the role's n_w waves takes units w, w + n_w, ...: one unit per wave when the grid covers the units (the
    // hardware dispatcher then balances the load, heaviest units first), several in a persistent grid
    const unsigned wpb = blockDim.x >> 6;
    const unsigned n_w = (gridDim.x >> 3) * XPR * wpb;
    const unsigned w0 = ((blockIdx.x >> 3) * XPR + ((x & 3) / NS)) * wpb + (threadIdx.x >> 6);
    if (HOT) {
        const int n4 = min(p.hot_rows, n_tab) * (W / 4);
        for (int i = threadIdx.x; i < n4; i += blockDim.x)
            reinterpret_cast<float4 *>(lds)[i] = reinterpret_cast<const float4 *>(xtab)[i];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xtab), 0, n_tab * W * 4, 0x00020000);
    const int lane_off = sl * 16;
    const unsigned nun = (unsigned)p.n_units[cls];
    const int4 *heads = p.head + p.unit_base[cls];
    const int64_t ybase = p.slab_off[cls][s];
    auto load_pair = [&](const v4i *q) -> v4i { return p.nt_ent ? __builtin_nontemporal_load(q) : *q; };

    for (unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)w0); t < nun; t += n_w) {
        const int4 h = heads[t];
        const int row0 = h.y, nh = HOT ? (h.z & 0xffff) : 0, nc = h.z >> 16, lp = h.w & 0xff, nrows = (h.w >> 8) & 0xff;
        const bool wide = (h.w >> 16) & 1;  // the row spans the 4 waves of this workgroup (uniform per workgroup: the plan aligns it)
        const v4i *eb = p.ent + (h.x >> 1);
        Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
        // One section (hot: LDS, cold: L2) = `len` slots per lane-group in batches of 8 (the last one of len % 8, even).
        // The pair of entries a lane holds for batch k sits at base + (LGW k) / 2 + lg (sb / 2) + q4.
        auto section = [&](const v4i *base, int len, auto is_lds) __attribute__((always_inline)) {
            constexpr bool kLds = decltype(is_lds)::value;
            if (len <= 0) return;
            int sb = min(8, len);
            v4i w = {0, 0, 0, 0};
            if (2 * q4 < sb) w = load_pair(base + lg * (sb >> 1) + q4);
            for (int k = 0; k < len; k += 8) {
                const int sbn = min(8, len - k - 8);  // slots of the next batch (<= 0: none)
                v4i wn = {0, 0, 0, 0};
                auto batch = [&](auto nc_) __attribute__((always_inline)) {
                    constexpr int n = decltype(nc_)::value;
                    v4f xv[n];
                    StaticFor<0, n>::run([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const int off = ent_col<j>(w) + lane_off;
                        if (kLds) xv[j] = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(lds) + off);
                        else xv[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                    });
                    if (sbn > 0 && 2 * q4 < sbn) wn = load_pair(base + ((LGW * (k + 8)) >> 1) + lg * (sbn >> 1) + q4);
                    StaticFor<0, n>::run([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        fma_row(acc, ent_val<j>(w), xv[j]);
                    });
                };
                if (sb == 8) batch(std::integral_constant<int, 8>{});
                else if (sb == 6) batch(std::integral_constant<int, 6>{});
                else if (sb == 4) batch(std::integral_constant<int, 4>{});
                else batch(std::integral_constant<int, 2>{});
                w = wn;
                sb = sbn;
            }
        };
        if (HOT) section(eb, nh, std::true_type{});
        section(eb + ((LGW * nh) >> 1), nc, std::false_type{});
        // ---- parts of a split row: butterfly over the adjacent lane-groups (fixed order) ----------------------------------------
        const int parts = 1 << lp;
        if (lp > 0) {
#pragma unroll
            for (int off = 1; off < LGW; off <<= 1) {
                const float a0 = __shfl_xor(acc.lo.x, off * G), a1 = __shfl_xor(acc.lo.y, off * G);
                const float a2 = __shfl_xor(acc.hi.x, off * G), a3 = __shfl_xor(acc.hi.y, off * G);
                if (off < parts) { acc.lo.x += a0; acc.lo.y += a1; acc.hi.x += a2; acc.hi.y += a3; }
            }
        }
        if (wide) {  // 4 waves x LGW parts of ONE row: per-wave partials through LDS, added in wave order
            __shared__ float s_wide[4][W];
            const int wv = (threadIdx.x >> 6) & 3;
            if (lg == 0) *reinterpret_cast<float4 *>(&s_wide[wv][sl * 4]) = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
            __syncthreads();
            if (wv == 0 && lg == 0) {
                float4 tsum = *reinterpret_cast<const float4 *>(&s_wide[0][sl * 4]);
                for (int q = 1; q < 4; ++q) {
                    const float4 o4 = *reinterpret_cast<const float4 *>(&s_wide[q][sl * 4]);
                    tsum.x += o4.x; tsum.y += o4.y; tsum.z += o4.z; tsum.w += o4.w;
                }
                acc.lo.x = tsum.x; acc.lo.y = tsum.y; acc.hi.x = tsum.z; acc.hi.y = tsum.w;
            }
            __syncthreads();
        }
        const int r = lg >> lp;
        if ((lg & (parts - 1)) == 0 && r < nrows && (!wide || ((threadIdx.x >> 6) & 3) == 0)) {
            const int row = row0 + r;
            const int64_t o = ybase + (int64_t)row * W + sl * 4;
            if (p.mode == SLAB_MEAN) {
                float4 sum = *reinterpret_cast<const float4 *>(p.prev[0] + o);
                for (int i = 1; i < p.n_prev; ++i) {
                    const float4 q = *reinterpret_cast<const float4 *>(p.prev[i] + o);
                    sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
                }
                sum.x = (sum.x + acc.lo.x) / p.denom; sum.y = (sum.y + acc.lo.y) / p.denom;
                sum.z = (sum.z + acc.hi.x) / p.denom; sum.w = (sum.w + acc.hi.y) / p.denom;
                const int node = p.orig[(cls ? p.n_class[0] : 0) + row];
                *reinterpret_cast<float4 *>(p.out + (int64_t)node * 64 + s * W + sl * 4) = sum;
            } else {
                *reinterpret_cast<float4 *>(p.ys + o) = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
            }
        }
    }
}

// ---- v3: the unit's entries staged in LDS by DMA ------------------------------------------------------------------------------
// PMC (r03): the gathers are short (300-430 cycles from the L2) but a wave's life is dominated by two dependent DRAM-latency
// reads, header -> entries, and then one more per batch when the entries are fetched batch by batch.  Here the whole entry
// block of the unit (contiguous: batches of the hot section, then of the cold section) is copied to LDS with
// global_load_lds_dwordx4 requests issued back to back right after the header arrives: one exposed latency per unit.
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

template <int W, int BUF>
__global__ __launch_bounds__(256) void slab_dma_kernel(const SlabParams p) {
    constexpr int G = W / 4, LGW = 64 / G, NS = 64 / W, XPR = 4 / NS;
    constexpr int WBUF = BUF + 64;  // bytes of LDS per wave (+ slack: lanes past a partial batch read, and ignore, 48 B more)
    __shared__ __align__(16) char stage[4 * WBUF];
    const int x = blockIdx.x & 7, cls = x >> 2, s = (x & 3) % NS;
    const int lane = threadIdx.x & 63, lg = lane / G, sl = lane % G, q4 = lane & 3, wave = threadIdx.x >> 6;
    const float *xtab = p.xs + p.slab_off[1 - cls][s];
    const int n_tab = p.n_class[1 - cls];
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x >> 3) * XPR + ((x & 3) / NS)) * 4 + wave));
    if (t >= (unsigned)p.n_units[cls]) return;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (p.trace) tr0 = __builtin_readcyclecounter();
    const int4 h = p.head[p.unit_base[cls] + t];
    const int row0 = h.y, nh = h.z & 0xffff, nc = h.z >> 16, lp = h.w & 0xff, nrows = h.w >> 8;
    if (p.trace) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr1 = __builtin_readcyclecounter() + (row0 & 0); }
    const char *ub = reinterpret_cast<const char *>(p.ent) + (int64_t)h.x * 8;
    const int total = LGW * (nh + nc) * 8;  // bytes of the unit's entry block
    const int nbh = (nh + 7) >> 3, nb = nbh + ((nc + 7) >> 3);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xtab), 0, n_tab * W * 4, 0x00020000);
    const int lane_off = sl * 16;
    char *mybuf = stage + wave * WBUF;
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)mybuf);  // LDS byte address (wave-uniform)
    Acc acc = {{0.f, 0.f}, {0.f, 0.f}};
    int pos = 0, b = 0;
    while (pos < total) {
        const int len = min(BUF, total - pos);
        for (int pc = 0; pc * 1024 < len; ++pc) lds_dma16(ub + pos + pc * 1024 + lane * 16, lds_base + pc * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.trace && pos == 0) tr2 = __builtin_readcyclecounter();
        int loc = 0;
        while (b < nb) {
            const bool cold = b >= nbh;
            const int k = cold ? b - nbh : b;
            const int sb = min(8, (cold ? nc : nh) - 8 * k);
            const int bytes = LGW * sb * 8;
            if (loc + bytes > len) break;
            const v4i w = *reinterpret_cast<const v4i *>(mybuf + loc + (lg * (sb >> 1) + q4) * 16);
            auto batch = [&](auto nc_) __attribute__((always_inline)) {
                constexpr int n = decltype(nc_)::value;
                v4f xv[n];
                StaticFor<0, n>::run([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    xv[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, ent_col<j>(w) + lane_off, 0, 0));
                });
                StaticFor<0, n>::run([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    fma_row(acc, ent_val<j>(w), xv[j]);
                });
            };
            // (no LDS tile in this kernel: the hot section, if the plan has one, is gathered from the L2 like the cold one)
            if (sb == 8) batch(std::integral_constant<int, 8>{});
            else if (sb == 6) batch(std::integral_constant<int, 6>{});
            else if (sb == 4) batch(std::integral_constant<int, 4>{});
            else batch(std::integral_constant<int, 2>{});
            loc += bytes;
            ++b;
        }
        pos += loc;
    }
    const int parts = 1 << lp;
    if (lp > 0) {
#pragma unroll
        for (int off = 1; off < LGW; off <<= 1) {
            const float a0 = __shfl_xor(acc.lo.x, off * G), a1 = __shfl_xor(acc.lo.y, off * G);
            const float a2 = __shfl_xor(acc.hi.x, off * G), a3 = __shfl_xor(acc.hi.y, off * G);
            if (off < parts) { acc.lo.x += a0; acc.lo.y += a1; acc.hi.x += a2; acc.hi.y += a3; }
        }
    }
    const int r = lg >> lp;
    if ((lg & (parts - 1)) == 0 && r < nrows) {
        const int row = row0 + r;
        const int64_t o = p.slab_off[cls][s] + (int64_t)row * W + sl * 4;
        if (p.mode == SLAB_MEAN) {
            float4 sum = *reinterpret_cast<const float4 *>(p.prev[0] + o);
            for (int i = 1; i < p.n_prev; ++i) {
                const float4 q = *reinterpret_cast<const float4 *>(p.prev[i] + o);
                sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
            }
            sum.x = (sum.x + acc.lo.x) / p.denom; sum.y = (sum.y + acc.lo.y) / p.denom;
            sum.z = (sum.z + acc.hi.x) / p.denom; sum.w = (sum.w + acc.hi.y) / p.denom;
            const int node = p.orig[(cls ? p.n_class[0] : 0) + row];
            *reinterpret_cast<float4 *>(p.out + (int64_t)node * 64 + s * W + sl * 4) = sum;
        } else {
            *reinterpret_cast<float4 *>(p.ys + o) = make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y);
        }
    }
    if (p.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *q = p.trace + (int64_t)(blockIdx.x * 4 + wave) * 6;
        q[0] = tr0; q[1] = tr1; q[2] = tr2; q[3] = __builtin_readcyclecounter(); q[4] = nb; q[5] = (unsigned)x | ((unsigned long long)t << 8);
    }
}

// row-major [N, 64] in original numbering -> slab layout in internal numbering (and back)
template <int W>
__global__ __launch_bounds__(256) void to_slab_kernel(const float *src, float *dst, const int32_t *orig, int n0, int n1,
                                                      int64_t off00, int64_t slab_stride0, int64_t off10, int64_t slab_stride1, int back) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4, sl = threadIdx.x & 15;
    if (g >= n0 + n1) return;
    const int cls = g >= n0, row = cls ? g - n0 : g;
    const int node = orig[g];
    const int c0 = sl * 4, s = c0 / W, w = c0 % W;
    const int64_t so = (cls ? off10 + s * slab_stride1 : off00 + s * slab_stride0) + (int64_t)row * W + w;
    if (back) *reinterpret_cast<float4 *>(dst + (int64_t)node * 64 + c0) = *reinterpret_cast<const float4 *>(src + so);
    else *reinterpret_cast<float4 *>(dst + so) = *reinterpret_cast<const float4 *>(src + (int64_t)node * 64 + c0);
}

extern "C" {

int slab_convert(const float *src, float *dst, const int32_t *orig, int n0, int n1, const int64_t *slab_off, int W, int back, void *stream) {
    const int n = n0 + n1;
    const dim3 grid((n * 16 + 255) / 256), bl(256);
    hipStream_t s = (hipStream_t)stream;
    const int ns = 64 / W;
    const int64_t st0 = ns > 1 ? slab_off[1] - slab_off[0] : 0, st1 = ns > 1 ? slab_off[5] - slab_off[4] : 0;
    if (W == 16) hipLaunchKernelGGL(to_slab_kernel<16>, grid, bl, 0, s, src, dst, orig, n0, n1, slab_off[0], st0, slab_off[4], st1, back);
    else if (W == 32) hipLaunchKernelGGL(to_slab_kernel<32>, grid, bl, 0, s, src, dst, orig, n0, n1, slab_off[0], st0, slab_off[4], st1, back);
    else hipLaunchKernelGGL(to_slab_kernel<64>, grid, bl, 0, s, src, dst, orig, n0, n1, slab_off[0], st0, slab_off[4], st1, back);
    return (int)hipGetLastError();
}

int slab_spmm(const SlabParams *pp, int W, int hot, int n_wg, int tpb, void *stream) {
    const SlabParams p = *pp;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = hot ? (size_t)p.hot_rows * W * 4 : 0;
#define LAUNCH(WW, HH)                                                                                                        \
    {                                                                                                                         \
        auto k = slab_spmm_kernel<WW, HH>;                                                                                    \
        if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(tpb), lds, s, p);                                                              \
    }
    if (W == 16) { if (hot) LAUNCH(16, true) else LAUNCH(16, false) }
    else if (W == 32) { if (hot) LAUNCH(32, true) else LAUNCH(32, false) }
    else { if (hot) LAUNCH(64, true) else LAUNCH(64, false) }
    return (int)hipGetLastError();
}

int slab_spmm_dma(const SlabParams *pp, int W, int buf, int n_wg, void *stream) {
    const SlabParams p = *pp;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(n_wg), b(256);
    if (W == 16 && buf == 4096) hipLaunchKernelGGL((slab_dma_kernel<16, 4096>), g, b, 0, s, p);
    else if (W == 16 && buf == 8192) hipLaunchKernelGGL((slab_dma_kernel<16, 8192>), g, b, 0, s, p);
    else if (W == 32 && buf == 4096) hipLaunchKernelGGL((slab_dma_kernel<32, 4096>), g, b, 0, s, p);
    else if (W == 32 && buf == 8192) hipLaunchKernelGGL((slab_dma_kernel<32, 8192>), g, b, 0, s, p);
    else if (W == 32 && buf == 2048) hipLaunchKernelGGL((slab_dma_kernel<32, 2048>), g, b, 0, s, p);
    else if (W == 64 && buf == 4096) hipLaunchKernelGGL((slab_dma_kernel<64, 4096>), g, b, 0, s, p);
    else return -1;
    return (int)hipGetLastError();
}

}  // extern "C"
