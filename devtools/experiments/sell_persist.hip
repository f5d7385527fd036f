// NOT BUILT (r04 experiment, kept for the record: DESIGN 6.10, profiles/r04_persist_probe.jsonl).  It compiled against
// recbole-gnn_amd/csrc at commit "persistent K-layer launch: measured negative" with SellDev::sync / persist_ok and option "sell_persist".

// sell_persist.hip — the K layers of a propagation chain (lightgcn.py:70-81 / the backward Horner chain) in ONE launch of
// the column-slab kernel (r04).
//
// Why.  With every gather of two of the three layers reading nothing (probe "sell_skip_hot", profiles/r04_skip_hot_probe.jsonl)
// the Gowalla propagation only falls from 92.5 to 81.0 us: a layer is bound by the address units' issue rate (a 1 KB wave-load
// costs them 16 cycles, hit or miss: 15.3 us for the 526 MB of gathers) plus ~10 us per LAUNCH in which they idle — the dependent
// launch itself (3.3 us), the ramp (a new wave walks header -> entries -> first gathers, ~2.5 us before the address units fill)
// and the tail (the last, lightest units are pure latency chains).  One launch for the K layers removes K - 1 of those:
//   * a persistent grid: 8 workgroups of 4 waves per CU (the same residency as the per-layer kernel), workgroup b on XCD b & 7
//     with the same (class, slab) roles;
//   * units are handed out by TICKETS, a group of four units (= the four waves of a workgroup, the granularity at which the
//     hardware dispatcher recycled wave slots as well; a wide row is one such group) per draw: per (layer, XCD) one counter
//     bumped by L2-local atomics (workgroup scope: every requester of a counter sits on the XCD whose L2 performs the atomic —
//     r03 measured ONE device-scope counter at ~100 ns per contended draw, which serialised the launch); heaviest groups
//     first, as the hardware dispatcher took them; the next ticket is drawn before the gathers of the current group start;
//   * a layer barrier per XCD then across XCDs: workgroups count in on an L2-local counter; the last one of an XCD releases (one
//     L2 write-back), bumps the one device-scope counter (8 bumps per layer), waits for the other seven, acquires (one L2
//     invalidate) and raises its XCD's flag, which the other workgroups of the XCD poll;
//   * all counters are reset by the last arriver, so the state is zero again when the launch ends (no memset node per replay).
// The summation order of every row is the plan's, as before: bit-identical to the per-layer launches.
// The grid must be co-resident (2 048 workgroups = the chip's capacity at <= 64 VGPRs); a workgroup that cannot be placed yet
// (another kernel holds the CU) is waited for by the spinning ones, which is safe as long as that kernel does not wait on this one.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "internal.h"
#include "sell_kernel.h"

namespace rbg {

constexpr int kPersistStride = 32;            // ints between counters (one 128-byte line each)
constexpr int kPersistSlots = 2;              // counters per (layer, XCD): 0 group tickets, 1 arrivals
constexpr int kPersistLayers = RBG_MAX_FUSED_LAYERS + 1;
constexpr int kPersistGlobal = kPersistLayers * 8 * kPersistSlots * kPersistStride;  // the device-scope layer counter
constexpr int kPersistGo = kPersistGlobal + kPersistStride;                            // per XCD: layers complete everywhere
constexpr int kPersistInts = kPersistGo + 8 * kPersistStride;

struct SellChainParams {
    SellParams base;  // FIRST: the device code reads it in place (sell_kernarg)
    SellChainLayer layer[kPersistLayers];
    int32_t *sync;    // [kPersistInts], zero between launches
    int32_t K;
    int32_t fences;   // bit 0 release (L2 write-back), bit 1 acquire (L2 invalidate), bit 2 L1 invalidate: 7 = correct; less = diagnostic
};
typedef const __attribute__((address_space(4))) SellChainParams SellChainParamsK;

__device__ __forceinline__ int ticket_add(int32_t *p) {  // performed by the L2 of this XCD (see the header)
    return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int W, int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void sell_persist_kernel(const SellChainParams cp_) {
    SellChainParamsK &cp = *(SellChainParamsK *)__builtin_amdgcn_kernarg_segment_ptr();  // (= cp_, read in place)
    SellParamsK &p = cp.base;
    constexpr int XR = 4 / NS;  // XCDs per (class, slab) role
    __shared__ float s_wide[4][W];
    const int x = blockIdx.x & 7, cls = x >> 2, xr = x & 3, s = xr & (NS - 1), xi = xr / NS;
    const int wave = threadIdx.x >> 6;
    const int n_wg_x = gridDim.x >> 3;  // workgroups on this XCD
    const unsigned nun = (unsigned)p.n_units[cls];
    const int4 *heads = p.head + p.unit_base[cls];
    const int64_t ybase = p.slab_off[cls][s];
    // this XCD's share of the role's units: groups of four units (the four waves of a workgroup; a wide row IS such a group)
    // g = 0, 1, ... <-> units (g XR + xi) 4 .. + 3, heaviest first
    const int groups = (int)((nun + 3) >> 2);
    const int g_local = groups > xi ? (groups - xi + XR - 1) / XR : 0;
    int32_t *const gcount = cp.sync + kPersistGlobal;
    const int K = cp.K;
    for (int kv = 0; kv < K; ++kv) {
        // (the polling loop below makes the compiler's divergence analysis give up on the layer counter: say it is uniform, or
        // the buffer resource lands in vector registers and every gather becomes a waterfall loop)
        const int k = __builtin_amdgcn_readfirstlane(kv);
        const SellLayer L = {cp.layer[k].xs, cp.layer[k].ys, cp.layer[k].x_rm, cp.layer[k].store_scaled, cp.layer[k].last,
                             cp.layer[k].n_prev, cp.layer[k].prev0_rm, cp.layer[k].prev_scaled};
        const bool compact = cp.layer[k].compact != 0;
        const __amdgpu_buffer_rsrc_t rs = sell_table_rsrc<W, NS>(p, L, cls, s, 0);
        const v4i *ents = L.x_rm ? p.ent0 : p.ent;
        int32_t *const sy = cp.sync + ((k * 8 + x) * kPersistSlots) * kPersistStride;
        // ---- the units of this layer: groups dealt to the workgroups of the XCD in snake order (round r: w, then n - 1 - w, ...),
        // heaviest first — no tickets (r04 second form drew a ticket per group: ~950 contended L2 atomics per XCD and layer at
        // 30-100 ns each were the +28 us per layer it cost)
        {
            const int w = blockIdx.x >> 3;
            for (int r = 0;; ++r) {
                const int g = r * n_wg_x + ((r & 1) ? n_wg_x - 1 - w : w);
                if (r * n_wg_x >= g_local) break;
                if (g >= g_local) continue;
                const unsigned t = (unsigned)(((g * XR + xi) << 2) + wave);
                if (t < nun) {  // (uniform over the workgroup except in the class's last group — which is never a wide row)
                    const int4 h = heads[t];
                    if (compact) sell_unit<W, NS, true, 1>(p, L, cls, s, h, rs, ents, ybase, 0, s_wide);
                    else sell_unit<W, NS, false, 1>(p, L, cls, s, h, rs, ents, ybase, 0, s_wide);
                }
            }
        }
        // ---- the layer is complete when every workgroup of every XCD has arrived ------------------------------------------------
        // every wave waits until the L2 has acknowledged its stores (a workgroup barrier alone orders them inside the CU's
        // write-through L1 only), then the workgroup meets: what thread 0 releases below is everything this workgroup wrote
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t *const go = cp.sync + kPersistGo + x * kPersistStride;  // this XCD's "layer k is complete everywhere" flag
            const int old = ticket_add(sy + kPersistStride);
            if (old == n_wg_x - 1) {
                // the last workgroup of this XCD speaks for it: ONE L2 write-back, one bump of the device-scope counter, one
                // wait, ONE L2 invalidate per XCD and layer (r04 first form: all 2 048 workgroups polled the device-scope counter
                // and acquired at agent scope — an L2 invalidate each: +50 us per layer)
                __hip_atomic_store(sy + kPersistStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cp.fences & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this XCD's L2 is written back
                const int gold = __hip_atomic_fetch_add(gcount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k + 1 < K) {
                    while (__hip_atomic_load(gcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8 * (k + 1)) __builtin_amdgcn_s_sleep(1);
                    if (cp.fences & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // stale lines of the other XCDs' rows are dropped from this L2
                    __hip_atomic_store(go, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    __hip_atomic_store(go, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (gold == 8 * K - 1) __hip_atomic_store(gcount, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (k + 1 < K) {
                // (agent scope on both sides: a workgroup-scope LOAD may be served by the CU's L1 for ever — that form hung; polling
                // the flag by an L2 read-modify-write made 255 pollers serialise at ~2 us per round: 1 265 us per propagation)
                while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1) __builtin_amdgcn_s_sleep(2);
            }
            if (k + 1 < K && (cp.fences & 4)) asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1 (a chain that reuses a buffer two layers on)
        }
        if (k + 1 < K) __syncthreads();
    }
}

__global__ void xcc_census_kernel(int *bad) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0 && (int)(x & 0xf) != (int)(blockIdx.x & 7)) atomicExch(bad, 1);
}
// the tickets rely on workgroup b running on XCD b & 7 (r01 census: 100 % at every grid size in SPX mode): checked once per device
static bool xcd_rule_holds(int device) {
    static std::mutex m;
    static int state[64] = {};  // 0 unknown, 1 yes, 2 no
    std::lock_guard<std::mutex> lock(m);
    if (device < 0 || device >= 64) return false;
    if (state[device]) return state[device] == 1;
    int *d = nullptr, h = 1;
    hipDeviceProp_t prop;
    bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount == 256 && dev_malloc(&d, sizeof(int)) == hipSuccess &&
              hipMemset(d, 0, sizeof(int)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(xcc_census_kernel, dim3(2048), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(xcc_census_kernel, dim3(4096), dim3(256), 0, 0, d);
        ok = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && h == 0;
    }
    (void)hipGetLastError();
    if (d) (void)hipFree(d);
    state[device] = ok ? 1 : 2;
    return ok;
}

// called once per plan (sell_adopt): the sync block and the number of wide units per class
int sell_persist_prepare(const rbg_graph *g, SellDev *sw) {
    sw->persist_ok = false;
    if (sw->W != 32 || sw->borrowed) return RBG_OK;
    if (!xcd_rule_holds(g->device)) return RBG_OK;
    int32_t *sync = nullptr;
    if (dev_malloc(&sync, sizeof(int32_t) * kPersistInts) != hipSuccess || hipMemset(sync, 0, sizeof(int32_t) * kPersistInts) != hipSuccess) {
        (void)hipGetLastError();
        if (sync) (void)hipFree(sync);
        return RBG_OK;  // (optional: the per-layer launches serve the chain)
    }
    sw->sync = sync;
    sw->persist_ok = true;
    return RBG_OK;
}

template <int NS>
static int persist_launch(const SellDev *sw, SellChainParams &cp, hipStream_t s) {
    hipLaunchKernelGGL((sell_persist_kernel<32, NS>), dim3(2048), dim3(256), 0, s, cp);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

bool sell_persist_applicable(const rbg_graph *g, int d, int K) {
    const SellDev *sw = g->sell;
    return opt_sell_persist() && sw && sw->persist_ok && sw->W == 32 && (d == 64 || d == 128 || d == 32) && K >= 2 && K <= kPersistLayers &&
           !opt_sell_skip_hot();
}

// the chain described by `layers` (filled by sell.hip's chain builders) in one launch
int sell_persist_chain(const rbg_graph *g, const SellParams &base, const SellChainLayer *layers, int K, int d, hipStream_t s) {
    static_assert(sizeof(SellChainParams) <= 4096, "kernel arguments");
    const SellDev *sw = g->sell;
    SellChainParams cp{};
    cp.base = base;
    cp.K = K;
    cp.fences = opt_sell_persist() >> 4 ? (opt_sell_persist() >> 4) & 7 : 7;  // (diagnostic: option value 1 + 16 * mask)
    cp.sync = sw->sync;
    for (int k = 0; k < K; ++k) cp.layer[k] = layers[k];
    if (d == 64) return persist_launch<2>(sw, cp, s);
    if (d == 128) return persist_launch<4>(sw, cp, s);
    return persist_launch<1>(sw, cp, s);
}

}  // namespace rbg
