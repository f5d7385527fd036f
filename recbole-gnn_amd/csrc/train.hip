// train.hip — the mini-batch step around the propagation, fused.
//
// Replaces, for LightGCN.calculate_loss (recbole_gnn/model/general_recommender/lightgcn.py:83-110) followed by
// loss.backward() and optimizer.step() in RecBole's Trainer._train_epoch [recbole==1.1.1]:
//   rbg_bpr_grad_f32   : pos/neg scores (lightgcn.py:98-99), BPRLoss(gamma=1e-10) (:100), and dLoss/d(out_mean)
//                        scattered into a dense [N,d] buffer (what torch's index backward does)
//   rbg_emb_reg_grad_f32: EmbLoss on the EGO embeddings with require_pow=True (:103-108): loss term and its
//                        sparse-row gradient added onto dE0 after the backward chain
//   rbg_adam_step_f32  : torch.optim.Adam (defaults: no weight decay, no amsgrad) over both embedding tables
// Row scatters use float atomics, like torch's GPU index_put_(accumulate=True): duplicated ids inside a batch
// make the low bits of those rows order-dependent.

#include <hip/hip_runtime.h>

#include <math.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>

#include "internal.h"
#include "ordered.h"

namespace rbg {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// hipMemsetAsync is NOT used on the caller's stream anywhere in this library: captured into a HIP graph, a memset node writes
// its value correctly on the first replay and garbage bytes on later ones (ROCm 7.2, measured: devtools/memset_probe.py — the
// fused NGCF step's EmbLoss sums read 0xD9D9D9D9 at its 51st replay).  A fill KERNEL is a kernel node like any other.
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ p, int64_t n_words) {
    const int64_t n4 = n_words >> 2;
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[(n4 << 2) + threadIdx.x] = 0u;
}

int zero_async(void *ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return RBG_OK;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3)) return fail(RBG_EINVAL, "zero_async: %zu bytes at %p (4-byte units)", bytes, ptr);
    uint32_t *p = static_cast<uint32_t *>(ptr);
    int64_t words = (int64_t)(bytes >> 2);
    const int64_t head = std::min<int64_t>(words, (int64_t)((16 - (reinterpret_cast<uintptr_t>(ptr) & 15)) & 15) >> 2);  // up to the 16-byte boundary
    if (head) {
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, s, p, head);
        p += head, words -= head;
    }
    if (words) {
        const unsigned blocks = (unsigned)std::min<int64_t>(((words >> 2) + 255) / 256 + 1, 4096);
        hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, s, p, words);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

constexpr int kElemsPerWave = 4;  // batch elements per wavefront: 16 per 256-thread workgroup, ONE loss atomic per workgroup
// (one atomic per element on the single loss word serialised the whole kernel: 80 us for 6 144 elements)
// Sums over the batch (loss values, block norms): every workgroup reduces its elements in a fixed order (block_sum); across
// workgroups the default is one float atomic per workgroup and value — order-dependent in the last bits.  Option "deterministic":
// the workgroups add their partial sums as 31.32 FIXED-POINT integers into a slot of g_fix_acc (integer addition is associative:
// any order gives the same bits), the last workgroup to arrive converts the total and adds it to the caller's cell, and leaves the
// slot zeroed.  No scratch memory from the caller, nothing allocated, capturable.  A workgroup's partial enters the fixed-point
// sum only while |partial| < 2^31 / gridDim.x (so the total cannot wrap the 64-bit accumulator: beyond it — a diverged run —
// the partial is added with a float atomic, still summed, no longer order-free); resolution 2^-32 of the value BEFORE its
// `post` factor (r06, ADVICE r05: a weighted regulariser of 1e-9 per workgroup is summed unweighted and scaled once at the end).
// Slots (r06): one per STREAM — launches of one stream are ordered, so they can share it — and one per stream
// CAPTURE (hipStreamGetCaptureInfo's id: a captured step keeps its slot for every replay, whatever stream replays it, and never
// meets the eager launches of the stream it was captured on).  kFixSlots keys are live at a time; the oldest key's slot is
// recycled after that (a graph captured more than kFixSlots captures / streams ago may then share its slot with a new one:
// only concurrent use would collide).
constexpr int kMaxWaves = 16;
constexpr int kFixSlots = 256;
__device__ unsigned long long g_fix_acc[kFixSlots][4];
__device__ unsigned g_fix_cnt[kFixSlots];
static std::mutex g_fix_mutex;
static std::unordered_map<unsigned long long, int> g_fix_key_slot;
static unsigned long long g_fix_slot_key[kFixSlots];
static unsigned g_fix_next = 0;
static int fix_slot(hipStream_t stream) {
    if (!opt_deterministic()) return -1;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    unsigned long long key;
    if (hipStreamGetCaptureInfo(stream, &st, &id) == hipSuccess && st == hipStreamCaptureStatusActive) key = (id << 1) | 1ull;
    else {
        (void)hipGetLastError();
        key = (unsigned long long)(uintptr_t)stream << 1;  // (pointers are at least 2-byte aligned: no clash with the capture keys' low bit)
    }
    std::lock_guard<std::mutex> lock(g_fix_mutex);
    auto it = g_fix_key_slot.find(key);
    if (it != g_fix_key_slot.end()) return it->second;
    const int slot = (int)(g_fix_next++ % kFixSlots);
    if (g_fix_next > (unsigned)kFixSlots) g_fix_key_slot.erase(g_fix_slot_key[slot]);  // the slot's previous key
    g_fix_slot_key[slot] = key;
    g_fix_key_slot[key] = slot;
    return slot;
}

// the workgroup's waves' partial sums, added in wave order (pairwise for the usual four)
__device__ __forceinline__ float block_sum(float part, float *red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = part;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) {
        if (nw == 4) s = (red[0] + red[1]) + (red[2] + red[3]);
        else
            for (int w = 0; w < nw; ++w) s += red[w];
    }
    return s;  // valid in thread 0
}

// dst[j] += the sum over all workgroups of part[j] (each lane-0-of-wave partial, summed per workgroup first); slot < 0: float atomics
// (post: a factor applied to the sum, not to the addends — the fixed-point sum keeps its 2^-32 resolution on values of order 1)
template <int NV>
__device__ __forceinline__ void commit_sums(const float (&part)[NV], float *const (&dst)[NV], int slot, const float post = 1.f) {
    __shared__ float red[NV][kMaxWaves];
    float tot[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) tot[j] = block_sum(part[j], red[j]);
    if (threadIdx.x != 0) return;
    if (slot < 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (tot[j] != 0.f) atomicAdd(dst[j], tot[j] * post);
        return;
    }
    const float bound = 2147483648.0f / (float)gridDim.x;  // gridDim.x partials below it cannot wrap the signed 64-bit sum
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (fabsf(tot[j]) < bound) atomicAdd(&g_fix_acc[slot][j], (unsigned long long)__float2ll_rn(tot[j] * 4294967296.0f));
        else atomicAdd(dst[j], tot[j] * post);  // beyond the range (a diverged run; NaN): still summed, no longer order-free
    }
    __threadfence();
    if (atomicAdd(&g_fix_cnt[slot], 1u) == gridDim.x - 1) {  // the last workgroup: every partial sum is in
        __threadfence();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const long long t = (long long)atomicExch(&g_fix_acc[slot][j], 0ull);
            const float f = (float)((double)t * 2.3283064365386963e-10 * (double)post);
            if (f != 0.f) atomicAdd(dst[j], f);
        }
        atomicExch(&g_fix_cnt[slot], 0u);
    }
}

__device__ __forceinline__ void block_add_loss(float part, float *loss, int slot, const float post = 1.f) {
    const float p1[1] = {part};
    float *const d1[1] = {loss};
    commit_sums<1>(p1, d1, slot, post);
}

// element groups of kElemsPerWave a wave walks: grp = first, first + stride, ...
#define RBG_FOR_GROUPS(grp, n_elems)                                                                  \
    for (int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); grp * kElemsPerWave < (n_elems); \
         grp += (int64_t)gridDim.x * (blockDim.x >> 6))

static inline dim3 grid_for(int64_t n_elems) { return dim3((unsigned)((n_elems + 4 * kElemsPerWave - 1) / (4 * kElemsPerWave))); }

// ---- BPR on the layer mean (lightgcn.py:98-100) ---------------------------------------------------------------------------
struct BprArgs {
    const float *mean;
    int64_t n_users;
    const int64_t *user, *pos, *neg;
    int64_t B;
    int d;
    float gamma;
    float *grad_mean;
};

// triple b: its loss term; the gradient rows of `which` (0 user, 1 positive, 2 negative; -1 all three; -2 none)
template <bool ORDERED>
__device__ __forceinline__ float bpr_elem(const BprArgs &a, int64_t b, int which, int lane) {
    const int d = a.d;
    const float *ue = a.mean + a.user[b] * d;
    const float *pe = a.mean + (a.n_users + a.pos[b]) * d;
    const float *ne = a.mean + (a.n_users + a.neg[b]) * d;
    float sp = 0.f, sn = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float u = ue[k];
        sp = fmaf(u, pe[k], sp);
        sn = fmaf(u, ne[k], sn);
    }
    sp = wave_sum(sp);
    sn = wave_sum(sn);
    const float x = sp - sn;
    const float sig = 1.0f / (1.0f + expf(-x));
    // L = -mean(log(gamma + sig));  dL/dx = -(sig (1 - sig)) / (gamma + sig) / B
    const float c = -(sig * (1.0f - sig)) / (a.gamma + sig) / (float)a.B;
    if (which != -2) {
        float *gu = a.grad_mean + a.user[b] * d;
        float *gp = a.grad_mean + (a.n_users + a.pos[b]) * d;
        float *gn = a.grad_mean + (a.n_users + a.neg[b]) * d;
        for (int k = lane; k < d; k += 64) {
            const float u = ue[k], p = pe[k], n = ne[k];
            if (which < 0 || which == 0) row_add<ORDERED>(gu + k, c * (p - n));
            if (which < 0 || which == 1) row_add<ORDERED>(gp + k, c * u);
            if (which < 0 || which == 2) row_add<ORDERED>(gn + k, -c * u);
        }
    }
    return -logf(a.gamma + sig) / (float)a.B;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void bpr_grad_kernel(const BprArgs a, float *__restrict__ loss, int slot) {
    const int lane = threadIdx.x & 63;
    float loss_part = 0.f;
    RBG_FOR_GROUPS(grp, a.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t b = grp * kElemsPerWave + e;
            if (b >= a.B) break;
            loss_part += bpr_elem<false>(a, b, SCATTER ? -1 : -2, lane);
        }
    block_add_loss(loss_part, loss, slot);
}

// occurrence m of [0, 3B): which = m / B of triple m % B.  Users share keys with users only, items with items.
struct Triples {
    const int64_t *user, *pos, *neg;
    int64_t n_users, B;
    __device__ __forceinline__ int segments(int64_t w, KeySeg (&seg)[2], int64_t &mine) const {
        if (w < B) {
            mine = user[w];
            seg[0] = KeySeg{user, B, 0};
            return 1;
        }
        mine = w < 2 * B ? pos[w - B] : neg[w - 2 * B];
        seg[0] = KeySeg{pos, B, B};
        seg[1] = KeySeg{neg, B, 2 * B};
        return 2;
    }
    __device__ __forceinline__ int which_of(int64_t m) const { return m < B ? 0 : (m < 2 * B ? 1 : 2); }
    __device__ __forceinline__ int64_t triple_of(int64_t m) const { return m < B ? m : (m < 2 * B ? m - B : m - 2 * B); }
};

struct BprRows : Triples {
    BprArgs a;
    __device__ __forceinline__ void apply(int64_t m, int lane) const { bpr_elem<true>(a, triple_of(m), which_of(m), lane); }
};

// ---- EmbLoss on the ego embeddings (lightgcn.py:103-108) ------------------------------------------------------------------
struct EmbRegArgs {
    const float *user_emb, *item_emb;
    int64_t n_users;
    const int64_t *user, *pos, *neg;
    int64_t B;
    int d;
    float *grad_e0;
};

// occurrence w of [0, 3B): grad_e0[row] += scale * row (if SCATTER); returns the row's sum of squares
template <bool ORDERED, bool SCATTER>
__device__ __forceinline__ float emb_reg_elem(const EmbRegArgs &a, int64_t w, float scale, int lane) {
    const int64_t b = w % a.B;
    const int which = (int)(w / a.B);  // 0 user, 1 pos item, 2 neg item
    const int64_t id = which == 0 ? a.user[b] : (which == 1 ? a.pos[b] : a.neg[b]);
    const float *row = which == 0 ? a.user_emb + id * a.d : a.item_emb + id * a.d;
    float *g = a.grad_e0 + (which == 0 ? id : a.n_users + id) * a.d;
    float sq = 0.f;
    for (int k = lane; k < a.d; k += 64) {
        const float e = row[k];
        sq = fmaf(e, e, sq);
        if (SCATTER) row_add<ORDERED>(g + k, scale * e);
    }
    return wave_sum(sq);
}

// EmbLoss(norm=2, require_pow=True): reg = (|U0[user]|^2 + |I0[pos]|^2 + |I0[neg]|^2) / B / 2
// d(reg_weight * reg)/d(row) = reg_weight / B * row   per occurrence
template <bool SCATTER>
__global__ __launch_bounds__(256) void emb_reg_grad_kernel(const EmbRegArgs a, float reg_weight, float *__restrict__ loss, int slot) {
    const int lane = threadIdx.x & 63;
    float loss_part = 0.f;
    RBG_FOR_GROUPS(grp, 3 * a.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t w = grp * kElemsPerWave + e;
            if (w >= 3 * a.B) break;
            loss_part += emb_reg_elem<false, SCATTER>(a, w, reg_weight / (float)a.B, lane) * 0.5f;  // (unweighted: reg_weight / B multiplies the SUM)
        }
    block_add_loss(loss_part, loss, slot, reg_weight / (float)a.B);
}

// EmbLoss(norm=2, require_pow=False): reg = (||U0[user]||_F + ||I0[pos]||_F + ||I0[neg]||_F) / B — torch.norm of each
// gathered [B, d] block (RecBole's default; LightGCN.yaml switches to the squared form).  Pass 1: the three sums of squares.
__global__ __launch_bounds__(256) void emb_sumsq_kernel(const EmbRegArgs a, float *__restrict__ sums, int slot) {
    const int lane = threadIdx.x & 63;
    float part[3] = {0.f, 0.f, 0.f};
    RBG_FOR_GROUPS(grp, 3 * a.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t w = grp * kElemsPerWave + e;
            if (w >= 3 * a.B) break;
            const int which = (int)(w / a.B);
            const float sq = emb_reg_elem<false, false>(a, w, 0.f, lane);
            part[0] += which == 0 ? sq : 0.f;
            part[1] += which == 1 ? sq : 0.f;
            part[2] += which == 2 ? sq : 0.f;
        }
    float *const dst[3] = {sums, sums + 1, sums + 2};
    commit_sums<3>(part, dst, slot);
}

// Pass 2: d(reg_weight * ||E||_F / B)/d(row) = reg_weight / B * row / ||E||_F per occurrence (torch: zero at ||E|| = 0).
__device__ __forceinline__ float nopow_scale(const float *sums, int which, float reg_weight, int64_t B) {
    const float nw = sqrtf(sums[which]);
    return nw > 0.f ? reg_weight / (float)B / nw : 0.f;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void emb_reg_grad_nopow_kernel(const EmbRegArgs a, float reg_weight, const float *__restrict__ sums,
                                                                 float *__restrict__ loss) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(loss, reg_weight * ((sqrtf(sums[0]) + sqrtf(sums[1])) + sqrtf(sums[2])) / (float)a.B);
    if (!SCATTER) return;
    RBG_FOR_GROUPS(grp, 3 * a.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t w = grp * kElemsPerWave + e;
            if (w >= 3 * a.B) break;
            emb_reg_elem<false, true>(a, w, nopow_scale(sums, (int)(w / a.B), reg_weight, a.B), lane);
        }
}

struct EmbRegRows : Triples {
    EmbRegArgs a;
    float reg_weight;
    const float *sums;  // NULL: the squared form
    __device__ __forceinline__ void apply(int64_t m, int lane) const {
        emb_reg_elem<true, true>(a, m, sums ? nopow_scale(sums, which_of(m), reg_weight, B) : reg_weight / (float)B, lane);
    }
};

// ---- the mini-batch loss of NGCF on the CONCATENATION of its layer outputs (ngcf.py:100-126) without forming it ----------
// u_e = cat_t(E_t[user]) etc.: a score is the sum over the tables of the per-table dots, a block norm the root of the sum over
// the tables of the per-table sums of squares.  Pass 1 (concat_bpr_begin_kernel): scores, BPR value, dLoss/d(pos - neg) per
// triple and the three sums of squares.  Pass 2 (concat_bpr_scatter_kernel), once per table, after that table's dense gradient
// from the layer above has been written: the rows' gradients added in place.
struct ConcatTables {
    const float *tab[RBG_MAX_CONCAT];
    int width[RBG_MAX_CONCAT];
    int n;
};

__global__ __launch_bounds__(256) void concat_bpr_begin_kernel(ConcatTables T, int64_t n_users, const int64_t *__restrict__ user,
                                                               const int64_t *__restrict__ pos, const int64_t *__restrict__ neg,
                                                               int64_t B, float gamma, int form, float *__restrict__ coef,
                                                               float *__restrict__ sums, float *__restrict__ loss, int slot) {
    const int lane = threadIdx.x & 63;
    float part[4] = {0.f, 0.f, 0.f, 0.f};  // loss, |U|^2, |P|^2, |N|^2
    RBG_FOR_GROUPS(grp, B)
    for (int e = 0; e < kElemsPerWave; ++e) {
        const int64_t b = grp * kElemsPerWave + e;
        if (b >= B) break;
        const int64_t ru = user[b], rp = n_users + pos[b], rn = n_users + neg[b];
        float sp = 0.f, sn = 0.f, qu = 0.f, qp = 0.f, qn = 0.f;
        for (int t = 0; t < T.n; ++t) {
            const int w = T.width[t];
            const float *ue = T.tab[t] + ru * w, *pe = T.tab[t] + rp * w, *ne = T.tab[t] + rn * w;
            for (int k = lane; k < w; k += 64) {
                const float u = ue[k], p = pe[k], n = ne[k];
                sp = fmaf(u, p, sp);
                sn = fmaf(u, n, sn);
                qu = fmaf(u, u, qu);
                qp = fmaf(p, p, qp);
                qn = fmaf(n, n, qn);
            }
        }
        sp = wave_sum(sp), sn = wave_sum(sn), qu = wave_sum(qu), qp = wave_sum(qp), qn = wave_sum(qn);
        const float x = sp - sn;
        if (form == 0) {  // recbole BPRLoss: -mean(log(gamma + sigmoid(x)))
            const float sig = 1.0f / (1.0f + expf(-x));
            if (lane == 0) coef[b] = -(sig * (1.0f - sig)) / (gamma + sig) / (float)B;
            part[0] += -logf(gamma + sig) / (float)B;
        } else {  // sgl.py:147-162: -sum(logsigmoid(x)); d/dx = -sigmoid(-x)
            const float en = expf(-fabsf(x));  // logsigmoid(x) = min(x, 0) - log1p(exp(-|x|))
            if (lane == 0) coef[b] = -(x >= 0.f ? en : 1.0f) / (1.0f + en);
            part[0] += log1pf(en) - fminf(x, 0.f);
        }
        part[1] += qu, part[2] += qp, part[3] += qn;
    }
    float *const dst[4] = {loss, sums, sums + 1, sums + 2};
    commit_sums<4>(part, dst, slot);
}

// EmbLoss(norm = 2) on the three concatenated blocks: require_pow False: (|U| + |P| + |N|) / B, d/d(row) = row / |block| / B;
// True: (|U|^2 + |P|^2 + |N|^2) / B / 2, d/d(row) = row / B
struct ConcatArgs {
    const float *tab;
    int w;
    int64_t n_users;
    const int64_t *user, *pos, *neg;
    int64_t B;
    float reg_weight;
    int require_pow;
    const float *coef, *sums;
    float *grad;
};

__device__ __forceinline__ void concat_scales(const ConcatArgs &a, float (&s3)[3]) {
    if (a.require_pow) {
        s3[0] = s3[1] = s3[2] = a.reg_weight / (float)a.B;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float nrm = sqrtf(a.sums[i]);
            s3[i] = nrm > 0.f ? a.reg_weight / (float)a.B / nrm : 0.f;
        }
    }
}

// triple b: the gradient rows of `which` (-1: all three)
template <bool ORDERED>
__device__ __forceinline__ void concat_elem(const ConcatArgs &a, const float (&s3)[3], int64_t b, int which, int lane) {
    const int w = a.w;
    const int64_t ru = a.user[b], rp = a.n_users + a.pos[b], rn = a.n_users + a.neg[b];
    const float c = a.coef[b];
    const float *ue = a.tab + ru * w, *pe = a.tab + rp * w, *ne = a.tab + rn * w;
    float *gu = a.grad + ru * w, *gp = a.grad + rp * w, *gn = a.grad + rn * w;
    for (int k = lane; k < w; k += 64) {
        const float u = ue[k], p = pe[k], n = ne[k];
        if (which < 0 || which == 0) row_add<ORDERED>(gu + k, fmaf(c, p - n, s3[0] * u));
        if (which < 0 || which == 1) row_add<ORDERED>(gp + k, fmaf(c, u, s3[1] * p));
        if (which < 0 || which == 2) row_add<ORDERED>(gn + k, fmaf(-c, u, s3[2] * n));
    }
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void concat_bpr_scatter_kernel(const ConcatArgs a, float *__restrict__ loss_reg) {
    const int lane = threadIdx.x & 63;
    float s3[3];
    concat_scales(a, s3);
    if (loss_reg && blockIdx.x == 0 && threadIdx.x == 0) {
        const float r = a.require_pow ? ((a.sums[0] + a.sums[1]) + a.sums[2]) * 0.5f : (sqrtf(a.sums[0]) + sqrtf(a.sums[1])) + sqrtf(a.sums[2]);
        atomicAdd(loss_reg, a.reg_weight * r / (float)a.B);
    }
    if (!SCATTER) return;
    RBG_FOR_GROUPS(grp, a.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t b = grp * kElemsPerWave + e;
            if (b >= a.B) break;
            concat_elem<false>(a, s3, b, -1, lane);
        }
}

struct ConcatRows : Triples {
    ConcatArgs a;
    __device__ __forceinline__ void apply(int64_t m, int lane) const {
        float s3[3];
        concat_scales(a, s3);
        concat_elem<true>(a, s3, triple_of(m), which_of(m), lane);
    }
};

// torch.optim.Adam single step (foreach/fused semantics): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).   Rows [0,n_users) of the [N,d] state live in the
// user table, the rest in the item table.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ user_emb, float *__restrict__ item_emb, int64_t n_users_d,
                                                   const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                                   int64_t nd, float lr_over_bc1, float inv_sqrt_bc2, float beta1, float beta2, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = grad[i];
        const float mi = beta1 * m[i] + (1.0f - beta1) * g;
        const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        float *p = i < n_users_d ? user_emb + i : item_emb + (i - n_users_d);
        *p = *p - lr_over_bc1 * (mi / denom);
    }
}

// The same update with the step count in DEVICE memory (a HIP-graph replay must not bake the bias corrections in): a one-thread
// launch increments *step and writes {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)}; the update reads them.  float4 lanes when the user
// table's float count is a multiple of 4 (d % 4 == 0).
// (r06: loss_total != NULL: the step's finished loss joins a running total here — the driver reads it once per epoch instead of
//  adding a device scalar per step)
__global__ void adam_tick_kernel(int64_t *__restrict__ step, float lr, float beta1, float beta2, float *__restrict__ factors,
                                 const float *__restrict__ loss, float *__restrict__ loss_total) {
    const int64_t t = *step + 1;
    *step = t;
    if (loss_total) *loss_total += *loss;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    factors[0] = (float)((double)lr / bc1);
    factors[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float *__restrict__ user_emb, float *__restrict__ item_emb, int64_t n_users_d,
                                                       const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                                       int64_t nd, const float *__restrict__ factors, float beta1, float beta2, float eps) {
    const float lr_over_bc1 = factors[0], inv_sqrt_bc2 = factors[1];
    const int64_t n4 = nd >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = q << 2;
        const float4 g = *reinterpret_cast<const float4 *>(grad + i);
        float4 mi = *reinterpret_cast<const float4 *>(m + i), vi = *reinterpret_cast<const float4 *>(v + i);
        float *pp = i < n_users_d ? user_emb + i : item_emb + (i - n_users_d);
        float4 pv = *reinterpret_cast<const float4 *>(pp);
#define RBG_ADAM1(c)                                                        \
        mi.c = beta1 * mi.c + (1.0f - beta1) * g.c;                         \
        vi.c = beta2 * vi.c + (1.0f - beta2) * g.c * g.c;                   \
        pv.c = pv.c - lr_over_bc1 * (mi.c / (sqrtf(vi.c) * inv_sqrt_bc2 + eps));
        RBG_ADAM1(x) RBG_ADAM1(y) RBG_ADAM1(z) RBG_ADAM1(w)
#undef RBG_ADAM1
        *reinterpret_cast<float4 *>(m + i) = mi;
        *reinterpret_cast<float4 *>(v + i) = vi;
        *reinterpret_cast<float4 *>(pp) = pv;
    }
}

// ---- r06: LightGCN's step without its glue launches --------------------------------------------------------------------------
// The fused step was 6 propagation launches + 7 small ones: zero(grad_mean, 18 MB), zero(loss), bpr_grad, emb_reg_grad, adam_tick,
// adam, and the driver's loss accumulation — 59 of its 231 us at the Gowalla shape (rocprofv3 of an epoch).  Two launches do the
// same arithmetic:
//   head: bpr_grad + the regulariser's VALUE + how often every node occurs in the batch (row_count; the regulariser's gradient is
//         reg_weight / B x occurrences x row: the tail adds it where it applies Adam).  The loss is summed in a scratch cell and
//         the last workgroup to arrive stores it (and adds it to a running total): nothing to zero beforehand.
//   tail: Adam over both tables with that gradient term; it zeroes the grad_mean rows the head wrote (so grad_mean is all-zero
//         again: no 18 MB fill per step), clears the OTHER occurrence table (the tables alternate with the step's parity: the one
//         it reads cannot be cleared while other lanes still read it), derives Adam's bias corrections from the device step count
//         in every workgroup and lets the last workgroup to finish increment it.
struct LeanHeadArgs {
    BprArgs bpr;
    const float *user_emb, *item_emb;
    int64_t n_nodes;
    float reg_weight;
    int32_t *row_count;        // [2][n_nodes]
    int64_t *step;
    float *scratch;            // [0]: the loss sum, [1] (as unsigned): workgroups done, [2]: the tail's count, [4], [5]: Adam's factors of this step
    float *loss, *loss_total;  // stored / accumulated by the last workgroup
    float lr, beta1, beta2;
};

__global__ __launch_bounds__(256) void lgcn_head_kernel(const LeanHeadArgs a) {
    __shared__ float red[kMaxWaves];
    const int lane = threadIdx.x & 63;
    const int64_t parity = *a.step & 1;
    int32_t *cnt = a.row_count + parity * a.n_nodes;
    const BprArgs &b = a.bpr;
    float loss_part = 0.f, reg_part = 0.f;
    if (b.d <= 64) {
        // one column per lane: the wave's four triples side by side — all their row loads are issued before the first is used (one
        // after the other they were four dependent chains of id -> rows -> sums -> atomics: 17.5 us for 2 048 triples)
        RBG_FOR_GROUPS(grp, b.B) {
            const bool col = lane < b.d;
            int64_t u[kElemsPerWave], ip[kElemsPerWave], in[kElemsPerWave];
            bool live[kElemsPerWave];
#pragma unroll
            for (int e = 0; e < kElemsPerWave; ++e) {
                const int64_t t = grp * kElemsPerWave + e;
                live[e] = t < b.B;
                u[e] = live[e] ? b.user[t] : 0, ip[e] = live[e] ? b.pos[t] : 0, in[e] = live[e] ? b.neg[t] : 0;
            }
            float ue[kElemsPerWave], pe[kElemsPerWave], ne[kElemsPerWave], ru[kElemsPerWave], rp[kElemsPerWave], rn[kElemsPerWave];
#pragma unroll
            for (int e = 0; e < kElemsPerWave; ++e) {
                const bool on = live[e] && col;
                ue[e] = on ? b.mean[u[e] * b.d + lane] : 0.f;
                pe[e] = on ? b.mean[(b.n_users + ip[e]) * b.d + lane] : 0.f;
                ne[e] = on ? b.mean[(b.n_users + in[e]) * b.d + lane] : 0.f;
                ru[e] = on ? a.user_emb[u[e] * b.d + lane] : 0.f;
                rp[e] = on ? a.item_emb[ip[e] * b.d + lane] : 0.f;
                rn[e] = on ? a.item_emb[in[e] * b.d + lane] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < kElemsPerWave; ++e) {
                const float x = wave_sum(ue[e] * (pe[e] - ne[e]));  // sp - sn
                const float sq = wave_sum(fmaf(ru[e], ru[e], fmaf(rp[e], rp[e], rn[e] * rn[e])));
                if (!live[e]) continue;
                const float sig = 1.0f / (1.0f + expf(-x));
                const float c = -(sig * (1.0f - sig)) / (b.gamma + sig) / (float)b.B;
                loss_part += -logf(b.gamma + sig) / (float)b.B;
                reg_part += sq * 0.5f;
                if (col) {
                    atomicAdd(b.grad_mean + u[e] * b.d + lane, c * (pe[e] - ne[e]));
                    atomicAdd(b.grad_mean + (b.n_users + ip[e]) * b.d + lane, c * ue[e]);
                    atomicAdd(b.grad_mean + (b.n_users + in[e]) * b.d + lane, -c * ue[e]);
                }
                if (lane == 0) {
                    atomicAdd(&cnt[u[e]], 1);
                    atomicAdd(&cnt[b.n_users + ip[e]], 1);
                    atomicAdd(&cnt[b.n_users + in[e]], 1);
                }
            }
        }
    } else {
    RBG_FOR_GROUPS(grp, b.B)
        for (int e = 0; e < kElemsPerWave; ++e) {
            const int64_t t = grp * kElemsPerWave + e;
            if (t >= b.B) break;
            loss_part += bpr_elem<false>(b, t, -1, lane);
            const int64_t u = b.user[t], ip = b.pos[t], in = b.neg[t];
            const float *ru = a.user_emb + u * b.d, *rp = a.item_emb + ip * b.d, *rn = a.item_emb + in * b.d;
            float sq = 0.f;
            for (int k = lane; k < b.d; k += 64) {
                const float x = ru[k], y = rp[k], z = rn[k];
                sq = fmaf(x, x, fmaf(y, y, fmaf(z, z, sq)));
            }
            reg_part += wave_sum(sq) * 0.5f;
            if (lane == 0) {
                atomicAdd(&cnt[u], 1);
                atomicAdd(&cnt[b.n_users + ip], 1);
                atomicAdd(&cnt[b.n_users + in], 1);
            }
        }
    }
    const float tot = block_sum(loss_part + reg_part * (a.reg_weight / (float)b.B), red);
    if (threadIdx.x != 0) return;
    // no __threadfence here (an L2 write-back per workgroup): the arrival count is bumped only after this workgroup's sum has
    // RETURNED from the atomic unit (its old value feeds the increment), so the last arrival's exchange sees every sum
    const float before = atomicAdd(&a.scratch[0], tot);
    unsigned *done = reinterpret_cast<unsigned *>(a.scratch) + 1;
    const unsigned one = __float_as_uint(before) == 0xffffffffu ? 3u : 1u;  // (always 1 for a finite sum: the data dependency is the point)
    if (atomicAdd(done, one) == gridDim.x - 1) {  // the last workgroup: every partial sum is in
        const float total = atomicExch(&a.scratch[0], 0.f);
        *a.loss = total;
        if (a.loss_total) *a.loss_total += total;
        // Adam's bias corrections of THIS step for the tail (two double-precision pow: once per step, not once per workgroup there)
        const int64_t t1 = *a.step + 1;
        const double t = (double)t1;
        a.scratch[4] = (float)((double)a.lr / (1.0 - pow((double)a.beta1, t)));
        a.scratch[5] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, t)));
        // the step is counted HERE (every workgroup of this launch has read the old count — its parity — before it arrived): the
        // tail needs no arrival counter of its own (8 192 atomics on one word were 38 of its 58 us)
        *a.step = t1;
        atomicExch(done, 0u);
    }
}

struct LeanTailArgs {
    float *user_emb, *item_emb;
    int64_t n_users_d, nd, n_nodes;
    int d;
    const float *grad_e0;
    float *grad_mean;
    int32_t *row_count;
    float reg_over_b;
    float *m, *v;
    const int64_t *step;
    const float *factors;
    float beta1, beta2, eps;
    int quads_shift;  // log2(d / 4) when d / 4 is a power of two, else -1
};

__global__ __launch_bounds__(256) void lgcn_tail_kernel(const LeanTailArgs a) {
    const int64_t t0 = *a.step - 1;  // (the head counted the step already)
    const float lr_over_bc1 = a.factors[0], inv_sqrt_bc2 = a.factors[1];  // (the head's last workgroup wrote them for this step)
    const int32_t *cnt = a.row_count + (t0 & 1) * a.n_nodes;
    int32_t *other = a.row_count + ((t0 & 1) ^ 1) * a.n_nodes;
    const float beta1 = a.beta1, beta2 = a.beta2, eps = a.eps;
    const int64_t n4 = a.nd >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = q << 2;
        const int64_t row = a.quads_shift >= 0 ? (q >> a.quads_shift) : (int64_t)((uint64_t)q / (unsigned)(a.d >> 2));
        const int c = cnt[row];
        float4 g = *reinterpret_cast<const float4 *>(a.grad_e0 + i);
        float4 mi = *reinterpret_cast<const float4 *>(a.m + i), vi = *reinterpret_cast<const float4 *>(a.v + i);
        float *pp = i < a.n_users_d ? a.user_emb + i : a.item_emb + (i - a.n_users_d);
        float4 pv = *reinterpret_cast<const float4 *>(pp);
        if (c) {  // EmbLoss(require_pow): reg_weight / B x row per occurrence; the head's rows of grad_mean back to zero
            const float w = a.reg_over_b * (float)c;
            g.x = fmaf(w, pv.x, g.x), g.y = fmaf(w, pv.y, g.y), g.z = fmaf(w, pv.z, g.z), g.w = fmaf(w, pv.w, g.w);
            *reinterpret_cast<float4 *>(a.grad_mean + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (i - row * a.d == 0) other[row] = 0;
#define RBG_ADAM1(c)                                                        \
        mi.c = beta1 * mi.c + (1.0f - beta1) * g.c;                         \
        vi.c = beta2 * vi.c + (1.0f - beta2) * g.c * g.c;                   \
        pv.c = pv.c - lr_over_bc1 * (mi.c / (sqrtf(vi.c) * inv_sqrt_bc2 + eps));
        RBG_ADAM1(x) RBG_ADAM1(y) RBG_ADAM1(z) RBG_ADAM1(w)
#undef RBG_ADAM1
        *reinterpret_cast<float4 *>(a.m + i) = mi;
        *reinterpret_cast<float4 *>(a.v + i) = vi;
        *reinterpret_cast<float4 *>(pp) = pv;
    }
}

// r06: the one-occurrence mask of a batch's ids (models._once_mask: what torch.unique selects, with static shapes) and the row weights of
// the masked InfoNCE in ONE single-block launch instead of five or six torch launches: every position writes its index into its id's
// slot (first = 1: an integer minimum, so the FIRST occurrence stays — option "deterministic"), the position that finds itself there
// is the occurrence kept; row_w = once (SimGCL's sum over the distinct ids) or once / count (XSimGCL's mean, xsimgcl.py:54).
// slot [n_ids] needs no reset: only the slots written here are read.
__global__ __launch_bounds__(1024) void once_mask_kernel(const int64_t *__restrict__ ids, int64_t B, long long *__restrict__ slot, int first,
                                                         int mean_form, float *__restrict__ once, float *__restrict__ row_w) {
    __shared__ float part[16];
    const int tid = threadIdx.x;
    if (first) {
        for (int64_t b = tid; b < B; b += 1024) slot[ids[b]] = 0x7fffffffffffffffLL;
        __syncthreads();
        for (int64_t b = tid; b < B; b += 1024) atomicMin(&slot[ids[b]], (long long)b);
    } else {
        for (int64_t b = tid; b < B; b += 1024) slot[ids[b]] = (long long)b;  // (racing stores of different positions: one of them stays)
    }
    __syncthreads();
    float cnt = 0.f;
    for (int64_t b = tid; b < B; b += 1024) {
        const float o = (__hip_atomic_load(&slot[ids[b]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (long long)b) ? 1.f : 0.f;
        once[b] = o;
        cnt += o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((tid & 63) == 0) part[tid >> 6] = cnt;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += part[k];
    for (int64_t b = tid; b < B; b += 1024) row_w[b] = mean_form ? once[b] / total : once[b];
}

// out = y + sign(y) * noise / max(|noise row|, 1e-12) * eps: the epilogue of rbg_spmm_noise_f32 on a product that exists already
// (r06: SimGCL's three passes share their first product A E_0).  One wave per row, d <= 128: lane c holds columns c and c + 64.
__global__ __launch_bounds__(256) void sign_noise_kernel(const float *__restrict__ y, const float *__restrict__ noise, int64_t n, int d,
                                                         float eps, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (row >= n) return;
    const bool in0 = lane < d, in1 = lane + 64 < d;
    const float n0 = in0 ? noise[row * d + lane] : 0.f, n1 = in1 ? noise[row * d + lane + 64] : 0.f;
    const float y0 = in0 ? y[row * d + lane] : 0.f, y1 = in1 ? y[row * d + lane + 64] : 0.f;
    float ss = fmaf(n0, n0, n1 * n1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float nsc = eps / fmaxf(sqrtf(ss), 1e-12f);
    auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };  // torch.sign
    if (in0) out[row * d + lane] = fmaf(sgn(y0) * n0, nsc, y0);
    if (in1) out[row * d + lane + 64] = fmaf(sgn(y1) * n1, nsc, y1);
}

}  // namespace rbg

using namespace rbg;

extern "C" {

int rbg_bpr_grad_f32(const float *out_mean, int64_t n_users, int64_t n_items, const int64_t *user, const int64_t *pos,
                     const int64_t *neg, int64_t B, int d, float *grad_mean, float *loss, void *stream) {
    clear_error();
    if (n_users < 0 || n_items < 0 || B < 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (!out_mean || !grad_mean || !loss || (B > 0 && (!user || !pos || !neg))) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    int zrc = zero_async(grad_mean, sizeof(float) * (size_t)(n_users + n_items) * d, s);
    if (zrc || (zrc = zero_async(loss, sizeof(float), s))) return zrc;
    if (B == 0) return RBG_OK;
    const BprArgs a{out_mean, n_users, user, pos, neg, B, d, 1e-10f, grad_mean};
    if (opt_deterministic()) {  // the loss summed in fixed point (order-free), every gradient row by the wavefront that owns it
        hipLaunchKernelGGL((bpr_grad_kernel<false>), grid_for(B), dim3(256), 0, s, a, loss, fix_slot(s));
        BprRows r{};
        r.user = user, r.pos = pos, r.neg = neg, r.n_users = n_users, r.B = B, r.a = a;
        launch_ordered_scatter(r, 3 * B, s);
    } else {
        hipLaunchKernelGGL((bpr_grad_kernel<true>), grid_for(B), dim3(256), 0, s, a, loss, -1);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_emb_reg_grad_f32(const float *user_emb, const float *item_emb, int64_t n_users, const int64_t *user,
                         const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight, float *grad_e0,
                         float *loss, void *stream) {
    clear_error();
    if (n_users < 0 || B < 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (B == 0 || reg_weight == 0.f) return RBG_OK;
    if (!user_emb || !item_emb || !user || !pos || !neg || !grad_e0 || !loss) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const EmbRegArgs a{user_emb, item_emb, n_users, user, pos, neg, B, d, grad_e0};
    if (opt_deterministic()) {
        hipLaunchKernelGGL((emb_reg_grad_kernel<false>), grid_for(3 * B), dim3(256), 0, s, a, reg_weight, loss, fix_slot(s));
        EmbRegRows r{};
        r.user = user, r.pos = pos, r.neg = neg, r.n_users = n_users, r.B = B, r.a = a, r.reg_weight = reg_weight, r.sums = nullptr;
        launch_ordered_scatter(r, 3 * B, s);
    } else {
        hipLaunchKernelGGL((emb_reg_grad_kernel<true>), grid_for(3 * B), dim3(256), 0, s, a, reg_weight, loss, -1);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_emb_reg_grad_nopow_f32(const float *user_emb, const float *item_emb, int64_t n_users, const int64_t *user,
                               const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight, float *grad_e0,
                               float *loss, float *workspace, void *stream) {
    clear_error();
    if (n_users < 0 || B < 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (B == 0 || reg_weight == 0.f) return RBG_OK;
    if (!user_emb || !item_emb || !user || !pos || !neg || !grad_e0 || !loss || !workspace) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (int zrc = zero_async(workspace, 3 * sizeof(float), s)) return zrc;
    const EmbRegArgs a{user_emb, item_emb, n_users, user, pos, neg, B, d, grad_e0};
    hipLaunchKernelGGL(emb_sumsq_kernel, grid_for(3 * B), dim3(256), 0, s, a, workspace, fix_slot(s));
    if (opt_deterministic()) {
        hipLaunchKernelGGL((emb_reg_grad_nopow_kernel<false>), dim3(1), dim3(256), 0, s, a, reg_weight, workspace, loss);
        EmbRegRows r{};
        r.user = user, r.pos = pos, r.neg = neg, r.n_users = n_users, r.B = B, r.a = a, r.reg_weight = reg_weight, r.sums = workspace;
        launch_ordered_scatter(r, 3 * B, s);
    } else {
        hipLaunchKernelGGL((emb_reg_grad_nopow_kernel<true>), grid_for(3 * B), dim3(256), 0, s, a, reg_weight, workspace, loss);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_concat_bpr_begin_f32(const float *const *tables, const int *widths, int n_tables, int64_t n_users, int64_t n_items,
                             const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B, int form, float *coef,
                             float *sums, float *loss, void *stream) {
    clear_error();
    if (form != 0 && form != 1) return fail(RBG_EINVAL, "form = %d (0: BPRLoss mean, 1: sum of -logsigmoid)", form);
    if (n_users < 0 || n_items < 0 || B < 0 || n_tables <= 0 || n_tables > RBG_MAX_CONCAT) return fail(RBG_ESHAPE, "bad shape (1..%d tables)", RBG_MAX_CONCAT);
    if (!tables || !widths || !sums || !loss || (B > 0 && (!user || !pos || !neg || !coef))) return fail(RBG_EINVAL, "NULL pointer");
    ConcatTables T{};
    T.n = n_tables;
    for (int t = 0; t < n_tables; ++t) {
        if (!tables[t] || widths[t] <= 0) return fail(RBG_EINVAL, "table %d: NULL or width <= 0", t);
        T.tab[t] = tables[t], T.width[t] = widths[t];
    }
    hipStream_t s = (hipStream_t)stream;
    int zrc = zero_async(sums, 3 * sizeof(float), s);
    if (zrc || (zrc = zero_async(loss, sizeof(float), s))) return zrc;
    if (B == 0) return RBG_OK;
    hipLaunchKernelGGL(concat_bpr_begin_kernel, grid_for(B), dim3(256), 0, s, T, n_users, user, pos, neg, B, 1e-10f, form, coef, sums, loss, fix_slot(s));
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_concat_bpr_scatter_f32(const float *table, int width, int64_t n_users, const int64_t *user, const int64_t *pos,
                               const int64_t *neg, int64_t B, float reg_weight, int require_pow, const float *coef,
                               const float *sums, float *grad_table, float *loss_reg, void *stream) {
    clear_error();
    if (n_users < 0 || B < 0 || width <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (B == 0) return RBG_OK;
    if (!table || !user || !pos || !neg || !coef || !sums || !grad_table) return fail(RBG_EINVAL, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const ConcatArgs a{table, width, n_users, user, pos, neg, B, reg_weight, require_pow ? 1 : 0, coef, sums, grad_table};
    if (opt_deterministic()) {
        hipLaunchKernelGGL((concat_bpr_scatter_kernel<false>), dim3(1), dim3(256), 0, s, a, loss_reg);
        ConcatRows r{};
        r.user = user, r.pos = pos, r.neg = neg, r.n_users = n_users, r.B = B, r.a = a;
        launch_ordered_scatter(r, 3 * B, s);
    } else {
        hipLaunchKernelGGL((concat_bpr_scatter_kernel<true>), grid_for(B), dim3(256), 0, s, a, loss_reg);
    }
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_adam_step_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                      float *exp_avg, float *exp_avg_sq, int64_t step, float lr, float beta1, float beta2, float eps,
                      void *stream) {
    clear_error();
    if (n_users < 0 || n_items < 0 || d <= 0 || step < 1) return fail(RBG_ESHAPE, "bad shape or step < 1");
    const int64_t nd = (n_users + n_items) * d;
    if (nd == 0) return RBG_OK;
    if (!grad || !exp_avg || !exp_avg_sq || (n_users && !user_emb) || (n_items && !item_emb)) return fail(RBG_EINVAL, "NULL pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const unsigned blocks = (unsigned)std::min<int64_t>((nd + 255) / 256, 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, user_emb, item_emb, n_users * d, grad,
                       exp_avg, exp_avg_sq, nd, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

static int adam_step_dev(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                         float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                         float eps, const float *loss, float *loss_total, void *stream);

int rbg_adam_step_dev_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                          float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                          float eps, void *stream) {
    return adam_step_dev(user_emb, item_emb, n_users, n_items, d, grad, exp_avg, exp_avg_sq, step, factors, lr, beta1, beta2, eps, nullptr,
                         nullptr, stream);
}

int rbg_adam_step_dev_total_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                                float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                                float eps, const float *loss, float *loss_total, void *stream) {
    if (!loss || !loss_total) {
        clear_error();
        return fail(RBG_EINVAL, "NULL pointer");
    }
    return adam_step_dev(user_emb, item_emb, n_users, n_items, d, grad, exp_avg, exp_avg_sq, step, factors, lr, beta1, beta2, eps, loss,
                         loss_total, stream);
}

static int adam_step_dev(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                         float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                         float eps, const float *loss, float *loss_total, void *stream) {
    clear_error();
    if (n_users < 0 || n_items < 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (d % 4) return fail(RBG_EUNSUPPORTED, "rbg_adam_step_dev_f32: d = %d is not a multiple of 4", d);
    if (!step || !factors) return fail(RBG_EINVAL, "NULL pointer");
    const int64_t nd = (n_users + n_items) * d;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, step, lr, beta1, beta2, factors, loss, loss_total);
    RBG_HIP(hipGetLastError());
    if (nd == 0) return RBG_OK;
    if (!grad || !exp_avg || !exp_avg_sq || (n_users && !user_emb) || (n_items && !item_emb)) return fail(RBG_EINVAL, "NULL pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>((nd / 4 + 255) / 256, 8192);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, s, user_emb, item_emb, n_users * d, grad, exp_avg, exp_avg_sq, nd,
                       factors, beta1, beta2, eps);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_lightgcn_step_head_f32(const float *out_mean, const float *user_emb, const float *item_emb, int64_t n_users, int64_t n_items,
                               const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight,
                               float *grad_mean, int32_t *row_count, int64_t *step, float *scratch, float *loss, float *loss_total,
                               float lr, float beta1, float beta2, void *stream) {
    clear_error();
    if (n_users < 0 || n_items < 0 || B <= 0 || d <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (!out_mean || !user_emb || !item_emb || !user || !pos || !neg || !grad_mean || !row_count || !step || !scratch || !loss)
        return fail(RBG_EINVAL, "NULL pointer");
    if (opt_deterministic()) return fail(RBG_EUNSUPPORTED, "the lean LightGCN step adds repeated rows with float atomics: use the separate calls in deterministic mode");
    LeanHeadArgs a{};
    a.bpr = BprArgs{out_mean, n_users, user, pos, neg, B, d, 1e-10f, grad_mean};
    a.user_emb = user_emb, a.item_emb = item_emb, a.n_nodes = n_users + n_items, a.reg_weight = reg_weight;
    a.row_count = row_count, a.step = step, a.scratch = scratch, a.loss = loss, a.loss_total = loss_total;
    a.lr = lr, a.beta1 = beta1, a.beta2 = beta2;
    hipLaunchKernelGGL(lgcn_head_kernel, grid_for(B), dim3(256), 0, (hipStream_t)stream, a);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_lightgcn_step_tail_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad_e0,
                               float *grad_mean, int32_t *row_count, float reg_weight, int64_t B, float *exp_avg, float *exp_avg_sq,
                               const int64_t *step, float *scratch, float lr, float beta1, float beta2, float eps, void *stream) {
    clear_error();
    if (n_users < 0 || n_items < 0 || d <= 0 || B <= 0) return fail(RBG_ESHAPE, "bad shape");
    if (d % 4) return fail(RBG_EUNSUPPORTED, "rbg_lightgcn_step_tail_f32: d = %d is not a multiple of 4", d);
    if (!user_emb || !item_emb || !grad_e0 || !grad_mean || !row_count || !exp_avg || !exp_avg_sq || !step || !scratch)
        return fail(RBG_EINVAL, "NULL pointer");
    LeanTailArgs a{};
    a.user_emb = user_emb, a.item_emb = item_emb, a.n_users_d = n_users * d, a.nd = (n_users + n_items) * d, a.n_nodes = n_users + n_items;
    a.d = d, a.grad_e0 = grad_e0, a.grad_mean = grad_mean, a.row_count = row_count, a.reg_over_b = reg_weight / (float)B;
    a.m = exp_avg, a.v = exp_avg_sq, a.step = step;
    a.factors = scratch + 4, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps;
    (void)lr;  // (in the factors the head wrote)
    const int quads = d >> 2;
    a.quads_shift = (quads & (quads - 1)) == 0 ? __builtin_ctz((unsigned)quads) : -1;
    if (a.nd == 0) return RBG_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((a.nd / 4 + 255) / 256, 8192);
    hipLaunchKernelGGL(lgcn_tail_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_sign_noise_f32(const float *Y, const float *noise, int64_t n, int d, float eps, float *out, void *stream) {
    clear_error();
    if (n < 0 || d <= 0) return fail(RBG_ESHAPE, "n = %lld, d = %d", (long long)n, d);
    if (d > 128) return fail(RBG_EUNSUPPORTED, "sign_noise: d = %d > 128", d);
    if (n == 0) return RBG_OK;
    if (!Y || !noise || !out) return fail(RBG_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(sign_noise_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Y, noise, n, d, eps, out);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

int rbg_once_mask_f32(const int64_t *ids, int64_t B, int64_t n_ids, int64_t *slot, int first_occurrence, int mean_form, float *once,
                      float *row_w, void *stream) {
    clear_error();
    if (B < 0 || n_ids < 0) return fail(RBG_ESHAPE, "B = %lld, n_ids = %lld", (long long)B, (long long)n_ids);
    if (B == 0) return RBG_OK;
    if (!ids || !slot || !once || !row_w) return fail(RBG_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(once_mask_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, B, reinterpret_cast<long long *>(slot),
                       first_occurrence, mean_form, once, row_w);
    RBG_HIP(hipGetLastError());
    return RBG_OK;
}

}  // extern "C"
