// shard.hip — the node-range sharded propagation behind the C ABI (SURVEY.md §8(b),(e)): one process (or thread) per GPU,
// a rank owns a set of nodes; per layer the trimmed halo rows travel peer to peer over RCCL (grouped ncclSend / ncclRecv =
// an all-to-all-v over xGMI) on a second HIP stream while the interior product runs on the caller's stream.
// The reference is single-device (no distributed code at all): there is no reference interface to mirror; the layer it
// distributes is LightGCNConv (recbole_gnn/model/layers.py:13-20) and the forward is lightgcn.py:70-81.
//
// r06 — the fused layer (option "shard_fused", default 1): the rank's block [A_interior | A_halo] is ONE rectangular handle
// planned for the column-slab kernel (sell_plan.hip, rectangular form) over the table [owned rows | halo rows]; a layer is
// pack -> exchange (into the tail of that table) -> ONE launch, no accumulate pass over Y, every entry on the fast kernel
// (r05: the halo block — (P - 1) / P of the entries on an unstructured graph — ran spmm_binned_kernel and re-read Y).  The
// two-handle form (interior product beside the exchange on a second stream) stays behind the option for machines where the
// overlap pays.
//
// RCCL is bound at run time (dlopen of librccl.so.1, preferring a copy the process already loaded — PyTorch ships one), so
// librbgnn.so itself has no link-time dependency on it: a box without RCCL only loses these entry points.

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>
#include <vector>

#include "internal.h"

namespace rbg {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static const RcclApi *rccl() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // a copy already in the process (e.g. PyTorch's)
        for (const char *n : names) {
            if (h) break;
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) return;
        RcclApi a;
        a.lib = h;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.Send && a.Recv && a.GroupStart && a.GroupEnd) g_rccl = a;
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}

#define RBG_NCCL(api, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess)                                                                                     \
            return ::rbg::fail(RBG_EHIP, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "RCCL error"); \
    } while (0)

}  // namespace rbg

struct rbg_comm {
    int nranks = 0, rank = 0, device = -1;
    ncclComm_t comm = nullptr;
};

struct rbg_shard {
    rbg_comm *comm = nullptr;
    rbg_graph *g_int = nullptr, *g_halo = nullptr;
    rbg_graph *g_cat = nullptr;   // [A_interior | A_halo]: n_owned x (n_owned + n_halo), columns = rows of a cat table
    std::vector<float *> cat;     // cat[k]: [n_owned + n_halo][d] — layer k's input: the owned rows, then the halo rows as received
    int64_t n_owned = 0, n_halo = 0, n_send = 0;
    int d_max = 0;
    int64_t *d_send_idx = nullptr;
    float *d_send = nullptr, *d_halo = nullptr;
    std::vector<int64_t> send_counts, recv_counts;
    hipStream_t comm_stream = nullptr;
    hipEvent_t x_ready = nullptr, halo_ready = nullptr;
};

using namespace rbg;

// at least n cat tables on the shard ([n_owned + n_halo][d_max] floats each); grows outside stream captures only
static int cat_reserve(rbg_shard *s, size_t n, hipStream_t stream) {
    if (s->cat.size() >= n) return RBG_OK;
    if (stream) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            return fail(RBG_EUNSUPPORTED, "the sharded propagation needs %zu layer tables but holds %zu: run it once outside the capture", n, s->cat.size());
    }
    try {
        s->cat.reserve(n);
    } catch (const std::bad_alloc &) {
        return fail(RBG_ENOMEM, "out of host memory");
    }
    while (s->cat.size() < n) {
        float *b = nullptr;
        const int rc = to_device_raw((void **)&b, nullptr, sizeof(float) * (size_t)std::max<int64_t>(s->n_owned + s->n_halo, 1) * s->d_max);
        if (rc) return rc;
        s->cat.push_back(b);
    }
    return RBG_OK;
}

extern "C" {

int rbg_comm_unique_id(void *id) {
    clear_error();
    if (!id) return fail(RBG_EINVAL, "id is NULL");
    const RcclApi *api = rccl();
    if (!api) return fail(RBG_EUNSUPPORTED, "RCCL (librccl.so.1) is not available on this machine");
    static_assert(sizeof(ncclUniqueId) == RBG_COMM_ID_BYTES, "RBG_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    RBG_NCCL(api, api->GetUniqueId(reinterpret_cast<ncclUniqueId *>(id)));
    return RBG_OK;
}

int rbg_comm_create(rbg_comm **out, int nranks, int rank, const void *id, int device) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id || device < 0) return fail(RBG_EINVAL, "bad rank / nranks / id / device");
    const RcclApi *api = rccl();
    if (!api) return fail(RBG_EUNSUPPORTED, "RCCL (librccl.so.1) is not available on this machine");
    int rc = set_device_for(device);
    if (rc) return rc;
    rbg_comm *c = new (std::nothrow) rbg_comm();
    if (!c) return fail(RBG_ENOMEM, "out of host memory");
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(RBG_EHIP, "ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
    }
    *out = c;
    return RBG_OK;
}

void rbg_comm_destroy(rbg_comm *c) {
    if (!c) return;
    const RcclApi *api = rccl();
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    delete c;
}

void rbg_shard_destroy(rbg_shard *s) {
    if (!s) return;
    if (s->comm && hipSetDevice(s->comm->device) == hipSuccess) {
        if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
        (void)hipFree(s->d_send_idx);
        (void)hipFree(s->d_send);
        (void)hipFree(s->d_halo);
        for (float *c : s->cat) (void)hipFree(c);
        if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
        if (s->x_ready) (void)hipEventDestroy(s->x_ready);
        if (s->halo_ready) (void)hipEventDestroy(s->halo_ready);
    }
    rbg_graph_destroy(s->g_int);
    rbg_graph_destroy(s->g_halo);
    rbg_graph_destroy(s->g_cat);
    delete s;
}

int rbg_graph_create_sharded(rbg_shard **out, rbg_comm *comm, int64_t n_owned, int64_t n_users_owned, const int64_t *int_rowptr,
                             const int32_t *int_col, const float *int_val, int64_t n_halo, const int64_t *halo_rowptr,
                             const int32_t *halo_col, const float *halo_val, const int64_t *send_idx, const int64_t *send_counts,
                             const int64_t *recv_counts, int d_max) {
    clear_error();
    if (!out) return fail(RBG_EINVAL, "out is NULL");
    *out = nullptr;
    if (!comm) return fail(RBG_EINVAL, "comm is NULL");
    if (n_owned < 0 || n_halo < 0 || n_users_owned < 0 || n_users_owned > n_owned || d_max <= 0) return fail(RBG_EINVAL, "bad sizes");
    if (!int_rowptr || !send_counts || !recv_counts || (n_halo && !halo_rowptr)) return fail(RBG_EINVAL, "NULL plan array");
    int64_t n_send = 0, n_recv = 0;
    for (int q = 0; q < comm->nranks; ++q) {
        if (send_counts[q] < 0 || recv_counts[q] < 0) return fail(RBG_EINVAL, "negative count");
        n_send += send_counts[q];
        n_recv += recv_counts[q];
    }
    if (n_recv != n_halo) return fail(RBG_EINVAL, "recv_counts sum to %lld but n_halo = %lld", (long long)n_recv, (long long)n_halo);
    if (n_send && !send_idx) return fail(RBG_EINVAL, "send_idx is NULL");
    for (int64_t i = 0; i < n_send; ++i)
        if (send_idx[i] < 0 || send_idx[i] >= n_owned) return fail(RBG_EINVAL, "send_idx[%lld] out of range", (long long)i);
    int rc = set_device_for(comm->device);
    if (rc) return rc;
    rbg_shard *s = new (std::nothrow) rbg_shard();
    if (!s) return fail(RBG_ENOMEM, "out of host memory");
    s->comm = comm;
    s->n_owned = n_owned;
    s->n_halo = n_halo;
    s->n_send = n_send;
    s->d_max = d_max;
    try {
        s->send_counts.assign(send_counts, send_counts + comm->nranks);
        s->recv_counts.assign(recv_counts, recv_counts + comm->nranks);
    } catch (const std::bad_alloc &) {
        delete s;
        return fail(RBG_ENOMEM, "out of host memory");
    }
    // r06: [A_interior | A_halo] as one rectangular handle (row r: its interior entries, then its halo entries with the column
    // moved past the owned rows).  If the column-slab planner serves it the shard is fused-only: no second copy of the graph.
    rc = RBG_OK;
    if (opt_shard_fused() && n_halo && n_owned) {
        try {
            std::vector<int64_t> rp((size_t)n_owned + 1);
            const int64_t nnz = int_rowptr[n_owned] + halo_rowptr[n_owned];
            std::vector<int32_t> cc((size_t)nnz);
            std::vector<float> vv((size_t)nnz);
            int64_t o = 0;
            for (int64_t r = 0; r < n_owned; ++r) {
                rp[(size_t)r] = o;
                for (int64_t e = int_rowptr[r]; e < int_rowptr[r + 1]; ++e, ++o) cc[(size_t)o] = int_col[e], vv[(size_t)o] = int_val[e];
                for (int64_t e = halo_rowptr[r]; e < halo_rowptr[r + 1]; ++e, ++o)
                    cc[(size_t)o] = (int32_t)(n_owned + halo_col[e]), vv[(size_t)o] = halo_val[e];
            }
            rp[(size_t)n_owned] = o;
            rc = rbg_graph_create_csr_classes(&s->g_cat, n_owned, n_owned + n_halo, rp.data(), cc.data(), vv.data(), n_users_owned, comm->device, 0);
        } catch (const std::bad_alloc &) {
            rc = fail(RBG_ENOMEM, "out of host memory");
        }
        if (!rc && !(s->g_cat->sell && s->g_cat->sell->rect)) {  // not served by the planner (rbg_graph_sell_status says why): two handles
            rbg_graph_destroy(s->g_cat);
            s->g_cat = nullptr;
        }
    }
    // the two-handle form: both blocks get the user / item row classes (user rows gather item rows only, local or halo, and
    // vice versa)
    if (!rc && !s->g_cat) rc = rbg_graph_create_csr_classes(&s->g_int, n_owned, n_owned, int_rowptr, int_col, int_val, n_users_owned, comm->device, 0);
    if (!rc && !s->g_cat && n_halo)
        rc = rbg_graph_create_csr_classes(&s->g_halo, n_owned, n_halo, halo_rowptr, halo_col, halo_val, n_users_owned, comm->device, 0);
    if (!rc) rc = to_device_raw((void **)&s->d_send_idx, send_idx, sizeof(int64_t) * (size_t)n_send);
    if (!rc) rc = to_device_raw((void **)&s->d_send, nullptr, sizeof(float) * (size_t)std::max<int64_t>(n_send, 1) * d_max);
    if (!rc && !s->g_cat) rc = to_device_raw((void **)&s->d_halo, nullptr, sizeof(float) * (size_t)std::max<int64_t>(n_halo, 1) * d_max);
    if (!rc && s->g_cat) rc = cat_reserve(s, 2, nullptr);
    if (!rc) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) hi = 0;  // "greatest" priority is the numerically lowest
        if (hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&s->x_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->halo_ready, hipEventDisableTiming) != hipSuccess)
            rc = fail(RBG_EHIP, "stream / event creation failed");
    }
    if (rc) {
        rbg_shard_destroy(s);
        return rc;
    }
    *out = s;
    return RBG_OK;
}

// pack the rows the peers need of X ([n_owned, d]) and exchange: peer q's rows land at halo + (rows before q's) * d.  On `cs`.
static int shard_exchange(rbg_shard *s, const float *X, float *halo, int d, hipStream_t cs) {
    const RcclApi *api = rccl();
    if (!api) return fail(RBG_EUNSUPPORTED, "RCCL is not available");
    const int nranks = s->comm->nranks;
    int rc;
    if (s->n_send && (rc = rbg_gather_rows_f32(X, d, s->d_send_idx, s->d_send, s->n_send, d, cs))) return rc;
    RBG_NCCL(api, api->GroupStart());
    // inside the group the first error is remembered, not returned: the group must be closed on every path, or this rank's (and
    // its peers') next call would wait on a collective that was never issued
    ncclResult_t first = ncclSuccess;
    int64_t so = 0, ro = 0;
    for (int q = 0; q < nranks; ++q) {
        const int64_t sc = s->send_counts[(size_t)q], rcv = s->recv_counts[(size_t)q];
        ncclResult_t r = ncclSuccess;
        if (sc && first == ncclSuccess) r = api->Send(s->d_send + so * d, (size_t)(sc * d), ncclFloat, q, s->comm->comm, cs);
        if (r != ncclSuccess) first = r;
        if (rcv && first == ncclSuccess) r = api->Recv(halo + ro * d, (size_t)(rcv * d), ncclFloat, q, s->comm->comm, cs);
        if (r != ncclSuccess) first = r;
        so += sc;
        ro += rcv;
    }
    const ncclResult_t end = api->GroupEnd();
    if (first == ncclSuccess) first = end;
    if (first != ncclSuccess)
        return fail(RBG_EHIP, "halo exchange failed: %s", api->GetErrorString ? api->GetErrorString(first) : "RCCL error");
    return RBG_OK;
}

// One fused layer: exchange the halo of cat[k]'s owned rows into its tail, then Y = [A_int | A_halo] cat[k] in ONE launch (the last
// layer of a propagation: the layer mean in its epilogue).  Everything on the caller's stream.
static int shard_layer_fused(rbg_shard *s, const float *xcat, float *Y, int d, const float *const *srcs, int n_srcs, float *out_mean,
                             hipStream_t ms) {
    int rc;
    if (s->comm->nranks > 1 || s->n_send > 0)
        if ((rc = shard_exchange(s, xcat, const_cast<float *>(xcat) + s->n_owned * d, d, ms))) return rc;
    if (out_mean) return rbg_spmm_mean_f32(s->g_cat, xcat, nullptr, srcs, n_srcs, out_mean, d, ms);
    return rbg_spmm_f32(s->g_cat, xcat, Y, d, 0, ms);
}

// One sharded layer.  srcs / out_mean != NULL: the layer is the last of a propagation and carries the layer mean.
static int shard_layer(rbg_shard *s, const float *X, float *Y, int d, const float *const *srcs, int n_srcs, float *out_mean,
                       hipStream_t ms) {
    const int nranks = s->comm->nranks;
    const bool exchange = nranks > 1 || s->n_send > 0;
    // "shard_single_stream": pack and exchange on the caller's stream, no fork (nothing overlaps, but the layer is a plain
    // stream-ordered sequence — the form that HIP-graph capture takes)
    const bool forked = !opt_shard_single_stream();
    if (exchange) {
        hipStream_t cs = forked ? s->comm_stream : ms;
        if (forked) {
            RBG_HIP(hipEventRecord(s->x_ready, ms));
            RBG_HIP(hipStreamWaitEvent(cs, s->x_ready, 0));
        }
        const int rce = shard_exchange(s, X, s->d_halo, d, cs);
        if (forked) (void)hipEventRecord(s->halo_ready, cs);  // (recorded on every path: the next call waits on it)
        if (rce) return rce;
    }
    int rc;
    const bool last = out_mean != nullptr;
    if (last && !s->g_halo) return rbg_spmm_mean_f32(s->g_int, X, nullptr, srcs, n_srcs, out_mean, d, ms);
    if ((rc = rbg_spmm_f32(s->g_int, X, Y, d, 0, ms))) return rc;  // overlaps with the exchange on the comm stream
    if (exchange && forked) RBG_HIP(hipStreamWaitEvent(ms, s->halo_ready, 0));
    if (!s->g_halo) return RBG_OK;
    if (last) return rbg_spmm_mean_f32(s->g_halo, s->d_halo, Y, srcs, n_srcs, out_mean, d, ms);
    return rbg_spmm_f32(s->g_halo, s->d_halo, Y, d, 1, ms);
}

int rbg_shard_status(const rbg_shard *s, char *buf, int len) {
    clear_error();
    if (!s || !buf || len <= 0) return fail(RBG_EINVAL, "NULL argument");
    char a[160] = "none", b[160] = "none";
    if (s->g_cat) {
        (void)rbg_graph_sell_status(s->g_cat, a, sizeof a);
        snprintf(buf, (size_t)len, "fused: %s", a);
        return RBG_OK;
    }
    if (s->g_int) (void)rbg_graph_sell_status(s->g_int, a, sizeof a);
    if (s->g_halo) (void)rbg_graph_sell_status(s->g_halo, b, sizeof b);
    snprintf(buf, (size_t)len, "two handles: interior %s, halo %s", a, b);
    return RBG_OK;
}

int rbg_spmm_sharded_f32(rbg_shard *s, const float *X, float *Y, int d, void *stream) {
    clear_error();
    if (!s) return fail(RBG_EINVAL, "shard is NULL");
    if (d <= 0 || d > s->d_max) return fail(RBG_ESHAPE, "d = %d (the shard's buffers hold d <= %d)", d, s->d_max);
    if (s->n_owned == 0 && s->comm->nranks == 1) return RBG_OK;
    if (!X || !Y) return fail(RBG_EINVAL, "X or Y is NULL");
    if (X == Y) return fail(RBG_EINVAL, "X and Y alias");
    int rc = set_device_for(s->comm->device);
    if (rc) return rc;
    if (s->g_cat) {  // the owned rows go in front of the halo's landing zone (one copy of n_owned rows), then one launch
        hipStream_t ms = (hipStream_t)stream;
        RBG_HIP(hipMemcpyAsync(s->cat[0], X, sizeof(float) * (size_t)s->n_owned * d, hipMemcpyDeviceToDevice, ms));
        return shard_layer_fused(s, s->cat[0], Y, d, nullptr, 0, nullptr, ms);
    }
    return shard_layer(s, X, Y, d, nullptr, 0, nullptr, (hipStream_t)stream);
}

int rbg_lightgcn_forward_sharded_f32(rbg_shard *s, const float *E0, float *out_mean, float *layers, int d, int K, void *stream) {
    clear_error();
    if (!s) return fail(RBG_EINVAL, "shard is NULL");
    if (d <= 0 || d > s->d_max) return fail(RBG_ESHAPE, "d = %d (the shard's buffers hold d <= %d)", d, s->d_max);
    if (K < 1 || K - 1 > RBG_MAX_FUSED_LAYERS) return fail(RBG_EINVAL, "1 <= K <= %d", RBG_MAX_FUSED_LAYERS + 1);
    if (!E0 || !out_mean || !layers) return fail(RBG_EINVAL, "NULL pointer");
    int rc = set_device_for(s->comm->device);
    if (rc) return rc;
    hipStream_t ms = (hipStream_t)stream;
    const int64_t nd = s->n_owned * d;
    const float *srcs[RBG_MAX_FUSED_LAYERS + 1];
    if (s->g_cat) {  // layer k reads cat[k] = [E_k owned | E_k halo] and writes E_(k+1)'s owned rows into cat[k + 1]; `layers` is not used
        if ((rc = cat_reserve(s, (size_t)K, ms))) return rc;
        RBG_HIP(hipMemcpyAsync(s->cat[0], E0, sizeof(float) * (size_t)nd, hipMemcpyDeviceToDevice, ms));
        for (int k = 0; k < K; ++k) {
            srcs[k] = s->cat[(size_t)k];
            const bool last = (k == K - 1);
            if ((rc = shard_layer_fused(s, s->cat[(size_t)k], last ? nullptr : s->cat[(size_t)k + 1], d, last ? srcs : nullptr, last ? K : 0,
                                        last ? out_mean : nullptr, ms)))
                return rc;
        }
        return RBG_OK;
    }
    srcs[0] = E0;
    const float *x = E0;
    for (int k = 0; k < K; ++k) {
        float *y = layers + (int64_t)k * nd;  // layer k + 1 (the last one is scratch for the interior product)
        const bool last = (k == K - 1);
        if ((rc = shard_layer(s, x, y, d, last ? srcs : nullptr, last ? K : 0, last ? out_mean : nullptr, ms))) return rc;
        if (!last) srcs[k + 1] = y;
        x = y;
    }
    return RBG_OK;
}

}  // extern "C"
