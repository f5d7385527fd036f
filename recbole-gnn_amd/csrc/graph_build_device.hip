// graph_build_device.hip — the normalized-adjacency builder on the GPU.
//
// Same contract as the host builder (graph_build.cpp: replaces get_norm_adj_mat, recbole_gnn/data/dataset.py:49-79,
// and SGL's per-epoch view rebuild, sgl.py:107-126), but the O(E log E) part runs in HBM:
//   1. one pass over the interactions: range check, degree count (integer atomics -> deterministic), and the two
//      directed edges of every kept interaction as 64-bit keys (row << 32 | col); dropped interactions get ~0
//   2. rocPRIM radix sort of the 2E keys on the significant bits only -> entries grouped by row, columns ascending
//      (what torch_sparse's SparseTensor constructor does on the CPU)
//   3. rocPRIM exclusive scan of the degrees -> rowptr
//   4. col = low half of the key, val = (dis[row] * 1) * dis[col], dis = 1/sqrt(deg) with IEEE divide and sqrt
//      (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), i.e. the fp32 bits of PyG's gcn_norm on CPU
// Only rowptr comes back to the host (N+1 ints) for the launch plan (plan_bins).

#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "internal.h"

namespace rbg {

__global__ void make_keys_kernel(const int64_t *__restrict__ uid, const int64_t *__restrict__ iid,
                                 const uint8_t *__restrict__ keep, int64_t n_inter, int64_t n_users, int64_t n_items,
                                 unsigned long long *__restrict__ keys, int32_t *__restrict__ deg, int *__restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_inter; e += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long k0 = ~0ull, k1 = ~0ull;
        if (!keep || keep[e]) {
            const int64_t u = uid[e], it = iid[e];
            if (u < 0 || u >= n_users || it < 0 || it >= n_items) {
                atomicMax(bad, 1);
            } else {
                const unsigned long long r = (unsigned long long)u, c = (unsigned long long)(it + n_users);
                k0 = (r << 32) | c;  // user row lists the item
                k1 = (c << 32) | r;  // item row lists the user
                atomicAdd(deg + u, 1);
                atomicAdd(deg + it + n_users, 1);
            }
        }
        keys[2 * e] = k0;
        keys[2 * e + 1] = k1;
    }
}

__global__ void fill_csr_kernel(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ deg, int64_t nnz,
                                int32_t *__restrict__ col, float *__restrict__ val) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        const int32_t r = (int32_t)(k >> 32), c = (int32_t)(k & 0xffffffffu);
        const float dr = 1.0f / sqrtf((float)deg[r]);
        const float dc = 1.0f / sqrtf((float)deg[c]);
        col[i] = c;
        val[i] = (dr * 1.0f) * dc;  // every node on an edge has deg >= 1, so no inf -> 0 case arises here
    }
}

namespace {
struct DevBuf {  // frees on scope exit
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(RBG_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        return RBG_OK;
    }
};
int bits_for(int64_t n) {
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}
}  // namespace

int build_device_csr(rbg_graph *g, int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                     const int64_t *iid, const uint8_t *keep) {
    if (n_users < 0 || n_items < 0 || n_inter < 0) return fail(RBG_EINVAL, "negative size");
    if (n_inter > 0 && (!uid || !iid)) return fail(RBG_EINVAL, "uid/iid is NULL");
    const int64_t n = n_users + n_items;
    if (n >= (int64_t)INT32_MAX) return fail(RBG_EUNSUPPORTED, "node count %lld >= 2^31", (long long)n);
    if (2 * n_inter >= (int64_t)INT32_MAX)
        return fail(RBG_EUNSUPPORTED, "nnz %lld >= 2^31 (int32 rowptr build)", (long long)(2 * n_inter));
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(RBG_ENODEV, "a device graph was requested but no GPU is visible");
    if (g->device >= n_dev) return fail(RBG_EINVAL, "device %d out of range (%d visible)", g->device, n_dev);
    int rc = set_device_for(g->device);
    if (rc) return rc;
    hipStream_t s = nullptr;

    DevBuf d_uid, d_iid, d_keep, d_keys, d_sorted, d_deg, d_bad, d_tmp;
    const size_t e = (size_t)n_inter;
    if ((rc = d_uid.alloc(e * 8)) || (rc = d_iid.alloc(e * 8)) || (rc = d_keys.alloc(2 * e * 8)) ||
        (rc = d_sorted.alloc(2 * e * 8)) || (rc = d_deg.alloc(((size_t)n + 1) * 4)) || (rc = d_bad.alloc(4)))
        return rc;
    if (keep && (rc = d_keep.alloc(e))) return rc;
    if (e) {
        RBG_HIP(hipMemcpyAsync(d_uid.p, uid, e * 8, hipMemcpyDefault, s));
        RBG_HIP(hipMemcpyAsync(d_iid.p, iid, e * 8, hipMemcpyDefault, s));
        if (keep) RBG_HIP(hipMemcpyAsync(d_keep.p, keep, e, hipMemcpyDefault, s));
    }
    RBG_HIP(hipMemsetAsync(d_deg.p, 0, ((size_t)n + 1) * 4, s));
    RBG_HIP(hipMemsetAsync(d_bad.p, 0, 4, s));
    if (e) {
        const unsigned blocks = (unsigned)std::min<int64_t>((n_inter + 255) / 256, 8192);
        hipLaunchKernelGGL(make_keys_kernel, dim3(blocks), dim3(256), 0, s, (const int64_t *)d_uid.p, (const int64_t *)d_iid.p,
                           (const uint8_t *)d_keep.p, n_inter, n_users, n_items, (unsigned long long *)d_keys.p,
                           (int32_t *)d_deg.p, (int *)d_bad.p);
        RBG_HIP(hipGetLastError());
    }
    int bad = 0;
    RBG_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    RBG_HIP(hipStreamSynchronize(s));
    if (bad) return fail(RBG_EINVAL, "an interaction id is out of range (uid in [0,%lld), iid in [0,%lld))", (long long)n_users, (long long)n_items);

    // rowptr = exclusive scan of deg over N+1 entries (the extra zero makes rowptr[N] = nnz)
    rc = to_device_raw((void **)&g->d_rowptr, nullptr, ((size_t)n + 1) * 4);
    if (rc) return rc;
    size_t tmp_bytes = 0;
    RBG_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, (int32_t *)d_deg.p, g->d_rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), s));
    size_t sort_bytes = 0;
    const int row_bits = bits_for(std::max<int64_t>(n, 2));
    if (e)
        RBG_HIP(rocprim::radix_sort_keys(nullptr, sort_bytes, (unsigned long long *)d_keys.p, (unsigned long long *)d_sorted.p, 2 * e, 0, 64, s));
    if ((rc = d_tmp.alloc(std::max(tmp_bytes, sort_bytes)))) return rc;
    RBG_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, (int32_t *)d_deg.p, g->d_rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), s));
    if (e) {
        // Dropped interactions carry ~0 keys; they must sort last, so when a mask is present all 64 bits take part,
        // otherwise only the significant bits of (row, col).
        const unsigned begin = 0, end = keep ? 64u : (unsigned)(32 + row_bits);
        RBG_HIP(rocprim::radix_sort_keys(d_tmp.p, sort_bytes, (unsigned long long *)d_keys.p, (unsigned long long *)d_sorted.p, 2 * e,
                                         begin, end, s));
    }
    g->h_rowptr.resize((size_t)n + 1);
    RBG_HIP(hipMemcpyAsync(g->h_rowptr.data(), g->d_rowptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, s));
    RBG_HIP(hipStreamSynchronize(s));
    g->n_rows = g->n_cols = n;
    g->n_users = n_users;
    g->nnz = g->h_rowptr[(size_t)n];
    if ((rc = to_device_raw((void **)&g->d_col, nullptr, (size_t)g->nnz * 4))) return rc;
    if ((rc = to_device_raw((void **)&g->d_val, nullptr, (size_t)g->nnz * 4))) return rc;
    if (g->nnz) {
        const unsigned blocks = (unsigned)std::min<int64_t>((g->nnz + 255) / 256, 16384);
        hipLaunchKernelGGL(fill_csr_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned long long *)d_sorted.p,
                           (const int32_t *)d_deg.p, g->nnz, g->d_col, g->d_val);
        RBG_HIP(hipGetLastError());
    }
    RBG_HIP(hipStreamSynchronize(s));
    return RBG_OK;
}

// map[e] = position of the transposed entry of CSR entry e = (r, c): the k-th entry (r, c) of row r pairs with the k-th
// entry (c, r) of row c (rows are column-sorted; duplicated interactions stay separate entries).  *bad counts entries
// whose transpose is missing (the structure is not symmetric).
__global__ void transpose_map_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t n_rows,
                                     int64_t nnz, int32_t *__restrict__ map, int *bad) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        // row of entry e: upper bound of e in rowptr
        int64_t lo = 0, hi = n_rows;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rowptr[mid + 1] <= e) lo = mid + 1; else hi = mid;
        }
        const int32_t r = (int32_t)lo, c = col[e];
        auto lower = [&](int32_t row, int32_t key) {
            int32_t a = rowptr[row], b = rowptr[row + 1];
            while (a < b) {
                const int32_t m = (a + b) >> 1;
                if (col[m] < key) a = m + 1; else b = m;
            }
            return a;
        };
        const int32_t k = (int32_t)e - lower(r, c);
        int32_t t = -1;
        if (c < n_rows) {
            t = lower(c, r) + k;
            if (t >= rowptr[c + 1] || col[t] != r) t = -1;
        }
        if (t < 0) atomicAdd(bad, 1);
        map[e] = t < 0 ? 0 : t;
    }
}

}  // namespace rbg

extern "C" int rbg_graph_transpose_map(const rbg_graph *g, int32_t *map, void *stream) {
    using namespace rbg;
    clear_error();
    if (!g) return fail(RBG_EINVAL, "graph is NULL");
    if (g->device < 0) return fail(RBG_ENODEV, "rbg_graph_transpose_map needs a device graph");
    if (g->n_rows != g->n_cols) return fail(RBG_ESHAPE, "graph is not square");
    if (g->nnz == 0) return RBG_OK;
    if (!map) return fail(RBG_EINVAL, "map is NULL");
    int rc = set_device_for(g->device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    int *bad = nullptr;
    RBG_HIP(hipMalloc((void **)&bad, sizeof(int)));
    hipError_t e1 = hipMemsetAsync(bad, 0, sizeof(int), s);
    const unsigned blocks = (unsigned)std::min<int64_t>((g->nnz + 255) / 256, 16384);
    hipLaunchKernelGGL(transpose_map_kernel, dim3(blocks), dim3(256), 0, s, g->d_rowptr, g->d_col, g->n_rows, g->nnz, map, bad);
    int h_bad = 0;
    hipError_t e2 = hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s);
    hipError_t e3 = hipStreamSynchronize(s);
    (void)hipFree(bad);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(RBG_EHIP, "transpose map failed: %s", hipGetErrorString(e3 != hipSuccess ? e3 : (e2 != hipSuccess ? e2 : e1)));
    if (h_bad) return fail(RBG_EINVAL, "%d entries have no transposed partner: the graph's structure is not symmetric", h_bad);
    return RBG_OK;
}
