/*
 * rbg_oracle.c — plain-C restatement of the reference's CPU path for the LightGCN propagation.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker / the timed CPU baseline ("kind": "port").  The product never links it.
 *
 * PARITY UNPINNED: the algorithm lives in third-party packages that are absent from
 * /root/reference and un-pinned by it (torch_sparse: SparseTensor + matmul -> csrc/cpu/spmm_cpu.cpp;
 * torch_geometric: gcn_norm).  This file restates their published algorithms from the reference's
 * call sites:
 *   ora_build_norm_csr   <- recbole_gnn/data/dataset.py:41-47,60-75  (SparseTensor(...).t(), gcn_norm)
 *   ora_spmm_csr_f32     <- recbole_gnn/model/layers.py:19-20        (torch_sparse.matmul, reduce='add')
 *   ora_lightgcn_forward <- recbole_gnn/model/general_recommender/lightgcn.py:70-81
 * int64 indices and fp32 values exactly like torch_sparse; multiply then add (no FMA contraction:
 * build with -ffp-contract=off), entries of a row visited in column order, rows in parallel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ora_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* Normalized adj_t as CSR.  keep may be NULL.  rowptr: N+1, col/val: 2*kept.  Returns nnz or -1. */
int64_t ora_build_norm_csr(int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                           const int64_t *iid, const uint8_t *keep, int64_t *rowptr, int64_t *col,
                           float *val) {
    const int64_t n = n_users + n_items;
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    if (!cnt) return -1;
    for (int64_t e = 0; e < n_inter; ++e) {
        if (keep && !keep[e]) continue;
        cnt[uid[e]]++;
        cnt[iid[e] + n_users]++;
    }
    rowptr[0] = 0;
    for (int64_t r = 0; r < n; ++r) rowptr[r + 1] = rowptr[r] + cnt[r];
    const int64_t nnz = rowptr[n];
    /* deg = row sums of the all-ones matrix (gcn_norm SparseTensor form); dis = deg^-0.5, inf -> 0.
       ATen's CPU pow(x, -0.5) is 1/sqrt(x) with IEEE sqrt and divide. */
    float *dis = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    if (!dis) { free(cnt); return -1; }
    for (int64_t r = 0; r < n; ++r) {
        float deg = (float)cnt[r];
        float s = 1.0f / sqrtf(deg);
        dis[r] = isinf(s) ? 0.0f : s;
    }
    memset(cnt, 0, sizeof(int64_t) * ((size_t)n + 1));
    for (int64_t e = 0; e < n_inter; ++e) {
        if (keep && !keep[e]) continue;
        int64_t u = uid[e], i = iid[e] + n_users;
        col[rowptr[u] + cnt[u]++] = i; /* row u (target) receives from source i */
        col[rowptr[i] + cnt[i]++] = u;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n; ++r) {
        int64_t b = rowptr[r], len = rowptr[r + 1] - b;
        if (len > 1) qsort(col + b, (size_t)len, sizeof(int64_t), cmp_i64);
        for (int64_t e = b; e < b + len; ++e) {
            float w = 1.0f;
            w = w * dis[r];      /* mul(adj_t, dis.view(-1,1)) */
            w = w * dis[col[e]]; /* mul(adj_t, dis.view(1,-1)) */
            val[e] = w;
        }
    }
    free(dis);
    free(cnt);
    return nnz;
}

/* out[m,:] = sum_e val[e] * mat[col[e],:]   (spmm_cpu loop order) */
/* target_clones: an AVX2 body is picked at load time where the host has it (the k loop vectorises across k, which does
 * not change any element's operation order: results are bit-identical to the scalar clone). */
__attribute__((target_clones("avx2", "default")))
void ora_spmm_csr_f32(int64_t M, int64_t K, const int64_t *restrict rowptr, const int64_t *restrict col,
                      const float *restrict val, const float *restrict mat, float *restrict out) {
#pragma omp parallel
    {
        float *restrict vals = (float *)malloc(sizeof(float) * (size_t)(K ? K : 1));
#pragma omp for schedule(dynamic, 64)
        for (int64_t m = 0; m < M; ++m) {
            for (int64_t k = 0; k < K; ++k) vals[k] = 0.0f;
            for (int64_t e = rowptr[m]; e < rowptr[m + 1]; ++e) {
                const int64_t c = col[e];
                const float v = val[e];
                const float *restrict src = mat + c * K;
                for (int64_t k = 0; k < K; ++k) vals[k] += v * src[k];
            }
            memcpy(out + m * K, vals, sizeof(float) * (size_t)K);
        }
        free(vals);
    }
}

/* lightgcn.py:70-81: E0 = cat(user, item); K x spmm; mean over the K+1 layers.
 * layers: [K+1][N][d] scratch provided by the caller (layer 0 = E0 copy). */
void ora_lightgcn_forward_f32(int64_t n_users, int64_t n_items, int64_t d, int n_layers,
                              const int64_t *rowptr, const int64_t *col, const float *val,
                              const float *user_w, const float *item_w, float *layers, float *out_mean) {
    const int64_t n = n_users + n_items, nd = n * d;
    memcpy(layers, user_w, sizeof(float) * (size_t)(n_users * d));
    memcpy(layers + n_users * d, item_w, sizeof(float) * (size_t)(n_items * d));
    for (int k = 0; k < n_layers; ++k)
        ora_spmm_csr_f32(n, d, rowptr, col, val, layers + (int64_t)k * nd, layers + (int64_t)(k + 1) * nd);
    const float denom = (float)(n_layers + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nd; ++i) {
        float s = layers[i];
        for (int k = 1; k <= n_layers; ++k) s += layers[(int64_t)k * nd + i];
        out_mean[i] = s / denom;
    }
}

/* ---- the same propagation with NUMA-aware placement (bench.py's cpu_baseline only; r05) -------------------
 * The loop above hands rows out dynamically and reads arrays whose pages sit where the Python thread first touched them: on a
 * two-socket host it stops scaling at 16 of 256 logical CPUs.  Here every thread owns ONE contiguous block of rows (blocks of
 * equal entry counts), and the arrays it streams — its block of col / val, its rows of every layer and of the mean — are
 * allocated untouched and first written by that thread, so their pages land on its NUMA node (run with OMP_PROC_BIND=spread /
 * OMP_PLACES=cores).  The arithmetic is the loop's above, entry for entry: results are bit-identical. */
typedef struct {
    int64_t n_users, n_items, d;
    int n_layers, n_threads;
    int64_t *row_lo;  /* [n_threads + 1] */
    int64_t *rowptr, *col;
    float *val, *layers, *mean;
} ora_numa;

void ora_numa_free(ora_numa *h) {
    if (!h) return;
    free(h->row_lo); free(h->rowptr); free(h->col); free(h->val); free(h->layers); free(h->mean);
    free(h);
}

ora_numa *ora_numa_prepare(int64_t n_users, int64_t n_items, int64_t d, int n_layers, const int64_t *rowptr, const int64_t *col,
                           const float *val) {
    const int64_t n = n_users + n_items, nnz = rowptr[n];
    ora_numa *h = (ora_numa *)calloc(1, sizeof(ora_numa));
    if (!h) return NULL;
    h->n_users = n_users, h->n_items = n_items, h->d = d, h->n_layers = n_layers, h->n_threads = ora_num_threads();
    const int T = h->n_threads;
    h->row_lo = (int64_t *)malloc(sizeof(int64_t) * (size_t)(T + 1));
    h->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    h->col = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    h->val = (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1));
    h->layers = (float *)malloc(sizeof(float) * (size_t)((n_layers + 1) * n * d + 1));
    h->mean = (float *)malloc(sizeof(float) * (size_t)(n * d + 1));
    if (!h->row_lo || !h->rowptr || !h->col || !h->val || !h->layers || !h->mean) { ora_numa_free(h); return NULL; }
    h->row_lo[0] = 0;
    for (int t = 1; t <= T; ++t) {  /* the first row whose entries start at or beyond t / T of all entries */
        const int64_t target = (int64_t)((double)nnz * t / T);
        int64_t lo = h->row_lo[t - 1], hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (rowptr[mid] < target) lo = mid + 1; else hi = mid; }
        h->row_lo[t] = t == T ? n : lo;
    }
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t r0 = h->row_lo[t], r1 = h->row_lo[t + 1];
        for (int64_t r = r0; r < r1; ++r) h->rowptr[r] = rowptr[r];
        if (t == T - 1) h->rowptr[n] = rowptr[n];
        for (int64_t e = rowptr[r0]; e < rowptr[r1]; ++e) h->col[e] = col[e], h->val[e] = val[e];
        for (int k = 0; k <= n_layers; ++k) memset(h->layers + ((int64_t)k * n + r0) * d, 0, sizeof(float) * (size_t)((r1 - r0) * d));
        memset(h->mean + r0 * d, 0, sizeof(float) * (size_t)((r1 - r0) * d));
    }
    return h;
}

__attribute__((target_clones("avx2", "default")))
static void ora_numa_rows(int64_t r0, int64_t r1, int64_t K, const int64_t *restrict rowptr, const int64_t *restrict col, const float *restrict val,
                          const float *restrict mat, float *restrict out, float *restrict vals) {
    for (int64_t m = r0; m < r1; ++m) {
        for (int64_t k = 0; k < K; ++k) vals[k] = 0.0f;
        for (int64_t e = rowptr[m]; e < rowptr[m + 1]; ++e) {
            const int64_t c = col[e];
            const float v = val[e];
            const float *restrict src = mat + c * K;
            for (int64_t k = 0; k < K; ++k) vals[k] += v * src[k];
        }
        memcpy(out + m * K, vals, sizeof(float) * (size_t)K);
    }
}

/* out_mean may be NULL (the timing loop): the mean stays in the handle's own, thread-placed buffer */
void ora_numa_forward_f32(ora_numa *h, const float *user_w, const float *item_w, float *out_mean) {
    const int64_t n = h->n_users + h->n_items, d = h->d, nd = n * d;
    const int T = h->n_threads, K = h->n_layers;
    const float denom = (float)(K + 1);
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t r0 = h->row_lo[t], r1 = h->row_lo[t + 1];
        float *vals = (float *)malloc(sizeof(float) * (size_t)(d ? d : 1));
        for (int64_t r = r0; r < r1; ++r)
            memcpy(h->layers + r * d, r < h->n_users ? user_w + r * d : item_w + (r - h->n_users) * d, sizeof(float) * (size_t)d);
        for (int k = 0; k < K; ++k) {
#pragma omp barrier
            ora_numa_rows(r0, r1, d, h->rowptr, h->col, h->val, h->layers + (int64_t)k * nd, h->layers + (int64_t)(k + 1) * nd, vals);
        }
        for (int64_t i = r0 * d; i < r1 * d; ++i) {
            float s = h->layers[i];
            for (int k = 1; k <= K; ++k) s += h->layers[(int64_t)k * nd + i];
            h->mean[i] = s / denom;
        }
        free(vals);
    }
    if (out_mean) memcpy(out_mean, h->mean, sizeof(float) * (size_t)nd);
}
