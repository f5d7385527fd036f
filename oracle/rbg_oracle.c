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
