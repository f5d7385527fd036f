/*
 * rbgnn.h — C ABI of librbgnn.so, the MI355X (gfx950) LightGCN / NGCF propagation engine.
 *
 * This is the drop-in boundary for ONE path of RUCAIBox/RecBole-GNN: the normalized-adjacency
 * message passing  E(k+1) = Â · E(k)  behind GeneralGraphRecommender
 * (get_norm_adj_mat / forward / full_sort_predict).  The reference has no FFI of its own (it is
 * Python calling torch_sparse / PyG); every entry point below names the reference interface it
 * replaces (paths are relative to the reference repo root).
 *
 * Conventions
 *   - plain C types only; no torch types cross this boundary.
 *   - every function returns RBG_OK (0) or a negative RBG_E* code; the message for the calling
 *     thread is available from rbg_last_error().  Nothing aborts or throws across the ABI.
 *   - device pointers are fp32, row-major, contiguous; the caller (PyTorch) owns every embedding /
 *     score buffer.  The library owns graph handles and their HBM arrays.
 *   - `stream` is a hipStream_t (0 = the null stream).  Kernels are enqueued and the call returns
 *     without synchronizing.  No allocation happens on a hot call.
 *   - a graph handle is immutable after creation except for a small per-handle scratch used by rows longer
 *     than `seg_len`: calls on ONE handle must be stream-ordered with each other; different handles are independent.
 */
#ifndef RBGNN_H
#define RBGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBG_ABI_VERSION 1

/* error codes */
#define RBG_OK            0
#define RBG_EINVAL       (-1)  /* bad argument (null pointer, negative size, id out of range) */
#define RBG_ENOMEM       (-2)  /* host or device allocation failed */
#define RBG_EHIP         (-3)  /* a HIP runtime call failed; message carries hipGetErrorString */
#define RBG_ESHAPE       (-4)  /* shape mismatch between the graph and the dense operands */
#define RBG_ENODEV       (-5)  /* a device graph/op was requested but no GPU is visible */
#define RBG_EUNSUPPORTED (-6)  /* valid request this build does not implement (e.g. nnz >= 2^31) */

/* rbg_graph_create* flags */
#define RBG_GRAPH_DEFAULT        0u
#define RBG_GRAPH_KEEP_HOST      1u  /* keep the host CSR next to the device copy (export without D2H) */
#define RBG_GRAPH_BUILD_ON_HOST  2u  /* force the host (C++) builder even for a device graph */
#define RBG_GRAPH_NATURAL_ORDER  4u  /* do not degree-bin rows; rows are processed in id order */
#define RBG_GRAPH_INPUTS_ON_DEVICE 8u /* uid / iid / keep are DEVICE pointers on the graph's GPU (device builder only; work
                                         that produced them must be complete or ordered before the null stream): an SGL view
                                         sampled on the device is built without the interactions crossing PCIe again */

/* rbg_lightgcn_forward_f32 flags */
#define RBG_FWD_DEFAULT          0u
#define RBG_FWD_KEEP_LAST_LAYER  1u  /* also write E_K into layers[K-1] (autograd needs only E_1..E_{K-1}) */
#define RBG_FWD_LAYERS_SCRATCH   2u  /* the caller will not read `layers`: the library may keep any layout in it (option "slab") */

/* rbg_bignn_conv_f32 flags */
#define RBG_BIGNN_CONV_ONLY      0u  /* exactly BiGNNConv.forward */
#define RBG_BIGNN_LEAKY_NORM     1u  /* + LeakyReLU(slope) + row L2-normalize (NGCF.forward's per-layer tail) */

#define RBG_MAX_FUSED_LAYERS     8   /* K up to this is fused into K launches; larger K still works (unfused mean) */

typedef struct rbg_graph rbg_graph;

/* ---------------------------------------------------------------------------------------------
 * library
 * ------------------------------------------------------------------------------------------- */
int         rbg_abi_version(void);
const char *rbg_last_error(void);           /* thread-local; never NULL */
int         rbg_device_count(int *count);   /* RBG_OK with *count = 0 when no GPU is visible */

/* Tuning knobs of the degree-binned SpMM (process-wide; read when a graph is created).
 *   short_max : rows with degree <= short_max are handled one row per lane-group (D/4 lanes)
 *   wave_max  : rows with degree <= wave_max by one wavefront; longer rows by a whole workgroup
 *   seg_len   : workgroup rows longer than this are split into seg_len-entry segments whose
 *               partial sums are combined in fixed order by the last segment to finish
 * Pass a negative value to keep a knob unchanged. */
int rbg_set_tuning(int short_max, int wave_max, int seg_len);
int rbg_get_tuning(int *short_max, int *wave_max, int *seg_len);

/* Named process-wide options (experimentation / A-B runs; defaults are the tuned values).
 *   "spmm_unroll" : neighbour rows in flight per lane-group before the FMAs (4 or 8)
 *   "xcd_split"   : S in 1..7 = user rows on XCDs [0,S), item rows on XCDs [S,8) (graphs built from interactions;
 *                   read at graph creation; default 4), 0 = one row class on all XCDs
 *   "nt_store"    : 1 = non-temporal output stores
 *   "col_split"   : SpMM column-half mode (even / odd XCDs own the lower / upper half of the columns of one row class,
 *                   halving the per-XCD gather working set): -1 = auto (d = 128 and a table <= 512 MB), 0 = off, 1 = on where eligible
 *   "bignn_dma"   : the NGCF configuration (d_in = 64; forward d_out in {16, 32, 48, 64}, backward d_out = 64): 1 (default) =
 *                   row tiles arrive by LDS-DMA (forward: weights in registers, one software-pipelined wave per SIMD,
 *                   16x16x4 fp32 MFMA on Y^T; backward: input AND weight gradients in one kernel); 0 = the general kernels
 *   "mfma_split"  : 1 (default) = rbg_score_f32 / rbg_full_sort_topk_f32 split both fp32 operands into three bf16 terms and
 *                   form the six products of order >= 2^-16 on the bf16 matrix cores (fp32-MFMA accuracy at 2.3x its
 *                   rate); 0 = the exact-fp32 MFMA chain
 *   "score_uniform" : 1 (default) = rbg_score_f32 with B >= 64 q users (q = 32 / gcd(n mod 32, 32)) gives every workgroup the users of
 *                   ONE alignment class of the contiguous [B, n] output and shifts its item tiles to that class's line boundary:
 *                   whole aligned lines are stored straight from the accumulators (r04: 178 -> 156 us at 4096 x 40 982 x 64);
 *                   0 = the shifted store stream (16 cross-lane reads per tile), which small batches always use
 *   "lse_onepass" : 1 (default) = rbg_infonce_f32 with gradients takes the denominators and the batch rows' gradient out of ONE
 *                   pass over the table (the unnormalised gradient accumulates beside the denominator; no separate forward
 *                   launch); 0 = forward launch + two gradient launches
 *   "lse_tr_read" : 1 (default) = the gradient launches of rbg_infonce*_f32 / rbg_lse_rows_backward_f32 read the second product's B
 *                   fragments out of the row-major bf16 planes with gfx950's LDS transpose read (ds_read_b64_tr_b16) instead of
 *                   publishing a transposed copy of every tile (r06: InfoNCE forward + backward at 2048 x 40 982 x 64
 *                   354 -> 330 us, at 2048 x 91 600 x 128 1 790 -> 1 396 us; same results bit for bit); 0 = the transposed copy
 *   "lse_f16"     : 3 (default), 1, 2 = the two gradient launches of rbg_infonce_f32 (NOT the weighted rbg_infonce_masked_f32, NOT
 *                   rbg_lse_rows*_f32 — their operands are the caller's numbers) run both products on the fp16 matrix cores with TWO
 *                   terms per operand: the rows are unit rows (|x| <= 1, scaled by 2^8) and the weights are exp(.) <= 1 resp.
 *                   softmax probabilities (scaled by 2^14; the call's weight / tau is divided out and multiplied back), so
 *                   x = h + l to 2^-22 |x| without leaving fp16's range, and three products (h h, h l, l h) replace the six of
 *                   the three-term bf16 form: half the matrix-core time, two LDS planes instead of three, 32 instead of 48
 *                   fragment registers — three workgroups per CU at d <= 64, two at d = 128 (one before); the row kernels that
 *                   normalise the table and the batch rows write the two fp16 planes as tile images and the launches take
 *                   their tiles by LDS-DMA (2 = every workgroup fetches, splits and publishes its tiles: 221 vs 212 us); 3 = as 1
 *                   with the tile loop software-pipelined inside a wave (the first product of tile t issued in front of the exp2 /
 *                   split of tile t - 1, three LDS tiles: 212 -> 209 us).  r06:
 *                   forward + backward 333 -> 209 us at 2048 x 40 982 x 64, 1 404 -> 686 us at 2048 x 91 600 x 128; errors against float64
 *                   autograd unchanged (~ 2e-7 of the largest gradient entry; tests at 1e-5 for both forms,
 *                   profiles/r06_lse_f16.jsonl).  Needs "mfma_split" = 1, "lse_tr_read" = 1, "lse_image" = 0, d % 4 == 0; 0 = the bf16 form
 *   "lse_image"   : 0 (default) ; 1 = the gradient launches of rbg_infonce*_f32 take the tiles of the normalised table / batch rows from
 *                   plane images (built once per call, as "topk_image") by LDS-DMA, three workgroups per CU at d <= 64 with the
 *                   chunking tuned for three.  Same values to the last chunk-summation bit; measured -4 % at 40 982 / 91 600 table
 *                   rows, +3 % at 29 858 / 31 669 (profiles/r06_lse_image_3wg.jsonl): off
 *   "sell_wpb"    : waves per workgroup of the column-slab launch: 1 (default), 2 or 4 — same results, 1 measured best everywhere (r06)
 *   "sell_c16"    : 1 (default) = the launches of the factored chain read their entries as 16-bit slab-row numbers where the plan
 *                   has them (both row classes below 65 536 rows: 2 instead of 4 bytes per entry; r05: 91.6 -> 89.6 us per
 *                   propagation at the Gowalla shape, 124.7 -> 121.1 at Yelp2018); 0 = 32-bit offsets.  Same bits.
 *   "deterministic" : 0 (default) = the mini-batch kernels (rbg_bpr_grad_f32, rbg_emb_reg_grad[_nopow]_f32, rbg_concat_bpr_*_f32,
 *                   rbg_infonce[_masked]_f32 with gradients) add the rows of repeated ids with float atomics, like torch's GPU
 *                   index_put_(accumulate=True): the low bits of those rows and of the loss sums vary from run to run;
 *                   1 = every such row is written by ONE wavefront (the one of the id's first occurrence) that applies the
 *                   occurrences in batch order, and loss values / block norms are summed across workgroups as 31.32 fixed-point
 *                   integers (integer addition does not depend on the order; |sum| < 2^31, absolute resolution 2^-32 of the
 *                   sum before its weight — a regulariser's reg_weight / B multiplies the total, not the addends): bit-identical from
 *                   run to run, no scratch memory, capturable; costs a scan of the batch's ids per wavefront (csrc/ordered.h).
 *                   The sums use one slot of a device table per stream and one per stream capture (a captured step keeps its own
 *                   for every replay); 256 such streams / captured graphs are live at a time, the oldest slot is recycled after
 *                   that.  A captured step has the mode it was captured in (train.py re-captures on a toggle).
 *                   The propagation, scoring and top-k kernels are bit-stable in either mode.
 *   "score_tiles" : item tiles one workgroup of rbg_score_f32 walks (0 = auto: whole rounds of resident workgroups)
 *   "topk_short_lists" : rbg_full_sort_topk_f32 at k <= 12, d <= 64 keeps 24-entry candidate lists (three workgroups per CU instead of
 *                   two; a tile's arrivals that do not fit are appended in rounds with a prune between): 1 (default) from
 *                   2048 users, 2 always, 0 never.  Same results.
 *   "topk_sample" : items the pre-pass of rbg_full_sort_topk_f32 looks at (multiple of 128, default 8192)
 *   "topk_screen" : 1 (default) = rbg_full_sort_topk_f32 at d <= 128, B >= 256, more than 2 x "topk_sample" items screens every
 *                   (user, item) pair with ONE bf16 x bf16 product and a rigorous bound of its error on the matrix core
 *                   (|s - s^| <= ||du|| ||i^|| + ||u|| ||di|| from the rows' actual rounding errors: no pair that can be in the top k is dropped) and rescoring the ~ k n /
 *                   sample survivors per user exactly in fp32 (csrc/topk_screen.hip; r06: 4096 users x 40 982 items 188 -> 91 us
 *                   per call at d = 64, 394 -> 122 us at d = 128, 242 -> 97 us on trained tables); 2 = for any batch size; 0 = the exact passes on 3-way split
 *                   operands for every pair.  Same items wherever two scores are not equal to the last bit; values agree to
 *                   ~ 1e-7 relative (the rescoring sums in a different order than the split products).
 *   "topk_image"  : 1 (default) = rbg_full_sort_topk_f32 at 64 < d <= 128, B >= 1024 splits the item table ONCE per call into three
 *                   bf16 planes stored tile by tile in the LDS layout (in the workspace) and both passes take their item tiles from
 *                   it by LDS-DMA (r06: 430 -> 394 us per call at 4096 users x 40 982 items x 128; at d <= 64 the per-workgroup
 *                   fetch + split was already hidden: 191.5 vs 192.0 us); 2 = at d <= 64 as well; 0 = every workgroup fetches,
 *                   splits and publishes its tiles itself.  Same results bit for bit.
 *   "sell"        : 1 (default) = rbg_lightgcn_forward_f32 / _backward_f32 / rbg_spmm_f32 use an attached column-slab plan
 *                   (rbg_graph_attach_sell) where it applies; 0 = the binned kernel
 *   "sell_rowmajor", "sell_factored" : see rbg_graph_attach_sell / rbg_graph_sell_set_factors (both default 1)
 *   "sell_auto"   : 1 (default) = rbg_graph_create* plans the column-slab propagation (rbg_graph_plan_sell) for every device graph
 *                   with a user / item boundary; 0 = no plan until rbg_graph_plan_sell / rbg_graph_attach_sell is called
 *   "fail_alloc_after" : test hook: the (n + 1)-th device allocation of the plan code from now fails once (-1 = off)
 *   "sell_nt"     : non-temporal hints in that kernel's epilogue (bit 0 stores, bit 1 the mean's addend loads; default 0, no effect measured)
 *   "slab"        : measured-and-off r03 variant of the binned kernel over column halves (default 0)
 *   "shard_single_stream" : 1 = the C-ABI sharded layer packs and exchanges on the caller's stream (capturable); default 0
 *   "shard_fused" : 1 (default) = rbg_graph_create_sharded builds the rank's block [A_interior | A_halo] as ONE rectangular handle
 *                   planned for the column-slab kernel; a layer is pack -> exchange -> ONE launch over the table [owned rows | halo
 *                   rows] on the caller's stream (no accumulate pass, every entry on sell_spmm_kernel); 0 = two handles (interior
 *                   product beside the exchange on the shard's stream, then Y += A_halo X_halo).  Read at creation. */
int rbg_set_option(const char *key, int64_t value);
int rbg_get_option(const char *key, int64_t *value);

/* ---------------------------------------------------------------------------------------------
 * graph construction
 * ------------------------------------------------------------------------------------------- */

/* Replaces GeneralGraphDataset.get_norm_adj_mat(enable_sparse=True)
 *   recbole_gnn/data/dataset.py:49-75 (+ edge_index_to_adj_t :41-47, gcn_norm [PyG] at :74).
 * uid/iid are the two id columns of inter_feat (HOST pointers, int64, uid in [0,n_users),
 * iid in [0,n_items)); node ids are users first, then items (iid + n_users, dataset.py:61).
 * Builds Â = D^-1/2 A D^-1/2 (no self loops, duplicates kept as separate edges) as CSR of
 * adj_t with int32 columns sorted within a row and fp32 values  val = (dis[row]*1)*dis[col],
 * dis = 1/sqrt(deg) with inf -> 0.  device = -1 keeps the graph on the host (build / export
 * only; ops need a device graph), device >= 0 places it in that GPU's HBM. */
int rbg_graph_create(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter,
                     const int64_t *uid, const int64_t *iid, int device, uint32_t flags);

/* Replaces SGL.random_graph_augment's graph rebuild for one view
 *   recbole_gnn/model/general_recommender/sgl.py:107-126 (ED / RW) and :97-106 (ND, via the mask).
 * keep[e] != 0 keeps interaction e; the view is re-normalized on ITS OWN degrees (sgl.py:119-124).
 * keep == NULL is rbg_graph_create. */
int rbg_graph_create_masked(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter,
                            const int64_t *uid, const int64_t *iid, const uint8_t *keep,
                            int device, uint32_t flags);

/* rbg_graph_create_masked plus a node partition for L2 locality: part[node] in [0, n_parts) (HOST array over all
 * n_users + n_items nodes; n_parts in {1,2,4,8}) names the community of every node, e.g. from a graph partitioner
 * or the shard owner array.  The rows of part p are pinned to XCDs [p*8/n_parts, (p+1)*8/n_parts) (workgroup b runs
 * on XCD b % 8), so an XCD's private 4 MB L2 only sees that community's embeddings plus the edge cut.  Results are
 * identical to the unpartitioned graph (same CSR, same per-row arithmetic); only the launch plan changes.  Parts
 * should carry similar nnz.  part == NULL or n_parts == 1: the default user-row / item-row XCD split. */
int rbg_graph_create_partitioned(rbg_graph **out, int64_t n_users, int64_t n_items, int64_t n_inter,
                                 const int64_t *uid, const int64_t *iid, const uint8_t *keep, const int32_t *part,
                                 int n_parts, int device, uint32_t flags);

/* A pre-built (already weighted) CSR, possibly rectangular: rows = output nodes, cols = index
 * space of the dense operand.  Used for node-range shards ([local | halo] column space) and for
 * callers that hold the reference's SparseTensor storage (rowptr,col,value of adj_t;
 * recbole_gnn/model/abstract_recommender.py:15-18).  HOST pointers. */
int rbg_graph_create_csr(rbg_graph **out, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                         const int32_t *col, const float *val, int device, uint32_t flags);

/* The same, with the caller declaring two row classes: rows [0, n_class0_rows) and the rest reference disjoint sets of
 * columns (a shard's user rows gather item embeddings only, its item rows user embeddings only), so the launch plan pins
 * the classes to different XCDs exactly as it does for graphs built from interactions (each XCD's L2 then serves one
 * table).  n_class0_rows < 0: one class (= rbg_graph_create_csr).  Results do not depend on it. */
int rbg_graph_create_csr_classes(rbg_graph **out, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                                 const int32_t *col, const float *val, int64_t n_class0_rows, int device, uint32_t flags);

/* The (edge_index, edge_weight) form, i.e. get_norm_adj_mat(enable_sparse=False|None)
 *   recbole_gnn/data/dataset.py:60-66,77-79.
 * Writes edge_index as int64 [2][2*n_inter] (row 0 = source, row 1 = target; first all u->i,
 * then all i->u, order preserved) and edge_weight fp32 [2*n_inter].  Pure host function. */
int rbg_norm_edges(int64_t n_users, int64_t n_items, int64_t n_inter, const int64_t *uid,
                   const int64_t *iid, int64_t *edge_index, float *edge_weight);

/* Build a graph from the reference's dense-branch pair (edge_index int64 [2][nnz], edge_weight
 * fp32 [nnz], HOST pointers): Y[target] += w * X[source]  (layers.py:16-17 + PyG scatter).
 * This is how a model holding `self.edge_index, self.edge_weight` hands its graph over. */
int rbg_graph_create_coo(rbg_graph **out, int64_t n_nodes, int64_t nnz, const int64_t *edge_index,
                         const float *edge_weight, int device, uint32_t flags);

int rbg_graph_info(const rbg_graph *g, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int *device);

/* Row-binning statistics (diagnostics / DESIGN.md): counts of lane-group rows, wavefront rows,
 * workgroup tasks, split rows, and the launch grid. Any pointer may be NULL. */
int rbg_graph_bins(const rbg_graph *g, int d, int64_t *n_short, int64_t *n_wave, int64_t *n_block_tasks,
                   int64_t *n_split_rows, int64_t *grid_blocks);

/* Name (as rocprofv3 prints it, without the namespace) of the kernel rbg_spmm_f32 launches for width d on this handle
 * under the current options — so that a benchmark can label its roofline line and find the kernel in a trace. */
int rbg_spmm_kernel_name(const rbg_graph *g, int d, char *buf, int len);

/* Copy the CSR out to HOST buffers: rowptr int64 [n_rows+1], col int32 [nnz], val fp32 [nnz].
 * Any pointer may be NULL.  Works for host and device graphs (D2H copy + sync for the latter). */
int rbg_graph_export_csr(const rbg_graph *g, int64_t *rowptr, int32_t *col, float *val);

/* Edge re-weighting without a rebuild — NGCF's per-forward edge dropout (ngcf.py:74-90: dropout_adj [PyG] filters the
 * directed edges with a Bernoulli mask, weights unchanged, no re-normalisation, then re-builds and re-sorts a
 * SparseTensor on every forward).  A view shares src's sparsity structure and launch plan and reads its values from the
 * caller's DEVICE array vals [nnz] (the handle's CSR entry order) at launch time: the caller rewrites that array between
 * stream-ordered launches (vals[e] = weight[e] * keep[e]); a dropped edge is a zero weight.  src must outlive the view.
 * rbg_graph_transpose_map fills map [nnz] (DEVICE int32) with the position of every entry's transposed partner, so the
 * values of the transposed view (the autograd backward of a NON-symmetric dropout mask) are vals_t[e] = vals[map[e]].
 * It synchronises the stream (called once per graph) and fails if the structure is not symmetric. */
int rbg_graph_create_reweighted(rbg_graph **out, const rbg_graph *src, const float *vals);
int rbg_graph_transpose_map(const rbg_graph *g, int32_t *map, void *stream);

/* Column-slab propagation: the planner inside the library (r04; csrc/sell_plan.hip).  Cuts this device graph's normalized CSR
 * (the product of get_norm_adj_mat, recbole_gnn/data/dataset.py:49-79, or of an SGL view rebuild, sgl.py:107-126) into the
 * SELL-C-sigma form the column-slab kernel reads (slab width W: 32 serves d = 32 / 64 / 128; 64 serves d = 128, slower), on
 * the device (rocPRIM sorts / scans + one-pass kernels; two small host round trips), then installs it with its derived arrays
 * (row-major twin, 4-byte offsets column, row factors r = deg^-1/2 when the values are r_i r_j, the CSR position of every
 * slot).  chunk = entries per piece of a split row (0 = default 128).  rbg_graph_create* calls this itself
 * for every device graph with a user / item boundary (option "sell_auto", default 1): a caller that binds only
 * rbg_graph_create + rbg_lightgcn_forward_f32 / rbg_spmm_f32 (layers.py:19-20, lightgcn.py:74-76) runs the column-slab kernel.
 * RBG_EUNSUPPORTED: the graph is outside what the plan serves (no boundary / not bipartite, a hub row longer than
 * 4 LGW x max(512, nnz / 8192) entries, a table beyond 32-bit slab offsets); the handle keeps the binned kernel and
 * rbg_graph_sell_status says why.  Not concurrently with launches on the handle. */
int rbg_graph_plan_sell(rbg_graph *g, int W, int chunk);
/* "planned" / "attached" / "view of a planned graph", or the reason the handle has no plan. */
int rbg_graph_sell_status(const rbg_graph *g, char *buf, int len);
/* Shape of the installed plan (any pointer may be NULL; n_units = int32[2]); RBG_EINVAL without a plan. */
int rbg_graph_sell_info(const rbg_graph *g, int *W, int *chunk, int64_t *n_ent, int32_t *n_units, int *factored, int *rowmajor);
/* The plan's device arrays (read-only, valid while the plan lives): ent [n_ent + 128][2], head [n_units][4], orig [n_rows],
 * factors [n_rows] or NULL, src [n_ent] (CSR position of every slot, -1 = padding) or NULL. */
int rbg_graph_sell_arrays(const rbg_graph *g, const int32_t **ent, const int32_t **head, const int32_t **orig, const float **factors,
                          const int32_t **src);
/* A re-weighted view (rbg_graph_create_reweighted) of a planned graph borrows the plan and owns a COPY of its valued entries:
 * call this after every rewrite of the view's `vals` array (NGCF edge dropout, ngcf.py:74-90) — it refreshes the copy on
 * `stream` (4 bytes read + 4 written per entry).  The view runs the column-slab kernel from its first refresh on; a caller
 * that never refreshes keeps the binned kernel, which reads `vals` at launch time.  A no-op without a plan. */
int rbg_graph_refresh_values(rbg_graph *view, void *stream);
/* Lifetime rule of borrowed plans: a view holds pointers INTO its base handle's plan arrays.  While a handle has live views that
 * borrow its plan, rbg_graph_detach_sell / rbg_graph_plan_sell / rbg_graph_attach_sell on it return RBG_EUNSUPPORTED and leave the
 * plan in place; destroy the views first (and, as for every view, before the base handle itself). */

/* Column-slab propagation (r03; csrc/sell.hip): attach an EXTERNALLY built SELL-C-sigma plan of this graph for slab width W
 * (the executable specification tests/sell_spec.py; the tests compare rbg_graph_plan_sell with it).  rbg_lightgcn_forward_f32 then runs the slab kernel whenever it is called with ONE graph,
 * RBG_FWD_LAYERS_SCRATCH and without RBG_FWD_KEEP_LAST_LAYER (option "sell", default 1); with the layers kept row-major the
 * same kernel serves the other flag combinations, rbg_lightgcn_backward_f32 and rbg_spmm_f32 at that width (option
 * "sell_rowmajor", default 1: E0 / the gradient / X are gathered where they lie through a twin of the entry array in the
 * reference's numbering, built at attach time, +8 bytes per entry).  The planner is
 * tests/sell_spec.py (torch ops on the handle's device CSR); `ent` [n_ent][2], `head` [n_units][4] and `orig` [n_rows] are
 * DEVICE arrays on the graph's device, `unit_base` / `n_units` host arrays of 2.  Every index the kernel dereferences is
 * range-checked on the device before the plan is adopted (copied: the caller keeps its arrays).  Graphs built from
 * interactions (a user / item boundary, square) only; a re-weighted view cannot carry a plan.
 * Unit header (r06) = {first entry, first row, slots << 16 | j, log2(parts) | rows << 8 | wide << 16 | U << 17}: a row of more
 * than chunk x LGW entries ("wide") is U consecutive units at the FRONT of its class (unit j of U); their partial sums meet
 * in a scratch slot per unit that belongs to the handle (every re-weighted view has its own) and the last unit to arrive adds
 * them in unit order — the result is bit-stable, but launches on ONE handle must be stream-ordered (two streams running the
 * same handle at once would share that scratch; the binned kernel's split rows have the same rule).  Rows of any length are
 * served (r03-r05: four units per wide row, longer hub rows kept the binned kernel). */
int rbg_graph_attach_sell(rbg_graph *g, int W, const int32_t *ent, int64_t n_ent, const int32_t *head, const int32_t *unit_base,
                          const int32_t *n_units, const int32_t *orig);
/* Row factors of an attached plan: r [n_rows] (device, the PLAN's row numbering = orig[]) with val_ij = r_i * r_j — the symmetric
 * normalisation D^-1/2 A D^-1/2 of dataset.py:41-79 (r = deg^-1/2, 0 for an empty row).  Checked on the device against every
 * stored value (1e-6 relative).  With factors the slab chains of rbg_lightgcn_forward_f32 / _backward_f32 keep r (.) E_k between
 * the layers and every launch after the first reads 4 bytes per entry (the column offset) instead of 8 (option "sell_factored",
 * default 1); results differ from the unfactored chain by rounding only. */
int rbg_graph_sell_set_factors(rbg_graph *g, const float *r);
int rbg_graph_detach_sell(rbg_graph *g);
int rbg_graph_has_sell(const rbg_graph *g, int d);
/* Name of the kernel rbg_lightgcn_forward_f32 launches per layer for this graph, width and flags (one graph). */
int rbg_lightgcn_forward_kernel_name(const rbg_graph *g, int d, uint32_t flags, char *buf, int len);

/* The handle's CSR arrays in HBM, read-only and valid while the handle lives: rowptr int32 [n_rows + 1], col int32 [nnz],
 * val fp32 [nnz] (any pointer argument may be NULL).  For device-side consumers of the normalized adjacency (the shard
 * planner cuts a rank's blocks out of it without a host round trip). */
int rbg_graph_device_arrays(const rbg_graph *g, const int32_t **rowptr, const int32_t **col, const float **val);

void rbg_graph_destroy(rbg_graph *g);

/* ---------------------------------------------------------------------------------------------
 * operators (device graphs only; X, Y, ... are DEVICE pointers)
 * ------------------------------------------------------------------------------------------- */

/* Replaces LightGCNConv.forward(x, edge_index, edge_weight)
 *   recbole_gnn/model/layers.py:13-20  (torch_sparse.matmul(adj_t, x, reduce='add') at :19-20,
 *   or edge_weight.view(-1,1) * x_j + scatter-add at :16-17).
 * Y[n_rows, d] = Â · X[n_cols, d]  (accumulate = 0)   or   Y += Â · X  (accumulate != 0).
 * X and Y must not alias.  Rows of Â with no entries produce zeros.  Also the autograd backward
 * (Â symmetric):  dL/dX = Â · dL/dY. */
int rbg_spmm_f32(const rbg_graph *g, const float *X, float *Y, int d, int accumulate, void *stream);

/* The last layer of a propagation with the layer mean in its epilogue (lightgcn.py:75-78 for one layer):
 *   out_mean = (srcs[0] + ... + srcs[n_srcs-1] + (partial + Â X)) / (n_srcs + 1)
 * srcs = E_0 and the earlier layer outputs ([n_rows, d] each, n_srcs <= RBG_MAX_FUSED_LAYERS + 1); partial ([n_rows, d] or
 * NULL) is a product already formed for the same rows — the node-range sharded path passes the interior product and
 * runs this call on the halo block, so its K-th layer needs no separate accumulate and mean passes. */
int rbg_spmm_mean_f32(const rbg_graph *g, const float *X, const float *partial, const float *const *srcs, int n_srcs,
                      float *out_mean, int d, void *stream);

/* Y = Z + Â X  (r06): one step of a Horner chain whose addend changes from layer to layer — the backward of a propagation whose
 * layers receive different incoming gradients (XSimGCL's contrast at layer_cl, xsimgcl.py:39-41; NCL's context layer, ncl.py:137-165):
 * dE_0 = g_0 + Â (g_1 + Â (g_2 + ...)).  Replaces "copy the addend into Y, then rbg_spmm_f32(accumulate = 1)": one 18 MB copy per
 * layer less at the Gowalla shape.  X, Z, Y: [n, d]; Y must alias neither input. */
int rbg_spmm_add_f32(const rbg_graph *g, const float *X, const float *Z, float *Y, int d, void *stream);

/* One perturbed layer of SimGCL / XSimGCL (simgcl.py:29-34, xsimgcl.py:34-38):
 *   Y = Â X;   Y += sign(Y) * F.normalize(noise, dim=-1) * eps        (noise [N, d]: the caller's torch.rand_like draw)
 * as the SpMM's epilogue.  sign() has zero gradient, so the backward of this op is the plain Â^T product. */
int rbg_spmm_noise_f32(const rbg_graph *g, const float *X, float *Y, const float *noise, int d, float eps, void *stream);

/* The epilogue of rbg_spmm_noise_f32 on a product that exists already (r06):  out = Y + sign(Y) * F.normalize(noise, dim=-1) * eps.
 * SimGCL's plain pass and its two perturbed passes (simgcl.py:45-55: three calls of forward()) start with the SAME product Â E_0;
 * the perturbed first layers are this row kernel on the plain pass's first layer instead of two more propagations.
 * Y, noise, out: [n, d] contiguous, d <= 128; out may alias Y. */
int rbg_sign_noise_f32(const float *Y, const float *noise, int64_t n, int d, float eps, float *out, void *stream);

/* The one-occurrence mask of a batch's ids and the row weights of rbg_infonce_masked_f32 in one launch (r06; SimGCL / XSimGCL contrast
 * over torch.unique of the batch without its data-dependent shape, simgcl.py:38-43, xsimgcl.py:50-54):
 *   once[b] = 1 for exactly ONE position b of every distinct id, 0 for the others  (first_occurrence = 1: the first position — the
 *             same positions in every run; 0: whichever store stays, cheaper; no loss depends on which: rows of equal ids are equal)
 *   row_w[b] = once[b]                (mean_form = 0: a sum over the distinct ids)
 *            = once[b] / sum(once)    (mean_form = 1: their mean)
 * ids [B] int64 in [0, n_ids); slot: scratch of n_ids int64 (no reset needed between calls); once, row_w: [B] fp32. */
int rbg_once_mask_f32(const int64_t *ids, int64_t B, int64_t n_ids, int64_t *slot, int first_occurrence, int mean_form, float *once,
                      float *row_w, void *stream);

/* Replaces LightGCN.get_ego_embeddings + LightGCN.forward
 *   recbole_gnn/model/general_recommender/lightgcn.py:60-68,70-81  (and SGL.forward,
 *   sgl.py:128-145, where layer k may use its own graph).
 * graphs: n_graphs == 1 (same Â every layer) or n_graphs == K (layer k uses graphs[k]).
 * user_emb [n_users, d], item_emb [n_items, d]: the two embedding tables (no concatenated copy is
 * made; n_users + n_items must equal the graph's node count, n_users is taken from the graph or,
 * for CSR/COO-built graphs, from `n_users`).
 * out_mean [N, d] = mean(E_0..E_K) (lightgcn.py:77-78); rows [0,n_users) are user_all_embeddings,
 * the rest item_all_embeddings (the split at :80 is a view).
 * layers: caller-provided [K][N][d] fp32 buffer; E_k is written to layers[k-1] for k < K (and for
 * k == K with RBG_FWD_KEEP_LAST_LAYER).  Needed by the next layer and by autograd. */
int rbg_lightgcn_forward_f32(const rbg_graph *const *graphs, int n_graphs, int64_t n_users,
                             const float *user_emb, const float *item_emb, float *out_mean,
                             float *layers, int d, int K, uint32_t flags, void *stream);

/* Backward of rbg_lightgcn_forward_f32 with respect to the two embedding tables (what torch autograd does through
 * LightGCN.forward, lightgcn.py:70-81, during loss.backward()): the propagation is linear, so no activations are
 * needed:  dE0 = (g + Â_0 (g + Â_1 (... (g + Â_{K-1} g)))) / (K+1),  g = dL/d(out_mean)  — K launches of the same
 * SpMM with a fused "+ g" epilogue.  The graphs must be symmetric (they are: dataset.py:62-64, sgl.py:113-115) or
 * the caller passes the transposed handles.  grad_e0 [N, d]: rows [0,n_users) are the user-table gradient, the rest
 * the item-table gradient.  work: [N, d] scratch, needed for K >= 2.  No buffer may alias another. */
int rbg_lightgcn_backward_f32(const rbg_graph *const *graphs, int n_graphs, const float *grad_out, float *grad_e0,
                              float *work, int d, int K, void *stream);

/* Replaces BiGNNConv.forward(x, edge_index, edge_weight)
 *   recbole_gnn/model/layers.py:54-58:  P = ÂX;  Y = lin1(P + X) + lin2(P ⊙ X)
 * and, with RBG_BIGNN_LEAKY_NORM, NGCF.forward's per-layer tail
 *   recbole_gnn/model/general_recommender/ngcf.py:96,98: LeakyReLU(slope) then F.normalize(p=2, dim=1)
 *   (message dropout, ngcf.py:97, is not applied: parity is defined at message_dropout = 0).
 * W1, W2: [d_out, d_in] row-major (nn.Linear.weight), b1, b2: [d_out].
 * Y [N, d_out] is written with row stride ldy floats (ldy >= d_out) so the caller can place layer
 * outputs directly inside the concatenated [N, sum(d)] buffer (ngcf.py:100).
 * X [N, d_in] is read with row stride ldx floats.  P_save (required, caller-provided [N, d_in]) receives ÂX: it is the
 * dense kernel's operand and what the autograd backward needs. */
int rbg_bignn_conv_f32(const rbg_graph *g, const float *X, int64_t ldx, const float *W1, const float *b1,
                       const float *W2, const float *b2, float *Y, int64_t ldy, float *P_save,
                       int d_in, int d_out, uint32_t flags, float slope, void *stream);

/* The dense half of that layer from a product P = ÂX [n_rows, d_in] (contiguous) the caller already holds — the sharded path
 * forms P with rbg_spmm_sharded_f32 and finishes BiGNNConv (layers.py:55-57) [+ the NGCF tail] on its own rows. */
int rbg_bignn_dense_f32(const float *P, const float *X, int64_t ldx, const float *W1, const float *b1, const float *W2,
                        const float *b2, float *Y, int64_t ldy, int64_t n_rows, int d_in, int d_out, uint32_t flags, float slope,
                        void *stream);

/* Replaces the scoring GEMM of full_sort_predict
 *   recbole_gnn/model/general_recommender/lightgcn.py:131 (ngcf.py:147, sgl.py:240):
 *   scores = u_embeddings @ restore_item_e.T
 * S[B, n] = U[B, d] · I[n, d]^T, fp32 MFMA (v_mfma_f32_32x32x2_f32), ldu/ldi = row strides in floats. */
int rbg_score_f32(const float *U, int64_t ldu, const float *I, int64_t ldi, float *S, int64_t B,
                  int64_t n, int d, void *stream);

/* Weight gradients of BiGNNConv (autograd of layers.py:54-58) with G = dL/dY [N, d_out] (row stride ldg), P = ÂX
 * [N, d_in] contiguous (the P_save of rbg_bignn_conv_f32) and X [N, d_in] (row stride ldx):
 *   dW1 = G^T (P + X)   dW2 = G^T (P ⊙ X)   [d_out, d_in] each;   db = sum_rows G  [d_out] (= db1 = db2; may be NULL)
 * One pass over G, P, X; split over row ranges, partials summed in a fixed order (no atomics).  d_in, d_out <= 128.
 * `workspace`: rbg_bignn_wgrad_workspace(n_rows, d_in, d_out) bytes. */
int rbg_bignn_wgrad_workspace(int64_t n_rows, int d_in, int d_out, int64_t *bytes);
int rbg_bignn_wgrad_f32(const float *G, int64_t ldg, const float *P, const float *X, int64_t ldx, int64_t n_rows, int d_in,
                        int d_out, float *dW1, float *dW2, float *db, void *workspace, void *stream);

/* Training forward of one NGCF layer: rbg_bignn_conv_f32 with RBG_BIGNN_LEAKY_NORM that additionally writes
 * inv_norm [N] = 1 / max(||a||, eps) per row — with the output Y all the tail's backward needs — and applies the
 * message dropout of ngcf.py:97 when drop_mask != NULL: a = LeakyReLU(z) ⊙ drop_mask, drop_mask [N, d_out] contiguous
 * holding 0 or 1/(1-p) (the caller draws it with its own RNG, as the reference's nn.Dropout does). */
int rbg_bignn_layer_f32(const rbg_graph *g, const float *X, int64_t ldx, const float *W1, const float *b1,
                        const float *W2, const float *b2, float *Y, int64_t ldy, float *P_save, float *inv_norm,
                        const float *drop_mask, int d_in, int d_out, float slope, void *stream);

/* Backward of that layer (autograd of layers.py:54-58 [+ ngcf.py:96,98 when inv_norm != NULL]) from GY = dL/dY:
 *   G = dL/dz (normalize [+ dropout with the forward's drop_mask] + LeakyReLU backward from the saved Y, inv_norm;
 *       G = GY without the tail)
 *   GX [N, d_in] = G W1 + (G W2) ⊙ P + Â^T (G W1 + (G W2) ⊙ X)        (the gradient of the layer input X)
 *   dW1 = G^T (P + X),  dW2 = G^T (P ⊙ X),  db = sum_rows G            (db may be NULL)
 * g_t: handle of Â^T (for the symmetric graphs of this path, the forward handle).  d_in, d_out <= 128.
 * `workspace`: rbg_bignn_backward_workspace(n_rows, d_in, d_out) bytes. */
int rbg_bignn_backward_workspace(int64_t n_rows, int d_in, int d_out, int64_t *bytes);
int rbg_bignn_backward_f32(const rbg_graph *g_t, const float *GY, int64_t ldgy, const float *Y, int64_t ldy,
                           const float *inv_norm, const float *drop_mask, const float *X, int64_t ldx, const float *P,
                           const float *W1, const float *W2, int d_in, int d_out, float slope, float *GX, float *dW1,
                           float *dW2, float *db, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * fused mini-batch training step.  All pointers are DEVICE pointers; `loss` is a
 * device scalar.  Row scatters use float atomics like torch's GPU index backward.
 * ------------------------------------------------------------------------------------------- */

/* Replaces lightgcn.py:93-100 + their autograd: pos/neg scores of (user, pos, neg) triples on out_mean [N,d]
 * (rows [0,n_users) users, then items), BPRLoss(gamma = 1e-10) = -mean(log(gamma + sigmoid(pos - neg))).
 * Zeroes grad_mean [N,d] and *loss, then writes dLoss/d(out_mean) and the loss value. */
int rbg_bpr_grad_f32(const float *out_mean, int64_t n_users, int64_t n_items, const int64_t *user, const int64_t *pos,
                     const int64_t *neg, int64_t B, int d, float *grad_mean, float *loss, void *stream);

/* Replaces lightgcn.py:103-108 with require_pow = True: reg_weight * EmbLoss(U0[user], I0[pos], I0[neg]) on the EGO
 * embeddings; ADDS its value to *loss and its sparse-row gradient onto grad_e0 [N,d] (call after
 * rbg_lightgcn_backward_f32). */
int rbg_emb_reg_grad_f32(const float *user_emb, const float *item_emb, int64_t n_users, const int64_t *user,
                         const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight, float *grad_e0,
                         float *loss, void *stream);

/* The same with require_pow = False (RecBole's EmbLoss default; lightgcn.py:103-108 passes self.require_pow):
 * reg_weight * (||U0[user]||_F + ||I0[pos]||_F + ||I0[neg]||_F) / B — the 2-norm of each gathered [B, d] block.
 * `workspace`: 3 floats of device scratch. */
int rbg_emb_reg_grad_nopow_f32(const float *user_emb, const float *item_emb, int64_t n_users, const int64_t *user,
                               const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight, float *grad_e0,
                               float *loss, float *workspace, void *stream);

/* NGCF's mini-batch loss (ngcf.py:106-126: BPRLoss on the scores of, and EmbLoss on, the rows of the CONCATENATION of the
 * layer outputs, ngcf.py:100) without forming the concatenation or gathering from it.  tables[t] (HOST array of DEVICE
 * pointers, 1..RBG_MAX_CONCAT) is E_t [N, widths[t]] contiguous, rows [0, n_users) users then items.
 * begin: zeroes sums[3] and *loss; writes coef[b] = dBPR/d(pos_b - neg_b), the three blocks' sums of squares and *loss =
 *        form 0: recbole's BPRLoss, -mean(log(1e-10 + sigmoid(pos - neg)));  form 1: -sum(logsigmoid(pos - neg)), the
 *        spelling of sgl.py:147-162 (one table: the propagated mean).
 * scatter (once per table, in any order, after begin): ADDS the loss gradient w.r.t. table t's rows onto grad_table
 *        [N, width] — the dense gradient the layer above has already written, or zeros for the last layer — with float
 *        atomics; when loss_reg is non-NULL (pass it on exactly one of the calls) also adds reg_weight * EmbLoss to it.
 *        require_pow = 0: EmbLoss's default (norms), 1: the squared form. */
#define RBG_MAX_CONCAT 8
int rbg_concat_bpr_begin_f32(const float *const *tables, const int *widths, int n_tables, int64_t n_users, int64_t n_items,
                             const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B, int form, float *coef,
                             float *sums, float *loss, void *stream);
int rbg_concat_bpr_scatter_f32(const float *table, int width, int64_t n_users, const int64_t *user, const int64_t *pos,
                               const int64_t *neg, int64_t B, float reg_weight, int require_pow, const float *coef,
                               const float *sums, float *grad_table, float *loss_reg, void *stream);

/* Replaces optimizer.step() of torch.optim.Adam (RecBole's default learner; no weight decay, no amsgrad) for the
 * two embedding tables in one pass.  grad / exp_avg / exp_avg_sq are [N,d]; step counts from 1. */
int rbg_adam_step_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                      float *exp_avg, float *exp_avg_sq, int64_t step, float lr, float beta1, float beta2, float eps,
                      void *stream);

/* r06 — LightGCN's training step (lightgcn.py:83-110 + Adam) without its glue launches: two calls around the backward propagation
 * replace rbg_bpr_grad_f32 (and the 18 MB fill of grad_mean it starts with), rbg_emb_reg_grad_f32 (require_pow = True) and
 * rbg_adam_step_dev_f32 — 221 -> 207 us per step, 0.113 -> 0.105 s per epoch at the Gowalla shape (profiles/r06_lean_step.jsonl);
 * same arithmetic.
 *   head: grad_mean += dBPR/d(out_mean) (rows of the batch; float atomics), *loss = BPR + reg_weight x EmbLoss (stored, not added:
 *         nothing to zero; also added to *loss_total when non-NULL), and how often every node occurs in the batch -> row_count.
 *         Counts the step (*step += 1) and leaves Adam's bias corrections of it in scratch.
 *   tail: Adam on both tables with gradient grad_e0 + reg_weight / B x occurrences x row; zeroes the rows of grad_mean the head
 *         wrote.
 * State the caller keeps between steps, all zero before the first: grad_mean [N, d], row_count [2][N] int32 (two tables used
 * alternately by the step's parity), step (device int64), scratch (8 floats; the head leaves Adam's bias corrections of the step
 * there for the tail: lr, beta1, beta2 must be the same in both calls).  Not in deterministic mode (RBG_EUNSUPPORTED). */
int rbg_lightgcn_step_head_f32(const float *out_mean, const float *user_emb, const float *item_emb, int64_t n_users, int64_t n_items,
                               const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B, int d, float reg_weight,
                               float *grad_mean, int32_t *row_count, int64_t *step, float *scratch, float *loss, float *loss_total,
                               float lr, float beta1, float beta2, void *stream);
int rbg_lightgcn_step_tail_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad_e0,
                               float *grad_mean, int32_t *row_count, float reg_weight, int64_t B, float *exp_avg, float *exp_avg_sq,
                               const int64_t *step, float *scratch, float lr, float beta1, float beta2, float eps, void *stream);

/* The same update with the step count in device memory, for callers that replay the step from a HIP graph (a host-side count
 * would bake the bias corrections of the captured step into every replay): `step` is a DEVICE int64 (0 before the first
 * call; incremented by the call), `factors` 2 floats of device scratch.  d must be a multiple of 4. */
int rbg_adam_step_dev_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                          float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                          float eps, void *stream);

/* The same, and the step's finished loss joins a running total in the launch that counts the step (r06: a training driver reads
 * *loss_total once per epoch instead of adding a device scalar per step; trainer.py's `total_loss += loss.item()` without the sync). */
int rbg_adam_step_dev_total_f32(float *user_emb, float *item_emb, int64_t n_users, int64_t n_items, int d, const float *grad,
                                float *exp_avg, float *exp_avg_sq, int64_t *step, float *factors, float lr, float beta1, float beta2,
                                float eps, const float *loss, float *loss_total, void *stream);

/* ---------------------------------------------------------------------------------------------
 * contrastive (InfoNCE) denominator without the [B, n] matrix
 * ------------------------------------------------------------------------------------------- */

/* lse[b] = log sum_{j<n} exp(scale * <Q[b,:], C[j,:]>),  computed as log sum exp(scale*x - shift) + shift.
 * Replaces sgl.py:195-198 / :204-207:  v2 = u_emd1.matmul(all_user2.T); v2 = sum(exp(v2 / ssl_tau), dim=1)
 * with Q = the normalised batch rows, C = the normalised table, scale = 1 / ssl_tau; the reference takes no shift
 * (shift = 0 reproduces it exactly, incl. overflow at tiny tau); for unit rows shift = scale keeps every term <= 1.
 * Exact-fp32 MFMA; the matrix is never written.  d <= 128.  `workspace`: rbg_lse_rows_workspace(B, n, d) bytes. */
int rbg_lse_rows_workspace(int64_t B, int64_t n, int d, int64_t *bytes);
int rbg_lse_rows_f32(const float *Q, int64_t ldq, int64_t B, const float *C, int64_t ldc, int64_t n, int d, float scale,
                     float shift, float *lse, void *workspace, void *stream);

/* Autograd of the above (what torch derives for sgl.py:195-198): with P[b][j] = exp(scale*x[b][j] - lse[b]),
 *   grad_Q[b,:] = grad_lse[b] * scale * sum_j P[b][j] C[j,:]      [B, d] contiguous (may be NULL)
 *   grad_C[j,:] = scale * sum_b grad_lse[b] P[b][j] Q[b,:]        [n, d] contiguous (may be NULL)
 * Tiles are recomputed; partial sums are combined in a fixed order (no atomics: bit-reproducible). */
int rbg_lse_rows_backward_f32(const float *Q, int64_t ldq, int64_t B, const float *C, int64_t ldc, int64_t n, int d,
                              float scale, float shift, const float *lse, const float *grad_lse, float *grad_Q,
                              float *grad_C, void *workspace, void *stream);

/* One half of SGL.calc_ssl_loss (sgl.py:191-199 users, :201-208 items), value AND gradients in one call:
 *   a = normalize(T1[idx]); p = normalize(T2[idx]); c = normalize(T2)                       (F.normalize, eps 1e-12)
 *   *loss += weight * sum_b ( log sum_j exp(<a_b, c_j> / tau)  -  <a_b, p_b> / tau )        (= -sum log(v1 / v2))
 *   grad_T1 [n, d] += d/dT1,  grad_T2 [n, d] += d/dT2 of that term (either may be NULL; both NULL = value only).
 * T1, T2: the two views of one table, [n, d] contiguous; idx [B] int64 rows (repeats allowed: float atomics on the
 * scattered rows, like torch's GPU index_add).  d <= 128.  `workspace`: rbg_infonce_workspace(B, n, d) bytes. */
int rbg_infonce_workspace(int64_t B, int64_t n, int d, int64_t *bytes);
int rbg_infonce_f32(const float *T1, const float *T2, int64_t n, int d, const int64_t *idx, int64_t B, float tau,
                    float weight, float *loss, float *grad_T1, float *grad_T2, void *workspace, void *stream);

/* The same with weights: *loss += weight * sum_b row_w[b] * ( log sum_j col_w[j] exp(<a_b, c_j> / tau) - <a_b, p_b> / tau ).
 * row_w [B] / col_w [n]: device arrays or NULL (= 1).  With T1, T2 the batch's gathered rows ([B, d], idx = 0..B-1, n = B) and
 * row_w = col_w = "one occurrence per distinct id" this is the contrast of simgcl.py:38-43,52-57 / xsimgcl.py:50-54,86-89 over
 * torch.unique of the batch (for the mean form divide row_w by its sum), without a data-dependent shape. */
int rbg_infonce_masked_f32(const float *T1, const float *T2, int64_t n, int d, const int64_t *idx, int64_t B, float tau, float weight,
                           const float *row_w, const float *col_w, float *loss, float *grad_T1, float *grad_T2, void *workspace,
                           void *stream);

/* InfoNCE of rows of one table against ALL rows of another, the positive of a row given by a map (r06; NCL's prototype contrast,
 * ncl.py:106-135: the batch's rows of E_0 against the k centroids, the positive = the row's cluster):
 *   a = normalize(T1[idx]);  c = normalize(T2)  (T2 [n2, d]; already-unit rows stay what they are to rounding)
 *   *loss += weight * sum_b ( log sum_j exp(<a_b, c_j> / tau)  -  <a_b, c_(pos_map[idx[b]])> / tau )
 *   grad_T1[idx[b]] += d/da_b ...  (float atomics; NULL = value only);  grad_T2 [n2, d] += ... or NULL (constant prototypes: the table
 *   pass is skipped).  pos_map: int64 per row of T1's table, values in [0, n2).  `workspace`: rbg_infonce_workspace(B, n2, d). */
int rbg_infonce_map_f32(const float *T1, const float *T2, int64_t n2, int d, const int64_t *idx, const int64_t *pos_map, int64_t B, float tau,
                        float weight, float *loss, float *grad_T1, float *grad_T2, void *workspace, void *stream);

/* The same contrast AMONG THE ROWS OF A BATCH, straight on the tables (r06): with a = normalize(TA[ids]), c = normalize(TB[ids])
 *   *loss += weight * sum_b row_w[b] * ( log sum_j col_w[j] exp(<a_b, c_j> / tau) - <a_b, c_b> / tau )      b, j = positions 0..B-1
 *   grad_TA[ids[b]] += d/da_b ...,  grad_TB[ids[j]] += d/dc_j ...   (float atomics: ids repeat; both tables or none)
 * = rbg_infonce_masked_f32 on the gathered rows (T1 = TA[ids], T2 = TB[ids], idx = 0..B-1, n = B) followed by two index_add_ of the
 * row gradients — without the two gathers, the two zero-filled gradient blocks and the two scatters (simgcl.py:38-57,
 * xsimgcl.py:50-54,86-89 with the one-occurrence mask of rbg_once_mask_f32 as row_w / col_w).  `workspace`: rbg_infonce_workspace(B, B, d). */
int rbg_infonce_batch_f32(const float *TA, const float *TB, int d, const int64_t *ids, int64_t B, float tau, float weight,
                          const float *row_w, const float *col_w, float *loss, float *grad_TA, float *grad_TB, void *workspace,
                          void *stream);

/* Full-sort evaluation of one batch without the [B, n_items] score matrix.
 * Replaces full_sort_predict (lightgcn.py:123-133) + RecBole's Trainer._full_sort_batch_eval [recbole==1.1.1]:
 *   scores = user_all[users] @ item_all.T;  scores[:, 0] = -inf;  scores[history_index] = -inf;  topk(scores, k)
 * `history` is the TRAINING graph handle (a user's history = its graph row; NULL = no history mask).
 * out_val [B, k] fp32 descending, out_idx [B, k] int64 item ids (-1 / -inf when fewer than k items remain).
 * Ties are broken towards the smaller item id.  k <= 32, d <= 256.  `workspace`: device buffer of at least
 * rbg_full_sort_topk_workspace(B, n_items, k) bytes (r06: includes the screen's bf16 image of the item table, 288 bytes per
 * item, and its candidate pool, ~ 4 KB per user — option "topk_screen"). */
int rbg_full_sort_topk_workspace(int64_t B, int64_t n_items, int k, int64_t *bytes);
int rbg_full_sort_topk_f32(const rbg_graph *history, const float *user_all, const float *item_all, const int64_t *users,
                           int64_t B, int64_t n_users, int64_t n_items, int d, int k, float *out_val, int64_t *out_idx,
                           void *workspace, void *stream);

/* Row gather: dst[i, :] = src[idx[i], :]  (restore_user_e[user], lightgcn.py:128; also packs the
 * halo send buffer of the node-range sharded path).  idx is a DEVICE int64 array. */
int rbg_gather_rows_f32(const float *src, int64_t lds, const int64_t *idx, float *dst, int64_t n_idx,
                        int d, void *stream);

/* One layer of the node-range sharded propagation with the host side in two calls; the caller's
 * collective (e.g. RCCL all_to_all of send_buf -> halo on the comm stream) goes in between.
 *   begin: comm stream waits for the main stream (X ready), packs send_buf[i] = X[send_idx[i]];  main: Y = A_int X
 *   end  : main stream waits for the comm stream (halo filled);  main: Y += A_halo halo   (g_halo may be NULL)
 * A context owns the two events used for the cross-stream ordering. */
typedef struct rbg_shard_ctx rbg_shard_ctx;
int rbg_shard_ctx_create(rbg_shard_ctx **out, int device);
void rbg_shard_ctx_destroy(rbg_shard_ctx *ctx);
int rbg_shard_layer_begin(rbg_shard_ctx *ctx, const rbg_graph *g_int, const float *X, float *Y, const int64_t *send_idx,
                          int64_t n_send, float *send_buf, int d, void *main_stream, void *comm_stream);
int rbg_shard_layer_end(rbg_shard_ctx *ctx, const rbg_graph *g_halo, const float *halo, float *Y, int d, void *main_stream,
                        void *comm_stream);

/* ---------------------------------------------------------------------------------------------
 * multi-GPU: node-range shards with a per-layer halo exchange over RCCL (SURVEY.md §8(b),(e)).  One process (or thread)
 * per GPU.  The reference has no distributed path; what is distributed is LightGCNConv (layers.py:13-20) and
 * LightGCN.forward (lightgcn.py:70-81).  RCCL is bound at run time; without it these calls return RBG_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
#define RBG_COMM_ID_BYTES 128
typedef struct rbg_comm rbg_comm;
typedef struct rbg_shard rbg_shard;

/* Rank 0 draws the communicator id (ncclGetUniqueId) and hands the RBG_COMM_ID_BYTES bytes to every rank by any channel the
 * host program has (MPI, torch.distributed, a file); then every rank calls rbg_comm_create (collective: ncclCommInitRank). */
int rbg_comm_unique_id(void *id);
int rbg_comm_create(rbg_comm **out, int nranks, int rank, const void *id, int device);
void rbg_comm_destroy(rbg_comm *comm);

/* This rank's shard from its plan (HOST arrays; sharded.py::plan_from_csr / build_plans produce them): the interior block
 * [n_owned x n_owned] (columns = local row indices) and the halo block [n_owned x n_halo] (columns = halo slots, grouped by
 * owner rank, ascending global id inside a rank) of the rows it owns — rows [0, n_users_owned) are users — plus the local
 * rows it packs for every peer (send_idx grouped by destination, send_counts [nranks]) and what it receives
 * (recv_counts [nranks], summing to n_halo).  d_max sizes the exchange buffers. */
int rbg_graph_create_sharded(rbg_shard **out, rbg_comm *comm, int64_t n_owned, int64_t n_users_owned, const int64_t *int_rowptr,
                             const int32_t *int_col, const float *int_val, int64_t n_halo, const int64_t *halo_rowptr,
                             const int32_t *halo_col, const float *halo_val, const int64_t *send_idx, const int64_t *send_counts,
                             const int64_t *recv_counts, int d_max);
void rbg_shard_destroy(rbg_shard *shard);

/* What the shard runs, as text: "fused: <plan status of the [interior | halo] handle>" or "two handles: interior <status>, halo
 * <status>" (the statuses of rbg_graph_sell_status). */
int rbg_shard_status(const rbg_shard *shard, char *buf, int len);

/* One sharded layer, collective over the communicator: Y[owned] = Â[owned, :] X with X = this rank's owned rows [n_owned, d].
 * Fused shards (option "shard_fused", r06): the halo rows are packed and exchanged (grouped ncclSend / ncclRecv) on `stream` into
 * the tail of the shard's [owned | halo] table, then ONE launch computes Y.  Two-handle shards: the exchange runs on the shard's
 * own high-priority stream while the interior product runs on `stream`; then Y += Â_halo X_halo.
 * Also the backward of itself (the global Â is symmetric). */
int rbg_spmm_sharded_f32(rbg_shard *shard, const float *X, float *Y, int d, void *stream);

/* lightgcn.py:70-81 over the shard: out_mean [n_owned, d] = mean(E_0 .. E_K) of the owned rows, K exchanges; the layer
 * mean rides in the last product's epilogue.  layers: [K][n_owned][d] scratch (two-handle shards keep E_1 .. E_{K-1} there; a
 * fused shard keeps its layers in its own [owned | halo] tables — K of them, grown by the first call, which must therefore run
 * outside a stream capture — and leaves `layers` untouched).  1 <= K <= RBG_MAX_FUSED_LAYERS + 1. */
int rbg_lightgcn_forward_sharded_f32(rbg_shard *shard, const float *E0, float *out_mean, float *layers, int d, int K, void *stream);

/* ---------------------------------------------------------------------------------------------
 * multi-GPU: halo PUSH without a collective (r06; csrc/ipc.hip).  A rank keeps its layer tables [owned rows | halo rows] in
 * memory it exports; its peers map that memory and their pack kernels (rbg_gather_rows_f32 with dst inside the mapping) store
 * the rows the owner needs straight into the owner's table tail — over xGMI between GPUs, in the same HBM when two processes
 * share a GPU.  Ordering: after its pushes a sender release-stores a sequence number into the receiver's flag word
 * (rbg_ipc_signal, on the sender's stream); the receiver's stream waits for all its senders' words (rbg_ipc_wait: a bounded
 * spin — after timeout_ms it sets *err (device int, 1 + the index of the late word) instead of hanging) and its layer launch
 * follows in stream order.  sharded.py::PushExchange drives it; the collective transports remain the default.
 * ------------------------------------------------------------------------------------------- */
#define RBG_IPC_HANDLE_BYTES 64
int rbg_ipc_alloc(void **ptr, int64_t bytes, int device);       /* zero-filled device memory that can be exported */
void rbg_ipc_free(void *ptr);
int rbg_ipc_export(void *ptr, void *handle);                      /* hipIpcGetMemHandle: RBG_IPC_HANDLE_BYTES bytes for the peers */
int rbg_ipc_open(const void *handle, void **ptr, int device);    /* hipIpcOpenMemHandle in a peer process */
int rbg_ipc_close(void *ptr);
int rbg_ipc_signal(void *flag, uint64_t value, void *stream);    /* *flag = value (8-byte word, possibly in a peer's memory), release */
int rbg_ipc_wait(const void *flags, int n, uint64_t value, int timeout_ms, int *err, void *stream);  /* until flags[0..n) >= value */

/* out[t] = scale * (srcs[0][t] + srcs[1][t] + ...), added left to right: the layer mean of lightgcn.py:80-81 over
 * separately held layer outputs (the sharded propagation) in one launch.  n_srcs <= RBG_MAX_FUSED_LAYERS + 1. */
int rbg_mean_f32(const float *const *srcs, int n_srcs, int64_t n_floats, float scale, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RBGNN_H */
