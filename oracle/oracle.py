"""CPU restatement of the RecBole-GNN LightGCN / NGCF / SGL propagation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker.  The product (``recbole-gnn_amd/``) never imports this module and has no CPU fallback.

PARITY UNPINNED.  The reference cannot be imported in the build container (``recbole``,
``torch_geometric`` and ``torch_sparse`` are absent and un-vendored) and its own
tests assert no values for this path (``tests/test_model.py:27-43`` are one-epoch smoke tests).
The arithmetic lives in third-party packages the reference does not pin:
``torch_geometric`` (README.md:39 ``pyg>=2.0.4``: ``gcn_norm``, ``MessagePassing``) and
``torch_sparse`` (un-pinned in ``.github/workflows/python-package.yml:40``: ``SparseTensor``,
``matmul -> spmm_cpu``).  Their published algorithms are restated here from the reference's own
call sites; each function cites the reference ``file:line`` it follows.  What pins this oracle
instead: (1) three mutually independent formulations that must agree (torch gather/index_add_
fp32 = the dense branch; row-sequential CSR fp32 = the sparse branch, also as plain C in
``oracle/rbg_oracle.c``; scipy CSR float64 = truth), (2) hand-derived known answers (SURVEY.md
Appendix C), (3) ``torch``'s own ``scatter_add_ / pow / index_add_`` used verbatim where PyG
calls them, (4) the reference's own test interactions (``tests/golden/ref_test_inter.npz``).

All functions take / return CPU tensors or numpy arrays.
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# graph construction
# --------------------------------------------------------------------------------------------

def build_edge_index(uid, iid, n_users):
    """recbole_gnn/data/dataset.py:60-66 — symmetric bipartite COO, users first then items.

    row = uid, col = iid + user_num; edge_index = [[row;col],[col;row]]; weight 1 per edge.
    """
    row = torch.as_tensor(uid, dtype=torch.int64)
    col = torch.as_tensor(iid, dtype=torch.int64) + int(n_users)
    e1 = torch.stack([row, col])
    e2 = torch.stack([col, row])
    edge_index = torch.cat([e1, e2], dim=1)
    edge_weight = torch.ones(edge_index.size(1), dtype=torch.float32)
    return edge_index, edge_weight


def gcn_norm(edge_index, edge_weight, num_nodes, dtype=torch.float32):
    """PyG ``gcn_norm(edge_index, edge_weight, num_nodes, add_self_loops=False)`` as called at
    recbole_gnn/data/dataset.py:77 and sgl.py:124 (tensor form).

    deg = scatter_add(w, col); dis = deg^-0.5 with inf -> 0; w' = dis[row] * w * dis[col].
    """
    row, col = edge_index[0], edge_index[1]
    w = edge_weight.to(dtype)
    deg = torch.zeros(num_nodes, dtype=dtype).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return edge_index, dis[row] * w * dis[col]


def coo_to_adj_t_csr(edge_index, edge_weight, num_nodes):
    """``SparseTensor(row, col, value).t()`` — recbole_gnn/data/dataset.py:41-47.

    adj[row=src, col=dst]; adj_t has one CSR row per *target* node listing its sources, sorted by
    column (torch_sparse sorts on construction).  Returns (rowptr int64, col int64, val).
    """
    src = edge_index[0].numpy()
    dst = edge_index[1].numpy()
    order = np.lexsort((src, dst))  # primary key: target row, secondary: source column
    rows = dst[order]
    cols = src[order]
    vals = edge_weight.numpy()[order]
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, cols.astype(np.int64), vals


def get_norm_adj_mat(uid, iid, n_users, n_items, enable_sparse=False, dtype=torch.float32):
    """GeneralGraphDataset.get_norm_adj_mat — recbole_gnn/data/dataset.py:49-79.

    enable_sparse falsy -> (edge_index [2,2E] int64, edge_weight [2E]) (dataset.py:77-79);
    truthy -> CSR of the normalized adj_t (rowptr, col, val) (dataset.py:68-75).  The SparseTensor
    form of gcn_norm multiplies rows then columns by dis (A.1) — the same fp32 products because
    the graph is symmetric and every raw weight is 1.
    """
    n = int(n_users) + int(n_items)
    edge_index, w = build_edge_index(uid, iid, n_users)
    edge_index, w = gcn_norm(edge_index, w, n, dtype=dtype)
    if enable_sparse:
        return coo_to_adj_t_csr(edge_index, w, n)
    return edge_index, w


def sgl_keep_from_indices(n_inter, keep_idx):
    mask = np.zeros(int(n_inter), dtype=np.uint8)
    mask[np.asarray(keep_idx, dtype=np.int64)] = 1
    return mask


def sgl_random_graph_augment(uid, iid, n_users, n_items, aug_type, drop_ratio, rng, enable_sparse=True):
    """SGL.random_graph_augment — recbole_gnn/model/general_recommender/sgl.py:93-126.

    ``rng`` is a numpy Generator standing in for the reference's global ``np.random`` state
    (sgl.py:94-95).  Returns (keep_mask uint8 [E], graph) where graph is re-normalized on the
    sub-graph's own degrees (sgl.py:119-124).
    """
    uid = np.asarray(uid, dtype=np.int64)
    iid = np.asarray(iid, dtype=np.int64)
    n_inter = uid.shape[0]
    if aug_type == "ND":  # sgl.py:97-106
        drop_user = rng.choice(np.arange(n_users), size=int(n_users * drop_ratio), replace=False)
        drop_item = rng.choice(np.arange(n_items), size=int(n_items * drop_ratio), replace=False)
        mask = np.isin(uid, drop_user)
        mask |= np.isin(iid, drop_item)
        keep = np.where(~mask)[0]
    elif aug_type in ("ED", "RW"):  # sgl.py:107-110
        keep = rng.choice(np.arange(n_inter), size=int(n_inter * (1 - drop_ratio)), replace=False)
    else:
        raise ValueError(aug_type)
    keep_mask = sgl_keep_from_indices(n_inter, keep)
    # the reference indexes with `keep` (order of the sample); the normalized matrix does not
    # depend on edge order, so the mask form is equivalent.
    kept = np.sort(keep)
    graph = get_norm_adj_mat(uid[kept], iid[kept], n_users, n_items, enable_sparse=enable_sparse)
    return keep_mask, graph


def dropout_adj(edge_index, edge_weight, keep_mask):
    """PyG ``dropout_adj(edge_index, edge_attr, p, training=True)`` as called at
    recbole_gnn/model/general_recommender/ngcf.py:81-82,89-90, with the Bernoulli draw
    ``mask = torch.rand(E) >= p`` replaced by an explicit ``keep_mask``: the surviving directed edges keep their
    weights — no rescale by 1/(1-p), no re-normalisation, the two directions of an interaction are independent."""
    keep = torch.as_tensor(keep_mask, dtype=torch.bool)
    return edge_index[:, keep], edge_weight[keep]


# --------------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------------

def conv_dense(x, edge_index, edge_weight):
    """LightGCNConv on the (edge_index, edge_weight) branch — recbole_gnn/model/layers.py:13-17
    + PyG propagate: x_j = x[edge_index[0]]; m = w.view(-1,1) * x_j;
    out = scatter_add(m, edge_index[1]).  fp32 (or x's dtype)."""
    x_j = x.index_select(0, edge_index[0])
    m = edge_weight.to(x.dtype).view(-1, 1) * x_j
    out = torch.zeros_like(x)
    out.index_add_(0, edge_index[1], m)
    return out


def conv_csr_sequential(x, rowptr, col, val):
    """LightGCNConv on the SparseTensor branch — recbole_gnn/model/layers.py:19-20 ->
    torch_sparse spmm_cpu: per row, acc[k] += val[e] * x[col[e], k] in column
    order, fp32 multiply then add.  Pure-Python row loop: small cases only (the C restatement in
    rbg_oracle.c is the same loop for large cases)."""
    xn = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    out = np.zeros((len(rowptr) - 1, xn.shape[1]), dtype=xn.dtype)
    for r in range(len(rowptr) - 1):
        acc = np.zeros(xn.shape[1], dtype=xn.dtype)
        for e in range(int(rowptr[r]), int(rowptr[r + 1])):
            acc = acc + xn.dtype.type(val[e]) * xn[int(col[e])]
        out[r] = acc
    return torch.from_numpy(out)


def conv_csr_f64(x, rowptr, col, val):
    """Truth: the same product in float64 through scipy CSR."""
    import scipy.sparse as sp

    n_rows = len(rowptr) - 1
    xn = np.asarray(x, dtype=np.float64)
    a = sp.csr_matrix((np.asarray(val, dtype=np.float64), np.asarray(col), np.asarray(rowptr)),
                      shape=(n_rows, xn.shape[0]))
    return a @ xn


def lightgcn_forward(user_w, item_w, conv, n_layers, return_layers=False):
    """LightGCN.forward — recbole_gnn/model/general_recommender/lightgcn.py:60-81 (SGL.forward
    sgl.py:128-145 when ``conv`` is a list of per-layer callables).

    ``conv``: callable x -> Â·x, or a list of K callables.  cat -> K x conv -> stack(dim=1) ->
    mean(dim=1) -> split."""
    all_e = torch.cat([user_w, item_w], dim=0)
    embs = [all_e]
    convs = conv if isinstance(conv, (list, tuple)) else [conv] * n_layers
    assert len(convs) == n_layers
    for c in convs:
        all_e = c(all_e)
        all_e = all_e if isinstance(all_e, torch.Tensor) else torch.from_numpy(np.asarray(all_e))
        embs.append(all_e)
    stacked = torch.stack(embs, dim=1)
    mean = torch.mean(stacked, dim=1)
    user_all, item_all = torch.split(mean, [user_w.shape[0], item_w.shape[0]])
    if return_layers:
        return user_all, item_all, embs
    return user_all, item_all


def full_sort_predict(user_all, item_all, user):
    """LightGCN.full_sort_predict — lightgcn.py:123-133: u = user_all[user];
    scores = u @ item_all.T; view(-1)."""
    u = user_all[torch.as_tensor(user, dtype=torch.int64)]
    scores = torch.matmul(u, item_all.transpose(0, 1))
    return scores.view(-1)


def bignn_conv(x, conv, w1, b1, w2, b2):
    """BiGNNConv.forward — recbole_gnn/model/layers.py:54-58.
    x_prop = Â x; lin1(x_prop + x) + lin2(x_prop * x); lin = x @ W.T + b (nn.Linear)."""
    p = conv(x)
    p = p if isinstance(p, torch.Tensor) else torch.from_numpy(np.asarray(p))
    t = torch.nn.functional.linear(p + x, w1, b1)
    i = torch.nn.functional.linear(torch.mul(p, x), w2, b2)
    return t + i


def ngcf_forward(user_w, item_w, conv, layer_params, slope=0.2):
    """NGCF.forward at node_dropout = 0, message_dropout = 0 —
    recbole_gnn/model/general_recommender/ngcf.py:92-104: per layer BiGNNConv -> LeakyReLU(0.2)
    -> (Dropout p=0 = identity) -> F.normalize(p=2, dim=1); concat of K+1 blocks; split."""
    all_e = torch.cat([user_w, item_w], dim=0)
    embs = [all_e]
    for (w1, b1, w2, b2) in layer_params:
        all_e = bignn_conv(all_e, conv, w1, b1, w2, b2)
        all_e = torch.nn.functional.leaky_relu(all_e, negative_slope=slope)
        all_e = torch.nn.functional.normalize(all_e, p=2, dim=1)
        embs.append(all_e)
    cat = torch.cat(embs, dim=1)
    return torch.split(cat, [user_w.shape[0], item_w.shape[0]])


def simgcl_forward(user_w, item_w, conv, n_layers, noises=None, eps=0.0, layer_cl=None):
    """SimGCL.forward / XSimGCL.forward — recbole_gnn/model/general_recommender/simgcl.py:24-38, xsimgcl.py:28-48.
    Layers 1..K only (embeddings_list starts empty); perturbed (noises = the K torch.rand_like draws, in order):
    e = e + sign(e) * F.normalize(noise, dim=-1) * eps after every product.  layer_cl (XSimGCL): also return the
    embedding after layer `layer_cl` (the ego embedding if it is outside 1..K)."""
    all_embs = torch.cat([user_w, item_w], dim=0)
    all_embs_cl = all_embs
    embs = []
    for layer_idx in range(n_layers):
        all_embs = conv(all_embs)
        all_embs = all_embs if isinstance(all_embs, torch.Tensor) else torch.from_numpy(np.asarray(all_embs))
        if noises is not None:
            all_embs = all_embs + torch.sign(all_embs) * torch.nn.functional.normalize(noises[layer_idx], dim=-1) * eps
        embs.append(all_embs)
        if layer_cl is not None and layer_idx == layer_cl - 1:
            all_embs_cl = all_embs
    mean = torch.mean(torch.stack(embs, dim=1), dim=1)
    out = torch.split(mean, [user_w.shape[0], item_w.shape[0]])
    if layer_cl is not None:
        return out + torch.split(all_embs_cl, [user_w.shape[0], item_w.shape[0]])
    return out


def simgcl_cl_loss(x1, x2, temperature, reduce="sum"):
    """calculate_cl_loss — simgcl.py:40-46 (sum) / xsimgcl.py:48-54 (mean)."""
    x1, x2 = torch.nn.functional.normalize(x1, dim=-1), torch.nn.functional.normalize(x2, dim=-1)
    pos = torch.exp((x1 * x2).sum(dim=-1) / temperature)
    ttl = torch.exp(torch.matmul(x1, x2.transpose(0, 1)) / temperature).sum(dim=1)
    v = -torch.log(pos / ttl)
    return v.sum() if reduce == "sum" else v.mean()


def calc_ssl_loss(user_list, pos_item_list, user_sub1, user_sub2, item_sub1, item_sub2, ssl_tau, ssl_weight):
    """SGL.calc_ssl_loss — recbole_gnn/model/general_recommender/sgl.py:176-209, statement by statement
    (normalize; v1 = exp(<a,p>/tau); v2 = sum_j exp(<a,c_j>/tau); -sum log(v1/v2); users then items)."""
    nrm = torch.nn.functional.normalize
    u1, u2, all_u2 = nrm(user_sub1[user_list], dim=1), nrm(user_sub2[user_list], dim=1), nrm(user_sub2, dim=1)
    v1 = torch.exp(torch.sum(u1 * u2, dim=1) / ssl_tau)
    v2 = torch.sum(torch.exp(u1.matmul(all_u2.T) / ssl_tau), dim=1)
    ssl_user = -torch.sum(torch.log(v1 / v2))
    i1, i2, all_i2 = nrm(item_sub1[pos_item_list], dim=1), nrm(item_sub2[pos_item_list], dim=1), nrm(item_sub2, dim=1)
    v3 = torch.exp(torch.sum(i1 * i2, dim=1) / ssl_tau)
    v4 = torch.sum(torch.exp(i1.matmul(all_i2.T) / ssl_tau), dim=1)
    ssl_item = -torch.sum(torch.log(v3 / v4))
    return (ssl_item + ssl_user) * ssl_weight


def lse_rows(q, c, scale):
    """log of the InfoNCE denominator (sgl.py:195-198): log sum_j exp(scale * <q_b, c_j>), float64."""
    x = torch.as_tensor(q, dtype=torch.float64) @ torch.as_tensor(c, dtype=torch.float64).T * scale
    return torch.log(torch.exp(x).sum(dim=1))


# --------------------------------------------------------------------------------------------
# synthetic inputs — mirrored by the product's own generator; kept here so the
# oracle side of a test never imports product code to make its inputs.
# --------------------------------------------------------------------------------------------

def xavier_uniform(rows, d, gen):
    """recbole xavier_uniform_initialization on nn.Embedding (lightgcn.py:57): U(+-sqrt(6/(rows+d)))."""
    bound = float(np.sqrt(6.0 / (rows + d)))
    return (torch.rand(rows, d, generator=gen, dtype=torch.float32) * 2 - 1) * bound


# --------------------------------------------------------------------------------------------
# NCL (recbole_gnn/model/general_recommender/ncl.py)
# --------------------------------------------------------------------------------------------

def ncl_forward(user_w, item_w, conv, n_layers, hyper_layers):
    """ncl.py:93-104: max(n_layers, 2 * hyper_layers) propagations, every layer kept; the recommendation embedding is the
    mean of layers 0..n_layers."""
    x = torch.cat([user_w, item_w], dim=0)
    embs = [x]
    for _ in range(max(n_layers, hyper_layers * 2)):
        x = conv(x)
        embs.append(x)
    mean = torch.mean(torch.stack(embs[: n_layers + 1], dim=1), dim=1)
    return mean[: user_w.shape[0]], mean[user_w.shape[0]:], embs


def ncl_proto_nce_loss(node_embedding, n_users, user, item, user_centroids, user_2cluster, item_centroids, item_2cluster,
                       ssl_temp, proto_reg):
    """ncl.py:106-135, statement by statement."""
    import torch.nn.functional as F
    ua, ia = node_embedding[:n_users], node_embedding[n_users:]
    out = 0.0
    for table, idx, cents, n2c in ((ua, user, user_centroids, user_2cluster), (ia, item, item_centroids, item_2cluster)):
        norm = F.normalize(table[idx])
        pos = torch.exp((norm * cents[n2c[idx]]).sum(dim=1) / ssl_temp)
        ttl = torch.exp(norm.matmul(cents.T) / ssl_temp).sum(dim=1)
        out = out + (-torch.log(pos / ttl).sum())
    return proto_reg * out


def ncl_ssl_layer_loss(current, previous, n_users, user, item, ssl_temp, ssl_reg, alpha):
    """ncl.py:137-165, statement by statement."""
    import torch.nn.functional as F
    losses = []
    for cur, prev, idx in ((current[:n_users], previous[:n_users], user), (current[n_users:], previous[n_users:], item)):
        e1, e2, e_all = F.normalize(cur[idx]), F.normalize(prev[idx]), F.normalize(prev)
        pos = torch.exp((e1 * e2).sum(dim=1) / ssl_temp)
        ttl = torch.exp(e1.matmul(e_all.T) / ssl_temp).sum(dim=1)
        losses.append(-torch.log(pos / ttl).sum())
    return ssl_reg * (losses[0] + alpha * losses[1])


def kmeans_lloyd(x, init, niter=25):
    """Lloyd's algorithm in float64 from given starting centroids — the published algorithm behind ``faiss.Kmeans.train``
    (ncl.py:69-71; faiss is third-party, un-pinned, absent here).  Returns (centroids, assignment, objective per round).
    Empty clusters keep their centroid (the callers' test data has none; faiss would split a populated cluster)."""
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(init, dtype=np.float64).copy()
    obj = []
    for _ in range(niter):
        d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
        a = d2.argmin(1)
        obj.append(float(d2[np.arange(len(x)), a].sum()))
        for j in range(len(c)):
            m = a == j
            if m.any():
                c[j] = x[m].mean(0)
    d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
    return c, d2.argmin(1), obj
