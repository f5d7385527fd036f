"""Host-side mirror of the reference's model interface for the accelerated path:
``GeneralGraphRecommender`` (recbole_gnn/model/abstract_recommender.py:7-20), ``LightGCN``
(general_recommender/lightgcn.py), ``NGCF`` (ngcf.py) and ``SGL``'s propagation (sgl.py:73-145,235-240).

Same attribute names (``edge_index``, ``edge_weight``, ``use_sparse``, ``restore_user_e`` ...),
same method names and argument meaning, same cache/invalidate behaviour; the arithmetic runs in
librbgnn.so.  ``config`` is any mapping (missing keys read as None, like RecBole's Config);
``interaction`` is any mapping of field name -> tensor.  RecBole's own pieces used on the path
(BPRLoss, EmbLoss, Xavier init; recbole==1.1.1) are restated minimally below.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


_ONCE_CACHE = {}


def _once_mask(ids, n_ids=None):
    """mask[i] = True for exactly ONE occurrence of every distinct id in `ids` — what ``torch.unique`` selects, with static
    shapes, so a loss over the unique ids can be written as a masked loss over the whole batch and the step stays capturable
    into a HIP graph.  With ``n_ids`` (an upper bound of the ids) on a GPU: every position writes its index into a table slot of
    its id and the one that stays is the occurrence kept — three launches instead of a sort's dozen; WHICH occurrence survives
    is not fixed, and no loss depends on it (the rows of equal ids are equal)."""
    if n_ids is not None and ids.is_cuda:
        # the slot table and the position vector are kept per (device, stream, n_ids, batch): four calls per SimGCL / XSimGCL step
        # otherwise allocate 8-16 MB each on million-node catalogs.  The table needs no reset (only slots written
        # below are read) and the calls of one stream are ordered; tensors made while a stream is capturing belong to the graph's
        # pool and are not kept
        capturing = torch.cuda.is_current_stream_capturing()
        key = (ids.device.index, torch.cuda.current_stream(ids.device).cuda_stream, int(n_ids), int(ids.shape[0]))
        hit = None if capturing else _ONCE_CACHE.get(key)
        if hit is None:
            hit = (torch.arange(ids.shape[0], device=ids.device), torch.empty(int(n_ids), dtype=torch.int64, device=ids.device))
            if not capturing:
                if len(_ONCE_CACHE) >= 16:
                    _ONCE_CACHE.clear()
                _ONCE_CACHE[key] = hit
        ar, slot = hit
        if get_option("deterministic"):
            # option "deterministic": the FIRST occurrence stays (an integer atomic minimum: order-free), so the masked sums run over
            # the same positions in every run and the step is bit-stable
            slot.scatter_reduce_(0, ids, ar, "amin", include_self=False)
        else:
            slot.scatter_(0, ids, ar)
        return slot.index_select(0, ids) == ar
    s, order = torch.sort(ids)
    first = torch.ones_like(s, dtype=torch.bool)
    first[1:] = s[1:] != s[:-1]
    mask = torch.empty_like(first)
    mask[order] = first
    return mask


def _rows(table, idx):
    """``table[idx]`` / ``embedding(idx)`` of the reference, spelled index_select: same values, but the backward is one
    atomic ``index_add_`` launch instead of torch's sort-based index_put / embedding backward (115 us per lookup at
    batch 2048 on the Gowalla shape, r01 kernel stats of the SGL step)."""
    return table.index_select(0, idx)
from .graph import GraphHandle, InteractionDataset, get_option


class _Config(dict):
    def __getitem__(self, k):
        return self.get(k, None)


class BPRLoss(nn.Module):
    """recbole.model.loss.BPRLoss: -log(gamma + sigmoid(pos - neg)).mean(), gamma = 1e-10."""

    def __init__(self, gamma=1e-10):
        super().__init__()
        self.gamma = gamma

    def forward(self, pos_score, neg_score):
        return -torch.log(self.gamma + torch.sigmoid(pos_score - neg_score)).mean()


class EmbLoss(nn.Module):
    """recbole.model.loss.EmbLoss(norm=2)."""

    def __init__(self, norm=2):
        super().__init__()
        self.norm = norm

    def forward(self, *embeddings, require_pow=False):
        emb_loss = torch.zeros(1, device=embeddings[-1].device)
        if require_pow:
            for e in embeddings:
                emb_loss = emb_loss + torch.pow(input=torch.norm(e, p=self.norm), exponent=self.norm)
            emb_loss = emb_loss / embeddings[-1].shape[0]
            emb_loss = emb_loss / self.norm
        else:
            for e in embeddings:
                emb_loss = emb_loss + torch.norm(e, p=self.norm)
            emb_loss = emb_loss / embeddings[-1].shape[0]
        return emb_loss


def xavier_uniform_initialization(module):
    if isinstance(module, nn.Embedding):
        nn.init.xavier_uniform_(module.weight.data)
    elif isinstance(module, nn.Linear):
        nn.init.xavier_uniform_(module.weight.data)
        if module.bias is not None:
            nn.init.constant_(module.bias.data, 0)


def xavier_normal_initialization(module):
    if isinstance(module, nn.Embedding):
        nn.init.xavier_normal_(module.weight.data)
    elif isinstance(module, nn.Linear):
        nn.init.xavier_normal_(module.weight.data)
        if module.bias is not None:
            nn.init.constant_(module.bias.data, 0)


class GeneralGraphRecommender(nn.Module):
    """abstract_recommender.py:7-20.  Asks the dataset for Â, decides ``use_sparse`` and moves the
    graph to the device.  With ``enable_sparse`` falsy the reference's ``(edge_index, edge_weight)``
    tensors are kept (API parity) and the same normalized graph is also placed in HBM as a handle —
    both branches run the one CSR kernel."""

    USER_ID, ITEM_ID, NEG_ITEM_ID = "user_id", "item_id", "neg_item_id"

    def __init__(self, config, dataset):
        super().__init__()
        config = _Config(config)
        self.config = config
        self.device = torch.device(config["device"] or "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X engine needs config['device'] = 'cuda' (no CPU path)")
        self.n_users = dataset.num(self.USER_ID)
        self.n_items = dataset.num(self.ITEM_ID)
        self.dataset = dataset
        self.use_sparse = bool(config["enable_sparse"] and dataset.is_sparse)
        xcd_part = config["xcd_partition"]  # engine extension: None | "auto" | node -> community array
        if self.use_sparse:
            self.edge_index, self.edge_weight = dataset.get_norm_adj_mat(enable_sparse=True, device=self.device,
                                                                         xcd_part=xcd_part)
            self.graph = self.edge_index
        else:
            ei, ew = dataset.get_norm_adj_mat(enable_sparse=config["enable_sparse"])
            self.edge_index, self.edge_weight = ei.to(self.device), ew.to(self.device)
            self.graph = GraphHandle.from_interactions(dataset.uid, dataset.iid, self.n_users, self.n_items,
                                                       device=self.device, xcd_part=xcd_part)


class LightGCN(GeneralGraphRecommender):
    """general_recommender/lightgcn.py:36-133."""

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        self.latent_dim = config["embedding_size"] or 64
        self.n_layers = config["n_layers"] if config["n_layers"] is not None else 2
        self.reg_weight = config["reg_weight"] if config["reg_weight"] is not None else 1e-5
        self.require_pow = bool(config["require_pow"])
        self.fused = config["fused_forward"] if config["fused_forward"] is not None else True

        self.user_embedding = nn.Embedding(self.n_users, self.latent_dim)
        self.item_embedding = nn.Embedding(self.n_items, self.latent_dim)
        self.gcn_conv = ops.LightGCNConv(dim=self.latent_dim)
        self.mf_loss = BPRLoss()
        self.reg_loss = EmbLoss()
        self.restore_user_e = None
        self.restore_item_e = None
        self.apply(xavier_uniform_initialization)
        self.other_parameter_name = ["restore_user_e", "restore_item_e"]
        self.to(self.device)

    def get_ego_embeddings(self):
        return torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)

    def forward(self):
        if self.fused:  # K launches, no cat / stack / mean passes (lightgcn.py:70-81 in one call)
            mean = ops.lightgcn_forward(self.graph, self.user_embedding.weight, self.item_embedding.weight,
                                        self.n_layers)
        else:  # the reference's op-by-op structure on the same kernel
            all_embeddings = self.get_ego_embeddings()
            embeddings_list = [all_embeddings]
            for _ in range(self.n_layers):
                all_embeddings = self.gcn_conv(all_embeddings, self.graph, None)
                embeddings_list.append(all_embeddings)
            mean = ops.layer_mean(embeddings_list)
        return torch.split(mean, [self.n_users, self.n_items])

    def calculate_loss(self, interaction):
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        user = interaction[self.USER_ID]
        pos_item = interaction[self.ITEM_ID]
        neg_item = interaction[self.NEG_ITEM_ID]
        user_all, item_all = self.forward()
        # BPR + reg with their gradients in three library launches when user_all / item_all are the halves of one [N, d] tensor
        fused = ops.bpr_emb_loss(user_all, item_all, self.user_embedding.weight, self.item_embedding.weight, user, pos_item, neg_item,
                                 self.reg_weight, self.require_pow) if type(self.mf_loss) is BPRLoss and self.mf_loss.gamma == 1e-10 else None
        if fused is not None:
            return fused[0] + fused[1]
        u_e, pos_e, neg_e = _rows(user_all, user), _rows(item_all, pos_item), _rows(item_all, neg_item)
        pos_scores = torch.mul(u_e, pos_e).sum(dim=1)
        neg_scores = torch.mul(u_e, neg_e).sum(dim=1)
        mf_loss = self.mf_loss(pos_scores, neg_scores)
        reg_loss = self.reg_loss(_rows(self.user_embedding.weight, user), _rows(self.item_embedding.weight, pos_item),
                                 _rows(self.item_embedding.weight, neg_item), require_pow=self.require_pow)
        return mf_loss + self.reg_weight * reg_loss

    def predict(self, interaction):
        user = interaction[self.USER_ID]
        item = interaction[self.ITEM_ID]
        user_all, item_all = self.forward()
        return torch.mul(user_all[user], item_all[item]).sum(dim=1)

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID]
        if self.restore_user_e is None or self.restore_item_e is None:
            with torch.no_grad():
                self.restore_user_e, self.restore_item_e = self.forward()
        u_embeddings = ops.gather_rows(self.restore_user_e, user)
        scores = ops.score(u_embeddings, self.restore_item_e)
        return scores.view(-1)


def _full_sort_topk(model, interaction, k, history=None):
    """Evaluation without the score matrix: cached propagation (as full_sort_predict) + fused scoring / masking / top-k.
    ``history``: graph whose user rows are masked (default: the training graph)."""
    if model.restore_user_e is None or model.restore_item_e is None:
        with torch.no_grad():
            out = model.forward()
            model.restore_user_e, model.restore_item_e = out[0], out[1]
    return ops.full_sort_topk(history if history is not None else model.graph, model.restore_user_e, model.restore_item_e,
                              interaction[model.USER_ID], k)


LightGCN.full_sort_topk = _full_sort_topk


class NGCF(GeneralGraphRecommender):
    """general_recommender/ngcf.py:36-149.  Defaults are NGCF.yaml's (``message_dropout`` 0.1, ``node_dropout`` 0.0).
    The reference's ``nn.Dropout(p)(x)`` (ngcf.py:97) is a fresh module, hence active even under ``model.eval()``
    (SURVEY.md Q3): value parity is defined at ``message_dropout = 0`` (the parity tests pass it explicitly)."""

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        self.embedding_size = config["embedding_size"] or 64
        self.hidden_size_list = [self.embedding_size] + list(config["hidden_size_list"] or [64, 64, 64])
        self.node_dropout = config["node_dropout"] or 0.0
        self.message_dropout = config["message_dropout"] if config["message_dropout"] is not None else 0.1  # NGCF.yaml
        self.reg_weight = config["reg_weight"] if config["reg_weight"] is not None else 1e-5
        self.fused = config["fused_forward"] if config["fused_forward"] is not None else True  # False: op-by-op like ngcf.py
        self._drop = None  # lazily built state of the edge-dropout path (node_dropout > 0)
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_size)
        self.item_embedding = nn.Embedding(self.n_items, self.embedding_size)
        self.GNNlayers = nn.ModuleList(
            ops.BiGNNConv(i, o) for i, o in zip(self.hidden_size_list[:-1], self.hidden_size_list[1:]))
        self.mf_loss = BPRLoss()
        self.reg_loss = EmbLoss()
        self.restore_user_e = None
        self.restore_item_e = None
        self.apply(xavier_normal_initialization)
        self.other_parameter_name = ["restore_user_e", "restore_item_e"]
        self.to(self.device)

    def get_ego_embeddings(self):
        return torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)

    def _draw_edge_keep(self, n_edges):
        """dropout_adj's mask over the directed edges (PyG: ``torch.rand(E) >= p``; ngcf.py:81,89)."""
        return torch.rand(n_edges, device=self.device) >= self.node_dropout

    def _dropout_graph(self):
        """ngcf.py:74-90 with node_dropout > 0 in training mode: every forward drops each DIRECTED edge independently with
        probability p and keeps the surviving weights as they are (PyG ``dropout_adj``: no rescale, no re-normalisation,
        the two directions of an interaction fall independently — SURVEY A.4), the reference by rebuilding and
        re-sorting a SparseTensor per forward.  Here the sparsity structure and launch plan stay: a dropped edge is a
        zero weight in a re-weighted view of the graph (``GraphHandle.reweighted``); the matrix is no longer symmetric,
        so the backward runs on the transposed view, whose weights are the same mask read through the transpose map."""
        if self._drop is None:
            base = self.graph
            val = base.values()
            buf, buf_t = torch.empty_like(val), torch.empty_like(val)
            fwd, bwd = base.reweighted(buf), base.reweighted(buf_t)
            fwd._transpose, bwd._transpose = bwd, fwd
            self._drop = dict(val=val, tmap=base.transpose_map().long(), buf=buf, buf_t=buf_t, graph=fwd)
        st = self._drop
        keep = self._draw_edge_keep(st["val"].shape[0]).to(torch.float32)
        torch.mul(st["val"], keep, out=st["buf"])
        torch.mul(st["val"], keep.index_select(0, st["tmap"]), out=st["buf_t"])  # A^T[r,c] = A[c,r]; the weights are symmetric
        st["graph"].refresh_values()  # (the views' column-slab plans hold a copy of the weights)
        st["graph"]._transpose.refresh_values()
        return st["graph"]

    def _layer_outputs(self):
        """[E(0), E(1), ..., E(K)] of ngcf.py:92-99 — what ngcf.py:100 concatenates — each [N, d_k]."""
        graph = self.graph
        if self.node_dropout != 0 and self.training:
            graph = self._dropout_graph()
        all_embeddings = self.get_ego_embeddings()
        embeddings_list = [all_embeddings]
        if self.fused and isinstance(graph, ops.GraphHandle) and max(self.hidden_size_list) <= 128:
            # each layer with its LeakyReLU -> dropout -> normalize tail is one forward and one backward library call.
            # The dropout mask is drawn on EVERY forward, training or not, like the reference's fresh
            # nn.Dropout(p)(x) (ngcf.py:97); NGCF.yaml ships message_dropout = 0.1.
            for gnn in self.GNNlayers:
                all_embeddings = ops.bignn_layer(all_embeddings, gnn.lin1.weight, gnn.lin1.bias, gnn.lin2.weight, gnn.lin2.bias,
                                                 graph, 0.2, p_drop=self.message_dropout)
                embeddings_list += [all_embeddings]
            return embeddings_list
        for gnn in self.GNNlayers:
            all_embeddings = gnn(all_embeddings, graph, None)
            all_embeddings = F.leaky_relu(all_embeddings, negative_slope=0.2)
            all_embeddings = nn.Dropout(self.message_dropout)(all_embeddings)
            all_embeddings = F.normalize(all_embeddings, p=2, dim=1)
            embeddings_list += [all_embeddings]
        return embeddings_list

    def forward(self):
        if not (self.node_dropout != 0 and self.training) and not torch.is_grad_enabled() and self.message_dropout == 0 and self.fused:
            return self._forward_fused()
        ngcf_all_embeddings = torch.cat(self._layer_outputs(), dim=1)
        return torch.split(ngcf_all_embeddings, [self.n_users, self.n_items])

    def _forward_fused(self):
        """Inference: every layer writes straight into its column block of the concatenated
        [N, sum(d)] buffer (ngcf.py:100) with the LeakyReLU + L2-normalize tail fused."""
        n = self.n_users + self.n_items
        widths = self.hidden_size_list
        out = torch.empty((n, sum(widths)), dtype=torch.float32, device=self.device)
        out[: self.n_users, : widths[0]] = self.user_embedding.weight
        out[self.n_users:, : widths[0]] = self.item_embedding.weight
        off = 0
        for gnn, d_in, d_out in zip(self.GNNlayers, widths[:-1], widths[1:]):
            x = out[:, off: off + d_in]
            y = out[:, off + d_in: off + d_in + d_out]
            ops.bignn_conv_raw(self.graph, x, gnn.lin1.weight, gnn.lin1.bias, gnn.lin2.weight, gnn.lin2.bias, out=y,
                               leaky_norm=True, slope=0.2)
            off += d_in
        return torch.split(out, [self.n_users, self.n_items])

    def calculate_loss(self, interaction):
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        user = interaction[self.USER_ID]
        pos_item = interaction[self.ITEM_ID]
        neg_item = interaction[self.NEG_ITEM_ID]
        # ngcf.py:113-117 looks the batch up in the [N, sum(d)] concatenation; the rows of a concatenation are the
        # concatenation of the rows, so the 3 B rows are gathered from each layer's output instead and the [N, sum(d)]
        # tensor (72 MB at the Gowalla shape, forward copy + backward un-concatenation) is never formed in training
        layers = self._layer_outputs()
        b = user.shape[0]
        idx = torch.cat([user, pos_item + self.n_users, neg_item + self.n_users])
        rows = torch.cat([_rows(layer, idx) for layer in layers], dim=1)
        u_e, pos_e, neg_e = rows[:b], rows[b:2 * b], rows[2 * b:]
        pos_scores = torch.mul(u_e, pos_e).sum(dim=1)
        neg_scores = torch.mul(u_e, neg_e).sum(dim=1)
        mf_loss = self.mf_loss(pos_scores, neg_scores)
        reg_loss = self.reg_loss(u_e, pos_e, neg_e)
        return mf_loss + self.reg_weight * reg_loss

    def predict(self, interaction):
        user = interaction[self.USER_ID]
        item = interaction[self.ITEM_ID]
        user_all, item_all = self.forward()
        return torch.mul(user_all[user], item_all[item]).sum(dim=1)

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID]
        if self.restore_user_e is None or self.restore_item_e is None:
            with torch.no_grad():
                self.restore_user_e, self.restore_item_e = self.forward()
        u_embeddings = ops.gather_rows(self.restore_user_e, user)
        scores = ops.score(u_embeddings, self.restore_item_e)
        return scores.view(-1)


class SGL(GeneralGraphRecommender):
    """The propagation side of general_recommender/sgl.py: view construction (:73-126), ``forward``
    with an optional per-layer graph list (:128-145) and ``full_sort_predict`` (:235-240), and ``calculate_loss`` (:211-233) as plain
    torch over the fused propagations (a fused InfoNCE kernel is a "next" row)."""

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        self._user = dataset.uid
        self._item = dataset.iid
        self.embed_dim = config["embedding_size"] or 64
        self.n_layers = int(config["n_layers"] if config["n_layers"] is not None else 3)
        self.aug_type = config["type"] or "ED"
        self.drop_ratio = config["drop_ratio"] if config["drop_ratio"] is not None else 0.1
        self.ssl_tau = config["ssl_tau"] if config["ssl_tau"] is not None else 0.5
        self.reg_weight = config["reg_weight"] if config["reg_weight"] is not None else 1e-5
        self.ssl_weight = config["ssl_weight"] if config["ssl_weight"] is not None else 0.05
        # engine extension: sample the augmentation masks on the GPU (torch's device RNG) and build the views from
        # device-resident interactions.  False = the reference's numpy calls on the host (np.random's global stream:
        # what RNG-parity tests need; 51 of the 51 ms of graph_construction at the Gowalla shape, r01).
        self.device_sampling = config["device_sampling"] if config["device_sampling"] is not None else True
        self._inter_dev = None
        self.reg_loss = EmbLoss()
        self.user_embedding = nn.Embedding(self.n_users, self.embed_dim)
        self.item_embedding = nn.Embedding(self.n_items, self.embed_dim)
        self.gcn_conv = ops.LightGCNConv(dim=self.embed_dim)
        self.restore_user_e = None
        self.restore_item_e = None
        self.apply(xavier_uniform_initialization)
        self.other_parameter_name = ["restore_user_e", "restore_item_e"]
        self.sub_graph1 = self.sub_graph2 = None
        # the three propagations of a step on three HIP streams (ops.lightgcn_forward_views): measured SLOWER — 328 vs 268 us for
        # the three forwards, 2.13 vs 1.70 ms per captured step at the Gowalla shape: each launch already spreads over every XCD
        # and keeps one slab of one table in each L2; three at once evict each other — so off by default
        self.concurrent_views = False
        self.to(self.device)

    def train(self, mode: bool = True):
        t = super().train(mode=mode)
        if mode:
            self.graph_construction()
        return t

    def graph_construction(self):
        if self.aug_type in ("ND", "ED"):
            self.sub_graph1 = [self.random_graph_augment()] * self.n_layers
            self.sub_graph2 = [self.random_graph_augment()] * self.n_layers
        elif self.aug_type == "RW":
            self.sub_graph1 = [self.random_graph_augment() for _ in range(self.n_layers)]
            self.sub_graph2 = [self.random_graph_augment() for _ in range(self.n_layers)]

    def random_graph_augment(self):
        """sgl.py:93-126: sample with numpy's global RNG exactly as the reference does, then rebuild
        and re-normalize the view (native builder, keep-mask form)."""
        if self.device_sampling:
            return self._random_graph_augment_device()

        def rand_sample(high, size=None, replace=True):
            return np.random.choice(np.arange(high), size=size, replace=replace)

        n_inter = len(self._user)
        keep_mask = np.zeros(n_inter, dtype=np.uint8)
        if self.aug_type == "ND":
            drop_user = rand_sample(self.n_users, size=int(self.n_users * self.drop_ratio), replace=False)
            drop_item = rand_sample(self.n_items, size=int(self.n_items * self.drop_ratio), replace=False)
            mask = np.isin(self._user.numpy(), drop_user)
            mask |= np.isin(self._item.numpy(), drop_item)
            keep_mask[~mask] = 1
        elif self.aug_type in ("ED", "RW"):
            keep = rand_sample(n_inter, size=int(n_inter * (1 - self.drop_ratio)), replace=False)
            keep_mask[keep] = 1
        graph = GraphHandle.from_interactions(self._user, self._item, self.n_users, self.n_items, device=self.device,
                                              keep=keep_mask)
        return graph, None

    def _random_graph_augment_device(self):
        """The same three augmentations with the draw on the GPU: a uniform sample WITHOUT replacement of exactly the
        reference's size (sgl.py:97-110) — ``torch.randperm(n, device)[:k]`` instead of ``np.random.choice(arange(n), k,
        replace=False)`` — and the view built by the device builder from interactions that stay in HBM."""
        if self._inter_dev is None:
            self._inter_dev = (self._user.to(self.device), self._item.to(self.device))
        u, i = self._inter_dev
        n_inter = u.shape[0]
        if self.aug_type == "ND":
            du = torch.zeros(self.n_users, dtype=torch.bool, device=self.device)
            di = torch.zeros(self.n_items, dtype=torch.bool, device=self.device)
            du[torch.randperm(self.n_users, device=self.device)[: int(self.n_users * self.drop_ratio)]] = True
            di[torch.randperm(self.n_items, device=self.device)[: int(self.n_items * self.drop_ratio)]] = True
            keep = ~(du[u] | di[i])
        else:  # ED / RW
            keep = torch.zeros(n_inter, dtype=torch.bool, device=self.device)
            keep[torch.randperm(n_inter, device=self.device)[: int(n_inter * (1 - self.drop_ratio))]] = True
        graph = GraphHandle.from_interactions(u, i, self.n_users, self.n_items, device=self.device, keep=keep)
        return graph, None

    def forward(self, graph=None):
        if graph is None:
            graphs = [self.graph]
        else:
            graphs = [g for g, _ in graph]
            if all(g is graphs[0] for g in graphs):
                graphs = graphs[:1]
        mean = ops.lightgcn_forward(graphs, self.user_embedding.weight, self.item_embedding.weight, self.n_layers)
        return torch.split(mean, [self.n_users, self.n_items], dim=0)

    @staticmethod
    def _info_nce(anchor, positive, candidates, tau):
        """-sum log( exp(<a,p>/tau) / sum_j exp(<a,c_j>/tau) ) over the batch, all vectors L2-normalised first
        (one half of calc_ssl_loss, sgl.py:176-209)."""
        a, p, c = F.normalize(anchor, dim=1), F.normalize(positive, dim=1), F.normalize(candidates, dim=1)
        if a.shape[1] > 128:  # the fused denominator covers d <= 128; wider rows take the reference's own formula
            pos = torch.exp((a * p).sum(dim=1) / tau)
            tot = torch.exp(a.matmul(c.T) / tau).sum(dim=1)
            return -torch.log(pos / tot).sum()
        # -log(exp(s) / sum exp) = lse - s; the [B, n] matrix of sgl.py:195-198 is never written (unit rows: shift = 1/tau)
        return (ops.lse_rows(a, c, 1.0 / tau, 1.0 / tau) - (a * p).sum(dim=1) / tau).sum()

    def calculate_loss(self, interaction):
        """sgl.py:211-233: BPR (sum-reduced logsigmoid form, :147-162) + reg on the ego embeddings + ssl_weight x
        (user InfoNCE + item InfoNCE between the two augmented views): three fused propagations; each InfoNCE
        half is one ``ops.info_nce`` call (rbg_infonce_f32: normalisation, positives, denominators, all gradients)."""
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        if self.sub_graph1 is None:
            self.graph_construction()
        user, pos, neg = interaction[self.USER_ID], interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]
        (u_all, i_all), (u1, i1), (u2, i2) = self.propagate_views()
        ue = _rows(u_all, user)
        bpr = -F.logsigmoid((ue * _rows(i_all, pos)).sum(1) - (ue * _rows(i_all, neg)).sum(1)).sum()
        reg = self.reg_loss(_rows(self.user_embedding.weight, user), _rows(self.item_embedding.weight, pos),
                            _rows(self.item_embedding.weight, neg))
        if u1.shape[1] <= 128:  # value and table gradients of each half in one library call (rbg_infonce_f32)
            ssl = ops.info_nce(u1, u2, user, self.ssl_tau) + ops.info_nce(i1, i2, pos, self.ssl_tau)
        else:
            ssl = self._info_nce(u1[user], u2[user], u2, self.ssl_tau) + self._info_nce(i1[pos], i2[pos], i2, self.ssl_tau)
        return bpr + self.reg_weight * reg + self.ssl_weight * ssl

    def propagate_views(self):
        """The three propagations of one SGL training step (sgl.py:219-221): the full graph and the two
        augmented views.  Returns [(user_all, item_all)] x 3."""
        if self.sub_graph1 is None:
            self.graph_construction()
        views = [[self.graph], [g for g, _ in self.sub_graph1], [g for g, _ in self.sub_graph2]]
        if self.concurrent_views and self.user_embedding.weight.is_cuda and all(all(g is v[0] for g in v) for v in views):
            # ND / ED: one graph per view -> three independent chains of K launches, issued on three HIP streams
            means = ops.lightgcn_forward_views([v[0] for v in views], self.user_embedding.weight, self.item_embedding.weight,
                                               self.n_layers)
            return [torch.split(m, [self.n_users, self.n_items], dim=0) for m in means]
        return [self.forward(), self.forward(self.sub_graph1), self.forward(self.sub_graph2)]

    def predict(self, interaction):
        if self.restore_user_e is None or self.restore_item_e is None:
            with torch.no_grad():
                self.restore_user_e, self.restore_item_e = self.forward()
        user = self.restore_user_e[interaction[self.USER_ID]]
        item = self.restore_item_e[interaction[self.ITEM_ID]]
        return torch.sum(user * item, dim=1)

    def full_sort_predict(self, interaction):
        if self.restore_user_e is None or self.restore_item_e is None:
            with torch.no_grad():
                self.restore_user_e, self.restore_item_e = self.forward()
        user = ops.gather_rows(self.restore_user_e, interaction[self.USER_ID])
        return ops.score(user, self.restore_item_e)


class SimGCL(LightGCN):
    """general_recommender/simgcl.py:15-61.  forward() averages layers 1..K (no E_0, simgcl.py:25-36); the perturbed pass
    adds sign(e) * normalize(U(0,1) noise) * eps after every layer — the noise is drawn with ``torch.rand_like`` in the
    reference's order, the add is the SpMM's epilogue (``ops.spmm_noise``)."""

    # simgcl.py:52-53 restricts the contrast to ``torch.unique`` of the batch's ids: a data-dependent shape (a device-to-host sync
    # per step, no HIP-graph capture).  ``static_unique`` (default) computes the SAME loss with static shapes: one occurrence of
    # every id is kept by a mask over rows (the sum) and columns (the denominators) — equal up to summation order.
    graph_capturable = True

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        if config["reg_weight"] is None:
            self.reg_weight = 1e-4  # SimGCL.yaml / XSimGCL.yaml (LightGCN.yaml ships 1e-5)
        self.cl_rate = config["lambda"] if config["lambda"] is not None else 0.5
        self.eps = config["eps"] if config["eps"] is not None else 0.1
        self.temperature = config["temperature"] if config["temperature"] is not None else 0.2
        self.static_unique = config["static_unique"] if config["static_unique"] is not None else True
        self.graph_capturable = bool(self.static_unique)

    def _layers(self, perturbed):
        all_embs = self.get_ego_embeddings()
        graph = self.graph if isinstance(self.graph, ops.GraphHandle) else ops.graph_from_pair(
            self.edge_index, self.edge_weight, all_embs.shape[0], all_embs.device)
        out = []
        for _ in range(self.n_layers):
            if perturbed:
                buf = torch.empty_like(all_embs)
                random_noise = torch.rand_like(buf)  # simgcl.py:31, drawn before the product here: same RNG stream
                all_embs = ops.spmm_noise(graph, all_embs, random_noise, self.eps)
            else:
                all_embs = ops.spmm(graph, all_embs)
            out.append(all_embs)
        return out

    def forward(self, perturbed=False):
        embeddings_list = self._layers(perturbed)
        mean = ops.layer_mean(embeddings_list)
        return torch.split(mean, [self.n_users, self.n_items])

    def calculate_cl_loss(self, x1, x2, once=None):
        """simgcl.py:38-43.  ``once``: the rows / columns that count (one occurrence per distinct id of the batch) when x1, x2
        hold the WHOLE batch instead of its unique ids."""
        if once is not None and x1.is_cuda and x1.shape[1] <= 128:
            # one library call: normalisations, positives, masked denominators and both gradients (rbg_infonce_masked_f32 on the
            # batch's gathered rows as both "tables"); ~40 elementwise / reduction launches per side in torch otherwise
            w = once.to(torch.float32)
            return ops.info_nce(x1, x2, torch.arange(x1.shape[0], device=x1.device), self.temperature, row_w=w, col_w=w)
        x1, x2 = F.normalize(x1, dim=-1), F.normalize(x2, dim=-1)
        pos_score = torch.exp((x1 * x2).sum(dim=-1) / self.temperature)
        logits = torch.exp(torch.matmul(x1, x2.transpose(0, 1)) / self.temperature)
        if once is None:
            return -torch.log(pos_score / logits.sum(dim=1)).sum()
        w = once.to(logits.dtype)
        return -(torch.log(pos_score / (logits * w[None, :]).sum(dim=1)) * w).sum()

    def calculate_loss(self, interaction):
        loss = super().calculate_loss(interaction)
        user, pos_item = interaction[self.USER_ID], interaction[self.ITEM_ID]
        u1, i1 = self.forward(perturbed=True)
        u2, i2 = self.forward(perturbed=True)
        if self.static_unique:
            user_cl_loss = self.calculate_cl_loss(_rows(u1, user), _rows(u2, user), _once_mask(user, self.n_users))
            item_cl_loss = self.calculate_cl_loss(_rows(i1, pos_item), _rows(i2, pos_item), _once_mask(pos_item, self.n_items))
        else:
            user, pos_item = torch.unique(user), torch.unique(pos_item)
            user_cl_loss = self.calculate_cl_loss(_rows(u1, user), _rows(u2, user))
            item_cl_loss = self.calculate_cl_loss(_rows(i1, pos_item), _rows(i2, pos_item))
        return loss + self.cl_rate * (user_cl_loss + item_cl_loss)


class XSimGCL(SimGCL):
    """general_recommender/xsimgcl.py:18-90: ONE perturbed pass serves the recommendation loss and the contrast between
    the final embedding and the one after layer ``layer_cl``; the CL loss is a mean (xsimgcl.py:54)."""

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        self.cl_rate = config["lambda"] if config["lambda"] is not None else 0.1
        self.eps = config["eps"] if config["eps"] is not None else 0.2
        self.layer_cl = config["layer_cl"] if config["layer_cl"] is not None else 1

    def forward(self, perturbed=False):
        all_embs_cl = self.get_ego_embeddings()
        embeddings_list = self._layers(perturbed)
        if 1 <= self.layer_cl <= self.n_layers:
            all_embs_cl = embeddings_list[self.layer_cl - 1]
        mean = ops.layer_mean(embeddings_list)
        user_all, item_all = torch.split(mean, [self.n_users, self.n_items])
        if perturbed:
            user_cl, item_cl = torch.split(all_embs_cl, [self.n_users, self.n_items])
            return user_all, item_all, user_cl, item_cl
        return user_all, item_all

    def calculate_cl_loss(self, x1, x2, once=None):
        """xsimgcl.py:50-54: SimGCL's contrast as a MEAN over the (distinct) rows."""
        total = super().calculate_cl_loss(x1, x2, once)
        return total / (x1.shape[0] if once is None else once.sum().to(total.dtype))

    def calculate_loss(self, interaction):
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        user, pos_item, neg_item = interaction[self.USER_ID], interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]
        user_all, item_all, user_cl, item_cl = self.forward(perturbed=True)
        u_e, pos_e = _rows(user_all, user), _rows(item_all, pos_item)
        fused = ops.bpr_emb_loss(user_all, item_all, self.user_embedding.weight, self.item_embedding.weight, user, pos_item, neg_item,
                                 self.reg_weight, self.require_pow) if type(self.mf_loss) is BPRLoss and self.mf_loss.gamma == 1e-10 else None
        if fused is not None:
            mf_loss, reg_term = fused
        else:
            neg_e = _rows(item_all, neg_item)
            mf_loss = self.mf_loss(torch.mul(u_e, pos_e).sum(dim=1), torch.mul(u_e, neg_e).sum(dim=1))
            reg_term = self.reg_weight * self.reg_loss(_rows(self.user_embedding.weight, user), _rows(self.item_embedding.weight, pos_item),
                                                       _rows(self.item_embedding.weight, neg_item), require_pow=self.require_pow)
        if self.static_unique:
            user_cl_loss = self.calculate_cl_loss(u_e, _rows(user_cl, user), _once_mask(user, self.n_users))
            item_cl_loss = self.calculate_cl_loss(pos_e, _rows(item_cl, pos_item), _once_mask(pos_item, self.n_items))
        else:
            user_u, item_u = torch.unique(user), torch.unique(pos_item)
            user_cl_loss = self.calculate_cl_loss(_rows(user_all, user_u), _rows(user_cl, user_u))
            item_cl_loss = self.calculate_cl_loss(_rows(item_all, item_u), _rows(item_cl, item_u))
        return mf_loss, reg_term, self.cl_rate * (user_cl_loss + item_cl_loss)


class NCL(GeneralGraphRecommender):
    """general_recommender/ncl.py:20-219.  Propagation = LightGCN's with every layer kept (:93-104); the structure
    contrast (:137-165) is SGL's InfoNCE form on two layers of the same propagation, one ``ops.info_nce`` call per side;
    the prototype contrast (:106-135) takes its denominators from the fused logsumexp-GEMM (``ops.lse_rows``); the
    prototypes come from k-means on the device (``ops.kmeans``) where the reference calls faiss (:66-81).
    ``calculate_loss`` returns the reference's 3-tuple (BPR + reg, ssl, proto): NCLTrainer sums it, without the last
    term during the first ``warm_up_step`` epochs, and calls ``e_step`` every ``m_step`` epochs (trainer.py:35-40,130-133)."""

    # (capturable: e_step rewrites the prototype tensors IN PLACE, so a captured step keeps reading the current ones; which terms
    # of the loss tuple the trainer sums — trainer.py:130-133 — is the stepper's ``reduce``, see train.GraphedStep)
    graph_capturable = True

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        config = self.config
        self.latent_dim = config["embedding_size"] or 64
        self.n_layers = config["n_layers"] if config["n_layers"] is not None else 3
        self.reg_weight = config["reg_weight"] if config["reg_weight"] is not None else 1e-4
        self.ssl_temp = config["ssl_temp"] if config["ssl_temp"] is not None else 0.1
        self.ssl_reg = config["ssl_reg"] if config["ssl_reg"] is not None else 1e-7
        self.hyper_layers = config["hyper_layers"] if config["hyper_layers"] is not None else 1
        self.alpha = config["alpha"] if config["alpha"] is not None else 1
        self.proto_reg = config["proto_reg"] if config["proto_reg"] is not None else 8e-8
        self.k = config["num_clusters"] if config["num_clusters"] is not None else 1000
        self.m_step = config["m_step"] if config["m_step"] is not None else 1
        self.warm_up_step = config["warm_up_step"] if config["warm_up_step"] is not None else 20
        self.user_embedding = nn.Embedding(self.n_users, self.latent_dim)
        self.item_embedding = nn.Embedding(self.n_items, self.latent_dim)
        self.gcn_conv = ops.LightGCNConv(dim=self.latent_dim)
        self.mf_loss = BPRLoss()
        self.reg_loss = EmbLoss()
        self.restore_user_e = None
        self.restore_item_e = None
        self.apply(xavier_uniform_initialization)
        self.other_parameter_name = ["restore_user_e", "restore_item_e"]
        self.user_centroids = self.user_2cluster = self.item_centroids = self.item_2cluster = None
        self.to(self.device)

    def e_step(self):
        for side, table in (("user", self.user_embedding.weight), ("item", self.item_embedding.weight)):
            cent, assign = self.run_kmeans(table.detach())
            old_c, old_a = getattr(self, f"{side}_centroids"), getattr(self, f"{side}_2cluster")
            if old_c is not None and old_c.shape == cent.shape and old_a.shape == assign.shape:
                old_c.copy_(cent)  # same storage: a HIP graph captured on the previous prototypes reads the new ones
                old_a.copy_(assign)
            else:
                setattr(self, f"{side}_centroids", cent)
                setattr(self, f"{side}_2cluster", assign)

    def run_kmeans(self, x):
        """ncl.py:66-81: k clusters of the rows of x; centroids L2-normalized, node -> cluster as int64."""
        cent, assign = ops.kmeans(x.contiguous(), self.k)
        return F.normalize(cent, p=2, dim=1), assign

    def get_ego_embeddings(self):
        return torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)

    def forward(self):
        all_embeddings = self.get_ego_embeddings()
        embeddings_list = [all_embeddings]
        for _ in range(max(self.n_layers, self.hyper_layers * 2)):
            all_embeddings = self.gcn_conv(all_embeddings, self.graph, None)
            embeddings_list.append(all_embeddings)
        mean = ops.layer_mean(embeddings_list[: self.n_layers + 1])
        user_all, item_all = torch.split(mean, [self.n_users, self.n_items])
        return user_all, item_all, embeddings_list

    def _proto_side(self, table, idx, centroids, node2cluster):
        a = F.normalize(_rows(table, idx))
        pos = (a * centroids[node2cluster[idx]]).sum(dim=1) / self.ssl_temp
        if a.shape[1] <= 128:  # log sum_j exp(<a, c_j> / T) without the [B, k] matrix; unit rows: shift = 1 / T
            lse = ops.lse_rows(a, centroids, 1.0 / self.ssl_temp, 1.0 / self.ssl_temp)
        else:
            lse = torch.logsumexp(a.matmul(centroids.T) / self.ssl_temp, dim=1)
        return (lse - pos).sum()  # = -sum log(exp(pos) / sum_j exp(.))

    def ProtoNCE_loss(self, node_embedding, user, item):
        if self.user_centroids is None:
            raise RuntimeError("NCL.e_step() has not run: no prototypes yet (NCLTrainer calls it before the first epoch)")
        user_all, item_all = torch.split(node_embedding, [self.n_users, self.n_items])
        loss_u = self._proto_side(user_all, user, self.user_centroids, self.user_2cluster)
        loss_i = self._proto_side(item_all, item, self.item_centroids, self.item_2cluster)
        return self.proto_reg * (loss_u + loss_i)

    def ssl_layer_loss(self, current_embedding, previous_embedding, user, item):
        cu, ci = torch.split(current_embedding, [self.n_users, self.n_items])
        pu, pi = torch.split(previous_embedding, [self.n_users, self.n_items])
        if cu.shape[1] <= 128:  # normalize, positives, denominators over ALL previous rows, and the gradients: one call per side
            loss_u = ops.info_nce(cu, pu, user, self.ssl_temp)
            loss_i = ops.info_nce(ci, pi, item, self.ssl_temp)
        else:
            loss_u = SGL._info_nce(cu[user], pu[user], pu, self.ssl_temp)
            loss_i = SGL._info_nce(ci[item], pi[item], pi, self.ssl_temp)
        return self.ssl_reg * (loss_u + self.alpha * loss_i)

    def calculate_loss(self, interaction):
        if self.restore_user_e is not None or self.restore_item_e is not None:
            self.restore_user_e, self.restore_item_e = None, None
        user, pos_item, neg_item = interaction[self.USER_ID], interaction[self.ITEM_ID], interaction[self.NEG_ITEM_ID]
        user_all, item_all, embeddings_list = self.forward()
        center_embedding = embeddings_list[0]
        context_embedding = embeddings_list[self.hyper_layers * 2]
        ssl_loss = self.ssl_layer_loss(context_embedding, center_embedding, user, pos_item)
        proto_loss = self.ProtoNCE_loss(center_embedding, user, pos_item)
        fused = ops.bpr_emb_loss(user_all, item_all, self.user_embedding.weight, self.item_embedding.weight, user, pos_item, neg_item,
                                 self.reg_weight, False)
        if fused is not None:
            return fused[0] + fused[1], ssl_loss, proto_loss
        u_e, pos_e, neg_e = _rows(user_all, user), _rows(item_all, pos_item), _rows(item_all, neg_item)
        mf_loss = self.mf_loss(torch.mul(u_e, pos_e).sum(dim=1), torch.mul(u_e, neg_e).sum(dim=1))
        reg_loss = self.reg_loss(_rows(self.user_embedding.weight, user), _rows(self.item_embedding.weight, pos_item),
                                 _rows(self.item_embedding.weight, neg_item))
        return mf_loss + self.reg_weight * reg_loss, ssl_loss, proto_loss

    def predict(self, interaction):
        user_all, item_all, _ = self.forward()
        return torch.mul(user_all[interaction[self.USER_ID]], item_all[interaction[self.ITEM_ID]]).sum(dim=1)

    def full_sort_predict(self, interaction):
        user = interaction[self.USER_ID]
        if self.restore_user_e is None or self.restore_item_e is None:
            with torch.no_grad():
                self.restore_user_e, self.restore_item_e, _ = self.forward()
        return ops.score(ops.gather_rows(self.restore_user_e, user), self.restore_item_e).view(-1)


NGCF.full_sort_topk = _full_sort_topk
SGL.full_sort_topk = _full_sort_topk
NCL.full_sort_topk = _full_sort_topk
