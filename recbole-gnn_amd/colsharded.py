"""Feature-COLUMN sharding of the propagation across the GPUs of one node (r04).

``Y = Â·X`` acts on every column of X independently, so the K layers of ``LightGCN.forward`` / ``SGL.forward``
(lightgcn.py:70-81, sgl.py:128-145) need NO communication at all when the embedding tables are cut by COLUMNS instead of by
rows: rank r holds the whole normalized adjacency (CSR 3.3 GB + column-slab plan ≈ 10 GB at the config-#5 shape, of 288 — replicated, as the
reference replicates the dataset on every host) and the ``d / P`` columns ``[r d/P, (r + 1) d/P)`` of the two tables, of every
layer, of the mean and of the Adam moments.  On an unstructured power-law graph this is the only sharding whose K layers scale:
the node-range mode (``sharded.py``, the mode the north_star names) ships 98 % of the table per layer at P = 4 on the
Amazon-Book shape (``profiles/r03_bench_n4_staged.json``: 36 061 owned vs 106 056 halo rows), this one ships nothing.  A rank's
slab is exactly what one XCD role of ``csrc/sell.hip`` already works on: d = 128 over P = 4 ranks is ``sell_spmm_kernel<32, 1, ·>``
on 32 columns, d = 64 over P = 2 likewise.

What DOES need the other ranks' columns is the loss, per mini-batch:
* BPR (lightgcn.py:98-100): a score is a dot product over all d columns = the SUM over ranks of the partial dots — one
  all-reduce of ``[2 B]`` floats per step; its gradient w.r.t. a rank's columns needs only the (replicated) score derivative;
* EmbLoss (lightgcn.py:103-108): norms over all columns = sqrt / sum of the all-reduced partial sums of squares (3 scalars);
* InfoNCE (sgl.py:176-209): normalisation needs the row norms (an all-reduce of ``[B]`` partial sums of squares) and the
  logits ``[B, n]`` are sums of partial logits — an all-reduce of the logit matrix, which is what makes this mode a LightGCN /
  BPR mode first; SGL's denominators are better served by the node-range mode's distributed logsumexp (sharded_train.py).
* full_sort_predict (lightgcn.py:123-133): the score block ``[B, n_items]`` is the all-reduced sum of the ranks' partial GEMMs,
  or — cheaper — an all-gather of the ``[B, d/P]`` user rows and of the item slabs once per evaluation.

Every rank evaluates the same scalar loss; gradients are exact (tests: gloo world sizes 2 and 4 on CPU against the single-
device restatement; two ranks sharing one GPU through the HIP backend).  UNMEASURED on more than one GPU (no multi-GPU box
in this round): the claim is structural — zero bytes on the wire in the K layers — not a scaling number.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def column_range(d, rank, world):
    if d % world:
        raise ValueError(f"embedding width {d} is not a multiple of the {world} ranks")
    w = d // world
    return rank * w, (rank + 1) * w


def _all_reduce_sum(t, group):
    """SUM all-reduce of a tensor that may live on a GPU while the group is a host (gloo) group."""
    if group is None or dist.get_world_size(group) == 1:
        return t
    if t.device.type == "cuda" and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


class _ReduceOverRanks(torch.autograd.Function):
    """total = sum over ranks of `partial` (replicated result).  Every rank goes on to evaluate the SAME loss from `total`, so
    the derivative of that one loss w.r.t. this rank's partial is the derivative w.r.t. the total: backward is the identity."""

    @staticmethod
    def forward(ctx, partial, group):
        return _all_reduce_sum(partial.detach().clone(), group)

    @staticmethod
    def backward(ctx, grad):
        return grad, None


class _ColumnLightGCN(torch.autograd.Function):
    """(user_all, item_all)[:, cols] from the rank's column slab of E0: K local products + the layer mean, no communication;
    backward = the transposed (= the same: Â is symmetric) chain on the slab."""

    @staticmethod
    def forward(ctx, e0_slab, prop, n_layers):
        ctx.prop, ctx.n_layers = prop, n_layers
        return prop.forward(e0_slab.detach(), n_layers)

    @staticmethod
    def backward(ctx, grad):
        return ctx.prop.backward(grad.contiguous(), ctx.n_layers), None, None


class ColumnShardedPropagation:
    """One rank's share: the replicated graph + columns ``[lo, hi)`` of everything dense.

    ``backend``: ``sharded.HipBackend`` (the product: librbgnn.so on this rank's GPU) or any object with the same
    ``make_graph / spmm`` calls (the tests inject a CPU test double).  ``graph``: the backend's handle of the FULL normalized
    adjacency (``backend.make_graph(csr, n, n_users)`` or a ``GraphHandle`` built from the interactions)."""

    def __init__(self, graph, n_users, n_items, d, backend, rank=0, world=1, group=None):
        self.graph, self.backend, self.group = graph, backend, group
        self.n_users, self.n_items, self.d = int(n_users), int(n_items), int(d)
        self.rank, self.world = int(rank), int(world)
        self.lo, self.hi = column_range(d, rank, world)
        self.width = self.hi - self.lo

    # -- layout ------------------------------------------------------------------------------------------------------------
    def slab_of(self, table):
        """This rank's columns of a full-width [rows, d] table (a contiguous copy)."""
        return table[:, self.lo:self.hi].contiguous()

    def gather_columns(self, slab):
        """[rows, d] from every rank's [rows, d / P] slab (evaluation: the item table once, the batch's user rows per batch)."""
        if self.world == 1:
            return slab
        staged = slab.device.type == "cuda" and dist.get_backend(self.group) != "nccl"
        src = slab.cpu() if staged else slab
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src.contiguous(), group=self.group)
        return torch.cat(parts, dim=1).to(slab.device)

    # -- the K layers: local ------------------------------------------------------------------------------------------------
    def _fast(self):
        return hasattr(self.backend, "_ops") and self.width in (32, 64, 128)

    def forward(self, e0_slab, n_layers):
        """mean(E0, ÂE0, ..., Â^K E0) on the slab (lightgcn.py:70-81): zero bytes exchanged."""
        if self._fast():  # the fused propagation of the HIP engine on a width the column-slab plan serves
            nu = self.n_users
            return self.backend._ops.lightgcn_forward_raw(self.graph, e0_slab[:nu].contiguous(), e0_slab[nu:].contiguous(), n_layers)[0]
        acc, cur = e0_slab.clone(), e0_slab
        for _ in range(n_layers):
            nxt = torch.empty_like(cur)
            self.backend.spmm(self.graph, cur, nxt, False)
            acc += nxt
            cur = nxt
        return acc / (n_layers + 1)

    def backward(self, grad_slab, n_layers):
        """dE0 = (g + Â(g + Â(... + Âg))) / (K + 1) on the slab."""
        cur = grad_slab
        for _ in range(n_layers):
            nxt = torch.empty_like(cur)
            self.backend.spmm(self.graph, cur, nxt, False)
            cur = nxt + grad_slab
        return cur / (n_layers + 1)

    def propagate(self, e0_slab, n_layers):
        """autograd-aware ``forward``"""
        return _ColumnLightGCN.apply(e0_slab, self, n_layers)

    # -- what the loss needs from the other ranks ------------------------------------------------------------------------------
    def row_dots(self, a_slab, b_slab):
        """sum over ALL d columns of a * b per row, replicated: the all-reduced partial dots ([B] floats on the wire)."""
        return _ReduceOverRanks.apply((a_slab * b_slab).sum(dim=1), self.group)

    def sum_squares(self, *slabs):
        """[len(slabs)] total sums of squares over all columns (3 floats on the wire)."""
        return _ReduceOverRanks.apply(torch.stack([(s * s).sum() for s in slabs]), self.group)

    def full_sort_scores(self, mean_slab, users):
        """scores[b, j] = <user_all[users[b]], item_all[j]> (lightgcn.py:123-133): the sum over ranks of the partial GEMMs."""
        nu = self.n_users
        part = mean_slab[users] @ mean_slab[nu:].T
        return _all_reduce_sum(part.contiguous(), self.group)


class ColumnShardedTrainer:
    """LightGCN's training step (lightgcn.py:83-110: BPR + EmbLoss, Adam) on a column-sharded model: per step ONE all-reduce of
    2 B score partials and one of 3 scalars; the propagation, its backward and the optimizer never leave the rank."""

    def __init__(self, prop, e0_slab, n_layers, lr=1e-3, reg_weight=1e-5, require_pow=False):
        self.prop, self.n_layers = prop, int(n_layers)
        self.reg_weight, self.require_pow = reg_weight, require_pow
        self.e0 = e0_slab.detach().clone().requires_grad_(True)
        self.opt = torch.optim.Adam([self.e0], lr=lr)  # (element-wise: the slab's update equals the full table's columns)

    def loss(self, user, pos_item, neg_item):
        p, nu = self.prop, self.prop.n_users
        dev = self.e0.device
        user, pos_item, neg_item = (t.to(dev, torch.int64) for t in (user, pos_item, neg_item))
        out = p.propagate(self.e0, self.n_layers)
        ue, pe, ne = out[user], out[nu + pos_item], out[nu + neg_item]
        b = user.shape[0]
        scores = _ReduceOverRanks.apply(torch.cat([(ue * pe).sum(1), (ue * ne).sum(1)]), p.group)
        bpr = -torch.log(1e-10 + torch.sigmoid(scores[:b] - scores[b:])).mean()  # recbole BPRLoss
        sq = p.sum_squares(self.e0[user], self.e0[nu + pos_item], self.e0[nu + neg_item])  # recbole EmbLoss(norm = 2)
        reg = (sq.sum() / b / 2) if self.require_pow else (torch.sqrt(sq).sum() / b)
        return (bpr + self.reg_weight * reg).reshape(())

    def step(self, user, pos_item, neg_item):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(user, pos_item, neg_item)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
