"""A sharded SGL / LightGCN TRAINING step (BASELINE config #5: "SGL ... 8 x MI355X" as something trainable, not only
propagatable).  The reference trains on one device (sgl.py:211-233, lightgcn.py:83-110); here the embedding tables, the
propagated embeddings and the Adam moments live node-sharded across the ranks of `sharded.py`, and one step is:

1. the three propagations of this rank's E0 rows (``sharded_sgl_forward``: full graph + two edge-drop views, halo
   exchanges shared / overlapped as there), or the single LightGCN one;
2. the mini-batch (the same user / pos / neg ids on every rank — the reference's dataset is replicated on the host) needs
   rows that other ranks own: every rank fills the rows it owns into a zeroed [B, d] buffer and ONE all-reduce per table
   makes the batch rows replicated (``assemble_rows``; a row has exactly one owner, so the sum is a routing).  B = 2048,
   d = 128: 1 MB per table — nothing next to a halo exchange;
3. BPR, EmbLoss and the InfoNCE numerators are computed on the replicated rows (every rank the same arithmetic);
4. the InfoNCE denominators run over ALL users / ALL items (sgl.py:195-198): a DISTRIBUTED logsumexp — each rank takes
   ``lse_rows`` (rbg_lse_rows_f32: no [B, n] matrix) over the candidate rows it owns, the [B] partial results are
   all-gathered and combined with a logsumexp over ranks; backward: the softmax weight of the rank's share scales its local
   backward, the anchors' gradient is all-reduced (each rank holds only its candidates' part of it);
5. autograd runs the transposed chain of the sharded propagations (``ShardedPropagation.backward``); the gradient of the
   replicated batch rows goes to their owner only;
6. Adam on the owned rows (RecBole trains nn.Embedding with dense gradients: a per-row sharded dense Adam is the same update).

Every rank evaluates the same scalar loss, so its value needs no reduction; gradients are exact (tested against the
single-device ``SGL.calculate_loss`` / ``LightGCN.calculate_loss``: gloo world size 2 on CPU, two ranks sharing a GPU through
the HIP backend)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import sharded as sh


def _all_reduce_sum(t, group):
    """SUM all-reduce of a tensor that may live on a GPU while the group is a host (gloo) group."""
    if t.device.type == "cuda" and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def _all_gather_vec(v, world, group):
    """[world, len(v)] of a 1-D tensor from every rank (same staging rule as above)."""
    staged = v.device.type == "cuda" and dist.get_backend(group) != "nccl"
    src = v.cpu() if staged else v
    outs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(outs, src.contiguous(), group=group)
    return torch.stack(outs).to(v.device)


class _AssembleRows(torch.autograd.Function):
    """rows[b] = table[g_b] for GLOBAL node ids g of a node-sharded table, replicated on every rank."""

    @staticmethod
    def forward(ctx, table_local, pos, idx_local, n_rows, group, world):
        buf = table_local.new_zeros((n_rows, table_local.shape[1]))
        if pos.numel():
            buf.index_copy_(0, pos, table_local.index_select(0, idx_local))
        if world > 1:
            _all_reduce_sum(buf, group)
        ctx.save_for_backward(pos, idx_local)
        ctx.n_local = table_local.shape[0]
        return buf

    @staticmethod
    def backward(ctx, grad):
        pos, idx_local = ctx.saved_tensors
        g = grad.new_zeros((ctx.n_local, grad.shape[1]))
        if pos.numel():  # the loss is evaluated identically on every rank: the owner alone takes the row's gradient
            g.index_add_(0, idx_local, grad.index_select(0, pos))
        return g, None, None, None, None, None


class _DistLSE(torch.autograd.Function):
    """lse[b] = log sum over ALL candidates of exp(scale <a_b, c_j>): anchors replicated, candidates node-sharded."""

    @staticmethod
    def forward(ctx, anchors, cand_local, scale, shift, group, world, lse_fn):
        with torch.enable_grad():
            a = anchors.detach().requires_grad_(True)
            c = cand_local.detach().requires_grad_(True)
            if c.shape[0]:
                part = lse_fn(a, c, scale, shift)
            else:  # this rank owns no candidate of this side
                part = (a.sum(dim=1) * 0.0) + float("-inf")
        parts = _all_gather_vec(part.detach(), world, group) if world > 1 else part.detach()[None]
        lse = torch.logsumexp(parts, dim=0)
        ctx.saved = (a, c, part, lse)
        ctx.group, ctx.world = group, world
        return lse

    @staticmethod
    def backward(ctx, grad):
        a, c, part, lse = ctx.saved
        ctx.saved = None
        if c.shape[0]:
            w = torch.exp(part.detach() - lse) * grad
            ga, gc = torch.autograd.grad(part, (a, c), w)
        else:
            ga, gc = torch.zeros_like(a), torch.zeros_like(c)
        if ctx.world > 1:  # each rank holds the part of the anchors' gradient that flows through ITS candidates
            ga = _all_reduce_sum(ga.contiguous(), ctx.group)
        return ga, gc, None, None, None, None, None


def _lse_rows(a, c, scale, shift):
    if a.device.type == "cuda" and a.shape[1] <= 128:
        from . import ops
        return ops.lse_rows(a, c, scale, shift)
    return torch.logsumexp((a @ c.T) * scale, dim=1)


class ShardedTrainer:
    """One rank's share of a LightGCN / SGL training run.  ``plan``: the full graph's ShardPlan; ``view_plans``: the two
    edge-drop view plans on the SAME partition (``build_plans(owner=..., keep=mask)``; None / empty = plain LightGCN);
    ``e0_local``: this rank's rows of the two embedding tables (users first, ``plan.owned`` order)."""

    def __init__(self, plan, backend, e0_local, n_users, n_items, n_layers, view_plans=None, group=None, transport="nccl",
                 lr=1e-3, reg_weight=1e-5, ssl_tau=0.5, ssl_weight=0.05, require_pow=False, overlap=False):
        self.plan, self.group, self.transport = plan, group, transport
        self.n_users, self.n_items, self.n_layers = int(n_users), int(n_items), int(n_layers)
        self.reg_weight, self.ssl_tau, self.ssl_weight, self.require_pow = reg_weight, ssl_tau, ssl_weight, require_pow
        self.main = sh.ShardedPropagation(plan, backend, group=group, transport=transport, overlap=overlap)
        self.views = [self._make_view(p, backend, overlap) for p in (view_plans or [])]
        self.maps = [v.halo_map_from(plan) for v in self.views]
        dev = e0_local.device
        self.e0 = e0_local.detach().clone().requires_grad_(True)
        self.opt = torch.optim.Adam([self.e0], lr=lr)
        local = np.full(self.n_users + self.n_items, -1, dtype=np.int64)
        local[plan.owned] = np.arange(plan.n_owned)
        self.local_of = torch.from_numpy(local).to(dev)
        self.nu_local = int(plan.n_users_owned)
        # the collectives of the loss run on the group of the propagation's transport when that is a torch group; the
        # RCCL transport takes the default group (device tensors), the staged one its host group
        self._cgroup = group
        self.world = plan.world

    def _make_view(self, p, backend, overlap):
        """One view = one plan (ND / ED: the same sub-graph at every layer) or a list of K plans (RW: one per layer)."""
        if isinstance(p, (list, tuple)):
            return sh.LayeredShardedPropagation(list(p), backend, group=self.group, transport=self.transport, overlap=overlap)
        return sh.ShardedPropagation(p, backend, group=self.group, transport=self.transport, overlap=overlap)

    def set_views(self, view_plans):
        """A new pair of views (sgl.py:73-80: rebuilt once per epoch)."""
        backend = self.main.backend
        self.views = [self._make_view(p, backend, self.main.overlap) for p in view_plans]
        self.maps = [v.halo_map_from(self.plan) for v in self.views]

    # -- pieces ------------------------------------------------------------------------------------------------------------
    def assemble(self, table_local, node_ids):
        """[len(node_ids), d] rows of a node-sharded table for GLOBAL node ids (users: id, items: n_users + id)."""
        loc = self.local_of.index_select(0, node_ids)
        pos = torch.nonzero(loc >= 0).flatten()
        return _AssembleRows.apply(table_local, pos, loc.index_select(0, pos), int(node_ids.shape[0]), self._cgroup, self.world)

    def dist_lse(self, anchors, cand_local, scale, shift=0.0):
        return _DistLSE.apply(anchors, cand_local, float(scale), float(shift), self._cgroup, self.world, _lse_rows)

    def _emb_loss(self, *embs):  # recbole EmbLoss(norm=2), as models.EmbLoss
        total = embs[0].new_zeros(1)
        for e in embs:
            total = total + (torch.pow(torch.norm(e, p=2), 2) if self.require_pow else torch.norm(e, p=2))
        total = total / embs[-1].shape[0]
        return total / 2 if self.require_pow else total

    # -- the loss (value identical on every rank) ----------------------------------------------------------------------------
    def loss(self, user, pos_item, neg_item):
        """sgl.py:211-233 (with views) / lightgcn.py:83-110 (without), ids as in the reference's interaction batch."""
        nu = self.n_users
        dev = self.e0.device
        user, pos_item, neg_item = (t.to(dev, torch.int64) for t in (user, pos_item, neg_item))
        b = user.shape[0]
        ids3 = torch.cat([user, pos_item + nu, neg_item + nu])
        if self.views:
            m, v1, v2 = sh.sharded_sgl_forward(self.main, self.views, self.e0, self.n_layers, maps=self.maps)
        else:
            m = sh.sharded_lightgcn_forward(self.main, self.e0, self.n_layers)
        rows = self.assemble(m, ids3)
        ue, pe, ne = rows[:b], rows[b:2 * b], rows[2 * b:]
        ego = self.assemble(self.e0, ids3)
        if self.views:  # sgl.py:147-162: sum-reduced logsigmoid
            bpr = -F.logsigmoid((ue * pe).sum(1) - (ue * ne).sum(1)).sum()
        else:           # recbole BPRLoss: -mean log(gamma + sigmoid(pos - neg))
            bpr = -torch.log(1e-10 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean()
        reg = self._emb_loss(ego[:b], ego[b:2 * b], ego[2 * b:])
        total = bpr + self.reg_weight * reg
        if self.views:
            ids2 = ids3[: 2 * b]
            r1, r2 = self.assemble(v1, ids2), self.assemble(v2, ids2)
            tau = self.ssl_tau
            ssl = total.new_zeros(())
            for side, (lo, hi) in enumerate(((0, b), (b, 2 * b))):  # users, then items (sgl.py:176-209)
                a, p = F.normalize(r1[lo:hi], dim=1), F.normalize(r2[lo:hi], dim=1)
                cand = v2[: self.nu_local] if side == 0 else v2[self.nu_local:]
                lse = self.dist_lse(a, F.normalize(cand, dim=1), 1.0 / tau, 1.0 / tau)
                ssl = ssl + (lse - (a * p).sum(dim=1) / tau).sum()
            total = total + self.ssl_weight * ssl
        return total.reshape(())

    def step(self, user, pos_item, neg_item):
        """One optimizer step; returns the loss value (a python float, the same on every rank)."""
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(user, pos_item, neg_item)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
