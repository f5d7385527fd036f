"""Columns x node-ranges: the hybrid sharding of the propagation over the GPUs of one node (r05).

Feature-column sharding (``colsharded.py``) exchanges nothing in the K layers but stops at ``d / 32`` ranks (the column-slab kernel
works on 32-column slabs: d = 64 is two ranks); node-range sharding (``sharded.py``) reaches any rank count but ships almost the
whole table per layer on an unstructured graph.  The two compose because ``Y = Â·X`` acts on every column independently: the world
of ``P = C x R`` ranks is C column groups of R node shards; rank ``(c, r)`` holds the rows of node shard r and the ``d / C`` columns
of column group c — of E_0, of every layer, of the mean — and exchanges halos only inside its column group, rows of ``d / C``
floats: at d = 64, C = 2 every halo is half as wide as in the pure node-range mode, the interior product is the
``sell_spmm_kernel<32, 1, ·>`` launch, and 8 GPUs are 2 x 4 instead of 1 x 8 (a quarter of the table per rank instead of an
eighth: fewer, larger shards, fewer peers per exchange).

What needs the other column groups is what ``colsharded.py`` lists (the loss: all-reduced partial dot products over the column
group's peers with the same node shard) — ``gather_columns`` / ``reduce_over_columns`` here; the K layers use one process
subgroup per column group.  UNMEASURED on more than one GPU (no multi-GPU box in five rounds); tested with gloo at 2 x 2 and
2 x 4 ranks against a single-device run of the same graph, forward and backward."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import sharded as sh
from .colsharded import column_range


def grid_of(rank, world, col_shards):
    """(column group c, node shard r, node shards R) of `rank`: column-group major, so a column group is a contiguous rank range
    (on one node: neighbouring GPUs exchange the halos)."""
    if col_shards < 1 or world % col_shards:
        raise ValueError(f"{world} ranks do not split into {col_shards} column groups")
    r_count = world // col_shards
    return rank // r_count, rank % r_count, r_count


class HybridShardedPropagation:
    """One rank's share of a C x R grid.  ``backend``: ``sharded.HipBackend`` (the product) or a test double with the same calls.
    Every rank of the default group must construct it (the subgroups are created collectively)."""

    def __init__(self, uid, iid, n_users, n_items, d, backend, col_shards, rank=None, world=None, transport="nccl", owner=None, overlap=None,
                 keep=None):
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        self.rank, self.world, self.d = int(rank), int(world), int(d)
        self.c, self.r, self.R = grid_of(self.rank, self.world, int(col_shards))
        self.C = int(col_shards)
        self.n_users, self.n_items = int(n_users), int(n_items)
        # every rank creates every subgroup, in the same order (torch.distributed's rule); the host-staged transport exchanges
        # through gloo whatever the default group's backend is
        gb = "gloo" if transport == "staged" else None
        node_groups = [dist.new_group(ranks=list(range(c * self.R, (c + 1) * self.R)), backend=gb) for c in range(self.C)]
        col_groups = [dist.new_group(ranks=[c * self.R + r for c in range(self.C)], backend=gb) for r in range(self.R)]
        self.node_group, self.col_group = node_groups[self.c], col_groups[self.r]
        if owner is None:
            owner = sh.degree_striped_partition(uid, iid, n_users, n_items, self.R) if self.R > 1 else None
        self.plan = sh.build_plans(uid, iid, n_users, n_items, self.R, owner=owner, ranks=[self.r], keep=keep)[self.r]
        self.prop = sh.ShardedPropagation(self.plan, backend, group=self.node_group, transport=transport, overlap=overlap)
        self.lo, self.hi = column_range(d, self.c, self.C)
        self.width = self.hi - self.lo

    @property
    def owned(self):
        return self.plan.owned

    def slab_of(self, table):
        """This rank's block of a full [N, d] table: its node shard's rows, its column group's columns (a contiguous copy)."""
        rows = torch.as_tensor(self.plan.owned, dtype=torch.int64, device=table.device)
        return table.index_select(0, rows)[:, self.lo:self.hi].contiguous()

    def forward(self, e0_block, n_layers):
        """mean(E_0 .. E_K) of the block (lightgcn.py:70-81): K halo exchanges inside the column group, rows of d / C floats."""
        return self.prop.forward(e0_block, n_layers)

    def propagate(self, e0_block, n_layers):
        """autograd-aware ``forward`` (the backward is the same chain: the operator is symmetric)"""
        return sh.sharded_lightgcn_forward(self.prop, e0_block, n_layers)

    def halo_bytes_per_layer(self):
        """what this rank receives per layer, beside the pure node-range mode's figure for the same rank count"""
        out = self.prop.halo_bytes_per_layer(self.width)
        out.update(columns=self.width, grid=f"{self.C} column groups x {self.R} node shards",
                   recv_bytes_if_full_width=out["recv_bytes"] * self.C)
        return out

    # -- what the loss / the evaluation need from the other column groups ----------------------------------------------------
    def gather_columns(self, block):
        """[rows, d] from the C column groups' [rows, d / C] blocks of the same node shard."""
        if self.C == 1:
            return block
        staged = block.device.type == "cuda" and dist.get_backend(self.col_group) != "nccl"
        src = (block.cpu() if staged else block).contiguous()
        parts = [torch.empty_like(src) for _ in range(self.C)]
        dist.all_gather(parts, src, group=self.col_group)
        return torch.cat(parts, dim=1).to(block.device)

    def reduce_over_columns(self, partial):
        """sum over the column groups of a partial result (row dots, sums of squares), replicated in the column group"""
        if self.C == 1:
            return partial
        staged = partial.device.type == "cuda" and dist.get_backend(self.col_group) != "nccl"
        t = partial.cpu() if staged else partial.clone()
        dist.all_reduce(t, group=self.col_group)
        return t.to(partial.device)
