"""Columns x node-ranges (recbole-gnn_amd/hybrid.py): C column groups of R node shards, halos exchanged inside a
column group as rows of d / C floats.  CPU, gloo: grids 2 x 2 and 2 x 4 (and the degenerate 1 x 2, 2 x 1) against the single-device
oracle — forward, backward (the same chain: the operator is symmetric), the halo-byte count against the pure node-range mode's,
and the column gathers / reductions the loss uses."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port
from oracle import coracle as C
from test_sharded import CpuBackend


def _worker(rank, world, col_shards, port, uid, iid, nu, ni, k_layers, d, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        hy = rbg.hybrid
        n = nu + ni
        rng = np.random.default_rng(1)
        e0 = rng.standard_normal((n, d)).astype(np.float32)
        w = rng.standard_normal((n, d)).astype(np.float32)
        h = hy.HybridShardedPropagation(uid, iid, nu, ni, d, CpuBackend(), col_shards, transport="staged")
        rp, cc, vv = C.build_norm_csr(uid, iid, nu, ni)
        ref = C.lightgcn_forward(rp, cc, vv, e0[:nu], e0[nu:], k_layers)
        gref = C.lightgcn_forward(rp, cc, vv, w[:nu], w[nu:], k_layers)  # d<w, M e0>/d e0 = M w
        x = h.slab_of(torch.from_numpy(e0)).requires_grad_(True)
        assert tuple(x.shape) == (h.plan.n_owned, d // col_shards)
        out = h.propagate(x, k_layers)
        (out * h.slab_of(torch.from_numpy(w))).sum().backward()
        err = float(np.abs(out.detach().numpy() - ref[h.owned][:, h.lo:h.hi]).max())
        gerr = float(np.abs(x.grad.numpy() - gref[h.owned][:, h.lo:h.hi]).max())
        full = h.gather_columns(out.detach())                       # the node shard's rows at full width
        ferr = float(np.abs(full.numpy() - ref[h.owned]).max())
        dots = h.reduce_over_columns((out.detach() * out.detach()).sum(1))  # row dots over all d columns
        derr = float(np.abs(dots.numpy() - (ref[h.owned] ** 2).sum(1)).max() / max(1.0, float((ref ** 2).sum(1).max())))
        hb = h.halo_bytes_per_layer()
        # the pure node-range mode over the same R shards would receive full-width rows
        assert hb["recv_bytes"] * col_shards == hb["recv_bytes_if_full_width"] and hb["columns"] == d // col_shards
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, h.c, h.r, err, gerr, ferr, derr, hb["halo_rows"], hb["recv_bytes"]))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,col_shards", [(4, 2), (8, 2), (2, 1), (2, 2), (4, 4)])
def test_hybrid_grid_over_gloo(ref_inter, world, col_shards):
    uid, iid, nu, ni = ref_inter
    d, k_layers = 64, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, col_shards, port, uid, iid, nu, ni, k_layers, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0
    assert len(res) == world
    r_count = world // col_shards
    seen = set()
    for rank, c, r, err, gerr, ferr, derr, halo_rows, recv_bytes in res:
        assert (c, r) == (rank // r_count, rank % r_count)
        seen.add((c, r))
        assert err <= 1e-5 and gerr <= 1e-5 and ferr <= 1e-5 and derr <= 1e-5, (rank, err, gerr, ferr, derr)
        assert (halo_rows > 0) == (r_count > 1)
        assert recv_bytes == halo_rows * (d // col_shards) * 4
    assert len(seen) == world
    # the same node shard receives the same halo rows in every column group
    by_r = {}
    for rank, c, r, *_rest, halo_rows, _b in res:
        by_r.setdefault(r, set()).add(halo_rows)
    assert all(len(v) == 1 for v in by_r.values())


def test_grid_of(rbg):
    hy = rbg.hybrid
    assert [hy.grid_of(r, 8, 2) for r in range(8)] == [(0, 0, 4), (0, 1, 4), (0, 2, 4), (0, 3, 4), (1, 0, 4), (1, 1, 4), (1, 2, 4), (1, 3, 4)]
    with pytest.raises(ValueError):
        hy.grid_of(0, 6, 4)
