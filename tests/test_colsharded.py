"""Feature-column sharding (recbole-gnn_amd/colsharded.py): every rank holds the whole
normalized adjacency and d / P columns of everything dense — the K layers exchange nothing, the BPR step all-reduces 2 B score
partials and three scalars.  Checked on CPU with gloo (world sizes 2 and 4; compute backend = the oracle's CPU SpMM, because
the product backend needs a GPU) against a single-process torch-autograd restatement of lightgcn.py:70-110."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port
from oracle import coracle as C
from test_sharded import CpuBackend


def reference_step(uid, iid, nu, ni, e0, batch, k_layers, reg_w, require_pow, lr):
    """lightgcn.py:70-110 on one device, float32, torch autograd over a dense copy of the normalized adjacency."""
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    n = nu + ni
    a = torch.zeros(n, n)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    a[torch.from_numpy(rows), torch.from_numpy(col.astype(np.int64))] = torch.from_numpy(val)
    x = torch.from_numpy(e0.copy()).requires_grad_(True)
    acc, cur = x, x
    for _ in range(k_layers):
        cur = a @ cur
        acc = acc + cur
    out = acc / (k_layers + 1)
    u, p, ng = (torch.from_numpy(np.asarray(t)) for t in batch)
    ue, pe, ne = out[u], out[nu + p], out[nu + ng]
    bpr = -torch.log(1e-10 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean()
    embs = (x[u], x[nu + p], x[nu + ng])
    reg = sum(torch.pow(torch.norm(e, p=2), 2) if require_pow else torch.norm(e, p=2) for e in embs) / len(u)
    if require_pow:
        reg = reg / 2
    loss = bpr + reg_w * reg
    loss.backward()
    grad = x.grad.clone()
    opt = torch.optim.Adam([x], lr=lr)
    opt.step()
    scores = (out[u[:5]] @ out[nu:].T).detach().numpy()
    return float(loss.detach()), out.detach().numpy(), grad.numpy(), x.detach().numpy(), scores


def _worker(rank, world, port, uid, iid, nu, ni, d, k_layers, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        cs = rbg.colsharded
        n = nu + ni
        rng = np.random.default_rng(4)
        e0 = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
        batch = (rng.integers(1, nu, 64), rng.integers(1, ni, 64), rng.integers(1, ni, 64))
        be = CpuBackend()
        graph = be.make_graph(C.build_norm_csr(uid, iid, nu, ni), n, nu)
        prop = cs.ColumnShardedPropagation(graph, nu, ni, d, be, rank=rank, world=world, group=dist.group.WORLD)
        res = {}
        for require_pow in (False, True):
            ref_loss, ref_out, ref_grad, ref_new, ref_scores = reference_step(uid, iid, nu, ni, e0, batch, k_layers, 1e-3, require_pow, 1e-2)
            tr = cs.ColumnShardedTrainer(prop, prop.slab_of(torch.from_numpy(e0)), k_layers, lr=1e-2, reg_weight=1e-3, require_pow=require_pow)
            tb = tuple(torch.from_numpy(np.asarray(t)) for t in batch)
            out = prop.forward(tr.e0.detach(), k_layers)
            loss = tr.loss(*tb)
            loss.backward()
            gerr = float(np.abs(tr.e0.grad.numpy() - ref_grad[:, prop.lo:prop.hi]).max())
            oerr = float(np.abs(out.numpy() - ref_out[:, prop.lo:prop.hi]).max())
            full = prop.gather_columns(out)
            ferr = float(np.abs(full.numpy() - ref_out).max())
            serr = float(np.abs(prop.full_sort_scores(out, tb[0][:5]).numpy() - ref_scores).max())
            tr.e0.grad = None
            v = tr.step(*tb)
            aerr = float(np.abs(tr.e0.detach().numpy() - ref_new[:, prop.lo:prop.hi]).max())
            res[require_pow] = (float(loss.detach()), ref_loss, gerr, float(np.abs(ref_grad).max()), oerr, ferr, serr, aerr, v)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, res))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,d", [(2, 16), (4, 16), (2, 64)])
def test_gloo_column_sharded_training_step(ref_inter, world, d):
    """Forward (zero communication), loss, dL/dE0, one Adam step, the gathered full-width mean and the all-reduced score block
    of P column shards equal the single-device restatement (1e-5)."""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, uid, iid, nu, ni, d, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = {}
    for rank, out in res:
        for rp, (loss, ref_loss, gerr, gscale, oerr, ferr, serr, aerr, v) in out.items():
            assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (rank, rp, loss, ref_loss)
            assert gerr <= 1e-5 * max(1.0, gscale) and oerr <= 1e-5 and ferr <= 1e-5 and serr <= 1e-5, (rank, rp, gerr, oerr, ferr, serr)
            assert aerr <= 2e-5, (rank, rp, aerr)
            vals.setdefault(rp, []).append(v)
    for rp, vs in vals.items():  # every rank evaluates the same scalar
        assert max(vs) - min(vs) <= 1e-6 * max(1.0, abs(vs[0])), (rp, vs)


def test_column_range_and_single_rank(rbg, ref_inter):
    cs = rbg.colsharded
    assert cs.column_range(128, 3, 4) == (96, 128) and cs.column_range(64, 0, 1) == (0, 64)
    with pytest.raises(ValueError):
        cs.column_range(64, 0, 3)
    uid, iid, nu, ni = ref_inter
    be = CpuBackend()
    n = nu + ni
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    prop = cs.ColumnShardedPropagation(be.make_graph((rowptr, col, val), n, nu), nu, ni, 8, be)
    e0 = torch.from_numpy(np.random.default_rng(0).standard_normal((n, 8)).astype(np.float32))
    ref = C.lightgcn_forward(rowptr, col, val, e0[:nu].numpy(), e0[nu:].numpy(), 2)
    assert np.abs(prop.forward(e0, 2).numpy() - ref).max() <= 1e-6
