"""Node-range sharding: the partition plan and the halo exchange, checked on CPU —
in-process for the plan algebra, and with two gloo processes for the N > 1 exchange path.  The compute
backend is injected (the oracle's CPU SpMM) because the product backend needs a GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port
from oracle import coracle as C


class CpuBackend:
    """Test double for sharded.HipBackend: same three calls, computed by the CPU oracle."""
    device = torch.device("cpu")

    def make_graph(self, csr, n_cols, n_user_rows=None):
        return (np.asarray(csr[0]), np.asarray(csr[1], dtype=np.int64), np.asarray(csr[2]), n_cols)

    def spmm(self, graph, x, out, accumulate):
        y = torch.from_numpy(C.spmm(graph[0], graph[1], graph[2], x.numpy()))
        if accumulate:
            out += y
        else:
            out.copy_(y)
        return out

    def gather_rows(self, src, idx):
        return src[idx]


def global_reference(uid, iid, nu, ni, e0, k_layers):
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    return C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], k_layers), (rowptr, col, val)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("layout", ["ranges", "striped", "degree-striped"])
def test_plan_algebra(rbg, ref_inter, world, layout):
    uid, iid, nu, ni = ref_inter
    n = nu + ni
    sh = rbg.sharded
    owner = {"ranges": None, "striped": sh.striped_partition(nu, ni, world),
             "degree-striped": sh.degree_striped_partition(uid, iid, nu, ni, world)}[layout]
    plans = sh.build_plans(uid, iid, nu, ni, world, owner=owner)
    x = np.random.default_rng(0).standard_normal((n, 8)).astype(np.float32)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    y_ref = C.spmm(rowptr, col, val, x)
    seen = np.zeros(n, dtype=int)
    for p, plan in plans.items():
        seen[plan.owned] += 1
        assert np.all(np.diff(plan.owned) > 0)
        assert plan.n_users_owned == np.count_nonzero(plan.owned < nu)
        # the rows of the two blocks together are exactly the global rows (bit-exact weights)
        for r_local in (0, plan.n_owned // 2, plan.n_owned - 1):
            g_row = plan.owned[r_local]
            ent = {}
            b, e = plan.int_csr[0][r_local], plan.int_csr[0][r_local + 1]
            for c, v in zip(plan.int_csr[1][b:e], plan.int_csr[2][b:e]):
                ent[int(plan.owned[c])] = v
            b, e = plan.halo_csr[0][r_local], plan.halo_csr[0][r_local + 1]
            for c, v in zip(plan.halo_csr[1][b:e], plan.halo_csr[2][b:e]):
                ent[int(plan.halo_ids[c])] = v
            gb, ge = rowptr[g_row], rowptr[g_row + 1]
            assert sorted(ent) == sorted(col[gb:ge].tolist())
            assert all(ent[int(c)] == v for c, v in zip(col[gb:ge], val[gb:ge]))
        # send lists mirror the peers' receive lists
        so = 0
        for q in range(world):
            sc = int(plan.send_counts[q])
            other = plans[q]
            ro = int(other.recv_counts[:p].sum())
            assert sc == int(other.recv_counts[p])
            assert np.array_equal(plan.owned[plan.send_idx[so:so + sc]], other.halo_ids[ro:ro + sc])
            so += sc
        assert int(plan.send_counts[p]) == 0 and int(plan.recv_counts[p]) == 0
        # Y[owned] = A_int X_local + A_halo X_halo
        y = C.spmm(plan.int_csr[0], plan.int_csr[1].astype(np.int64), plan.int_csr[2], x[plan.owned])
        if plan.n_halo:
            y = y + C.spmm(plan.halo_csr[0], plan.halo_csr[1].astype(np.int64), plan.halo_csr[2], x[plan.halo_ids])
        np.testing.assert_allclose(y, y_ref[plan.owned], atol=1e-6)
    assert np.all(seen == 1)  # a partition: every node owned exactly once
    if layout == "ranges" and world > 1:
        nnz = [p.int_csr[0][-1] + p.halo_csr[0][-1] for p in plans.values()]
        assert max(nnz) < 1.6 * (sum(nnz) / world)  # nnz-balanced
    if layout == "degree-striped" and world > 1:
        # rows within one of each other per side, nnz within a few per cent, and the degree-only statistics agree with the plans
        rows = [p.n_owned for p in plans.values()]
        nnz = [int(p.int_csr[0][-1] + p.halo_csr[0][-1]) for p in plans.values()]
        assert max(rows) - min(rows) <= 2 and max(nnz) < 1.15 * (sum(nnz) / world)
        st = sh.partition_stats(uid, iid, nu, ni, owner, world)
        assert st["rows"] == rows and st["nnz"] == nnz


def test_degree_striped_partition_balances_the_halo(rbg):
    """contiguous nnz-balanced ranges of popularity-sorted ids give rank 0 a few very heavy rows and a
    halo of almost the whole other side; dealing the degree order round-robin balances rows, nnz AND halo rows."""
    sh = rbg.sharded
    nu, ni, e = 2001, 3001, 60_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=5)
    for world in (2, 4):
        ranges = sh.build_plans(uid, iid, nu, ni, world)
        owner, name, stats = sh.choose_partition(uid, iid, nu, ni, world)
        assert name == "striped" and set(stats) == {"ranges", "striped"}
        striped = sh.build_plans(uid, iid, nu, ni, world, owner=owner)
        ratio = lambda plans: max(p.n_halo / max(p.n_owned, 1) for p in plans.values())  # noqa: E731
        assert ratio(striped) < 0.5 * ratio(ranges)
        assert max(p.n_owned for p in striped.values()) < max(p.n_owned for p in ranges.values())


def test_block_structure_trims_the_halo(rbg):
    sh = rbg.sharded
    world = 4
    nu, ni, e = 401, 801, 6000
    sizes = {}
    for p_in in (0.0, 0.9, 1.0):
        uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=3, n_blocks=world, p_in=p_in)
        plans = sh.build_plans(uid, iid, nu, ni, world, owner=sh.striped_partition(nu, ni, world))
        sizes[p_in] = sum(p.n_halo for p in plans.values())
    assert sizes[1.0] == 0 and sizes[0.9] < 0.5 * sizes[0.0]


def _worker(rank, world, port, uid, iid, nu, ni, k_layers, layout, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        sh = rbg.sharded
        owner = None if layout == "ranges" else sh.striped_partition(nu, ni, world)
        plan = sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank])[rank]
        e0 = np.random.default_rng(1).standard_normal((nu + ni, 16)).astype(np.float32)  # same on every rank
        prop = sh.ShardedPropagation(plan, CpuBackend(), transport="staged")
        mean_local = prop.forward(torch.from_numpy(e0[plan.owned]), k_layers)
        ref, _ = global_reference(uid, iid, nu, ni, e0, k_layers)
        err = float(np.abs(mean_local.numpy() - ref[plan.owned]).max())
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, err, plan.n_halo, int(plan.send_counts.sum())))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,world", [("ranges", 2), ("striped", 2), ("striped", 3)])
def test_two_process_gloo_exchange(ref_inter, layout, world):
    """(world 3: more than one peer per rank — send / receive lists grouped by owner, uneven splits)"""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, uid, iid, nu, ni, 3, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == world
    for rank, err, n_halo, n_send in res:
        assert err <= 1e-5, (rank, err)
        assert n_halo > 0 and n_send > 0  # the exchange path really ran


# ---- round 2: plan_from_csr, edge-drop views, backward, sharded scoring ------------------------------------------------------

def _plans_equal(a, b):
    assert a.rank == b.rank and a.world == b.world and a.n_users_owned == b.n_users_owned
    assert np.array_equal(a.owned, b.owned) and np.array_equal(a.halo_ids, b.halo_ids)
    assert np.array_equal(a.recv_counts, b.recv_counts) and np.array_equal(a.send_counts, b.send_counts)
    assert np.array_equal(a.send_idx, b.send_idx)
    for x, y in ((a.int_csr, b.int_csr), (a.halo_csr, b.halo_csr)):
        assert np.array_equal(np.asarray(x[0]), np.asarray(y[0])) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("layout", ["ranges", "striped"])
def test_plan_from_csr_equals_numpy_planner(rbg, ref_inter, world, layout):
    """The planner that cuts a rank's blocks out of the built global CSR (torch ops; on a GPU box they run in HBM) gives
    exactly the numpy planner's plan: same blocks bit for bit, same halo / send lists."""
    uid, iid, nu, ni = ref_inter
    sh = rbg.sharded
    owner = sh.default_partition(uid, iid, nu, ni, world) if layout == "ranges" else sh.striped_partition(nu, ni, world)
    ref = sh.build_plans(uid, iid, nu, ni, world, owner=owner)
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=None)  # host builder; on the GPU box: device_csr()
    rowptr, col, val = (torch.from_numpy(a) for a in g.export_csr())
    for r in range(world):
        _plans_equal(sh.plan_from_csr(rowptr, col, val, nu, owner, r, world), ref[r])


def test_edge_drop_view_plans_match_the_single_device_view(rbg, ref_inter):
    """build_plans(keep=mask): an SGL ED view (sgl.py:107-126) sharded — the blocks of all ranks together are the masked
    single-device graph bit for bit (weights from the view's own GLOBAL degrees, sgl.py:119-124)."""
    uid, iid, nu, ni = ref_inter
    sh = rbg.sharded
    keep = np.zeros(len(uid), dtype=np.uint8)
    keep[np.random.default_rng(5).permutation(len(uid))[: int(len(uid) * 0.9)]] = 1
    world = 2
    owner = sh.default_partition(uid, iid, nu, ni, world)  # the FULL graph's partition: the views share the embedding shards
    plans = sh.build_plans(uid, iid, nu, ni, world, owner=owner, keep=keep)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni, keep=keep)
    for p, plan in plans.items():
        assert np.array_equal(plan.owned, np.flatnonzero(owner == p))
        for r_local in range(0, plan.n_owned, 7):
            g_row = plan.owned[r_local]
            ent = {}
            for csr, ids in ((plan.int_csr, plan.owned), (plan.halo_csr, plan.halo_ids)):
                for c, v in zip(csr[1][csr[0][r_local]:csr[0][r_local + 1]], csr[2][csr[0][r_local]:csr[0][r_local + 1]]):
                    ent[int(ids[c])] = v
            gb, ge = rowptr[g_row], rowptr[g_row + 1]
            assert sorted(ent) == col[gb:ge].tolist() and all(ent[int(c)] == v for c, v in zip(col[gb:ge], val[gb:ge]))


def _worker2(rank, world, port, uid, iid, nu, ni, k_layers, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        sh = rbg.sharded
        n, d = nu + ni, 16
        rng = np.random.default_rng(1)
        e0 = rng.standard_normal((n, d)).astype(np.float32)
        w = rng.standard_normal((n, d)).astype(np.float32)
        keep = np.zeros(len(uid), dtype=np.uint8)
        keep[np.random.default_rng(5).permutation(len(uid))[: int(len(uid) * 0.9)]] = 1
        owner = sh.default_partition(uid, iid, nu, ni, world)
        out = {}
        for name, mask in (("full", None), ("view", keep)):
            plan = sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank], keep=mask)[rank]
            prop = sh.ShardedPropagation(plan, CpuBackend(), transport="staged")
            x = torch.from_numpy(e0[plan.owned]).requires_grad_(True)
            mean = sh.sharded_lightgcn_forward(prop, x, k_layers)
            (mean * torch.from_numpy(w[plan.owned])).sum().backward()
            rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni, keep=mask)
            ref = C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], k_layers)
            # the propagation operator M = (I + A + ... + A^K) / (K + 1) is symmetric: d<w, M e0>/d e0 = M w
            gref = C.lightgcn_forward(rowptr, col, val, w[:nu], w[nu:], k_layers)
            out[name] = (float(np.abs(mean.detach().numpy() - ref[plan.owned]).max()),
                         float(np.abs(x.grad.numpy() - gref[plan.owned]).max()))
            if name == "full":  # NGCF forward over the shard (ngcf.py:92-104): sharded product + local dense half per layer
                from oracle import oracle as O
                g2 = torch.Generator().manual_seed(3)
                params = [(torch.randn(16, 16, generator=g2) * 0.3, torch.randn(16, generator=g2) * 0.1,
                           torch.randn(16, 16, generator=g2) * 0.3, torch.randn(16, generator=g2) * 0.1) for _ in range(2)]
                conv = lambda t: torch.from_numpy(C.spmm(rowptr, col, val, t.numpy()))  # noqa: E731
                u_ref, i_ref = O.ngcf_forward(torch.from_numpy(e0[:nu]), torch.from_numpy(e0[nu:]), conv, params)
                got = prop.ngcf_forward(torch.from_numpy(e0[plan.owned]), params)
                out["ngcf"] = float((got - torch.cat([u_ref, i_ref])[plan.owned]).abs().max())
            if name == "full":  # sharded full-sort scoring: local users against the all-gathered item table
                m = mean.detach()
                table = prop.gather_item_table(m, nu, ni)
                users = torch.arange(min(5, plan.n_users_owned))
                s = prop.full_sort_scores(m, users, nu, ni, item_table=table)
                s_ref = ref[plan.owned[users.numpy()]] @ ref[nu:].T
                out["score"] = (float(np.abs(table.numpy() - ref[nu:]).max()), float(np.abs(s.numpy() - s_ref).max()))
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, out))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


def _worker3(rank, world, port, uid, iid, nu, ni, k_layers, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        sh = rbg.sharded
        n, d = nu + ni, 16
        rng = np.random.default_rng(2)
        e0 = rng.standard_normal((n, d)).astype(np.float32)
        ws = [rng.standard_normal((n, d)).astype(np.float32) for _ in range(3)]
        masks = []
        for seed in (5, 6):
            keep = np.zeros(len(uid), dtype=np.uint8)
            keep[np.random.default_rng(seed).permutation(len(uid))[: int(len(uid) * 0.9)]] = 1
            masks.append(keep)
        owner = sh.default_partition(uid, iid, nu, ni, world)
        plans = [sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank], keep=m)[rank] for m in (None, *masks)]
        props = [sh.ShardedPropagation(pl, CpuBackend(), transport="staged") for pl in plans]
        calls = {"n": 0}
        for pr in props:  # count the exchanges (each is one collective of the real transport)
            orig = pr._exchange_staged
            def counted(x, halo, _o=orig):
                calls["n"] += 1
                return _o(x, halo)
            pr._exchange_staged = counted
        owned = plans[0].owned
        x = torch.from_numpy(e0[owned]).requires_grad_(True)
        outs = sh.sharded_sgl_forward(props[0], props[1:], x, k_layers)
        n_shared = calls["n"]
        loss = sum((o * torch.from_numpy(w[owned])).sum() for o, w in zip(outs, ws))
        loss.backward()
        # the same three propagations one by one (3 K exchanges), and the oracle on the global graphs
        calls["n"] = 0
        plain = [pr.forward(torch.from_numpy(e0[owned]), k_layers).clone() for pr in props]
        n_plain = calls["n"]
        errs, gref = [], np.zeros((n, d), dtype=np.float32)
        for o, pl_out, m, w in zip(outs, plain, (None, *masks), ws):
            rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni, keep=m)
            ref = C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], k_layers)
            errs.append((bool(torch.equal(o.detach(), pl_out)), float(np.abs(o.detach().numpy() - ref[owned]).max())))
            gref += C.lightgcn_forward(rowptr, col, val, w[:nu], w[nu:], k_layers)  # M is symmetric: d<w, M e0>/d e0 = M w
        gerr = float(np.abs(x.grad.numpy() - gref[owned]).max())
        subset = all(set(pl.halo_ids.tolist()) <= set(plans[0].halo_ids.tolist()) for pl in plans[1:])
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, dict(n_shared=n_shared, n_plain=n_plain, errs=errs, gerr=gerr, subset=subset)))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_sgl_forward_shares_the_first_exchange(ref_inter):
    """sharded_sgl_forward: SGL's three propagations of one E0 (sgl.py:129, :219-221) over world size 2 — the views take
    their first-layer halo from the full graph's exchange (3 K - 2 exchanges instead of 3 K), results are
    bit-identical to three separate propagations, match the oracle on the global graphs, and the summed gradient is right."""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    k_layers = 3
    procs = [ctx.Process(target=_worker3, args=(r, 2, port, uid, iid, nu, ni, k_layers, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert out["subset"], rank
        assert out["n_plain"] == 3 * k_layers and out["n_shared"] == 3 * k_layers - 2, (rank, out["n_shared"], out["n_plain"])
        for same, err in out["errs"]:
            assert same and err <= 1e-5, (rank, out["errs"])
        assert out["gerr"] <= 3e-5, (rank, out["gerr"])


def test_two_process_gloo_backward_views_and_scoring(ref_inter):
    """World size 2 over gloo: sharded gradients (the backward is the same sharded product, by symmetry of the global
    matrix), an edge-drop view propagated over its own plan, and full-sort scoring over the all-gathered item table."""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, uid, iid, nu, ni, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for name in ("full", "view"):
            assert out[name][0] <= 1e-5 and out[name][1] <= 1e-5, (rank, name, out[name])
        assert out["score"][0] <= 1e-5 and out["score"][1] <= 1e-4, (rank, out["score"])
        assert out["ngcf"] <= 1e-5, (rank, out["ngcf"])


# ---- round 3: the sharded TRAINING step (sharded_train.py) ---------------------------------------------------------------------

def _reference_step(uid, iid, nu, ni, e0, masks, batch, k_layers, tau, ssl_w, reg_w, sgl):
    """The single-device loss and dL/dE0 by plain torch autograd (float64 matrices): sgl.py:211-233 / lightgcn.py:83-110."""
    from oracle import oracle as O
    import scipy.sparse as sp
    n = nu + ni

    def dense_adj(keep):
        rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni, keep=keep)
        return torch.from_numpy(sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(n, n)).toarray())

    x = torch.from_numpy(e0.astype(np.float64)).requires_grad_(True)

    def prop(a):
        layers, cur = [x], x
        for _ in range(k_layers):
            cur = a @ cur
            layers.append(cur)
        return torch.stack(layers, dim=1).mean(dim=1)

    user, pos, neg = (torch.from_numpy(np.asarray(t, dtype=np.int64)) for t in batch)
    m = prop(dense_adj(None))
    ue, pe, ne = m[user], m[nu + pos], m[nu + neg]
    ego = (x[user], x[nu + pos], x[nu + neg])
    reg = sum(torch.norm(e, p=2) for e in ego) / len(neg)
    if sgl:
        v1, v2 = prop(dense_adj(masks[0])), prop(dense_adj(masks[1]))
        bpr = -torch.nn.functional.logsigmoid((ue * pe).sum(1) - (ue * ne).sum(1)).sum()
        ssl = O.calc_ssl_loss(user, pos, v1[:nu], v2[:nu], v1[nu:], v2[nu:], tau, ssl_w)
        loss = bpr + reg_w * reg + ssl
    else:
        loss = -torch.log(1e-10 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean() + reg_w * reg
    loss.backward()
    return float(loss), x.grad.numpy()


def _worker_train(rank, world, port, uid, iid, nu, ni, k_layers, layout, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        from recbole_gnn_amd import sharded_train as st
        sh = rbg.sharded
        n, d = nu + ni, 16
        rng = np.random.default_rng(3)
        e0 = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
        masks = []
        for seed in (5, 6):
            keep = np.zeros(len(uid), dtype=np.uint8)
            keep[np.random.default_rng(seed).permutation(len(uid))[: int(len(uid) * 0.9)]] = 1
            masks.append(keep)
        b = 64
        batch = (rng.integers(1, nu, b), rng.integers(1, ni, b), rng.integers(1, ni, b))
        batch[0][:4] = batch[0][4:8]  # repeated users and items in one batch (their gradients add up)
        batch[1][:4] = batch[1][4:8]
        owner = sh.degree_striped_partition(uid, iid, nu, ni, world) if layout == "striped" else sh.default_partition(uid, iid, nu, ni, world)
        plans = [sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank], keep=m)[rank] for m in (None, *masks)]
        out = {}
        for name, views in (("sgl", plans[1:]), ("lightgcn", None)):
            tr = st.ShardedTrainer(plans[0], CpuBackend(), torch.from_numpy(e0[plans[0].owned]), nu, ni, k_layers, view_plans=views,
                                   transport="staged", lr=1e-2, reg_weight=1e-3, ssl_tau=0.5, ssl_weight=0.05)
            tb = tuple(torch.from_numpy(t) for t in batch)
            loss = tr.loss(*tb)
            loss.backward()
            ref_loss, ref_grad = _reference_step(uid, iid, nu, ni, e0, masks, batch, k_layers, 0.5, 0.05, 1e-3, name == "sgl")
            gerr = float(np.abs(tr.e0.grad.numpy() - ref_grad[plans[0].owned]).max())
            scale = float(np.abs(ref_grad).max())
            # one optimizer step moves the owned rows exactly as a dense Adam on the full table would
            full = torch.from_numpy(e0.copy()).requires_grad_(True)
            opt = torch.optim.Adam([full], lr=1e-2)
            full.grad = torch.from_numpy(ref_grad.astype(np.float32))
            opt.step()
            tr.e0.grad = None
            v = tr.step(*tb)
            aerr = float(np.abs(tr.e0.detach().numpy() - full.detach().numpy()[plans[0].owned]).max())
            out[name] = (float(loss.detach()), ref_loss, gerr, scale, aerr, v)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, out))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,world", [("ranges", 2), ("striped", 2), ("striped", 4)])
def test_two_process_gloo_training_step(ref_inter, layout, world):
    """SGL.calculate_loss (sgl.py:211-233: three propagations, BPR, reg, InfoNCE over ALL users / items)
    and LightGCN.calculate_loss (lightgcn.py:83-110) over two (and four) node shards — replicated batch rows by all-reduce,
    distributed logsumexp, sharded Adam — match the single-device value and dL/dE0 to 1e-5."""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_train, args=(r, world, port, uid, iid, nu, ni, 2, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    values = {}
    for rank, out in res:
        for name, (loss, ref_loss, gerr, scale, aerr, v) in out.items():
            assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (rank, name, loss, ref_loss)
            assert gerr <= 1e-5 * max(1.0, scale), (rank, name, gerr, scale)
            assert aerr <= 2e-5, (rank, name, aerr)  # one Adam step of lr 1e-2: rows move by ~1e-2
            values.setdefault(name, []).append(v)
    for name, vs in values.items():  # every rank evaluates the same scalar
        assert max(vs) - min(vs) <= 1e-6 * max(1.0, abs(vs[0])), (name, vs)


def _worker_rw(rank, world, port, uid, iid, nu, ni, k_layers, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recbole_gnn_amd as rbg
        sh = rbg.sharded
        n, d = nu + ni, 16
        rng = np.random.default_rng(4)
        e0 = rng.standard_normal((n, d)).astype(np.float32)
        w = rng.standard_normal((n, d)).astype(np.float32)
        masks = []
        for seed in range(k_layers):  # RW: an independent edge sample per layer (sgl.py:89-91)
            keep = np.zeros(len(uid), dtype=np.uint8)
            keep[np.random.default_rng(20 + seed).permutation(len(uid))[: int(len(uid) * 0.9)]] = 1
            masks.append(keep)
        owner = sh.degree_striped_partition(uid, iid, nu, ni, world)
        full = sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank])[rank]
        layer_plans = [sh.build_plans(uid, iid, nu, ni, world, owner=owner, ranks=[rank], keep=m)[rank] for m in masks]
        main = sh.ShardedPropagation(full, CpuBackend(), transport="staged")
        view = sh.LayeredShardedPropagation(layer_plans, CpuBackend(), transport="staged")
        x = torch.from_numpy(e0[full.owned]).requires_grad_(True)
        m_full, m_rw = sh.sharded_sgl_forward(main, [view], x, k_layers)
        (m_rw * torch.from_numpy(w[full.owned])).sum().backward()
        csrs = [C.build_norm_csr(uid, iid, nu, ni, keep=m) for m in masks]
        cur, acc = e0.copy(), e0.copy()
        for rp, cc, vv in csrs:  # sgl.py:137-139: layer k on ITS graph
            cur = C.spmm(rp, cc, vv, cur)
            acc += cur
        ref = acc / (k_layers + 1)
        t = w.copy()
        for rp, cc, vv in reversed(csrs):  # (w + A_1 (w + A_2 (... (w + A_K w)))) / (K + 1)
            t = w + C.spmm(rp, cc, vv, t)
        gref = t / (k_layers + 1)
        rp, cc, vv = C.build_norm_csr(uid, iid, nu, ni)
        ref_full = C.lightgcn_forward(rp, cc, vv, e0[:nu], e0[nu:], k_layers)
        res = (float(np.abs(m_rw.detach().numpy() - ref[full.owned]).max()), float(np.abs(x.grad.numpy() - gref[full.owned]).max()),
               float(np.abs(m_full.detach().numpy() - ref_full[full.owned]).max()))
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, res))
        if rank == 0:
            out_q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_rw_views_have_one_plan_per_layer(ref_inter):
    """sgl.py:89-91 / :137-139 ("RW": one sub-graph per layer) over two shards: LayeredShardedPropagation inside
    sharded_sgl_forward (E0's halo shared with the full graph's propagation), forward and gradient against the oracle."""
    uid, iid, nu, ni = ref_inter
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_rw, args=(r, 2, port, uid, iid, nu, ni, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, (err, gerr, ferr) in res:
        assert err <= 1e-5 and gerr <= 1e-5 and ferr <= 1e-5, (rank, err, gerr, ferr)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_fused_block_equals_the_two_handle_form(rbg, ref_inter, world):
    """r06: ShardPlan.cat_csr() is [A_interior | A_halo] over the table [owned rows | halo rows]; the fused layer (one product)
    and the two-handle layer (interior + halo accumulate) agree row by row, and both equal the rows of the global product."""
    uid, iid, nu, ni = ref_inter
    sh = rbg.sharded
    plans = sh.build_plans(uid, iid, nu, ni, world, owner=sh.degree_striped_partition(uid, iid, nu, ni, world))
    x = np.random.default_rng(1).standard_normal((nu + ni, 16)).astype(np.float32)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    y_ref = C.spmm(rowptr, col, val, x)
    be = CpuBackend()
    for p, plan in plans.items():
        rp, cc, vv = plan.cat_csr()
        ip, ic, iv = plan.int_csr
        hp, hc, hv = plan.halo_csr
        assert rp[-1] == ip[-1] + hp[-1] and np.all(np.diff(rp) == np.diff(ip) + np.diff(hp))
        for r in (0, plan.n_owned // 2, plan.n_owned - 1):   # a row = its interior entries, then its halo entries moved past the owned rows
            a, b = rp[r], rp[r + 1]
            ni_r = ip[r + 1] - ip[r]
            assert np.array_equal(cc[a:a + ni_r], ic[ip[r]:ip[r + 1]]) and np.array_equal(cc[a + ni_r:b], np.asarray(hc[hp[r]:hp[r + 1]]) + plan.n_owned)
        halo = torch.from_numpy(x[plan.halo_ids]) if plan.n_halo else torch.zeros((1, 16))
        xo = torch.from_numpy(x[plan.owned])
        fused = sh.ShardedPropagation(plan, be, transport="staged", fused=True)
        pair = sh.ShardedPropagation(plan, be, transport="staged", fused=False)
        assert fused.fused == (plan.n_halo > 0) and not pair.fused
        y1 = fused.spmm(xo, halo_rows=halo).numpy()
        y2 = pair.spmm(xo, halo_rows=halo).numpy()
        assert np.abs(y1 - y_ref[plan.owned]).max() <= 1e-5 and np.abs(y2 - y_ref[plan.owned]).max() <= 1e-5
        # column windows (a table beyond the rectangular plan's 32-bit offsets: config #5's shards): the same rows from 3+ launches
        wins = plan.cat_windows(max((plan.n_owned + plan.n_halo) // 3, 1))
        assert len(wins) >= 3 and wins[0][0] == 0 and wins[-1][1] == plan.n_owned + plan.n_halo
        assert sum(len(w[2][1]) for w in wins) == rp[-1] and all(a[1] == b_[0] for a, b_ in zip(wins, wins[1:]))
        windowed = sh.ShardedPropagation(plan, be, transport="staged", fused=True, cat_window_rows=max((plan.n_owned + plan.n_halo) // 3, 1))
        assert windowed.fused == (plan.n_halo > 0) and (len(windowed.g_cats) >= 3 or plan.n_halo == 0)
        y3 = windowed.spmm(xo, halo_rows=halo).numpy()
        assert np.abs(y3 - y_ref[plan.owned]).max() <= 1e-5
