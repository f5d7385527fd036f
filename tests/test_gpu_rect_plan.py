"""r06: the RECTANGULAR form of the column-slab plan (csrc/sell_plan.hip, SellDev::rect) and the fused sharded layer built on it.

A rank's block [A_interior | A_halo] — rows = the nodes it owns, columns = rows of the table [owned rows | halo rows] — is the
operator of layers.py:19-20 restricted to the rank's rows.  It must be planned ("planned", sell_spmm_kernel), equal the C oracle's
row loop on the same CSR to 1e-5, be bit-stable from launch to launch, and a whole propagation stitched from P virtual ranks on
one GPU must equal the single-graph forward (lightgcn.py:70-81)."""
import numpy as np
import pytest
import torch

from oracle import coracle as C

pytestmark = pytest.mark.gpu


def _graph(rbg, nu=3001, ni=4001, n_inter=150_000, seed=5):
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, n_inter, seed=seed)
    # one hub user beyond 1 024 entries (the 32-piece workgroup rows) and one item row of a few hundred (2 - 8 pieces)
    extra_i = np.arange(1, 2401, dtype=np.int64)
    extra_u = np.full_like(extra_i, 7)
    pairs = np.unique(np.stack([np.concatenate([uid, extra_u]), np.concatenate([iid, extra_i])], 1), axis=0)
    return pairs[:, 0].copy(), pairs[:, 1].copy(), nu, ni


def _plans(rbg, world):
    uid, iid, nu, ni = _graph(rbg)
    sh = rbg.sharded
    owner = sh.degree_striped_partition(uid, iid, nu, ni, world)
    return sh.build_plans(uid, iid, nu, ni, world, owner=owner), (uid, iid, nu, ni)


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_cat_block_is_planned_and_matches_the_row_loop(rbg, cuda, world, d):
    plans, _ = _plans(rbg, world)
    plan = plans[world - 1]
    rp, col, val = plan.cat_csr()
    n_cols = plan.n_owned + plan.n_halo
    # cat_csr = the two blocks side by side
    ip, ic, iv = plan.int_csr
    hp, hc, hv = plan.halo_csr
    assert rp[-1] == ip[-1] + hp[-1] and col.max() < n_cols
    g = rbg.GraphHandle.from_csr(rp, col, val, n_cols, device=cuda, n_class0_rows=plan.n_users_owned)
    assert g.sell_status() == "planned"
    assert g.spmm_kernel_name(d).startswith("sell_spmm_kernel")
    rng = np.random.default_rng(d + world)
    x = rng.standard_normal((n_cols, d)).astype(np.float32)
    ref = C.spmm(rp, col.astype(np.int64), val, x)
    xt = torch.from_numpy(x).to(cuda)
    y = torch.empty((plan.n_owned, d), device=cuda)
    be = rbg.sharded.HipBackend(cuda)
    be.spmm(g, xt, y, False)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5
    # the two blocks separately give the same rows (interior + halo, the r05 form)
    ref2 = C.spmm(ip, np.asarray(ic, dtype=np.int64), iv, x[: plan.n_owned]) + C.spmm(hp, np.asarray(hc, dtype=np.int64), hv, x[plan.n_owned:])
    assert np.abs(ref2 - ref).max() <= 1e-5
    # bit-stable
    y2 = torch.empty_like(y)
    be.spmm(g, xt, y2, False)
    assert torch.equal(y, y2)
    # accumulate
    y3 = torch.ones_like(y)
    be.spmm(g, xt, y3, True)
    assert np.abs(y3.cpu().numpy() - (ref + 1.0)).max() <= 1e-5
    # the layer mean in the epilogue
    srcs = [torch.from_numpy(rng.standard_normal((plan.n_owned, d)).astype(np.float32)).to(cuda) for _ in range(3)]
    out = torch.empty_like(y)
    be.spmm_mean(g, xt, None, srcs, out)
    want = (sum(s.cpu().numpy().astype(np.float64) for s in srcs) + ref) / 4.0
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5


def test_halo_block_alone_is_planned(rbg, cuda):
    """The two-handle form's halo block [n_owned x n_halo] is a rectangular plan as well (r05: binned kernel)."""
    plans, _ = _plans(rbg, 4)
    plan = plans[1]
    hp, hc, hv = plan.halo_csr
    g = rbg.GraphHandle.from_csr(hp, hc, hv, plan.n_halo, device=cuda, n_class0_rows=plan.n_users_owned)
    assert g.sell_status() == "planned"
    x = np.random.default_rng(3).standard_normal((plan.n_halo, 64)).astype(np.float32)
    ref = C.spmm(hp, np.asarray(hc, dtype=np.int64), hv, x)
    y = torch.full((plan.n_owned, 64), 2.0, device=cuda)
    rbg.sharded.HipBackend(cuda).spmm(g, torch.from_numpy(x).to(cuda), y, True)
    assert np.abs(y.cpu().numpy() - (ref + 2.0)).max() <= 1e-5


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("fused", [True, False, "windows"])
def test_virtual_ranks_on_one_gpu_equal_the_single_graph_forward(rbg, cuda, world, fused):
    """P ShardedPropagation objects in ONE process; the exchange is played by indexing the assembled global layer
    (``halo_rows``), so every launch of the fused (or two-handle) layer runs as on P GPUs.  3 layers + mean."""
    plans, (uid, iid, nu, ni) = _plans(rbg, world)
    sh = rbg.sharded
    d, K = 64, 3
    be = sh.HipBackend(cuda)
    # "windows": the layer table cut into column windows of ~ a third (what a table beyond 32-bit offsets gets): 3+ launches per layer
    kw = {"cat_window_rows": (plans[0].n_owned + plans[0].n_halo) // 3} if fused == "windows" else {}
    props = [sh.ShardedPropagation(plans[p], be, transport="staged", fused=bool(fused), **kw) for p in range(world)]
    for pr in props:
        st = pr.kernel_status()
        assert st["form"] == ("fused" if fused else "two handles")
        if fused == "windows":
            assert st["windows"] >= 3 and all(v == "planned" for v in st["cat"]), st
        else:
            assert all(v == "planned" for k, v in st.items() if k != "form" and v is not None), st
    e0 = np.random.default_rng(11).standard_normal((nu + ni, d)).astype(np.float32)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    ref = C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], K)
    x_glob = torch.from_numpy(e0).to(cuda)
    acc = x_glob.clone()
    for _ in range(K):
        nxt = torch.empty_like(x_glob)
        for p, pr in enumerate(props):
            pl = plans[p]
            own = torch.as_tensor(pl.owned, device=cuda)
            halo = x_glob[torch.as_tensor(pl.halo_ids, device=cuda)]
            if halo.shape[0] == 0:
                halo = x_glob.new_zeros((1, d))
            y = pr.spmm(x_glob[own].contiguous(), halo_rows=halo)
            nxt[own] = y
        x_glob = nxt
        acc += x_glob
    got = (acc / (K + 1)).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5


def test_rect_plan_refuses_what_it_cannot_serve(rbg, cuda):
    """No row classes -> no plan (the binned kernel, with the reason kept); results still right."""
    plans, _ = _plans(rbg, 2)
    plan = plans[0]
    rp, col, val = plan.cat_csr()
    n_cols = plan.n_owned + plan.n_halo
    g = rbg.GraphHandle.from_csr(rp, col, val, n_cols, device=cuda)  # classes not declared
    assert g.sell_status() != "planned"
    x = np.random.default_rng(4).standard_normal((n_cols, 64)).astype(np.float32)
    y = torch.empty((plan.n_owned, 64), device=cuda)
    rbg.sharded.HipBackend(cuda).spmm(g, torch.from_numpy(x).to(cuda), y, False)
    assert np.abs(y.cpu().numpy() - C.spmm(rp, col.astype(np.int64), val, x)).max() <= 1e-5
