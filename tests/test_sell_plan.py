"""The planner of the column-slab propagation (tests/sell_spec.py, moved out of the product package in r05) on CPU tensors: the plan alone must reproduce
Y = A X (float64 emulation of exactly the layout csrc/sell.hip reads), cover every entry once, keep a wide row's U units
together at the front of its class (r06: any U, each unit knows its index and U), and pad with {K_PAST, 0}.  The kernel itself is checked on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import coracle as C


def _graph(rbg, name):
    uid, iid, nu, ni = rbg.synth.make(name)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    return uid, iid, nu, ni, rowptr, col, val


@pytest.mark.parametrize("W", [32, 64])
@pytest.mark.parametrize("chunk", [4, 16, 64])
@pytest.mark.parametrize("name", ["toy", "ml-100k"])
def test_plan_reproduces_the_product(rbg, name, W, chunk):
    import sell_spec as sell
    uid, iid, nu, ni, rowptr, col, val = _graph(rbg, name)
    plan = sell.build_plan(torch.from_numpy(rowptr), torch.from_numpy(col), torch.from_numpy(val), nu, ni, W=W, chunk=chunk)
    n = nu + ni
    x = np.random.default_rng(0).standard_normal((n, 3))
    a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(n, n))
    assert np.abs(sell.emulate(plan, x) - a @ x).max() < 1e-12
    lgw = 64 // (W // 4)
    ent, head = plan["ent"].numpy(), plan["head"].numpy().astype(np.int64)
    # every real entry exactly once; padding = {K_PAST, 0.0}
    real = ent[: plan["n_ent"], 0] != sell.K_PAST
    assert int(real.sum()) == int(rowptr[-1]) and np.all(ent[: plan["n_ent"]][~real, 1] == 0)
    assert np.all(ent[plan["n_ent"]:] == 0) and len(ent) == plan["n_ent"] + 128
    # orig is a permutation that keeps the two classes apart
    orig = plan["orig"].numpy()
    assert np.array_equal(np.sort(orig), np.arange(n)) and np.all(orig[:nu] < nu) and np.all(orig[nu:] >= nu)
    for c in (0, 1):
        h = head[plan["unit_base"][c]: plan["unit_base"][c] + plan["n_units"][c]]
        nc, lp, nrows, wide = (h[:, 2] >> 16) & 0xFFFF, h[:, 3] & 0xFF, (h[:, 3] >> 8) & 0xFF, (h[:, 3] >> 16) & 1
        wj, wu = h[:, 2] & 0xFFFF, h[:, 3] >> 17
        assert np.all(h[:, 0] % 2 == 0) and np.all(nc % 2 == 0)
        assert np.all((1 << lp) <= lgw) and np.all(nrows <= (lgw >> lp)) and np.all(h[:, 1] + nrows <= [nu, ni][c])
        # wide rows: U = ceil(degree / (chunk lgw)) consecutive units at the front of the class, one row each, numbered 0 .. U - 1
        nw = int(wide.sum())
        assert np.all(wide[:nw] == 1) and np.all(nrows[:nw] == 1) and np.all(wj[nw:] == 0) and np.all(wu[nw:] == 0)
        first = np.flatnonzero(wj[:nw] == 0)
        deg_c = np.diff(rowptr)[([0, nu][c]):([nu, nu + ni][c])][plan["orig"].numpy()[([0, nu][c]):([nu, nu + ni][c])] - [0, nu][c]]
        for f in first:
            u_ = int(wu[f])
            assert u_ >= 2 and np.array_equal(wj[f:f + u_], np.arange(u_)) and np.all(wu[f:f + u_] == u_) and np.all(h[f:f + u_, 1] == h[f, 1])
            assert u_ == -(-int(deg_c[h[f, 1]]) // (chunk * lgw))
        assert int(wu[first].sum()) == nw
        # rows in processing order: every row of the class appears in exactly one place
        rows = np.concatenate([np.arange(r0, r0 + k) for r0, k in zip(h[nw:, 1], nrows[nw:])] + [h[first, 1]]) if len(h) else np.zeros(0)
        assert np.array_equal(np.sort(rows), np.arange([nu, ni][c]))
    if chunk == 4:
        assert any(((head[:, 3] >> 16) & 1).tolist())  # the small chunk really produced wide rows


def test_plan_is_deterministic_and_matches_the_heaviest_first_order(rbg):
    import sell_spec as sell
    uid, iid, nu, ni, rowptr, col, val = _graph(rbg, "ml-100k")
    t = [torch.from_numpy(a) for a in (rowptr, col, val)]
    p1, p2 = sell.build_plan(*t, nu, ni, W=32), sell.build_plan(*t, nu, ni, W=32)
    for k in ("ent", "head", "orig"):
        assert torch.equal(p1[k], p2[k])
    head = p1["head"].numpy().astype(np.int64)
    for c in (0, 1):
        h = head[p1["unit_base"][c]: p1["unit_base"][c] + p1["n_units"][c]]
        slots, kind = (h[:, 2] >> 16) & 0xFFFF, h[:, 3] & 0x100FF  # kind = (wide, log2 parts)
        assert slots.max() <= sell.CHUNK  # no piece longer than the chunk (r06: a row gets as many units as it needs)
        # (parts, degree) descending: wide rows, then 8-, 4-, 2-part rows, then whole rows, each group longest first
        parts_rank = np.where(h[:, 3] >> 16 & 1, 99, h[:, 3] & 0xFF)
        assert np.all(np.diff(parts_rank) <= 0)
        for k in np.unique(kind):
            if not (k >> 16) & 1:  # (wide rows: a row with one unit more has shorter pieces than the next row down)
                assert np.all(np.diff(slots[kind == k]) <= 2)  # (the pieces of one row differ by one entry: +-2 after the rounding)


def test_row_factors_are_found_for_the_symmetric_normalisation_only(rbg):
    """val_ij = r_i r_j with r = deg^-1/2 (dataset.py:41-79): the planner returns r in the plan's numbering, and nothing for
    values that do not factor (a re-weighted graph) — rbg_graph_sell_set_factors re-checks every value on the device."""
    import sell_spec as sell
    uid, iid, nu, ni, rowptr, col, val = _graph(rbg, "ml-100k")
    t = [torch.from_numpy(a) for a in (rowptr, col, val)]
    plan = sell.build_plan(*t, nu, ni, W=32)
    r = plan["factors"].numpy().astype(np.float64)
    deg = np.diff(rowptr)[plan["orig"].numpy()]
    want = np.where(deg > 0, np.maximum(deg, 1) ** -0.5, 0.0)
    assert np.abs(r - want).max() < 1e-7 and r[deg == 0].max(initial=0.0) == 0.0
    rows = np.repeat(np.arange(nu + ni), np.diff(rowptr))
    inv = np.empty(nu + ni, dtype=np.int64)
    inv[plan["orig"].numpy()] = np.arange(nu + ni)
    assert np.abs(r[inv[rows]] * r[inv[col]] - val).max() < 1e-6 * val.max()
    other = val * np.random.default_rng(0).uniform(0.5, 1.5, len(val)).astype(np.float32)
    assert sell.build_plan(t[0], t[1], torch.from_numpy(other), nu, ni, W=32)["factors"] is None


def test_hub_rows_get_as_many_units_as_they_need(rbg):
    """r06: a row of any length is cut into U = ceil(degree / (chunk LGW)) units (r03-r05: four units, and a hub beyond
    4 LGW x max(512, nnz / 8192) entries made the planner say NotApplicable — a 20 000-entry hub in a small graph, or the hubs
    of a node-range shard whose nnz shrinks with the rank count while its hubs do not)."""
    import sell_spec as sell
    lgw = 8
    hub = 512 * 4 * lgw + 1000
    nu, ni = 400, hub + 10
    u = np.concatenate([np.full(hub, 1), np.arange(1000) % (nu - 2) + 2]).astype(np.int64)
    i = np.concatenate([np.arange(hub) + 1, (np.arange(1000) * 7919) % (ni - 1) + 1]).astype(np.int64)
    key = np.unique(u * ni + i)
    u, i = key // ni, key % ni
    rowptr, col, val = C.build_norm_csr(u, i, nu, ni)
    plan = sell.build_plan(*[torch.from_numpy(a) for a in (rowptr, col, val)], nu, ni, W=32)
    head = plan["head"].numpy().astype(np.int64)
    wide = (head[:, 3] >> 16) & 1
    want = -(-int(np.diff(rowptr).max()) // (sell.CHUNK * lgw))
    assert int(wide.sum()) == want and int(head[0, 3] >> 17) == want
    x = np.random.default_rng(0).standard_normal((nu + ni, 2))
    a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(nu + ni, nu + ni))
    assert np.abs(sell.emulate(plan, x) - a @ x).max() < 1e-12
