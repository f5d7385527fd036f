"""The planner of the column-slab propagation (tests/sell_spec.py, moved out of the product package in r05) on CPU tensors: the plan alone must reproduce
Y = A X (float64 emulation of exactly the layout csrc/sell.hip reads), cover every entry once, keep wide rows aligned to
four units, and pad with {K_PAST, 0}.  The kernel itself is checked on the GPU (tests/test_gpu_parity.py)."""
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
        nc, lp, nrows, wide = h[:, 2] >> 16, h[:, 3] & 0xFF, (h[:, 3] >> 8) & 0xFF, (h[:, 3] >> 16) & 1
        assert np.all(h[:, 0] % 2 == 0) and np.all(nc % 2 == 0) and np.all((h[:, 2] & 0xFFFF) == 0)
        assert np.all((1 << lp) <= lgw) and np.all(nrows <= (lgw >> lp)) and np.all(h[:, 1] + nrows <= [nu, ni][c])
        # wide rows: whole groups of four units at the front of the class, one row each
        nw = int(wide.sum())
        assert nw % 4 == 0 and np.all(wide[:nw] == 1) and np.all(nrows[:nw] == 1)
        assert np.all(h[:nw, 1].reshape(-1, 4) == h[:nw:4, 1][:, None])
        # rows in processing order: every row of the class appears in exactly one place
        rows = np.concatenate([np.arange(r0, r0 + k) for r0, k in zip(h[nw:, 1], nrows[nw:])] + [h[:nw:4, 1]]) if len(h) else np.zeros(0)
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
        slots, kind = h[:, 2] >> 16, h[:, 3] & 0x100FF  # kind = (wide, log2 parts)
        deg = np.diff(rowptr)[([0, nu][c]):([nu, nu + ni][c])]
        assert slots.max() <= sell.CHUNK or deg.max() > sell.CHUNK * 8 * 4  # no piece longer than the chunk unless even 32 parts cannot hold the row
        # (parts, degree) descending: wide rows, then 8-, 4-, 2-part rows, then whole rows, each group longest first
        parts_rank = np.where(h[:, 3] >> 16 & 1, 99, h[:, 3] & 0xFF)
        assert np.all(np.diff(parts_rank) <= 0)
        for k in np.unique(kind):
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


def test_hub_row_limit_scales_with_the_graph(rbg):
    """A row is summed serially by 4 LGW lane-groups: beyond 4 LGW x max(MAX_PIECE, nnz / 8192) entries the planner says
    NotApplicable (the caller keeps the binned kernel, which splits hub rows over workgroups) — a 20 000-entry hub is refused
    in a small graph and planned in a large one."""
    import sell_spec as sell
    lgw = 8
    hub = sell.MAX_PIECE * 4 * lgw + 1000
    nu, ni = 400, hub + 10

    def graph(extra):
        u = np.concatenate([np.full(hub, 1), np.arange(extra) % (nu - 2) + 2]).astype(np.int64)
        i = np.concatenate([np.arange(hub) + 1, (np.arange(extra) * 7919) % (ni - 1) + 1]).astype(np.int64)
        key = np.unique(u * ni + i)
        u, i = key // ni, key % ni
        rowptr, col, val = C.build_norm_csr(u, i, nu, ni)
        return [torch.from_numpy(a) for a in (rowptr, col, val)]

    with pytest.raises(sell.NotApplicable):
        sell.build_plan(*graph(1000), nu, ni, W=32)
    big = graph(2_400_000)  # nnz / 8192 > hub / 32: the hub's 32 pieces are no longer the longest chain by far
    assert big[1].numel() // 8192 * 4 * lgw >= hub
    plan = sell.build_plan(*big, nu, ni, W=32)
    head = plan["head"].numpy().astype(np.int64)
    assert ((head[:, 3] >> 16) & 1).sum() >= 4  # the hub is a wide row: four units
