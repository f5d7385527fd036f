"""Host side of the engine (graph builder, binning plan, generator) against the oracle and the
golden vectors — bit-exact: this is integer / index work plus IEEE 1/sqrt.  CPU only."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from conftest import known_graphs
from oracle import coracle as C
from oracle import oracle as O


def test_builder_matches_golden(rbg, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni)
    rowptr, col, val = h.export_csr()
    assert (h.n_rows, h.n_cols, h.nnz) == (nu + ni, nu + ni, 2 * len(g["uid"]))
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"]) and np.array_equal(val, g["val"])
    ei, ew = rbg.norm_edges(g["uid"], g["iid"], nu, ni)
    oei, _ = O.build_edge_index(g["uid"], g["iid"], nu)
    assert torch.equal(ei, oei) and np.array_equal(ew.numpy(), g["edge_weight"])
    # SGL edge-drop view, re-normalized on its own degrees (sgl.py:107-126)
    v = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, keep=g["sgl_keep"])
    vrp, vcol, vval = v.export_csr()
    assert np.array_equal(vrp, g["sgl_rowptr"]) and np.array_equal(vcol, g["sgl_col"]) and np.array_equal(vval, g["sgl_val"])


@pytest.mark.parametrize("name", list(known_graphs()))
def test_builder_known_answers(rbg, name):
    kg = known_graphs()[name]
    n = kg["n_users"] + kg["n_items"]
    h = rbg.GraphHandle.from_interactions(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"])
    rowptr, col, val = h.export_csr()
    a = np.zeros((n, n))
    for r in range(n):
        for e in range(rowptr[r], rowptr[r + 1]):
            a[r, col[e]] += val[e]
    np.testing.assert_allclose(a, kg["dense"], atol=1e-7)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 15), st.integers(1, 15), st.data())
def test_builder_random_graphs(rbg, nu, ni, data):
    e = data.draw(st.integers(0, 60))
    uid = np.asarray(data.draw(st.lists(st.integers(0, nu - 1), min_size=e, max_size=e)), dtype=np.int64)
    iid = np.asarray(data.draw(st.lists(st.integers(0, ni - 1), min_size=e, max_size=e)), dtype=np.int64)
    keep = np.asarray(data.draw(st.lists(st.integers(0, 1), min_size=e, max_size=e)), dtype=np.uint8)
    for k in (None, keep):
        h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, keep=k)
        rowptr, col, val = h.export_csr()
        crp, ccol, cval = C.build_norm_csr(uid, iid, nu, ni, keep=k)
        assert np.array_equal(rowptr, crp) and np.array_equal(col, ccol) and np.array_equal(val, cval)
    ei, ew = rbg.norm_edges(uid, iid, nu, ni)
    oei, oew = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=False)
    assert torch.equal(ei, oei) and torch.equal(ew, oew)
    # the dense pair and a raw CSR hand the same matrix over
    h2 = rbg.GraphHandle.from_edge_index(ei, ew, nu + ni)
    h1 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni)
    for a, b in zip(h1.export_csr(), h2.export_csr()):
        assert np.array_equal(a, b)
    rp, c, v = h1.export_csr()
    h3 = rbg.GraphHandle.from_csr(rp, c, v, nu + ni)
    for a, b in zip(h1.export_csr(), h3.export_csr()):
        assert np.array_equal(a, b)


def test_empty_and_degenerate(rbg):
    h = rbg.GraphHandle.from_interactions([], [], 0, 0)
    assert (h.n_rows, h.nnz) == (0, 0)
    h = rbg.GraphHandle.from_interactions([], [], 3, 4)
    rowptr, col, val = h.export_csr()
    assert h.n_rows == 7 and h.nnz == 0 and np.array_equal(rowptr, np.zeros(8, dtype=np.int64))
    h = rbg.GraphHandle.from_interactions([1, 2], [1, 2], 3, 3, keep=np.zeros(2, dtype=np.uint8))
    assert h.nnz == 0
    with pytest.raises(rbg.RbgError):
        rbg.GraphHandle.from_csr([0, 2], [0, 5], [1.0, 1.0], 3)  # column out of range
    with pytest.raises(rbg.RbgError):
        rbg.GraphHandle.from_csr([0, 2, 1], [0, 1], [1.0, 1.0], 3)  # rowptr not monotone


def test_transpose_of_rectangular_csr(rbg):
    rowptr = np.array([0, 2, 3], dtype=np.int64)
    col = np.array([0, 2, 1], dtype=np.int32)
    val = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    h = rbg.GraphHandle.from_csr(rowptr, col, val, 3)
    t = h.transpose()
    trp, tcol, tval = t.export_csr()
    assert (t.n_rows, t.n_cols) == (3, 2)
    assert np.array_equal(trp, [0, 1, 2, 3]) and np.array_equal(tcol, [0, 1, 0]) and np.array_equal(tval, [1, 3, 2])
    assert t.transpose() is h


def test_generator_is_deterministic_and_shaped(rbg):
    u1, i1, nu, ni = rbg.synth.make("toy")
    u2, i2, _, _ = rbg.synth.make("toy")
    assert np.array_equal(u1, u2) and np.array_equal(i1, i2)
    assert len(u1) == 5999 and u1.min() >= 1 and i1.min() >= 1 and u1.max() < nu and i1.max() < ni
    assert len(np.unique(u1 * ni + i1)) == len(u1)  # unique pairs
    ub, ib, _, _ = rbg.synth.make("toy", n_blocks=2, p_in=1.0)
    assert np.all((ub - 1) % 2 == (ib - 1) % 2)  # every interaction stays inside its block
    b_layer, b_prop = rbg.synth.algorithmic_bytes(70841, 2054740, 64, 3)
    assert b_layer == 4 * 70842 + 8 * 2054740 + 8 * 70841 * 64
    assert round(b_layer / 1e6, 2) == 52.99 and round(b_prop / 1e6, 2) == 249.65


def test_dataset_mirror(rbg, ref_inter):
    uid, iid, nu, ni = ref_inter
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    assert ds.num("user_id") == nu and ds.num("item_id") == ni and ds.is_sparse
    ei, ew = ds.get_norm_adj_mat()                       # reference default: enable_sparse=None
    assert ei.shape == (2, 2 * len(uid)) and ew.dtype == torch.float32
    g, none = ds.get_norm_adj_mat(enable_sparse=True)    # host handle when no device is given
    assert none is None and isinstance(g, rbg.GraphHandle) and g.symmetric


def test_partitioned_create_validates(rbg):
    uid, iid, nu, ni = [1, 2, 1], [1, 2, 2], 3, 3
    part = np.array([0, 0, 1, 0, 1, 1], dtype=np.int32)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, xcd_part=part)
    ref = rbg.GraphHandle.from_interactions(uid, iid, nu, ni)
    for a, b in zip(h.export_csr(), ref.export_csr()):
        assert np.array_equal(a, b)                      # the partition never changes the matrix
    with pytest.raises(rbg.RbgError):
        rbg.GraphHandle.from_interactions(uid, iid, nu, ni, xcd_part=np.array([0, 0, 2, 0, 1, 1]))  # 3 parts
    with pytest.raises(ValueError):
        rbg.GraphHandle.from_interactions(uid, iid, nu, ni, xcd_part=np.array([0, 1]))


def test_find_communities_recovers_planted_structure(rbg):
    nu, ni, e = 2001, 3001, 60_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=3, n_blocks=4, p_in=0.97, layout="contiguous")
    rng = np.random.default_rng(5)  # scramble the ids: the structure is no longer visible in the numbering
    pu = np.concatenate([[0], rng.permutation(nu - 1) + 1])
    pi = np.concatenate([[0], rng.permutation(ni - 1) + 1])
    lab, cut, imb = rbg.find_communities(pu[uid], pi[iid], nu, ni, n_parts=4)
    assert lab.shape == (nu + ni,) and lab.min() >= 0 and lab.max() < 4
    assert cut < 0.08 and imb < 1.1   # planted cut: 3 % x 3/4
    # an unstructured graph: no partition worth using -> "auto" keeps the default plan (and never changes the matrix)
    u2, i2 = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=4)
    _, cut2, _ = rbg.find_communities(u2, i2, nu, ni, n_parts=8)
    assert cut2 > 0.6
    ha = rbg.GraphHandle.from_interactions(pu[uid], pi[iid], nu, ni, xcd_part="auto")
    hd = rbg.GraphHandle.from_interactions(pu[uid], pi[iid], nu, ni)
    for a, b in zip(ha.export_csr(), hd.export_csr()):
        assert np.array_equal(a, b)


def test_device_generator_has_the_numpy_generators_properties(rbg):
    """synth.powerlaw_bipartite_device (bench.py's config-#5 workload: the generator's algorithm with torch's RNG, here on the
    CPU device): exactly n_inter unique pairs, ids in [1, n), PAD rows empty, the same power-law degree profile."""
    nu, ni, e = 2001, 3001, 60_000
    u, i = rbg.synth.powerlaw_bipartite_device(nu, ni, e, "cpu", seed=7)
    assert len(u) == e and len(np.unique(u.astype(np.int64) * ni + i)) == e
    assert u.min() >= 1 and i.min() >= 1 and u.max() < nu and i.max() < ni
    u2, i2 = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=7)
    d1, d2 = np.sort(np.bincount(u, minlength=nu))[::-1], np.sort(np.bincount(u2, minlength=nu))[::-1]
    assert abs(d1[:20].mean() - d2[:20].mean()) < 0.2 * d2[:20].mean()      # same head
    assert abs(np.median(d1) - np.median(d2)) <= 2                          # same bulk
    u3, i3 = rbg.synth.powerlaw_bipartite_device(nu, ni, e, "cpu", seed=7)
    assert np.array_equal(u, u3) and np.array_equal(i, i3)                  # deterministic
