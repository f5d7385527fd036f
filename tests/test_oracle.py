"""The oracle pinned against itself: hand-derived known answers, three
independent formulations, invariants, and the committed golden vectors.  CPU only."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from conftest import known_graphs
from oracle import coracle as C
from oracle import oracle as O


def dense_from_csr(rowptr, col, val, n):
    a = np.zeros((n, n))
    for r in range(n):
        for e in range(rowptr[r], rowptr[r + 1]):
            a[r, col[e]] += val[e]
    return a


@pytest.mark.parametrize("name", list(known_graphs()))
def test_known_answer_matrix(name):
    kg = known_graphs()[name]
    n = kg["n_users"] + kg["n_items"]
    rowptr, col, val = O.get_norm_adj_mat(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"], enable_sparse=True)
    np.testing.assert_allclose(dense_from_csr(rowptr, col, val, n), kg["dense"], atol=1e-7)
    crp, ccol, cval = C.build_norm_csr(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"])
    assert np.array_equal(rowptr, crp) and np.array_equal(col, ccol) and np.array_equal(val, cval)
    ei, ew = O.get_norm_adj_mat(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"], enable_sparse=False)
    a = np.zeros((n, n))
    for s, t, w in zip(ei[0].tolist(), ei[1].tolist(), ew.tolist()):
        a[t, s] += w
    np.testing.assert_allclose(a, kg["dense"], atol=1e-7)


def test_known_answer_propagation():
    # single interaction: Y = [0, X[3], 0, X[1]]; K=2 mean at node 1 = (2 X[1] + X[3]) / 3
    kg = known_graphs()["single"]
    x = torch.arange(16, dtype=torch.float32).reshape(4, 4) + 1
    rowptr, col, val = O.get_norm_adj_mat(kg["uid"], kg["iid"], 2, 2, enable_sparse=True)
    y = O.conv_csr_sequential(x, rowptr, col, val)
    assert torch.equal(y, torch.stack([torch.zeros(4), x[3], torch.zeros(4), x[1]]))
    u_all, i_all = O.lightgcn_forward(x[:2], x[2:], lambda t: O.conv_csr_sequential(t, rowptr, col, val), 2)
    torch.testing.assert_close(u_all[1], (2 * x[1] + x[3]) / 3)
    # PAD rows: mean[pad] = E0[pad] / (K+1)
    torch.testing.assert_close(u_all[0], x[0] / 3)
    torch.testing.assert_close(i_all[0], x[2] / 3)
    # star: Â² restricted to the hub is the identity
    kg = known_graphs()["star"]
    n = kg["n_users"] + kg["n_items"]
    rowptr, col, val = O.get_norm_adj_mat(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"], enable_sparse=True)
    x = torch.randn(n, 3, generator=torch.Generator().manual_seed(1))
    y2 = O.conv_csr_sequential(O.conv_csr_sequential(x, rowptr, col, val), rowptr, col, val)
    torch.testing.assert_close(y2[1], x[1], atol=1e-6, rtol=0)


def test_three_formulations_agree(ref_inter):
    uid, iid, nu, ni = ref_inter
    rowptr, col, val = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=True)
    ei, ew = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=False)
    x = torch.randn(nu + ni, 24, generator=torch.Generator().manual_seed(7))
    y_dense = O.conv_dense(x, ei, ew)                      # layers.py:16-17 branch
    y_seq = O.conv_csr_sequential(x, rowptr, col, val)     # layers.py:19-20 branch, python loop
    y_c = torch.from_numpy(C.spmm(rowptr, col, val, x.numpy()))  # same loop in C
    y_64 = O.conv_csr_f64(x.numpy(), rowptr, col, val)
    assert torch.equal(y_seq, y_c)
    assert (y_dense - y_c).abs().max() < 2e-6
    assert np.abs(y_c.numpy() - y_64).max() < 2e-6


def test_invariants(ref_inter):
    uid, iid, nu, ni = ref_inter
    n = nu + ni
    rowptr, col, val = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=True)
    import scipy.sparse as sp
    a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(n, n))
    assert abs(a - a.T).max() < 1e-7                       # symmetric
    deg = np.diff(rowptr).astype(np.float64)
    s = np.sqrt(deg)
    np.testing.assert_allclose(a @ s, np.where(deg > 0, s, 0.0), atol=1e-5)  # Â D^1/2 1 = D^1/2 1
    assert deg[0] == 0 and deg[nu] == 0                    # the two PAD rows are empty
    assert a[:nu, :nu].nnz == 0 and a[nu:, nu:].nnz == 0   # bipartite blocks
    sv = sp.linalg.svds(a, k=1, return_singular_vectors=False)[0]
    assert sv <= 1 + 1e-6                                  # spectral radius <= 1


@settings(max_examples=25, deadline=None)
@given(st.integers(2, 12), st.integers(2, 12), st.data())
def test_random_graphs_with_duplicates(nu, ni, data):
    e = data.draw(st.integers(0, 40))
    uid = data.draw(st.lists(st.integers(0, nu - 1), min_size=e, max_size=e))
    iid = data.draw(st.lists(st.integers(0, ni - 1), min_size=e, max_size=e))
    n = nu + ni
    rowptr, col, val = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=True)
    crp, ccol, cval = C.build_norm_csr(np.asarray(uid, dtype=np.int64), np.asarray(iid, dtype=np.int64), nu, ni)
    assert np.array_equal(rowptr, crp) and np.array_equal(col, ccol) and np.array_equal(val, cval)
    ei, ew = O.get_norm_adj_mat(uid, iid, nu, ni, enable_sparse=False)
    x = torch.randn(n, 5, generator=torch.Generator().manual_seed(e))
    y1 = O.conv_dense(x, ei, ew)
    y2 = torch.from_numpy(C.spmm(rowptr, col, val, x.numpy()))
    assert (y1 - y2).abs().max() < 1e-5
    # linearity
    z = torch.randn(n, 5, generator=torch.Generator().manual_seed(e + 1))
    y3 = torch.from_numpy(C.spmm(rowptr, col, val, (2 * x + z).numpy()))
    yz = torch.from_numpy(C.spmm(rowptr, col, val, z.numpy()))
    assert (y3 - (2 * y2 + yz)).abs().max() < 1e-4


def test_golden_vectors_reproduce(golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    rowptr, col, val = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=True)
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"]) and np.array_equal(val, g["val"])
    _, ew = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=False)
    assert np.array_equal(ew.numpy(), g["edge_weight"])
    e0 = g["e0_d16"]
    for k in (1, 2, 3):
        mean, layers = C.lightgcn_forward(rowptr, col, val, e0[:nu], e0[nu:], k, return_layers=True)
        assert np.array_equal(mean, g[f"mean_k{k}_d16"])
        assert np.array_equal(layers[k], g[f"e{k}_d16"])
    assert np.abs(g["mean_k3_d16"] - g["mean_k3_d16_f64"]).max() < 1e-6
    mean64 = C.lightgcn_forward(rowptr, col, val, g["e0_d64"][:nu], g["e0_d64"][nu:], 3)
    assert np.array_equal(mean64, g["mean_k3_d64"])
    scores = O.full_sort_predict(torch.from_numpy(mean64[:nu]), torch.from_numpy(mean64[nu:]), g["score_users"])
    np.testing.assert_allclose(scores.numpy(), g["scores_d64"], atol=1e-6)
    keep = g["sgl_keep"].astype(bool)
    vrp, vcol, vval = O.get_norm_adj_mat(g["uid"][keep], g["iid"][keep], nu, ni, enable_sparse=True)
    assert np.array_equal(vrp, g["sgl_rowptr"]) and np.array_equal(vcol, g["sgl_col"]) and np.array_equal(vval, g["sgl_val"])
    assert keep.sum() == int(len(keep) * 0.9)


def test_ngcf_oracle_golden(golden):
    g = golden
    nu = int(g["n_users"])
    rowptr, col, val = g["rowptr"], g["col"].astype(np.int64), g["val"]
    conv = lambda t: torch.from_numpy(C.spmm(rowptr, col, val, t.numpy()))  # noqa: E731
    e0 = torch.from_numpy(g["e0_d16"])
    params = [tuple(torch.from_numpy(g[f"ngcf_{nm}_{li}"]) for nm in ("w1", "b1", "w2", "b2")) for li in (0, 1)]
    np.testing.assert_allclose(O.bignn_conv(e0, conv, *params[0]).numpy(), g["bignn_conv0"], atol=1e-6)
    u_all, i_all = O.ngcf_forward(e0[:nu], e0[nu:], conv, params)
    out = torch.cat([u_all, i_all]).numpy()
    np.testing.assert_allclose(out, g["ngcf_out"], atol=1e-6)
    assert out.shape[1] == 16 + 16 + 8
    # every propagated block is row-normalized (or an all-zero row)
    nrm = np.linalg.norm(out[:, 16:32], axis=1)
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm < 1e-12))


# ---- restatements added with the "next" rows (SGL InfoNCE, SimGCL / XSimGCL) -------------------------------------------

def test_ssl_loss_known_answers():
    """calc_ssl_loss (sgl.py:176-209): with identical views and one-hot rows the loss has a closed form."""
    n, tau = 4, 0.5
    eye = torch.eye(n, dtype=torch.float64)
    users = torch.tensor([0, 2])
    items = torch.tensor([1])
    # a = p = e_u, candidates = all e_j: v1 = exp(1/tau), v2 = exp(1/tau) + (n-1) exp(0)
    per_row = -np.log(np.exp(1 / tau) / (np.exp(1 / tau) + (n - 1)))
    got = O.calc_ssl_loss(users, items, eye, eye, eye, eye, tau, 0.3)
    assert abs(float(got) - 0.3 * 3 * per_row) < 1e-12
    # scale invariance of the normalised rows, and lse_rows == torch.logsumexp
    q = torch.randn(5, 7, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    c = torch.randn(9, 7, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    a = O.calc_ssl_loss(torch.arange(5), torch.arange(5), q, q * 3.0, q, q * 3.0, tau, 1.0)
    b = O.calc_ssl_loss(torch.arange(5), torch.arange(5), q * 0.5, q, q * 0.5, q, tau, 1.0)
    assert abs(float(a) - float(b)) < 1e-9
    assert torch.allclose(O.lse_rows(q, c, 2.0), torch.logsumexp(2.0 * q @ c.T, dim=1))


def test_simgcl_forward_restatement(ref_inter):
    """simgcl.py:24-38 / xsimgcl.py:28-48: no E0 in the mean; zero-eps noise changes nothing; the noise term has norm eps
    per row and the sign of the clean embedding; layer_cl picks the right layer."""
    uid, iid, nu, ni = ref_inter
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    conv = lambda t: torch.from_numpy(O.conv_csr_f64(t.numpy(), rowptr, col, val))
    gen = torch.Generator().manual_seed(3)
    uw, iw = torch.randn(nu, 8, dtype=torch.float64, generator=gen), torch.randn(ni, 8, dtype=torch.float64, generator=gen)
    e0 = torch.cat([uw, iw])
    e1 = conv(e0)
    e2 = conv(e1)
    u, i = O.simgcl_forward(uw, iw, conv, 2)
    assert torch.allclose(torch.cat([u, i]), (e1 + e2) / 2)
    noises = [torch.rand(nu + ni, 8, dtype=torch.float64, generator=gen) for _ in range(2)]
    u0, i0 = O.simgcl_forward(uw, iw, conv, 2, noises=noises, eps=0.0)
    assert torch.allclose(torch.cat([u0, i0]), (e1 + e2) / 2)
    out = O.simgcl_forward(uw, iw, conv, 2, noises=noises, eps=0.1, layer_cl=1)
    p1 = torch.cat(out[2:])  # the perturbed layer-1 embedding
    delta = p1 - e1
    rows = e1.abs().sum(1) > 0
    assert torch.allclose(delta[rows].norm(dim=1) ** 2, (0.01 * (torch.nn.functional.normalize(noises[0], dim=-1)[rows] ** 2)
                                                          * (e1[rows] != 0)).sum(1))
    assert torch.all(torch.sign(delta[rows]) * torch.sign(e1[rows]) >= 0)
    assert torch.all(delta[~rows] == 0)  # PAD rows: sign(0) = 0
    assert abs(float(O.simgcl_cl_loss(e1[:6], e1[:6], 0.2, "mean")) - float(O.simgcl_cl_loss(e1[:6], e1[:6], 0.2)) / 6) < 1e-12


def test_dropout_adj_keeps_weights_and_directions():
    """ngcf.py:81-82 / PyG dropout_adj: a filter — no rescale, no re-normalisation, directions independent."""
    ei, ew = O.get_norm_adj_mat(np.array([1, 1, 2]), np.array([1, 2, 1]), 3, 3, enable_sparse=False)
    keep = torch.tensor([True, False, True, True, True, False])
    ei2, ew2 = O.dropout_adj(ei, ew, keep)
    assert torch.equal(ei2, ei[:, keep]) and torch.equal(ew2, ew[keep]) and ei2.shape[1] == 4
    x = torch.eye(6)
    full, part = O.conv_dense(x, ei, ew), O.conv_dense(x, ei2, ew2)
    dropped = full - part  # exactly the two dropped directed edges, at their original weights
    assert int((dropped != 0).sum()) == 2 and not torch.equal(part, part.T)


def test_ncl_restatements_known_answers():
    """ncl.py:93-104,106-165 restated: the forward keeps every layer; the structure contrast of identical views reduces to
    -sum log softmax diag; Lloyd's objective never increases and converges on separated clusters."""
    g = torch.Generator().manual_seed(0)
    uw, iw = torch.randn(4, 8, generator=g), torch.randn(5, 8, generator=g)
    ident = lambda t: t  # noqa: E731
    u, i, embs = O.ncl_forward(uw, iw, ident, 3, 2)
    assert len(embs) == 5 and torch.allclose(torch.cat([u, i]), torch.cat([uw, iw]))  # max(3, 2*2) propagations, mean of 0..3
    x = torch.cat([uw, iw])
    user, item = torch.tensor([0, 2]), torch.tensor([1, 3])
    loss = O.ncl_ssl_layer_loss(x, x, 4, user, item, 0.5, 1.0, 2.0)
    xn = torch.nn.functional.normalize(x[:4])
    want_u = -(torch.log_softmax(xn[user] @ xn.T / 0.5, dim=1)[torch.arange(2), user]).sum()
    xi = torch.nn.functional.normalize(x[4:])
    want_i = -(torch.log_softmax(xi[item] @ xi.T / 0.5, dim=1)[torch.arange(2), item]).sum()
    assert torch.allclose(loss, want_u + 2.0 * want_i, atol=1e-5)
    cents = torch.nn.functional.normalize(torch.randn(3, 8, generator=g))
    n2c_u, n2c_i = torch.tensor([0, 1, 2, 0]), torch.tensor([2, 2, 1, 0, 1])
    proto = O.ncl_proto_nce_loss(x, 4, user, item, cents, n2c_u, cents, n2c_i, 0.5, 1.0)
    want = -(torch.log_softmax(xn[user] @ cents.T / 0.5, dim=1)[torch.arange(2), n2c_u[user]]).sum() \
        - (torch.log_softmax(xi[item] @ cents.T / 0.5, dim=1)[torch.arange(2), n2c_i[item]]).sum()
    assert torch.allclose(proto, want, atol=1e-5)
    rng = np.random.default_rng(0)
    centers = rng.standard_normal((4, 6)) * 10
    pts = centers[rng.integers(0, 4, 200)] + rng.standard_normal((200, 6)) * 0.1
    c, a, obj = O.kmeans_lloyd(pts, pts[[0, 50, 100, 150]] + 0.0, niter=10)
    assert all(b <= a_ + 1e-9 for a_, b in zip(obj, obj[1:]))
    for j in range(4):
        if (a == j).any():
            assert np.allclose(c[j], pts[a == j].mean(0))


def test_numa_aware_baseline_is_the_same_arithmetic(ref_inter):
    """oracle/rbg_oracle.c ora_numa_* (bench.py's NUMA-aware CPU baseline variant: thread-owned row blocks, first-touch
    placement): bit-identical to the plain restatement at several thread counts, also with more threads than rows per block."""
    from oracle import coracle
    uid, iid, nu, ni = ref_inter
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    rng = np.random.default_rng(0)
    uw, iw = rng.standard_normal((nu, 64)).astype(np.float32), rng.standard_normal((ni, 64)).astype(np.float32)
    ref = coracle.lightgcn_forward(rowptr, col, val, uw, iw, 3)
    keep = coracle.num_threads()
    try:
        for t in (1, 3, 8):
            coracle.set_num_threads(t)
            nf = coracle.NumaForward(rowptr, col, val, nu, ni, 64, 3)
            assert np.array_equal(nf(uw, iw), ref)
            assert nf(uw, iw, want_result=False) is None
            nf.close()
    finally:
        coracle.set_num_threads(keep)
