"""Regenerates the committed golden fixtures (run in the BUILD container only).

  python tests/golden/make_golden.py

* ref_test_inter.npz — the interactions of the reference's own test fixture
  (/root/reference/tests/test_data/test/test.inter, a data file its tests hold), remapped to dense
  ids the way RecBole does (0 = [PAD], tokens numbered in order of first appearance).  Data only.
* golden_ref_test.npz / golden_toy.npz — inputs and expected outputs of the path on that graph and
  on a synthetic power-law graph of the same size: CSR of the normalized adjacency, per-layer
  embeddings, layer means, full-sort scores, NGCF outputs, one SGL edge-drop view.
  Expected values come from the oracle (oracle/oracle.py, oracle/rbg_oracle.c): fp32 in the
  reference's loop order, plus float64 truth for the tolerance check.  The reference itself cannot
  be imported here, so these vectors pin the oracle against regressions and pin the
  HIP path to the oracle — they are not outputs of the reference ("parity unpinned").
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import coracle as C  # noqa: E402
from oracle import oracle as O  # noqa: E402

REF_INTER = "/root/reference/tests/test_data/test/test.inter"


def factorize(tokens):
    ids, seen = [], {}
    for t in tokens:
        if t not in seen:
            seen[t] = len(seen) + 1  # 0 is [PAD]
        ids.append(seen[t])
    return np.asarray(ids, dtype=np.int64), len(seen) + 1


def ref_interactions():
    users, items = [], []
    with open(REF_INTER) as f:
        next(f)
        for line in f:
            u, i = line.split("\t")[:2]
            users.append(u)
            items.append(i)
    uid, n_users = factorize(users)
    iid, n_items = factorize(items)
    return uid, iid, n_users, n_items


def synth_toy(n_users, n_items, n_inter, seed=2020):
    rng = np.random.default_rng(seed)
    pu = (np.arange(n_users - 1) + 10.0) ** -0.75
    pi = (np.arange(n_items - 1) + 10.0) ** -0.75
    keys = set()
    while len(keys) < n_inter:
        u = rng.choice(n_users - 1, size=n_inter, p=pu / pu.sum())
        i = rng.choice(n_items - 1, size=n_inter, p=pi / pi.sum())
        for a, b in zip(u, i):
            if len(keys) < n_inter:
                keys.add((int(a) + 1, int(b) + 1))
    arr = np.asarray(sorted(keys), dtype=np.int64)
    arr = arr[rng.permutation(len(arr))]
    return arr[:, 0].copy(), arr[:, 1].copy()


def golden_for(uid, iid, n_users, n_items, seed):
    n = n_users + n_items
    out = dict(uid=uid, iid=iid, n_users=np.int64(n_users), n_items=np.int64(n_items))
    rowptr, col, val = O.get_norm_adj_mat(uid, iid, n_users, n_items, enable_sparse=True)
    crp, ccol, cval = C.build_norm_csr(uid, iid, n_users, n_items)
    assert np.array_equal(rowptr, crp) and np.array_equal(col, ccol) and np.array_equal(val, cval)
    out.update(rowptr=rowptr, col=col.astype(np.int32), val=val)
    ei, ew = O.get_norm_adj_mat(uid, iid, n_users, n_items, enable_sparse=False)
    out.update(edge_weight=ew.numpy())
    gen = torch.Generator().manual_seed(seed)
    for d in (16, 64):
        e0 = torch.randn(n, d, generator=gen, dtype=torch.float32)
        out[f"e0_d{d}"] = e0.numpy()
        mean3, layers = C.lightgcn_forward(rowptr, col, val, e0[:n_users].numpy(), e0[n_users:].numpy(), 3,
                                           return_layers=True)
        if d == 16:
            for k in (1, 2, 3):
                out[f"e{k}_d16"] = layers[k]
                out[f"mean_k{k}_d16"] = C.lightgcn_forward(rowptr, col, val, e0[:n_users].numpy(),
                                                           e0[n_users:].numpy(), k)
            # float64 truth of the 3-layer mean
            x = e0.numpy().astype(np.float64)
            acc, cur = x.copy(), x
            for _ in range(3):
                cur = O.conv_csr_f64(cur, rowptr, col, val)
                acc = acc + cur
            out["mean_k3_d16_f64"] = acc / 4.0
        out[f"mean_k3_d{d}"] = mean3
        # dense-branch formulation must agree with the CSR one
        u_all, i_all = O.lightgcn_forward(e0[:n_users], e0[n_users:], lambda t: O.conv_dense(t, ei, ew), 3)
        assert (torch.cat([u_all, i_all]) - torch.from_numpy(mean3)).abs().max() < 2e-6
        if d == 64:
            users = np.asarray([1, 2, n_users - 1], dtype=np.int64)
            out["score_users"] = users
            out["scores_d64"] = O.full_sort_predict(torch.from_numpy(mean3[:n_users]), torch.from_numpy(mean3[n_users:]),
                                                    users).numpy()
    # NGCF: 2 layers 16 -> 16 -> 8, xavier-normal weights, small non-zero biases
    e0 = torch.from_numpy(out["e0_d16"])
    params, d_prev = [], 16
    for li, d_out in enumerate((16, 8)):
        w1 = torch.randn(d_out, d_prev, generator=gen) * float(np.sqrt(2.0 / (d_out + d_prev)))
        w2 = torch.randn(d_out, d_prev, generator=gen) * float(np.sqrt(2.0 / (d_out + d_prev)))
        b1 = torch.randn(d_out, generator=gen) * 0.01
        b2 = torch.randn(d_out, generator=gen) * 0.01
        params.append((w1, b1, w2, b2))
        for nm, t in zip(("w1", "b1", "w2", "b2"), (w1, b1, w2, b2)):
            out[f"ngcf_{nm}_{li}"] = t.numpy()
        d_prev = d_out
    conv = lambda t: torch.from_numpy(C.spmm(rowptr, col, val, t.numpy()))  # noqa: E731
    out["bignn_conv0"] = O.bignn_conv(e0, conv, *params[0]).numpy()
    u_all, i_all = O.ngcf_forward(e0[:n_users], e0[n_users:], conv, params)
    out["ngcf_out"] = torch.cat([u_all, i_all]).numpy()
    # one SGL edge-drop view (sgl.py:107-126)
    keep_mask, (vrp, vcol, vval) = O.sgl_random_graph_augment(uid, iid, n_users, n_items, "ED", 0.1,
                                                              np.random.default_rng(seed + 1))
    out.update(sgl_keep=keep_mask, sgl_rowptr=vrp, sgl_col=vcol.astype(np.int32), sgl_val=vval)
    return out


def main():
    uid, iid, n_users, n_items = ref_interactions()
    assert len(uid) == 5999, len(uid)
    np.savez_compressed(os.path.join(HERE, "ref_test_inter.npz"), uid=uid, iid=iid, n_users=np.int64(n_users),
                        n_items=np.int64(n_items))
    print("ref test fixture:", n_users, "users", n_items, "items", len(uid), "interactions")
    np.savez_compressed(os.path.join(HERE, "golden_ref_test.npz"), **golden_for(uid, iid, n_users, n_items, 2020))
    tu, ti = synth_toy(n_users, n_items, len(uid))
    np.savez_compressed(os.path.join(HERE, "golden_toy.npz"), **golden_for(tu, ti, n_users, n_items, 2021))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
