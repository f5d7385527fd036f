#!/usr/bin/env python
"""r06: randomised comparison of the screened top-k (option "topk_screen" 2: at every batch size) with the float64 reference
(scores, PAD and history masked, topk) over random shapes, widths, k, history densities and value distributions."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
rbg.set_option("topk_screen", 2)
rbg.set_option("topk_sample", 1024)
t_end = time.time() + budget
n_cases = n_rows = worst = 0
bad = []
while time.time() < t_end:
    d = int(rng.choice([8, 20, 33, 64, 64, 64, 100, 128]))
    ni = int(rng.integers(2100, 30000))
    nu = int(rng.integers(50, 4000))
    b = int(rng.choice([1, 7, 33, 64, 255, 256, 1000, 2500]))
    k = int(rng.choice([1, 5, 10, 20, 32]))
    dist = str(rng.choice(["normal", "cauchy", "popular", "lowrank", "scaled"]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    u, it = torch.randn((nu, d), generator=g), torch.randn((ni, d), generator=g)
    if dist == "cauchy":
        u, it = u / torch.randn((nu, d), generator=g).abs().clamp_min(1e-2), it / torch.randn((ni, d), generator=g).abs().clamp_min(1e-2)
    elif dist == "popular":
        it[: ni // 40] += 3.0 * u.mean(0) / u.mean(0).norm()
        u += 1.5 * u.mean(0) / u.mean(0).norm()
    elif dist == "lowrank":
        basis = torch.randn((3, d), generator=g)
        u, it = torch.randn((nu, 3), generator=g) @ basis, torch.randn((ni, 3), generator=g) @ basis
    elif dist == "scaled":
        u *= torch.logspace(-6, 6, nu)[:, None]
        it *= torch.logspace(-3, 3, ni)[torch.randperm(ni, generator=g)][:, None]
    users = torch.from_numpy(rng.integers(1, nu, b))
    hist = None
    uid = iid = np.zeros(0, dtype=np.int64)
    if rng.random() < 0.6:
        per = int(rng.integers(1, 60))
        uid = np.repeat(np.arange(1, nu), per)
        iid = rng.integers(1, ni, uid.shape[0])
        if rng.random() < 0.5:  # a hub whose history is a large part of the catalogue, and its best items
            hub = int(users[0])
            items = rng.choice(np.arange(1, ni), size=min(ni - 1, int(rng.integers(500, 6000))), replace=False)
            uid, iid = np.concatenate([uid, np.full(items.shape[0], hub)]), np.concatenate([iid, items])
            it[torch.from_numpy(items)] += 0.5 * u[hub]
        pairs = np.unique(np.stack([uid, iid], 1), axis=0)
        uid, iid = pairs[:, 0], pairs[:, 1]
        hist = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    ud, itd = u.to(dev), it.to(dev)
    vals, idx = rbg.full_sort_topk(hist, ud, itd, users.to(dev), k)
    sc = u[users].double() @ it.double().T
    sc[:, 0] = -np.inf
    if hist is not None:
        pos = {int(x): i for i, x in enumerate(users.tolist())}
        rows_of = {}
        for i, x in enumerate(users.tolist()):
            rows_of.setdefault(int(x), []).append(i)
        sel = np.isin(uid, users.numpy())
        for uu, ii in zip(uid[sel].tolist(), iid[sel].tolist()):
            for r in rows_of[uu]:
                sc[r, ii] = -np.inf
    rv, ri = torch.topk(sc, k, dim=1)
    vals_c, idx_c = vals.cpu().double(), idx.cpu()
    fin_sc = sc[torch.isfinite(sc)]
    scale = fin_sc.abs().max().clamp_min(1e-300) if fin_sc.numel() else torch.tensor(1.0, dtype=torch.float64)
    finite = torch.isfinite(rv)
    err = float(((vals_c - rv).abs()[finite] / scale).max()) if finite.any() else 0.0
    worst = max(worst, err)
    srt = torch.sort(sc, dim=1, descending=True).values
    gap_ok = (srt[:, k - 1] - srt[:, k]) > 1e-6 * scale
    wrong = 0
    for r in torch.nonzero(gap_ok).flatten().tolist():
        if set(idx_c[r].tolist()) != set(ri[r].tolist()):
            wrong += 1
    n_cases += 1
    n_rows += b
    if err > 2e-5 or wrong:
        bad.append({"d": d, "ni": ni, "nu": nu, "b": b, "k": k, "dist": dist, "history": hist is not None, "err": err, "wrong_rows": wrong})
    if hist is not None:
        hist.destroy()
print(json.dumps({"what": "screened top-k fuzz", "cases": n_cases, "rows": n_rows, "max_rel_err": worst, "bad": bad[:10], "n_bad": len(bad)}))
rbg.set_option("topk_screen", 1)
rbg.set_option("topk_sample", 8192)
