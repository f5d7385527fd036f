#!/usr/bin/env python
"""r06: the fused top-k with and without the plane image (option "topk_image"): time per call (HIP graph replay of 10 calls),
results compared bit for bit.  4096 users x the Gowalla item table, d = 64 (and d = 128), k = 10; fresh LightGCN tables."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")


def replay_us(fn, calls=10, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            keep = fn()
    out = []
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / (3 * calls))
    del keep
    return sorted(out)[len(out) // 2]


uid, iid, nu, ni = rbg.synth.make("gowalla")
for d in (64, 128):
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    torch.manual_seed(0)
    model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": 3}, ds)
    with torch.no_grad():
        ue, ie = model.forward()
    for nb in (4096, 1024):
        users = torch.randint(1, nu, (nb,), generator=torch.Generator().manual_seed(1)).to(dev)
        rec = {"d": d, "users": nb, "k": 10}
        res = {}
        for img in (0, 1, 0, 1):
            rbg.set_option("topk_image", 2 * img)
            v, i = rbg.full_sort_topk(model.graph, ue, ie, users, 10)
            res[img] = (v.clone(), i.clone())
            rec.setdefault(f"image{img}_us", []).append(round(replay_us(lambda: rbg.full_sort_topk(model.graph, ue, ie, users, 10)), 2))
        rec["bit_identical"] = bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
        print(json.dumps(rec), flush=True)
rbg.set_option("topk_image", 1)
