#!/usr/bin/env python
"""[HISTORICAL: the library option this probe drives (`sell_skip_hot`) was removed with the experiment; kept as the harness that
produced profiles/r04_*_probe.jsonl — it does not run against the current library.]
Upper bound of a hot-row tile: the propagation with the gathers of the plan's first T slab rows (its numbering is degree-
descending: the hottest rows) reading nothing (option "sell_skip_hot": WRONG results, timing only)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")

def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]

for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "amazon-book"]):
    uid, iid, nu, ni = rbg.synth.make(name)
    n, d = nu + ni, 64
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    deg = np.bincount(np.concatenate([uid, iid + nu]), minlength=n)
    du, di = np.sort(deg[:nu])[::-1], np.sort(deg[nu:])[::-1]
    uw, iw = torch.randn(nu, d, device=dev), torch.randn(ni, d, device=dev)
    o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
    rec = {"workload": name}
    for T in (0, 256, 512, 1024, 2048, 4096, 1 << 20):
        rbg.set_option("sell_skip_hot", T)
        share = (du[:T].sum() + di[:T].sum()) / deg.sum()
        rec[f"T{T}"] = {"prop_us": timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=o, layers=L), 100 if n < 500000 else 10),
                        "share_of_gathers_skipped_in_2_of_3_layers": float(share)}
    rbg.set_option("sell_skip_hot", 0)
    print(json.dumps(rec), flush=True)
