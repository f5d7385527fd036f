#!/usr/bin/env python
"""BiGNN dense half alone (rbg_bignn_dense_f32) against the cap on its workgroups (option bignn_grid)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
be = rbg.sharded.HipBackend(dev)
def time_us(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(out)[1]
for name in ("gowalla", "amazon-book"):
    nu, ni, _ = rbg.synth.shape(name)
    n = nu + ni
    g = torch.Generator().manual_seed(0)
    p, x = torch.randn(n, 64, generator=g).to(dev), torch.randn(n, 64, generator=g).to(dev)
    w1, w2 = (torch.randn(64, 64, generator=g) * 0.1).to(dev), (torch.randn(64, 64, generator=g) * 0.1).to(dev)
    b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    y = torch.empty(n, 64, device=dev)
    rec = dict(kind="bignn_dense", shape=name, rows=n)
    ref = None
    for rnd in range(2):
        for cap in (0, 256, 512, 768, 1024, 1536):
            rbg.set_option("bignn_grid", cap)
            be.bignn_dense(p, x, w1, b1, w2, b2, y)
            if ref is None: ref = y.clone()
            assert torch.equal(ref, y)
            rec[f"us_cap{cap}"] = min(time_us(lambda: be.bignn_dense(p, x, w1, b1, w2, b2, y)), rec.get(f"us_cap{cap}", 1e30))
    rbg.set_option("bignn_grid", 0)
    print(json.dumps(rec), flush=True)
