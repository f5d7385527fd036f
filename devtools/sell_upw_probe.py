#!/usr/bin/env python
"""Units per wave of the column-slab kernel (option "sell_units_per_wave": fewer, longer waves against the dispatcher's
4 100 waves/us).  JSON lines -> gpurun_out/sell_upw_probe.jsonl"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "amazon-book"]
log = open(os.path.join(ROOT, "gpurun_out", "sell_upw_probe.jsonl"), "a")


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]


for name in shapes:
    uid, iid, nu, ni = rbg.synth.make(name)
    n = nu + ni
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    for d in (64, 128):
        gen = torch.Generator().manual_seed(1)
        uw, iw = torch.randn(nu, d, generator=gen), torch.randn(ni, d, generator=gen)
        ref = coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), 3)
        uwd, iwd = uw.to(dev), iw.to(dev)
        o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
        g.attach_sell(d)
        for upw in [int(c) for c in os.environ.get("UPW", "1,2,3,4,6,8").split(",")]:
            rbg.set_option("sell_units_per_wave", upw)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
            err = float(np.abs(o.cpu().numpy() - ref).max())
            us = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 100)
            rec = {"workload": name, "d": d, "units_per_wave": upw, "prop_us": us, "err": err}
            print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
        rbg.set_option("sell_units_per_wave", 1)
        for nt in [int(c) for c in os.environ.get("NT", "").split(",") if c]:  # non-temporal epilogue accesses (option "sell_nt")
            rbg.set_option("sell_nt", nt)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
            err = float(np.abs(o.cpu().numpy() - ref).max())
            us = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 100)
            rec = {"workload": name, "d": d, "sell_nt": nt, "prop_us": us, "err": err}
            print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
        rbg.set_option("sell_nt", 0)
        g.detach_sell()
