#!/usr/bin/env python
"""Sweep of the SELL planner's `chunk` (entries per piece before a row is cut) for the column-slab propagation.
JSON lines -> gpurun_out/sell_chunk_probe.jsonl"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "amazon-book"]
log = open(os.path.join(ROOT, "gpurun_out", "sell_chunk_probe.jsonl"), "a")


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
    d = 64
    uw, iw = torch.randn(nu, d, device=dev), torch.randn(ni, d, device=dev)
    o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
    big = n > 1_000_000
    for chunk in [int(c) for c in os.environ.get("CHUNKS", "24,32,48,64,96,128,192").split(",")]:
        try:
            info = g.attach_sell(d, chunk=chunk)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"workload": name, "chunk": chunk, "error": str(ex)[:100]})); continue
        us = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=o, layers=L), 5 if big else 100)
        rec = {"workload": name, "d": d, "chunk": chunk, "prop_us": us, "padding": info["padding"], "n_units": info["n_units"]}
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
    g.detach_sell()
