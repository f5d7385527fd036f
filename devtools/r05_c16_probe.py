#!/usr/bin/env python
"""r05: compact launches on 16-bit slab-row numbers (option sell_c16 = 1, default) against 32-bit offsets (0): propagation (K = 3),
plain layer and backward chain, us, HIP-graph replays; outputs compared bit for bit.  -> gpurun_out/r05_c16_probe.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
log = open(os.path.join(ROOT, "gpurun_out", "r05_c16_probe.jsonl"), "a")


def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]


for name in sys.argv[1:] or ["gowalla", "yelp2018"]:
    uid, iid, nu, ni = rbg.synth.make(name)
    n, d = nu + ni, 64
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    gen = torch.Generator().manual_seed(1)
    uwd, iwd = torch.randn(nu, d, generator=gen).to(dev), torch.randn(ni, d, generator=gen).to(dev)
    o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
    fwd = lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
    outs = {}
    for rep in range(3):
        for c16 in (1, 0):
            rbg.set_option("sell_c16", c16)
            fwd(); torch.cuda.synchronize()
            outs[c16] = o.clone()
            rec = {"what": "sell_c16", "workload": name, "sell_c16": c16, "prop_us": round(timeit(fwd), 2)}
            print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
    rbg.set_option("sell_c16", 1)
    rec = {"what": "sell_c16", "workload": name, "bit_identical": bool(torch.equal(outs[0], outs[1]))}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
