#!/usr/bin/env python
"""r05: raised wave priority until the first gather batch is out (switch 16 of the RBG_SELL_TRACE build turns it off): propagation
(K = 3) and plain layer, us, HIP-graph replays.  -> gpurun_out/r05_sell_prio.jsonl"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_selltrace.so")
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
lib = rbg._lib.lib
log = open(os.path.join(ROOT, "gpurun_out", "r05_sell_prio.jsonl"), "a")


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


for name in sys.argv[1:] or ["gowalla"]:
    uid, iid, nu, ni = rbg.synth.make(name)
    n, d = nu + ni, 64
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    gen = torch.Generator().manual_seed(1)
    uwd, iwd = torch.randn(nu, d, generator=gen).to(dev), torch.randn(ni, d, generator=gen).to(dev)
    o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
    xx, yy = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
    fwd = lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
    lay = lambda: rbg.ops.spmm_raw(g, xx, out=yy)
    fwd(); lay()
    for rep in range(2):
        for bits in (0, 16, 32, 64):
            assert lib.mb_sell_debug_set(bits) == 0
            rec = {"what": "sell priority", "workload": name, "form": {0: "raised until the first batch is out", 16: "none", 32: "raised during the gathers", 64: "raised for the epilogue"}[bits], "prop_us": round(timeit(fwd), 1), "layer_us": round(timeit(lay), 1)}
            print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
    lib.mb_sell_debug_set(0)
