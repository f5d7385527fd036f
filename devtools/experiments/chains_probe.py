#!/usr/bin/env python
"""[HISTORICAL: the library option this probe drives (`sell_two_chains`) was removed with the experiment; kept as the harness that
produced profiles/r04_*_probe.jsonl — it does not run against the current library.]
Option "sell_two_chains": the K layers as two per-class launch chains on two streams vs K launches; us, HIP-graph replay."""
import ctypes, json, os, sys
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

def eager(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "yelp2018", "amazon-book"]):
    uid, iid, nu, ni = rbg.synth.make(name)
    n = nu + ni
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    for d in (64, 128):
        uw, iw = torch.randn(nu, d, device=dev), torch.randn(ni, d, device=dev)
        o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
        gout, ge0, work = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
        arr = (ctypes.c_void_p * 1)(g.ptr)
        def fwd():
            rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=o, layers=L)
        def bwd():
            rbg._lib.check(rbg._lib.lib.rbg_lightgcn_backward_f32(arr, 1, ctypes.c_void_p(gout.data_ptr()), ctypes.c_void_p(ge0.data_ptr()),
                                                                   ctypes.c_void_p(work.data_ptr()), d, 3,
                                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        rec = {"workload": name, "d": d}
        ref = refb = None
        for mode in (0, 1, 2, 0, 1, 2):
            rbg.set_option("sell_two_chains", mode)
            fwd(); bwd(); torch.cuda.synchronize()
            if mode == 0:
                ref, refb = o.clone(), ge0.clone()
            else:
                rec[f"equal{mode}"] = bool(torch.equal(o, ref)) and bool(torch.equal(ge0, refb))
            rec.setdefault(f"fwd_us_mode{mode}", []).append(round(timeit(fwd, 100 if n < 500000 else 10), 2))
            rec.setdefault(f"bwd_us_mode{mode}", []).append(round(timeit(bwd, 100 if n < 500000 else 10), 2))
        rec["fwd_eager_us"] = {}
        for mode in (0, 1, 2):
            rbg.set_option("sell_two_chains", mode)
            rec["fwd_eager_us"][mode] = round(eager(fwd), 2)
        rbg.set_option("sell_two_chains", 0)
        print(json.dumps(rec), flush=True)
