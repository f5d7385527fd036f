#!/usr/bin/env python
"""[HISTORICAL: the library option this probe drives (`sell_persist`) was removed with the experiment; kept as the harness that
produced profiles/r04_*_probe.jsonl — it does not run against the current library.]
The persistent K-layer launch (option "sell_persist") against the K launches: propagation and backward chain, us."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
MASKS = [int(m) for m in os.environ.get("FENCE_MASKS", "").split(",") if m]

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
        ref = None
        for persist in [0, 1] + [1 + 16 * m for m in MASKS]:
            rbg.set_option("sell_persist", persist)
            fwd(); torch.cuda.synchronize()
            if persist == 0:
                ref = o.clone()
            else:
                rec["equal"] = bool(torch.equal(o, ref))
            rec.setdefault(f"fwd_us_persist{persist}", []).append(timeit(fwd, 100 if n < 500000 else 10))
            rec.setdefault(f"bwd_us_persist{persist}", []).append(timeit(bwd, 100 if n < 500000 else 10))
        rbg.set_option("sell_persist", 0)
        print(json.dumps(rec), flush=True)
