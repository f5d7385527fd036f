#!/usr/bin/env python
"""r05: the resident-round launch (sell_stream.h) and the W = 16 plan against the r04 kernel, every quoted shape.
Per shape and d: propagation (K = 3) / plain layer / backward chain in us by HIP-graph replay; bit-identity of every stream form
with the one-wave-per-unit launch of the same plan; error against the C oracle.  JSON lines -> gpurun_out/r05_probe.jsonl
usage: r05_probe.py [shapes] [dims] [quick]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "yelp2018", "amazon-book", "g-1.3m"]
dims = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["64"])]
quick = len(sys.argv) > 3 and sys.argv[3] == "quick"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "r05_probe.jsonl"), "a")


def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]


def emit(rec):
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()


for name in shapes:
    uid, iid, nu, ni = rbg.synth.make(name)
    n = nu + ni
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    big = n > 1_000_000
    iters = 10 if big else 100
    for d in dims:
        if d == 128 and big:
            continue
        gen = torch.Generator().manual_seed(1)
        uw, iw = torch.randn(nu, d, generator=gen), torch.randn(ni, d, generator=gen)
        uwd, iwd = uw.to(dev), iw.to(dev)
        o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
        xx, yy = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
        gout, ge0, work = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
        arr = (ctypes.c_void_p * 1)(g.ptr)
        ref = coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), 3)

        def fwd():
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)

        def lay():
            rbg.ops.spmm_raw(g, xx, out=yy)

        def bwd():
            rbg._lib.check(rbg._lib.lib.rbg_lightgcn_backward_f32(arr, 1, ctypes.c_void_p(gout.data_ptr()), ctypes.c_void_p(ge0.data_ptr()),
                                                                   ctypes.c_void_p(work.data_ptr()), d, 3,
                                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        for W in (32,):  # (the W = 16 plan: 64-byte gathers cost the memory path a line slot each — 134 vs 92 us, session 1)
            rec = {"workload": name, "d": d, "nodes": n, "nnz": g.nnz, "W": W}
            rbg._lib.check(rbg._lib.lib.rbg_graph_plan_sell(g.ptr, W, 0))
            rec["plan"] = g.sell_info()
            rbg.set_option("sell_stream", 0)
            rec["kernel"] = g.propagation_kernel_name(d)
            o.fill_(7.0); yy.fill_(7.0); ge0.fill_(7.0)
            fwd(); lay(); bwd(); torch.cuda.synchronize()
            o0, y0, g0 = o.clone(), yy.clone(), ge0.clone()
            rec["err_vs_oracle"] = float(np.abs(o0.cpu().numpy() - ref).max())
            rec["prop_us_unit"] = timeit(fwd, iters)
            rec["spmm_us_unit"] = timeit(lay, iters)
            rec["bwd_us_unit"] = timeit(bwd, iters)
            emit(rec)
            forms = [(7, 0, 1)]
            for wgs, fit, sched in forms:
                rbg.set_option("sell_stream", 1)
                rbg.set_option("sell_stream_wgs", wgs)
                rbg.set_option("sell_stream_fit", fit)
                rbg.set_option("sell_stream_sched", sched)
                r2 = {"workload": name, "d": d, "W": W, "stream_wgs": wgs, "stream_fit": fit, "stream_sched": sched}
                o.fill_(7.0); yy.fill_(7.0); ge0.fill_(7.0)
                fwd(); lay(); bwd(); torch.cuda.synchronize()
                r2["bit_identical"] = [bool(torch.equal(o, o0)), bool(torch.equal(yy, y0)), bool(torch.equal(ge0, g0))]
                r2["max_diff"] = [float((o - o0).abs().max()), float((yy - y0).abs().max()), float((ge0 - g0).abs().max())]
                r2["prop_us"] = timeit(fwd, iters)
                r2["spmm_us"] = timeit(lay, iters)
                r2["bwd_us"] = timeit(bwd, iters)
                emit(r2)
            rbg.set_option("sell_stream", 0)
            rbg.set_option("sell_stream_sched", 1)
        rbg._lib.check(rbg._lib.lib.rbg_graph_plan_sell(g.ptr, 32, 0))
    del g
    torch.cuda.empty_cache()
