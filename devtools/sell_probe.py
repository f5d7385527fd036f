#!/usr/bin/env python
"""The column-slab propagation in the library (csrc/sell.hip, option "sell") against the binned path: plan time, propagation
time and parity against the C oracle, several shapes and widths.  JSON lines -> gpurun_out/sell_probe.jsonl"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["toy", "ml-100k", "gowalla", "yelp2018", "amazon-book", "g-1.3m"]
dims = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["64", "128"])]
log = open(os.path.join(ROOT, "gpurun_out", "sell_probe.jsonl"), "a")

def timeit(fn, iters):
    """GPU time per call: `iters` calls captured into one HIP graph and replayed (the Python wrapper costs ~110 us per call,
    more than the slab propagation itself at the Gowalla shape)."""
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


def timeit_eager(fn, iters):
    for _ in range(max(3, iters // 5)): fn()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]

for name in shapes:
    uid, iid, nu, ni = rbg.synth.make(name)
    n = nu + ni
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    for d in dims:
        if d == 128 and n > 1_000_000:
            continue
        gen = torch.Generator().manual_seed(1)
        uw, iw = torch.randn(nu, d, generator=gen), torch.randn(ni, d, generator=gen)
        uwd, iwd = uw.to(dev), iw.to(dev)
        o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
        rec = {"workload": name, "d": d, "nodes": n, "nnz": g.nnz}
        rbg.set_option("sell", 0)
        rec["kernel_binned"] = g.propagation_kernel_name(d)
        big = n > 1_000_000
        rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
        ref = coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), 3)
        rec["err_binned"] = float(np.abs(o.cpu().numpy() - ref).max())
        rec["prop_us_binned"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
        rbg.set_option("sell", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec["plan"] = g.attach_sell(d)
        torch.cuda.synchronize(); rec["plan_ms"] = (time.perf_counter() - t0) * 1e3
        rec["kernel_sell"] = g.propagation_kernel_name(d)
        for k in (1, 2, 3):
            o.fill_(7.0)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, k, out=o, layers=L[:k]); torch.cuda.synchronize()
            rk = ref if k == 3 else coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), k)
            rec[f"err_sell_k{k}"] = float(np.abs(o.cpu().numpy() - rk).max())
        a = o.clone()
        rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
        rec["bit_stable"] = bool(torch.equal(a, o))
        rec["prop_us_sell"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
        rec["prop_us_sell_host_issued"] = timeit_eager(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
        rec["speedup"] = rec["prop_us_binned"] / rec["prop_us_sell"]
        # the plain layer (rbg_spmm_f32) and the keep_layers chain over the plan against the binned kernel
        xx, yy = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
        for rm in (0, 1):
            rbg.set_option("sell_rowmajor", rm)
            rec[f"spmm_us_rm{rm}"] = timeit(lambda: rbg.ops.spmm_raw(g, xx, out=yy), 10 if big else 100)
            rec[f"spmm_kernel_rm{rm}"] = g.spmm_kernel_name(d)
            rec[f"keep_layers_us_rm{rm}"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, keep_layers=True, out=o, layers=L), 10 if big else 100)
        # the factored chain (compact entries after the first launch) against the valued one
        for fac in (0, 1):
            rbg.set_option("sell_factored", fac)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
            rec[f"err_fac{fac}"] = float(np.abs(o.cpu().numpy() - ref).max())
            rec[f"prop_us_fac{fac}"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
        # E0 converted to slabs first (option "sell_rowmajor" = 0) against gathered where it lies; the backward chain alike
        gout, ge0, work = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
        arr = (ctypes.c_void_p * 1)(g.ptr)
        def bwd():
            rbg._lib.check(rbg._lib.lib.rbg_lightgcn_backward_f32(arr, 1, ctypes.c_void_p(gout.data_ptr()), ctypes.c_void_p(ge0.data_ptr()),
                                                                   ctypes.c_void_p(work.data_ptr()), d, 3,
                                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        for rm in (0, 1):
            rbg.set_option("sell_rowmajor", rm)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L); torch.cuda.synchronize()
            rec[f"err_rm{rm}"] = float(np.abs(o.cpu().numpy() - ref).max())
            rec[f"prop_us_sell_rm{rm}"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
            bwd(); torch.cuda.synchronize()
            rec[f"bwd_us_sell_rm{rm}"] = timeit(bwd, 10 if big else 100)
        if d == 128:  # two slabs of 64 (a W = 64 plan) against the four slabs of 32 measured above
            g.attach_sell(128, W=64)
            rec["kernel_w64"] = g.propagation_kernel_name(d)
            rec["prop_us_sell_w64"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if big else 100)
            rec["spmm_us_w64"] = timeit(lambda: rbg.ops.spmm_raw(g, xx, out=yy), 10 if big else 100)
        g.detach_sell()
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
    del g
