#!/usr/bin/env python
"""r04: the launch forms of the column-slab kernel (options sell_depth x sell_first x sell_class_serial) on every quoted shape:
propagation (K = 3) / plain layer / backward chain in us (HIP-graph replay), error against the C oracle; the native planner's
time against the torch specification; the NGCF forward over the plan.  JSON lines -> gpurun_out/r04_probe.jsonl"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "yelp2018", "amazon-book", "g-1.3m"]
dims = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["64"])]
forms = [(1, 0), (2, 0), (1, 1), (2, 1)]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "r04_probe.jsonl"), "a")


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


for name in shapes:
    uid, iid, nu, ni = rbg.synth.make(name)
    n = nu + ni
    rbg.set_option("sell_auto", 0)
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    rbg.set_option("sell_auto", 1)
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

        def bwd():
            rbg._lib.check(rbg._lib.lib.rbg_lightgcn_backward_f32(arr, 1, ctypes.c_void_p(gout.data_ptr()), ctypes.c_void_p(ge0.data_ptr()),
                                                                   ctypes.c_void_p(work.data_ptr()), d, 3,
                                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        rec = {"workload": name, "d": d, "nodes": n, "nnz": g.nnz}
        # planners
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.attach_sell(d, planner="spec")
        torch.cuda.synchronize(); rec["plan_ms_spec_torch"] = (time.perf_counter() - t0) * 1e3
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rec["plan"] = g.plan_sell()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        rec["plan_ms_native"] = sorted(ts)[1]
        rbg.set_option("sell", 0)
        rec["prop_us_binned"] = timeit(fwd, iters)
        rbg.set_option("sell", 1)
        for serial in ((0, 1) if n > 100_000 else (0,)):
            rbg.set_option("sell_class_serial", serial)
            for depth, first in forms:
                rbg.set_option("sell_depth", depth)
                rbg.set_option("sell_first", first)
                key = f"d{depth}f{first}s{serial}"
                o.fill_(7.0)
                fwd(); torch.cuda.synchronize()
                rec[f"err_{key}"] = float(np.abs(o.cpu().numpy() - ref).max())
                rec[f"prop_us_{key}"] = timeit(fwd, iters)
                rec[f"spmm_us_{key}"] = timeit(lambda: rbg.ops.spmm_raw(g, xx, out=yy), iters)
                rec[f"bwd_us_{key}"] = timeit(bwd, iters)
                rec[f"kernel_{key}"] = g.propagation_kernel_name(d)
        rbg.set_option("sell_class_serial", -1)
        rbg.set_option("sell_depth", 1)
        rbg.set_option("sell_first", 0)
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
    if name == "yelp2018":  # NGCF (config #3): the fused inference forward and one training layer over the plan vs the binned kernel
        ds = rbg.InteractionDataset(uid, iid, nu, ni)
        torch.manual_seed(0)
        model = rbg.NGCF({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "hidden_size_list": [64, 64, 64],
                          "node_dropout": 0.0, "message_dropout": 0.0, "reg_weight": 1e-5}, ds)
        model.eval()
        rec = {"workload": name, "what": "ngcf_forward", "sell_status": model.graph.sell_status()}
        with torch.no_grad():
            for sell in (1, 0):
                rbg.set_option("sell", sell)
                rec[f"ngcf_forward_us_sell{sell}"] = timeit(lambda: model.forward(), 50)
                for depth, first in (forms if sell else ()):
                    rbg.set_option("sell_depth", depth); rbg.set_option("sell_first", first)
                    rec[f"ngcf_forward_us_d{depth}f{first}"] = timeit(lambda: model.forward(), 50)
                rbg.set_option("sell_depth", 1); rbg.set_option("sell_first", 0)
        rbg.set_option("sell", 1)
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
    del g
    torch.cuda.empty_cache()
