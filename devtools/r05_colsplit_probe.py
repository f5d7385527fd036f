#!/usr/bin/env python
"""r05 / VERDICT r04 #1a: does cutting the gathered table by COLUMN RANGE pay at the large shapes?  Y = A X as two launches
Y = A_0 X; Y += A_1 X where A_q keeps the entries whose column falls in part q of its class's columns (hot / cold by degree, or
two random halves of equal weight): every launch gathers from half a table (its live set per L2 halves), the second one reads and
rewrites Y.  Existing product calls only (two planned handles from CSR, the accumulate epilogue); µs by HIP-graph replay, error
against the one-launch product.  JSON lines -> gpurun_out/r05_colsplit.jsonl"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "amazon-book", "g-1.3m"]
log = open(os.path.join(ROOT, "gpurun_out", "r05_colsplit.jsonl"), "a")


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
    n, d = nu + ni, 64
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    rowptr, col, val = np.asarray(rowptr, np.int64), np.asarray(col, np.int64), np.asarray(val, np.float32)
    deg = np.diff(rowptr)
    rows = np.repeat(np.arange(n), deg)
    g = rbg.GraphHandle.from_csr(rowptr, col, val, n, device=dev, n_class0_rows=nu)
    x, y = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
    iters = 10 if n > 1_000_000 else 50
    rbg.ops.spmm_raw(g, x, out=y); torch.cuda.synchronize()
    y_ref = y.clone()
    rec = {"workload": name, "nodes": n, "nnz": int(rowptr[-1]), "one_launch_us": round(timeit(lambda: rbg.ops.spmm_raw(g, x, out=y), iters), 1),
           "kernel": g.spmm_kernel_name(d), "status": g.sell_status()}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
    rng = np.random.default_rng(0)
    for how in ("hot_half_of_columns", "hot_quarter_of_columns", "random_half"):
        part = np.zeros(n, dtype=np.int8)  # part of every COLUMN node
        for lo, hi in ((0, nu), (nu, n)):
            order = lo + np.argsort(-deg[lo:hi], kind="stable")
            if how == "hot_half_of_columns":
                part[order[(hi - lo) // 2:]] = 1
            elif how == "hot_quarter_of_columns":
                part[order[(hi - lo) // 4:]] = 1
            else:
                part[lo + rng.permutation(hi - lo)[(hi - lo) // 2:]] = 1
        hs = []
        for q in (0, 1):
            m = part[col] == q
            rp = np.zeros(n + 1, dtype=np.int64)
            np.add.at(rp, rows[m] + 1, 1)
            hs.append(rbg.GraphHandle.from_csr(np.cumsum(rp), col[m], val[m], n, device=dev, n_class0_rows=nu))

        def two():
            rbg.ops.spmm_raw(hs[0], x, out=y)
            rbg.ops.spmm_raw(hs[1], x, out=y, accumulate=True)

        two(); torch.cuda.synchronize()
        err = float((y - y_ref).abs().max())
        rec = {"workload": name, "split": how, "entries": [int(h.nnz) for h in hs], "two_launch_us": round(timeit(two, iters), 1),
               "first_us": round(timeit(lambda: rbg.ops.spmm_raw(hs[0], x, out=y), iters), 1),
               "second_us": round(timeit(lambda: rbg.ops.spmm_raw(hs[1], x, out=y, accumulate=True), iters), 1),
               "max_diff_vs_one_launch": err, "kernels": [h.spmm_kernel_name(d) for h in hs]}
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
        del hs
    del g
    torch.cuda.empty_cache()
