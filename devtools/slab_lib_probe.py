#!/usr/bin/env python
"""Option "slab" of rbg_lightgcn_forward_f32 (layers kept as two column slabs, column-half kernel over contiguous half rows)
against the default path: propagation time and parity against the float64 oracle, several shapes / widths."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from oracle import coracle

dev = torch.device("cuda:0")
shapes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "yelp2018", "amazon-book", "g-1.3m"]
out_path = os.path.join(ROOT, "gpurun_out", "slab_lib_probe.jsonl")
log = open(out_path, "a")

def timeit(fn, iters):
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
    for d in (64, 128):
        if d == 128 and n > 1_000_000:
            continue
        gen = torch.Generator().manual_seed(1)
        uw, iw = torch.randn(nu, d, generator=gen), torch.randn(ni, d, generator=gen)
        ref = coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), 3)
        uwd, iwd = uw.to(dev), iw.to(dev)
        o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
        rec = {"workload": name, "d": d, "nodes": n, "nnz": g.nnz}
        for slab in (0, 1):
            rbg.set_option("slab", slab)
            rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
            torch.cuda.synchronize()
            rec[f"err_slab{slab}"] = float(np.abs(o.cpu().numpy() - ref).max())
            rec[f"prop_us_slab{slab}"] = timeit(lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L), 10 if n > 1_000_000 else 100)
            for k in (1, 2):  # other depths: parity only
                rbg.ops.lightgcn_forward_raw(g, uwd, iwd, k, out=o, layers=L[:k])
                torch.cuda.synchronize()
                rk = coracle.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), k)
                rec[f"err_k{k}_slab{slab}"] = float(np.abs(o.cpu().numpy() - rk).max())
        rbg.set_option("slab", 0)
        rec["speedup"] = rec["prop_us_slab0"] / rec["prop_us_slab1"]
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
    del g
