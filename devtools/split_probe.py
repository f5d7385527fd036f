#!/usr/bin/env python
"""rbg_score_f32 / rbg_full_sort_topk_f32 with the operands split into three bf16 terms (option mfma_split = 1, default)
vs the exact-fp32 MFMA chain (0) vs rocBLAS (torch.matmul): time and error against float64."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")

def time_us(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(out)[1]

g = torch.Generator().manual_seed(0)
for (B, n, d) in [(4096, 40982, 64), (1024, 40982, 64), (4096, 91600, 64), (4096, 40982, 128), (4096, 40982, 256), (1024, 38049, 256)]:
    u, it = torch.randn(B, d, generator=g).to(dev), torch.randn(n, d, generator=g).to(dev)
    ref = (u[:64].double().cpu() @ it.double().cpu().T)
    rec = dict(kind="score", B=B, n=n, d=d)
    for rnd in range(3):  # interleaved rounds: the first thing timed after a shape change runs ~15 % slow (clock ramp)
        for split, tag in ((1, "split"), (0, "fp32")):
            rbg.set_option("mfma_split", split)
            s = rbg.score(u, it)
            rec[f"rel_err_{tag}"] = float((s[:64].double().cpu() - ref).abs().max() / ref.abs().max())
            us = time_us(lambda: rbg.score(u, it))
            rec[f"us_{tag}"] = min(us, rec.get(f"us_{tag}", 1e30))
        rec["us_rocblas"] = min(time_us(lambda: u @ it.T), rec.get("us_rocblas", 1e30))
    rec["rel_err_rocblas"] = float(((u[:64] @ it.T).double().cpu() - ref).abs().max() / ref.abs().max())
    print(json.dumps(rec), flush=True)
rbg.set_option("mfma_split", 1)
uid, iid, nu, ni = rbg.synth.make("gowalla")
graph = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
for d in (64, 256):
    ua, ia = torch.randn(nu, d, generator=g).to(dev), torch.randn(ni, d, generator=g).to(dev)
    for B in (128, 1024, 4096):
        users = torch.randint(1, nu, (B,), generator=g).to(dev)
        rec = dict(kind="topk", B=B, d=d, k=10)
        outs = {}
        for split in (1, 0):
            rbg.set_option("mfma_split", split)
            outs[split] = rbg.full_sort_topk(graph, ua, ia, users, 10)
            rec[f"us_split{split}"] = time_us(lambda: rbg.full_sort_topk(graph, ua, ia, users, 10), iters=10, warm=2)
        rec["same_items"] = float((outs[0][1] == outs[1][1]).float().mean())
        rec["max_val_diff"] = float((outs[0][0] - outs[1][0]).abs().max())
        print(json.dumps(rec), flush=True)
rbg.set_option("mfma_split", 1)
