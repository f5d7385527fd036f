#!/usr/bin/env python
"""Times (and, under rocprofv3 --pmc, exposes) the two MFMA kernels: scoring GEMM and the BiGNN dense layer."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
def time_us(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
uid, iid, nu, ni = rbg.synth.make("gowalla")
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
n = nu + ni
u = torch.randn(4096, 64, device=dev); it = torch.randn(ni, 64, device=dev)
print(json.dumps(dict(kind="score", B=4096, us=time_us(lambda: rbg.score(u, it)), us_torch=time_us(lambda: torch.matmul(u, it.T)))))
u1 = torch.randn(128, 64, device=dev)
print(json.dumps(dict(kind="score", B=128, us=time_us(lambda: rbg.score(u1, it)), us_torch=time_us(lambda: torch.matmul(u1, it.T)))))
x = torch.randn(n, 64, device=dev)
w1, w2 = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, 64, device=dev) * 0.1
b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
yo = torch.empty(n, 64, device=dev)
t_layer = time_us(lambda: rbg.ops.bignn_conv_raw(g, x, w1, b1, w2, b2, out=yo, leaky_norm=True))
y = torch.empty_like(x)
t_spmm = time_us(lambda: rbg.ops.spmm_raw(g, x, out=y))
print(json.dumps(dict(kind="bignn_layer", us=t_layer, us_spmm=t_spmm, us_dense=t_layer - t_spmm)))
# fused full-sort evaluation vs the unfused path (score matrix -> mask -> torch.topk)
for B in (128, 1024, 4096):
    users = torch.randint(1, nu, (B,), device=dev)
    ua = torch.randn(nu, 64, device=dev)
    hist_rows = torch.from_numpy(np.asarray(uid)).to(dev); hist_cols = torch.from_numpy(np.asarray(iid)).to(dev)
    def unfused():
        s = rbg.score(rbg.gather_rows(ua, users), it)
        s[:, 0] = float("-inf")
        return torch.topk(s, 10, dim=1)
    t_unf = time_us(unfused, iters=10, warm=2)
    t_fus = time_us(lambda: rbg.full_sort_topk(g, ua, it, users, 10), iters=10, warm=2)
    print(json.dumps(dict(kind="full_sort_topk", B=B, k=10, us_fused=t_fus, us_score_plus_torch_topk_no_history_mask=t_unf,
                          users_per_s=B / (t_fus * 1e-6))))
