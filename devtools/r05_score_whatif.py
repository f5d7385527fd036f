#!/usr/bin/env python
"""r05: what each part of score_uni_kernel costs — the RBG_SCORE_TRACE build's what-if switches (results are wrong on purpose):
1 = no product, 2 = no stores, 4 = no fetch / publish after the first tile.  4096 x 40 982 x 64.  -> gpurun_out/r05_score_whatif.jsonl"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_scoretrace.so")
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg

dev = torch.device("cuda:0")
lib = rbg._lib.lib
log = open(os.path.join(ROOT, "gpurun_out", "r05_score_whatif.jsonl"), "a")
B, n, d = 4096, 40982, 64
u, it = torch.randn(B, d, device=dev), torch.randn(n, d, device=dev)
out = torch.empty(B, n, device=dev)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]


import ctypes
vp = ctypes.c_void_p
stream = vp(torch.cuda.current_stream().cuda_stream)
sc = lambda: rbg._lib.check(lib.rbg_score_f32(vp(u.data_ptr()), d, vp(it.data_ptr()), d, vp(out.data_ptr()), B, n, d, stream))
for bits in (0, 1, 2, 4, 3, 5, 6, 7, 0):
    assert lib.mb_score_debug_set(bits) == 0
    rec = {"what": "score_whatif", "bits": bits, "us": round(timeit(sc), 1)}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
lib.mb_score_debug_set(0)
rec = {"what": "fill_ of the same matrix", "us": round(timeit(lambda: out.fill_(1.0)), 1)}
print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
