#!/usr/bin/env python
"""Store-only twins, second set (store_twin2.hip): k lines of a row back to back, the kernel's order, and the four waves of a
workgroup on adjacent pieces of the same rows.  JSON lines -> gpurun_out/r05_store_twin2.jsonl"""
import ctypes, json, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libstore_twin2.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "store_twin2.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mb_store_twin2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
B, n = 4096, 40982
out = open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "r05_store_twin2.jsonl"), "a")


def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]


ld = n
S = torch.empty(B * ld + 64, device=dev)
S = S[(-(S.data_ptr() // 4)) % 32:][: B * ld]  # 128-byte aligned base
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for layout, walks in ((0, (56, 224, 1281)), (1, (224, 1281))):
    for tpw in walks:
        for k in (1, 2, 4, 8, 16):
            for order in (0, 1):
                if k == 1 and order == 1: continue
                for nt in (0, 1):
                    us = timed(lambda: lib.mb_store_twin2(ctypes.c_void_p(S.data_ptr()), B, n, ld, tpw, k, order, layout, nt, st))
                    rec = dict(kind="store_twin2", layout=layout, tiles_per_walk=tpw, k=k, order=["rows_outer", "tiles_outer"][order], nt=nt,
                               us=round(us, 1), GBps=round(B * n * 4 / us / 1e3))
                    print(json.dumps(rec), flush=True); out.write(json.dumps(rec) + "\n")
us = timed(lambda: S.fill_(1.0))
rec = dict(kind="fill", us=round(us, 1), GBps=round(B * ld * 4 / us / 1e3))
print(json.dumps(rec), flush=True); out.write(json.dumps(rec) + "\n")
