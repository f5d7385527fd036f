#!/usr/bin/env python
"""Store-only twins of the scoring kernel with wider stores (store_twin.hip): dword / dwordx4 x 8 rows / dwordx4 x 4 rows,
plain and non-temporal, at the reference's row stride and at a line-aligned one.  JSON lines -> gpurun_out/store_twin.jsonl"""
import ctypes, json, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libstore_twin.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "store_twin.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mb_store_twin.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, n = 4096, 40982
out = open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "store_twin.jsonl"), "a")


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


for ld in (n, (n + 31) // 32 * 32):
    S = torch.empty(B * ld + 64, device=dev)
    S = S[(-(S.data_ptr() // 4)) % 32:][: B * ld]  # 128-byte aligned base
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tpw in (8, 32, 128):
        for mode in (0, 1, 2):
            for nt in (0, 1):
                us = timed(lambda: lib.mb_store_twin(ctypes.c_void_p(S.data_ptr()), B, n, ld, tpw, mode, nt, st))
                rec = dict(kind="store_twin", ld=ld, tiles_per_wave=tpw, mode=["dword_2rows", "x4_8rows", "x4_4rows_256B"][mode], nt=nt, us=round(us, 1),
                           GBps=round(B * n * 4 / us / 1e3))
                print(json.dumps(rec), flush=True); out.write(json.dumps(rec) + "\n")
    us = timed(lambda: S.fill_(1.0))
    rec = dict(kind="fill", ld=ld, us=round(us, 1), GBps=round(B * ld * 4 / us / 1e3))
    print(json.dumps(rec), flush=True); out.write(json.dumps(rec) + "\n")
