#!/usr/bin/env python
"""Write-side ceiling of the scoring GEMM: store-only twin of score_kernel vs fill, aligned vs unaligned rows."""
import ctypes, json, os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmb.so"))
lib.mb_store_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, n = 4096, 40982


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for ld in (n, (n + 31) // 32 * 32):
    S = torch.empty(B * ld, device=dev)
    for tpw in (2, 8, 32, 128):
        us = timed(lambda: lib.mb_store_tiles(ctypes.c_void_p(S.data_ptr()), B, n, ld, tpw, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        print(json.dumps(dict(kind="store_tiles", B=B, n=n, ld=ld, tiles_per_wave=tpw, us=round(us, 1), GBps=round(B * n * 4 / us / 1e3))), flush=True)
    us = timed(lambda: S.fill_(1.0))
    print(json.dumps(dict(kind="fill", floats=B * ld, us=round(us, 1), GBps=round(B * ld * 4 / us / 1e3))), flush=True)
