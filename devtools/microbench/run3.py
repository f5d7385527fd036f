#!/usr/bin/env python
"""Per-XCD L2 capacity curve: every XCD re-reads its own private region of S MB."""
import ctypes, json, os
import torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(HERE, "libmb.so"))
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.mb_xcd_private.argtypes = [vp, i64, ci, ci, vp, vp]
log = open(os.path.join(ROOT, "gpurun_out", "microbench3.jsonl"), "a")
dev = torch.device("cuda:0"); out = torch.zeros(16, device=dev); st = vp(torch.cuda.current_stream().cuda_stream)
for mb in (0.5, 1, 2, 3, 3.5, 4, 5, 6, 8, 12, 16, 32):
    floats = int(mb * (1 << 20)) // 4
    buf = torch.ones(8 * floats, device=dev)
    reps = max(2, int(256 / mb))
    for blocks in (2048,):
        for _ in range(2): lib.mb_xcd_private(vp(buf.data_ptr()), floats, reps, blocks, vp(out.data_ptr()), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): lib.mb_xcd_private(vp(buf.data_ptr()), floats, reps, blocks, vp(out.data_ptr()), st)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 5
        r = dict(kind="xcd_private", mb_per_xcd=mb, reps=reps, blocks=blocks, us=round(us, 1), gbps=round(8 * floats * 4 * reps / (us * 1e-6) / 1e9))
        print(json.dumps(r), flush=True); log.write(json.dumps(r) + "\n")
    del buf
