#!/usr/bin/env python
"""Dispatch-rate probe (dispatch.hip): time of kernels that do nothing, by grid shape and register budget."""
import ctypes, json, os, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(HERE, "libdispatch.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.mb_dispatch.argtypes = [vp, ci, ci, ci, ci, vp]
dev = torch.device("cuda:0"); out = torch.zeros(16, device=dev)
st = vp(torch.cuda.current_stream().cuda_stream)
log = open(os.path.join(ROOT, "gpurun_out", "dispatch_probe.jsonl"), "a")
def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for grid, block in ((256, 256), (2048, 256), (6000, 256), (12000, 256), (24000, 64), (3000, 512), (1500, 1024), (256, 1024), (24000, 256)):
    for vg in (4, 32, 60, 120, -16):
        for spin in (0, 8):
            us = timeit(lambda: lib.mb_dispatch(vp(out.data_ptr()), grid, block, vg, spin, st))
            rec = dict(kind="dispatch", grid=grid, block=block, waves=grid * block // 64, vgprs=vg, spin=spin, us=round(us, 2),
                       waves_per_us=round(grid * block / 64 / us, 1))
            print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
