#!/usr/bin/env python
"""Speed-of-light calibration for the SpMM access pattern (see mb.hip).  JSON lines -> gpurun_out/microbench.jsonl"""
import ctypes, json, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
so = os.path.join(HERE, "libmb.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "mb.hip"), "-o", so])
lib = ctypes.CDLL(so)
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.mb_gather.argtypes = [vp, vp, i64, vp, ci, ci, vp]
lib.mb_stream.argtypes = [vp, i64, ci, ci, vp, vp]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "microbench.jsonl"), "a")
def emit(**kw):
    s = json.dumps(kw); print(s, flush=True); log.write(s + "\n"); log.flush()
dev = torch.device("cuda:0")
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
out = torch.zeros(1 << 22, device=dev)
st = vp(torch.cuda.current_stream().cuda_stream)
# streaming reads: buffer sizes spanning L2 (4 MB/XCD), Infinity Cache (256 MB), HBM
for mb in (2, 16, 64, 192, 1024, 4096):
    buf = torch.ones(mb * (1 << 20) // 4, device=dev)
    reps = max(1, 2048 // mb)
    for blocks in (2048, 8192):
        us = timeit(lambda: lib.mb_stream(vp(buf.data_ptr()), buf.numel(), reps, blocks, vp(out.data_ptr()), st), iters=10, warm=2)
        emit(kind="stream", mb=mb, reps=reps, blocks=blocks, us=us, gbps=mb * (1 << 20) * reps / (us * 1e-6) / 1e9)
    del buf
# random row gathers: M = 2M gathers of 256 B from a table of R rows
m = 2_054_740
rng = np.random.default_rng(0)
for rows in (4096, 16384, 70841, 262144, 1300000, 8000000):
    tab = torch.ones(rows, 64, device=dev)
    for dist_name in ("uniform", "powerlaw"):
        if dist_name == "uniform":
            idx = rng.integers(0, rows, m)
        else:
            w = (np.arange(rows) + 10.0) ** -0.75; c = np.cumsum(w); c /= c[-1]
            idx = np.searchsorted(c, rng.random(m)).clip(0, rows - 1)
        idx_t = torch.from_numpy(idx.astype(np.int32)).to(dev)
        for per_group in (32, 128):
            for unroll in (4, 8, 16):
                us = timeit(lambda: lib.mb_gather(vp(tab.data_ptr()), vp(idx_t.data_ptr()), m, vp(out.data_ptr()), per_group, unroll, st), iters=30, warm=3)
                emit(kind="gather", rows=rows, table_mb=rows * 256 / 1e6, dist=dist_name, per_group=per_group, unroll=unroll, us=us,
                     gather_gbps=m * 256 / (us * 1e-6) / 1e9)
    del tab
