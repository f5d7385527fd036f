#!/usr/bin/env python
"""Experiment: does sweeping the table in the same order on every lane-group (loose lock-step) raise the L2 hit
rate of random row gathers?  And how much does halving the table (user/item XCD specialisation) buy?"""
import ctypes, json, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(HERE, "libmb.so"))
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.mb_gather.argtypes = [vp, vp, i64, vp, ci, ci, vp]
log = open(os.path.join(ROOT, "gpurun_out", "microbench2.jsonl"), "a")
def emit(**kw):
    s = json.dumps(kw); print(s, flush=True); log.write(s + "\n"); log.flush()
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
out = torch.zeros(1 << 22, device=dev)
st = vp(torch.cuda.current_stream().cuda_stream)
m = 2_054_740
rng = np.random.default_rng(0)
for rows in (40982, 70841, 1300000):
    tab = torch.ones(rows, 64, device=dev)
    w = (np.arange(rows) + 10.0) ** -0.75; c = np.cumsum(w); c /= c[-1]
    base = np.searchsorted(c, rng.random(m)).clip(0, rows - 1)
    for per_group in (32, 128, 512, 1024, 4096):
        for mode in ("random", "sorted_per_group"):
            idx = base.copy()
            if mode == "sorted_per_group":
                pad = (-len(idx)) % per_group
                tmp = np.concatenate([idx, np.full(pad, rows - 1)]).reshape(-1, per_group)
                tmp.sort(axis=1)
                idx = tmp.reshape(-1)[:m]
            idx_t = torch.from_numpy(idx.astype(np.int32)).to(dev)
            us = timeit(lambda: lib.mb_gather(vp(tab.data_ptr()), vp(idx_t.data_ptr()), m, vp(out.data_ptr()), per_group, 8, st))
            groups = (m + per_group - 1) // per_group
            emit(kind="gather2", rows=rows, per_group=per_group, mode=mode, us=round(us, 1), blocks=(groups + 15) // 16,
                 gather_gbps=round(m * 256 / (us * 1e-6) / 1e9))
    del tab
