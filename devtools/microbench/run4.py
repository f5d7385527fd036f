#!/usr/bin/env python
"""Does fetching cold rows with non-temporal loads protect the hot rows in L2?"""
import ctypes, json, os
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(HERE, "libmb.so"))
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.mb_gather_nt.argtypes = [vp, vp, i64, vp, ci, ci, ci, vp]
log = open(os.path.join(ROOT, "gpurun_out", "microbench4.jsonl"), "a")
dev = torch.device("cuda:0"); out = torch.zeros(1 << 22, device=dev); st = vp(torch.cuda.current_stream().cuda_stream)
m = 2_054_740
rng = np.random.default_rng(0)
for rows in (40982,):
    tab = torch.ones(rows, 64, device=dev)
    w = (np.arange(rows) + 10.0) ** -0.75; c = np.cumsum(w); c /= c[-1]
    idx = np.searchsorted(c, rng.random(m)).clip(0, rows - 1)
    idx_t = torch.from_numpy(idx.astype(np.int32)).to(dev)
    for hot, pol in [(1 << 30, 0)] + [(h, q) for q in range(6) for h in (0, 8192, 16384)]:
        fn = lambda: lib.mb_gather_nt(vp(tab.data_ptr()), vp(idx_t.data_ptr()), m, vp(out.data_ptr()), 128, hot, pol, st)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 30
        r = dict(kind="gather_pol", pol=["nt","sc0","sc1","sc0 sc1","sc0 sc1 nt","sc1 nt"][pol], rows=rows, hot_rows=hot, hot_mass=float(c[min(hot, rows) - 1]) if hot else 0.0, us=round(us, 1),
                 gbps=round(m * 256 / (us * 1e-6) / 1e9))
        print(json.dumps(r), flush=True); log.write(json.dumps(r) + "\n")
    del tab
