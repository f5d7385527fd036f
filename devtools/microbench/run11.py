#!/usr/bin/env python
"""Phase clock of score_kernel (B = 4096 x 40 982, d = 64): cycles per phase per wave, mean over waves."""
import ctypes, json, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import recbole_gnn_amd  # noqa: F401
lib = ctypes.CDLL(os.path.join(HERE, "libscore_trace.so"))
vp = ctypes.c_void_p
dev = torch.device("cuda:0")
B, n, d = 4096, 40982, int(sys.argv[1]) if len(sys.argv) > 1 else 64
u, it = torch.randn(B, d, device=dev), torch.randn(n, d, device=dev)
s = torch.empty(B, n, device=dev)
trace = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
for _ in range(3):
    trace.zero_()
    lib.mb_score_trace(vp(u.data_ptr()), vp(it.data_ptr()), vp(s.data_ptr()), ctypes.c_int64(B), ctypes.c_int64(n), d, vp(trace.data_ptr()), vp(0))
    torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = t[t.sum(1) > 0]
names = ["prologue (first fetch, publish, barrier)", "product (MFMA + LDS fragment reads)", "publish (split + LDS write of tile t+1)",
         "fetch issue (tile t+2)", "aligned emit (shuffles + stores)", "barrier", "flush", "-"]
tot = t.sum(1).mean()
for k in range(7):
    print(json.dumps(dict(kind="score_phase_clock", d=d, phase=names[k], mean_kcyc_per_wave=round(t[:, k].mean() / 1e3, 1),
                          share=round(t[:, k].mean() / tot, 3))))
print(json.dumps(dict(kind="score_phase_clock", d=d, phase="total", mean_kcyc_per_wave=round(tot / 1e3, 1), waves=int(t.shape[0]))))
