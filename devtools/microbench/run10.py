#!/usr/bin/env python
"""Clock trace of the LDS-DMA BiGNN dense kernel: per-stamp mean / max over waves, in us from the earliest wave start."""
import ctypes, json, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import recbole_gnn_amd  # noqa: F401  (loads librbgnn.so, whose internals the trace build links against)
lib = ctypes.CDLL(os.path.join(HERE, "libbignn_trace.so"))
vp = ctypes.c_void_p
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70841
x, p = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
w1, w2 = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, 64, device=dev) * 0.1
b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
y = torch.empty(n, 64, device=dev)
trace = torch.zeros(2048 * 32, dtype=torch.int64, device=dev)
for _ in range(5):
    lib.mb_bignn_trace(vp(p.data_ptr()), vp(x.data_ptr()), vp(w1.data_ptr()), vp(b1.data_ptr()), vp(w2.data_ptr()), vp(b2.data_ptr()),
                       vp(y.data_ptr()), ctypes.c_int64(n), 1, vp(trace.data_ptr()), vp(0))
    torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(256, 8, 32).astype(np.float64)
# s_memtime runs at the shader clock and is per XCD (workgroup b sits on XCD b % 8): times relative to the XCD's first wave
t0 = np.stack([t[x::8, :, 0].min() for x in range(8)])
rel = t - t0[np.arange(256) % 8][:, None, None]
rel[t == 0] = np.nan
rel = rel.reshape(2048, 32)
names = ["start", "w+tile0 landed", "barrier", "first fetch done"] + [f"t{i}:{nm}" for i in range(7) for nm in ("96 mfma + prev epilogue issued", "next tile fetched", "32 mfma + dma issued", "-")]
prev = None
for k in range(32):
    col = rel[:, k]
    ok = ~np.isnan(col)
    if ok.sum() == 0:
        continue
    rec = dict(stamp=k, name=names[k], waves=int(ok.sum()), mean_kcyc=round(float(np.nanmean(col)) / 1e3, 2),
               min_kcyc=round(float(np.nanmin(col)) / 1e3, 2), max_kcyc=round(float(np.nanmax(col)) / 1e3, 2))
    if k > 0:
        d = col - rel[:, k - 1]
        if (~np.isnan(d)).sum():
            rec["since_prev_mean_kcyc"] = round(float(np.nanmean(d)) / 1e3, 2)
    print(json.dumps(rec), flush=True)
