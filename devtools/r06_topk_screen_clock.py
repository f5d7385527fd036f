#!/usr/bin/env python
"""r06: phase clock of screen_merge_kernel (the RBG_SCREEN_DBG build): thread 0 of every workgroup stamps clock64 at the phase
boundaries.  -> mean / p90 microseconds per phase (100 MHz constant clock assumed for clock64... measured against the kernel span)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_screentrace.so")
import torch

sys.path.insert(0, os.path.dirname(HERE))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": 3}, ds)
with torch.no_grad():
    ue, ie = model.forward()
users = torch.randint(1, nu, (4096,), generator=torch.Generator().manual_seed(1)).to(dev)
lib = rbg._lib.lib
import ctypes
trace = torch.zeros((256, 16), dtype=torch.int64, device=dev)
for _ in range(3):
    rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
assert lib.mb_screen_trace_set(ctypes.c_void_p(trace.data_ptr())) == 0
rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
lib.mb_screen_trace_set(ctypes.c_void_p(0))
t = trace.cpu().numpy().astype("float64")
names = ["stage", "load entries", "score + history", "bucket", "sort", "out"]
span = t[:, 6].max() - t[:, 0].min()
rec = {"what": "screen_merge_kernel phase clock", "d": d, "kernel_span_ticks": span, "entries_mean": t[:, 7].mean(), "entries_max": t[:, 7].max()}
for k, nm in enumerate(names):
    dt = t[:, k + 1] - t[:, k]
    rec[nm] = {"mean_ticks": round(float(dt.mean()), 1), "p90": round(float(sorted(dt)[int(0.9 * len(dt))]), 1)}
rec["start_spread_ticks"] = float(t[:, 0].max() - t[:, 0].min())
print(json.dumps(rec))
