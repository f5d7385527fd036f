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
if os.environ.get("OVERFIT"):  # the tables the bench's training extras leave behind
    g = torch.Generator().manual_seed(1)
    batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
             "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
    fused = rbg.FusedBPRAdam(model, lr=1e-3)
    for _ in range(330):
        fused.step(batch)
with torch.no_grad():
    ue, ie = model.forward()
NB = int(os.environ.get('NB', '4096'))
users = torch.randint(1, nu, (NB,), generator=torch.Generator().manual_seed(1)).to(dev)
lib = rbg._lib.lib
import ctypes
trace = torch.zeros((2 * ((NB + 31) // 32), 16), dtype=torch.int64, device=dev)
for _ in range(3):
    rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
tmain = torch.zeros((1024, 4), dtype=torch.int64, device=dev)
assert lib.mb_screen_trace_main_set(ctypes.c_void_p(tmain.data_ptr())) == 0
assert lib.mb_screen_debug_set(int(os.environ.get("SCREEN_DBG", "0"))) == 0
assert lib.mb_screen_trace_set(ctypes.c_void_p(trace.data_ptr())) == 0
rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
lib.mb_screen_trace_set(ctypes.c_void_p(0))
lib.mb_screen_trace_main_set(ctypes.c_void_p(0))
tm = tmain.cpu().numpy().astype("float64")
tm = tm[tm[:, 0] > 0]
main_rec = {"what": "screen_main_kernel clock (wave 0 of every workgroup)", "workgroups": int(tm.shape[0]),
            "prologue_ticks_mean": float((tm[:, 1] - tm[:, 0]).mean()), "loop_ticks_mean": float((tm[:, 2] - tm[:, 1]).mean()),
            "loop_ticks_p90": float(sorted(tm[:, 2] - tm[:, 1])[int(0.9 * tm.shape[0])]),
            "span_ticks": float(tm[:, 2].max() - tm[:, 0].min()), "start_spread_ticks": float(tm[:, 0].max() - tm[:, 0].min()),
            "dbg_bits": int(os.environ.get("SCREEN_DBG", "0"))}
print(json.dumps(main_rec))
t = trace.cpu().numpy().astype("float64")
names = ["stage", "load entries", "score + history", "bucket", "sort", "out"]
span = t[:, 6].max() - t[:, 0].min()
rec = {"what": "screen_merge_kernel phase clock", "d": d, "kernel_span_ticks": span, "entries_mean": t[:, 7].mean(), "entries_max": t[:, 7].max()}
for k, nm in enumerate(names):
    dt = t[:, k + 1] - t[:, k]
    rec[nm] = {"mean_ticks": round(float(dt.mean()), 1), "p90": round(float(sorted(dt)[int(0.9 * len(dt))]), 1)}
rec["entries_top"] = sorted(t[:, 7].tolist())[-5:]
rec["region_entries_top"] = sorted(t[:, 8].tolist())[-5:]
rec["workgroups_with_unbounded_users"] = int((t[:, 9] > 0).sum())
rec["unbounded_users"] = int(t[:, 9].sum())
rec["overflowed_regions"] = int(t[:, 10].sum())
import numpy as np
ov = [(int(w), int(t[w, 11]), float(np.array([int(t[w, 12])], dtype=np.uint32).view(np.float32)[0])) for w in range(t.shape[0]) if t[w, 10] > 0]
rec["overflowed(workgroup, first chunk, min tau of its 16 users)"] = ov[:12]
if ov:
    with torch.no_grad():
        w, ch, _ = ov[0]
        tile, half = w // 2, w % 2
        rows = [tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half for r in range(16)]
        us = users[rows]
        nt = (ni + 31) // 32
        tpc = (nt + 63) // 64
        lo_i, hi_i = ch * tpc * 32, min((ch + 1) * tpc * 32, ni)
        sc = ue[us].double() @ ie[lo_i:hi_i].double().T
        rec["first_overflow"] = {"users": us.tolist(), "user_degree": [int(model.graph.export_csr()[0][u + 1] - model.graph.export_csr()[0][u]) for u in us.tolist()][:16],
                                 "user_norms": [round(float(x), 3) for x in ue[us].norm(dim=1)], "score_max_per_user": [round(float(x), 3) for x in sc.max(dim=1).values],
                                 "chunk_items": [lo_i, hi_i]}
rec["start_spread_ticks"] = float(t[:, 0].max() - t[:, 0].min())
print(json.dumps(rec))
