#!/usr/bin/env python
"""r05: per-wave phase clock of score_topk_kernel (RBG_TOPK_TRACE build), propagated embeddings, 4096 users, k = 10; with the what-if
switch 8 (nothing passes) beside it.  kcycles per wave and cycles per tile.  -> gpurun_out/r05_topk_clock.jsonl"""
import ctypes, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_topktrace.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import recbole_gnn_amd as rbg

dev = torch.device("cuda:0")
lib = rbg._lib.lib
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
users = torch.randint(1, nu, (4096,), generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    ua, it = model.forward()
    ua, it = ua.contiguous(), it.contiguous()
trace = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
lib.mb_topk_trace_set(ctypes.c_void_p(trace.data_ptr()))
names = ["prologue", "fetch issue (tile t+1)", "product (MFMA issue + LDS fragment reads)", "filter: threshold MFMA, max3 tree, ballot",
         "filter: tiles with a passing entry (row masks, appends)", "publish (split + LDS write of tile t+1)", "barrier", "-"]
log = open(os.path.join(ROOT, "gpurun_out", "r05_topk_clock.jsonl"), "a")
for bits in (0, 8):
    lib.mb_topk_debug_set(bits)
    for _ in range(3):
        trace.zero_()
        rbg.full_sort_topk(model.graph, ua, it, users, 10)
        torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t[t.sum(1) > 0]
    tiles = 1281 * 128 / t.shape[0]
    rec = {"what": "topk phase clock", "bits": bits, "waves": int(t.shape[0]), "tiles_per_wave": round(tiles, 1),
           "phases_kcyc_per_wave[cycles_per_tile]": {names[j]: [round(t[:, j].mean() / 1e3, 1), round(t[:, j].mean() / tiles)] for j in range(7)},
           "total_kcyc": round(t.sum(1).mean() / 1e3, 1), "max_wave_kcyc": round(t.sum(1).max() / 1e3, 1)}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
lib.mb_topk_debug_set(0)
