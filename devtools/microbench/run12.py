#!/usr/bin/env python
"""Phase clock of score_topk_kernel (4096 users x 40 982 items, d = 64, k = 10, history masked): cycles per phase per wave."""
import ctypes, json, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from recbole_gnn_amd import _lib
lib = ctypes.CDLL(os.path.join(HERE, os.environ.get("TOPK_TRACE_LIB", "libtopk_trace.so")))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.rbg_full_sort_topk_f32.argtypes = [vp, vp, vp, vp, i64, i64, i64, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
lib.rbg_full_sort_topk_workspace.argtypes = [i64, i64, ctypes.c_int, ctypes.POINTER(i64)]
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
torch.manual_seed(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ua, it = torch.randn(nu, d, device=dev) * 0.1, torch.randn(ni, d, device=dev) * 0.1
B, k = 4096, 10
users = torch.randint(1, nu, (B,), device=dev)
nbytes = i64()
lib.rbg_full_sort_topk_workspace(B, ni, k, ctypes.byref(nbytes))
work = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
vals, idx = torch.empty(B, k, device=dev), torch.empty(B, k, dtype=torch.int64, device=dev)
trace = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
lib.mb_topk_trace_set(vp(trace.data_ptr()))
if len(sys.argv) > 2:
    rbg.set_option("topk_short_lists", int(sys.argv[2]))
for _ in range(3):
    trace.zero_()
    rc = lib.rbg_full_sort_topk_f32(g.ptr, vp(ua.data_ptr()), vp(it.data_ptr()), vp(users.data_ptr()), B, nu, ni, d, k,
                                    vp(vals.data_ptr()), vp(idx.data_ptr()), vp(work.data_ptr()), vp(0))
    assert rc == 0, rc
    torch.cuda.synchronize()
ref_v, ref_i = rbg.full_sort_topk(g, ua, it, users, k)
assert os.environ.get("TOPK_TRACE_LIB") or torch.equal(ref_i, idx)
t = trace.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = t[t.sum(1) > 0]
names = ["prologue (first fetch, publish, barrier)", "fetch issue (tile t+1)", "product (MFMA + LDS fragment reads)", "filter (compare, ballot, appends)",
         "publish (split + LDS write of tile t+1)", "barrier"]
tot = t.sum(1).mean()
for j in range(6):
    print(json.dumps(dict(kind="topk_phase_clock", d=d, phase=names[j], mean_kcyc_per_wave=round(t[:, j].mean() / 1e3, 1), share=round(t[:, j].mean() / tot, 3))))
print(json.dumps(dict(kind="topk_phase_clock", d=d, phase="total", mean_kcyc_per_wave=round(tot / 1e3, 1), waves=int(t.shape[0]))))
