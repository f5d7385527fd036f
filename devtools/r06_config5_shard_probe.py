#!/usr/bin/env python
"""r06: one rank's layer product of BASELINE config #5's graph (10 M users / 5 M items / 200 M interactions, d = 128) cut into
8 node shards, alone on one GPU: the fused layer over column windows of the [owned | halo] table (the table is beyond the
rectangular plan's 32-bit offsets: 15 M rows x 512 B) against the two-handle form (interior on the column-slab kernel, halo block
on the binned kernel — what r05 ran).  Also the parity of the two against each other.  ~ 10 minutes, ~ 40 GB of HBM."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0   # 1.0 = the full shape
nu, ni, e = int(10_000_000 * scale), int(5_000_000 * scale), int(200_000_000 * scale)
d, world, rank = 128, 8, 0
t0 = time.time()
uid, iid = rbg.synth.powerlaw_bipartite_device(nu, ni, e, dev)
print(f"generated in {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
t0 = time.time()
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
sh = rbg.sharded
owner = sh.degree_striped_partition(uid, iid, nu, ni, world)
plan = sh.plan_from_csr(*g.device_csr(), nu, owner, rank, world)
# T1: the same layer (rbg_spmm_f32, d = 128) on the whole graph
xg, yg = torch.randn(nu + ni, d, device=dev), torch.empty(nu + ni, d, device=dev)
for _ in range(2):
    rbg.ops.spmm_raw(g, xg, out=yg)
torch.cuda.synchronize()
_a, _b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_a.record()
for _ in range(3):
    rbg.ops.spmm_raw(g, xg, out=yg)
_b.record()
torch.cuda.synchronize()
T1 = _a.elapsed_time(_b) * 1e3 / 3
T1_kernel, T1_status = g.spmm_kernel_name(d), g.sell_status()
del g, xg, yg
torch.cuda.empty_cache()
print(f"graph + plan in {time.time() - t0:.0f} s: owned {plan.n_owned}, halo {plan.n_halo}", file=sys.stderr, flush=True)
be = sh.HipBackend(dev)
rec = {"shape": [nu, ni, e], "d": d, "world": world, "rank": rank, "owned_rows": int(plan.n_owned), "halo_rows": int(plan.n_halo),
       "nnz": int(plan.int_csr[0][-1] + plan.halo_csr[0][-1]), "T1_whole_graph_layer_us": T1, "T1_kernel": T1_kernel, "T1_status": T1_status}


def time_us(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


no, nh = plan.n_owned, plan.n_halo
xc = torch.randn(no + nh, d, device=dev)
t0 = time.time()
fused = sh.ShardedPropagation(plan, be, transport="staged", fused=True, cat_window_rows=4_000_000)
rec["fused_build_s"] = round(time.time() - t0, 1)
rec["fused_status"] = fused.kernel_status()
y1 = torch.empty(no, d, device=dev)
if fused.fused:
    wins = fused.g_cats

    def run_fused():
        for w, (lo, hi, gw) in enumerate(wins):
            be.spmm(gw, xc[lo:hi], y1, w > 0)

    rec["fused_windows_us"] = time_us(run_fused)
    run_fused()
t0 = time.time()
pair = sh.ShardedPropagation(plan, be, transport="staged", fused=False)
rec["pair_build_s"] = round(time.time() - t0, 1)
rec["pair_status"] = pair.kernel_status()
y2 = torch.empty(no, d, device=dev)


def run_pair():
    be.spmm(pair.g_int, xc[:no], y2, False)
    be.spmm(pair.g_halo, xc[no:], y2, True)


rec["two_handles_us"] = time_us(run_pair)
run_pair()
torch.cuda.synchronize()
rec["max_abs_diff_between_the_forms"] = float((y1 - y2).abs().max()) if fused.fused else None
rec["ceiling(T1 / T_8,0)"] = {"two_handles": T1 / rec["two_handles_us"], "fused_windows": (T1 / rec["fused_windows_us"]) if "fused_windows_us" in rec else None}
print(json.dumps(rec), flush=True)
