#!/usr/bin/env python
"""Launches of the plain SpMM layer of one workload / width, for the PMC passes of devtools/traffic_session.sh.
Prints the kernel name the library runs for this shape (the key of profiles/traffic.json)."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="gowalla")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--launches", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make(args.workload)
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
n = nu + ni
x, y = torch.randn(n, args.dim, device=dev), torch.empty(n, args.dim, device=dev)
for _ in range(args.launches // 2):  # ping-pong, as the layers of a propagation do
    rbg.ops.spmm_raw(g, x, out=y)
    rbg.ops.spmm_raw(g, y, out=x)
torch.cuda.synchronize()
print(json.dumps({"workload": args.workload, "dim": args.dim, "kernel": g.spmm_kernel_name(args.dim), "nodes": n, "nnz": g.nnz}))
