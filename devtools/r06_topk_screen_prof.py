#!/usr/bin/env python
"""r06: 20 calls of the screened top-k (4096 users x Gowalla items, d = 64 or argv[1]) for rocprofv3 --kernel-trace --stats."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BITS = int(os.environ.get("SCREEN_DBG", "0"))
if BITS:  # the what-if build (devtools/microbench/build_topk_screen_trace.sh)
    os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_screentrace.so")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": 3}, ds)
with torch.no_grad():
    ue, ie = model.forward()
if BITS:
    assert rbg._lib.lib.mb_screen_debug_set(BITS) == 0
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
users = torch.randint(1, nu, (NB,), generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(20):
    rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
if os.environ.get("STATS"):
    # how many candidates the screen hands to the merge: items scoring at least the 10th best of the first 8192 (history included)
    with torch.no_grad():
        sc = ue[users] @ ie.T
        sc[:, 0] = float("-inf")
        tau = torch.topk(sc[:, :8192], 10 + 0, dim=1).values[:, -1:]
        cand = (sc >= tau).sum(dim=1).float()
        print({"candidates_per_user_mean": float(cand.mean()), "max": float(cand.max()), "p99": float(cand.quantile(0.99)), "total": float(cand.sum())})
