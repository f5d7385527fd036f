#!/usr/bin/env python
"""r06: 20 calls of the screened top-k (4096 users x Gowalla items, d = 64 or argv[1]) for rocprofv3 --kernel-trace --stats."""
import os
import sys

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
users = torch.randint(1, nu, (4096,), generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(20):
    rbg.full_sort_topk(model.graph, ue, ie, users, 10)
torch.cuda.synchronize()
