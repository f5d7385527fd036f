#!/usr/bin/env python
"""r06: where the screen starts to pay: small batches, option "topk_screen" 2 (always) against 0 (the exact passes)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402
from r06_topk_screen_probe import replay_us  # noqa: E402  (prints its own sweep first when imported: keep this script's output apart)

dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
for d in (64, 128):
    torch.manual_seed(0)
    model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": 3}, ds)
    with torch.no_grad():
        ue, ie = model.forward()
    for nb in (128, 256, 512, 768, 1024, 2048):
        users = torch.randint(1, nu, (nb,), generator=torch.Generator().manual_seed(1)).to(dev)
        rec = {"d": d, "users": nb, "k": 10}
        for mode in (0, 2):
            rbg.set_option("topk_screen", mode)
            rec[f"screen{mode}_us"] = round(replay_us(lambda: rbg.full_sort_topk(model.graph, ue, ie, users, 10)), 1)
        print(json.dumps(rec), flush=True)
rbg.set_option("topk_screen", 1)
