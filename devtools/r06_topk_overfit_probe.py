#!/usr/bin/env python
"""r06: the screened top-k on the tables the bench's training extras leave behind (one batch overfitted for ~ 300 steps): candidate
statistics from the float64 scores (how many pairs reach the user's pre-pass bound, per user and per candidate region) and the call's time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
users = torch.randint(1, nu, (4096,), generator=g).to(dev)
fused = rbg.FusedBPRAdam(model, lr=1e-3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 330):
    fused.step(batch)
with torch.no_grad():
    model.restore_user_e = model.restore_item_e = None
    v1, i1 = model.full_sort_topk({"user_id": users}, 10)
    ue, ie = model.restore_user_e, model.restore_item_e
    rbg.set_option("topk_screen", 0)
    v0, i0 = model.full_sort_topk({"user_id": users}, 10)
    rbg.set_option("topk_screen", 1)
    sc = (ue[users].double() @ ie.double().T)
    sc[:, 0] = float("-inf")
    nrm = ue[users].double().norm(dim=1, keepdim=True) * ie.double().norm(dim=1)[None, :]
    lb = sc - 0.0041 * nrm
    tau = torch.topk(lb[:, :8192], 10, dim=1).values[:, -1:]
    cand = (sc + 0.0041 * nrm >= tau)
    per_user = cand.sum(dim=1).float()
    # regions: 32 users x 21 tiles of 32 items
    nt = (ni + 31) // 32
    tpc = (nt + 63) // 64
    pad = torch.zeros((4096, tpc * 64 * 32 - ni), dtype=torch.bool, device=dev)
    reg = torch.cat([cand, pad], 1).view(128, 32, -1, tpc * 32).sum(dim=(1, 3))
    rec = {"what": "overfitted tables", "user_norm_max_over_median": float(ue.norm(dim=1).max() / ue.norm(dim=1).median()),
           "item_norm_max_over_median": float(ie.norm(dim=1).max() / ie.norm(dim=1).median()),
           "candidates_per_user": {"mean": float(per_user.mean()), "p99": float(per_user.quantile(0.99)), "max": float(per_user.max())},
           "region_max": int(reg.max()), "regions_over_512": int((reg > 512).sum()), "same_items": float((i0 == i1).all(dim=1).float().mean()),
           "tau_min": float(tau.min()), "tau_neg_inf": int(torch.isinf(tau).sum())}
    import time
    for mode in (0, 1):
        rbg.set_option("topk_screen", mode)
        model.full_sort_topk({"user_id": users}, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.full_sort_topk({"user_id": users}, 10)
        torch.cuda.synchronize()
        rec[f"screen{mode}_us"] = (time.perf_counter() - t0) * 1e5
    worst = int(per_user.argmax())
    rec["worst_user"] = {"candidates": float(per_user[worst]), "norm": float(ue[users[worst]].norm()), "tau": float(tau[worst]),
                         "score_max": float(sc[worst].max()), "score_median": float(sc[worst].median())}
print(json.dumps(rec))
