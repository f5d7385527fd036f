#!/usr/bin/env python
"""Seconds per NCL driver epoch with the prototype term on (warm_up_step = 0) and off (the default's first 20 epochs)."""
import sys, time, torch, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
for warm in (20, 0):
    torch.manual_seed(0); np.random.seed(0)
    m = rbg.NCL({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True, "warm_up_step": warm}, ds)
    rbg.driver.fit(m, uid, iid, epochs=1, lr=1e-3, device_sampler=True)
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rbg.driver.fit(m, uid, iid, epochs=1, lr=1e-3, device_sampler=True)
        torch.cuda.synchronize(); out.append(round(time.perf_counter() - t0, 4))
    print("NCL warm_up_step", warm, "epoch_s", out, flush=True)
