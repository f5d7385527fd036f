#!/usr/bin/env python
"""Seconds per driver epoch (502 batches, device sampler, Gowalla shape) of the models named on the command line."""
import sys, time, torch, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
for name in sys.argv[1:]:
    torch.manual_seed(0); np.random.seed(0)
    m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
    rbg.driver.fit(m, uid, iid, epochs=1, lr=1e-3, device_sampler=True)
    out = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rbg.driver.fit(m, uid, iid, epochs=1, lr=1e-3, device_sampler=True)
        torch.cuda.synchronize(); out.append(round(time.perf_counter() - t0, 4))
    print(name, "epoch_s", out, flush=True)
