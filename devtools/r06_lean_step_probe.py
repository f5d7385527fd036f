#!/usr/bin/env python
"""r06: LightGCN's fused training step in its lean form (two launches around the backward propagation) against the separate calls:
time per step (HIP events over 200 eager steps) and one driver epoch (502 batches, device sampler), Gowalla shape."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
rec = {"what": "LightGCN fused step, lean vs separate calls"}
for rnd in range(2):
    for lean in (False, True):
        torch.manual_seed(0)
        model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
        st = rbg.FusedBPRAdam(model, lr=1e-3)
        st.lean = lean
        for _ in range(20):
            st.step(batch)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            st.step(batch)
        b.record()
        torch.cuda.synchronize()
        rec.setdefault("lean_step_us" if lean else "separate_step_us", []).append(round(a.elapsed_time(b) * 5.0, 1))
import numpy as np
for lean in (False, True):
    torch.manual_seed(0); np.random.seed(0)
    model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
    orig = rbg.train.FusedBPRAdam.__init__
    def patched(self, *a, _lean=lean, **k):
        orig(self, *a, **k)
        self.lean = _lean
    rbg.train.FusedBPRAdam.__init__ = patched
    try:
        rbg.driver.fit(model, uid, iid, epochs=1, lr=1e-3, device_sampler=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = rbg.driver.fit(model, uid, iid, epochs=2, lr=1e-3, device_sampler=True)
        torch.cuda.synchronize()
        rec["lean_epoch_s" if lean else "separate_epoch_s"] = round((time.perf_counter() - t0) / 2, 4)
        rec["lean_losses" if lean else "separate_losses"] = [round(x, 4) for x in hist]
    finally:
        rbg.train.FusedBPRAdam.__init__ = orig
print(json.dumps(rec))
