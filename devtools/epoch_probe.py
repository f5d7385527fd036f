"""driver.fit at the Gowalla shape: seconds per epoch (502 batches of 2048) per model, device sampler against the numpy sampler."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
names = sys.argv[1:] or ["LightGCN", "NGCF", "SGL"]
for name in names:
    for dev_sampler in (True, False):
        torch.manual_seed(0); np.random.seed(0)
        m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
        times = []
        def log(msg, t=[time.perf_counter()]):
            torch.cuda.synchronize(); now = time.perf_counter(); times.append(now - t[0]); t[0] = now
        epochs = 3 if dev_sampler else 1
        hist = rbg.driver.fit(m, uid, iid, epochs=epochs, lr=1e-3, log=log, device_sampler=dev_sampler)
        print(json.dumps({"model": name, "sampler": "device" if dev_sampler else "numpy", "batches_per_epoch": (len(uid) + 2047) // 2048,
                          "s_per_epoch": [round(t, 3) for t in times], "stepper": type(rbg.fused_stepper(m)).__name__, "loss": [round(h, 2) for h in hist]}), flush=True)
