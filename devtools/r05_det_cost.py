"""Cost of option "deterministic" per autograd-free training step (Gowalla shape, batch 2048, eager launches): us per step in both modes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
log = open(os.path.join(ROOT, "gpurun_out", "r05_deterministic_steps.jsonl"), "a")


def time_us(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]


for name in ["LightGCN", "NGCF", "SGL", "SimGCL", "XSimGCL"]:
    rec = {"model": name, "batch": 2048}
    for det in (0, 1):
        rbg.set_option("deterministic", det)
        torch.manual_seed(0); np.random.seed(0)
        m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
        m.train()
        st = rbg.fused_stepper(m, lr=1e-3, graphed=False)
        rec["stepper"] = type(st).__name__
        rec["deterministic_us" if det else "default_us"] = round(time_us(lambda: st.step(batch)), 1)
    rbg.set_option("deterministic", 0)
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
