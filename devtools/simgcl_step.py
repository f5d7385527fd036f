"""SimGCL / XSimGCL training step replayed from a HIP graph at the Gowalla shape (for devtools/kstats.sh)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "SimGCL"
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
m.train()
g = torch.Generator().manual_seed(1)
batch = {k: torch.randint(1, n, (2048,), generator=g).to(dev) for k, n in (("user_id", nu), ("item_id", ni), ("neg_item_id", ni))}
gs = rbg.GraphedStep(m, batch, lr=1e-3)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    gs.step(batch)
torch.cuda.synchronize(); a.record()
for _ in range(30):
    gs.step(batch)
b.record(); torch.cuda.synchronize()
print(json.dumps({"model": name, "graphed_step_us": round(a.elapsed_time(b) * 1e3 / 30, 1)}))
