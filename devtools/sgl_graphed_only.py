"""SGL training step replayed from a HIP graph, nothing else (for rocprofv3 kernel stats: devtools/kstats.sh)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
np.random.seed(0); torch.manual_seed(0)
model = rbg.SGL({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "type": "ED", "drop_ratio": 0.1,
                 "ssl_tau": 0.2, "ssl_weight": 0.05, "reg_weight": 1e-4}, ds)
model.train(); model.graph_construction()
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
import json
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for name, gs in (("autograd_graphed", rbg.GraphedStep(model, batch, lr=1e-3)), ("fused_eager", rbg.FusedSGLAdam(model, lr=1e-3)),
                 ("fused_graphed", rbg.FusedSGLAdam(model, lr=1e-3, graphed=True))):
    if len(sys.argv) > 1 and sys.argv[1] != name:
        continue
    for _ in range(5):
        gs.step(batch)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        gs.step(batch)
    b.record(); torch.cuda.synchronize()
    out[name + "_us"] = round(a.elapsed_time(b) * 1e3 / 20, 1)
print(json.dumps(out))
