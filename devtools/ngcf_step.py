"""NGCF / LightGCN (autograd path) training step on the Gowalla shape: total time (for kstats: per-kernel)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "ngcf"
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
cfg = {"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "reg_weight": 1e-5, "require_pow": False,
       "hidden_size_list": [64, 64, 64], "node_dropout": 0.0, "message_dropout": float(os.environ.get("MSG_DROPOUT", "0.0")),
       "fused_forward": os.environ.get("FUSED", "1") == "1"}
model = (rbg.NGCF if which == "ngcf" else rbg.LightGCN)(cfg, ds)
model.train()
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    model.calculate_loss(batch).backward()
    opt.step()


for _ in range(3):
    step()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20):
    step()
b.record(); torch.cuda.synchronize()
out = {"kind": f"{which}_train_step", "us_eager": round(a.elapsed_time(b) * 1e3 / 20, 1)}
if len(sys.argv) > 2 and sys.argv[2] == "graph":
    stepper = rbg.GraphedStep(model, batch, lr=1e-3)
    for _ in range(3):
        stepper.step(batch)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        stepper.step(batch)
    b.record(); torch.cuda.synchronize()
    out["us_graphed"] = round(a.elapsed_time(b) * 1e3 / 20, 1)
if len(sys.argv) > 2 and sys.argv[2] == "fused":
    fg = rbg.FusedNGCFAdam(model, lr=1e-3, graphed=True)
    for _ in range(4):
        fg.step(batch)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        fg.step(batch)
    b.record(); torch.cuda.synchronize()
    out["us_fused_graphed"] = round(a.elapsed_time(b) * 1e3 / 20, 1)
print(json.dumps(out))
