"""SGL training step (calculate_loss + backward) on the Gowalla shape: total and the SSL part alone."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
np.random.seed(0)
torch.manual_seed(0)
model = rbg.SGL({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "type": "ED", "drop_ratio": 0.1,
                 "ssl_tau": 0.2, "ssl_weight": 0.05, "reg_weight": 1e-4}, ds)
model.train()
model.graph_construction()
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    model.calculate_loss(batch).backward()
    opt.step()


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


out = {"kind": "sgl_train_step", "us_step": round(timed(step), 1)}
fused = rbg.ops.info_nce


def torch_nce(a, p, c, tau):
    F = torch.nn.functional
    a, p, c = F.normalize(a, dim=1), F.normalize(p, dim=1), F.normalize(c, dim=1)
    return -torch.log(torch.exp((a * p).sum(1) / tau) / torch.exp(a.matmul(c.T) / tau).sum(1)).sum()


rbg.ops.info_nce = lambda t1, t2, idx, tau: torch_nce(t1[idx], t2[idx], t2, tau)  # the reference formula in torch
out["us_step_torch_ssl"] = round(timed(step), 1)
rbg.ops.info_nce = fused
gs = rbg.GraphedStep(model, batch, lr=1e-3)
out["us_step_graphed"] = round(timed(lambda: gs.step(batch)), 1)
print(json.dumps(out))
