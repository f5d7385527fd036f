"""model.full_sort_topk on PROPAGATED embeddings (the bench extra's data: Xavier tables through K layers, a few training steps),
24-entry against 48-entry candidate lists, interleaved."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
g = torch.Generator().manual_seed(1)
batch = {k: torch.randint(1, n, (2048,), generator=g).to(dev) for k, n in (("user_id", nu), ("item_id", ni), ("neg_item_id", ni))}
fused = rbg.FusedBPRAdam(model, lr=1e-3)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    fused.step(batch)
users = torch.randint(1, nu, (4096,), generator=g).to(dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out, res = {}, {}
with torch.no_grad():
    for rep in range(3):
        for mode in (1, 0):
            rbg.set_option("topk_short_lists", mode)
            for _ in range(3):
                res[mode] = model.full_sort_topk({"user_id": users}, 10)
            torch.cuda.synchronize(); a.record()
            for _ in range(20):
                model.full_sort_topk({"user_id": users}, 10)
            b.record(); torch.cuda.synchronize()
            out.setdefault("short_us" if mode else "long_us", []).append(round(a.elapsed_time(b) * 1e3 / 20, 1))
out["same"] = bool(torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0]))
print(json.dumps(out))
