"""top-k timing on the embeddings the bench has when it reaches its top-k extra: after ~ 300 fused training steps on one batch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
users = torch.randint(1, nu, (4096,), generator=g).to(dev)
fused = rbg.FusedBPRAdam(model, lr=1e-3)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]


out = {}
for steps in (0, 310):
    for _ in range(steps): fused.step(batch)
    with torch.no_grad():
        ua, it = model.forward(); ua, it = ua.contiguous(), it.contiguous()
        out[f"after_{steps}_steps_us"] = round(timeit(lambda: rbg.full_sort_topk(model.graph, ua, it, users, 10)), 1)
        model.restore_user_e = model.restore_item_e = None
        out[f"after_{steps}_steps_model_call_us"] = round(timeit(lambda: model.full_sort_topk({"user_id": users}, 10), iters=10), 1)
print(json.dumps(out), flush=True)
