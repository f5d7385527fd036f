cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
users = torch.randint(1, nu, (4096,), generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    ua, it = model.forward(); ua, it = ua.contiguous(), it.contiguous()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]
for sample in (4096, 6144, 8192, 12288, 16384, 8192):
    rbg.set_option("topk_sample", sample)
    print(json.dumps({"topk_sample": sample, "us": round(timeit(lambda: rbg.full_sort_topk(model.graph, ua, it, users, 10)), 1)}), flush=True)
PY
