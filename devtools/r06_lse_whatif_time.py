"""us per InfoNCE forward + backward (2048 x 40 982 x 64 and x 128) of whatever library RBGNN_LIB names (what-if builds: wrong results)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
rec = {"lib": os.path.basename(os.environ.get("RBGNN_LIB", "librbgnn.so"))}
for n, d in ((40982, 64), (40982, 128)):
    g = torch.Generator().manual_seed(1)
    t1 = (torch.randn(n, d, generator=g) * 0.1).to(dev).requires_grad_(True)
    t2 = (torch.randn(n, d, generator=g) * 0.1).to(dev).requires_grad_(True)
    idx = torch.randint(1, n, (2048,), generator=g).to(dev)
    def step():
        t1.grad = t2.grad = None
        rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): step()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    rec[f"d{d}_us"] = round(sorted(ts)[2], 1)
print(json.dumps(rec), flush=True)
