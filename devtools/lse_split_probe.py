import json, os, sys, torch
sys.path.insert(0, ".")
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
def time_us(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    res = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize(); res.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(res)[1]
for name, n, d, B in (("gowalla items", 40982, 64, 2048), ("amazon items d128", 91600, 128, 2048)):
    t1 = torch.randn(n, d, device=dev, requires_grad=True); t2 = torch.randn(n, d, device=dev, requires_grad=True)
    idx = torch.randint(1, n, (B,), device=dev)
    def run2():
        t1.grad = t2.grad = None
        rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
    rec = dict(kind="info_nce_fwd_bwd", shape=name)
    for rnd in range(2):
        for split in (1, 0):
            rbg.set_option("mfma_split", split)
            rec[f"us_split{split}"] = min(time_us(run2), rec.get(f"us_split{split}", 1e30))
    rbg.set_option("mfma_split", 1)
    print(json.dumps(rec), flush=True)
