"""rbg_infonce_f32 with gradients (2048 batch rows against 40 982 / 29 858 table rows, d = 64): us per forward + backward."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
out = {"tag": sys.argv[1] if len(sys.argv) > 1 else ""}
for n in (40982, 29858):
    d, B = 64, 2048
    t1 = torch.randn(n, d, device=dev, requires_grad=True)
    t2 = torch.randn(n, d, device=dev, requires_grad=True)
    idx = torch.randint(1, n, (B,), device=dev)
    def step():
        t1.grad = t2.grad = None
        rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): step()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    out[f"infonce_fwd_bwd_us(n={n})"] = round(sorted(ts)[2], 1)
print(json.dumps(out), flush=True)
