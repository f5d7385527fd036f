"""InfoNCE forward + backward (ops.info_nce) at [B = 2048] x n x d: one-pass form against the three-launch form, interleaved."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
n, d, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 40982), (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 2048
t1 = torch.randn(n, d, device=dev, requires_grad=True)
t2 = torch.randn(n, d, device=dev, requires_grad=True)
idx = torch.randint(1, n, (B,), device=dev)


def run():
    t1.grad = t2.grad = None
    rbg.ops.info_nce(t1, t2, idx, 0.2).backward()


a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {"n": n, "d": d, "B": B}
for rep in range(3):
    for mode in (1, 0):
        rbg.set_option("lse_onepass", mode)
        for _ in range(5):
            run()
        torch.cuda.synchronize(); a.record()
        for _ in range(30):
            run()
        b.record(); torch.cuda.synchronize()
        out.setdefault("onepass_us" if mode else "three_launch_us", []).append(round(a.elapsed_time(b) * 1e3 / 30, 1))
rbg.set_option("lse_onepass", 1)
print(json.dumps(out))
