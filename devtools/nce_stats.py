import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
n, d, B = int(sys.argv[1]), int(sys.argv[2]), 2048
if len(sys.argv) > 3:
    rbg.set_option("lse_probe", int(sys.argv[3]))
t1 = torch.randn(n, d, device=dev, requires_grad=True)
t2 = torch.randn(n, d, device=dev, requires_grad=True)
idx = torch.randint(1, n, (B,), device=dev)
for _ in range(20):
    t1.grad = t2.grad = None
    rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
torch.cuda.synchronize()
