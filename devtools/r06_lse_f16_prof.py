"""50 InfoNCE forward + backward calls (2048 x 40 982 x 64, tau 0.2) for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
n, d, B = 40982, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2048
g = torch.Generator().manual_seed(1)
t1 = (torch.randn(n, d, generator=g) * 0.1).to(dev).requires_grad_(True)
t2 = (torch.randn(n, d, generator=g) * 0.1).to(dev).requires_grad_(True)
idx = torch.randint(1, n, (B,), generator=g).to(dev)
for _ in range(50):
    t1.grad = t2.grad = None
    rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
torch.cuda.synchronize()
