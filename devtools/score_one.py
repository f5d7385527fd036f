import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40982
u, it = torch.randn(B, 64, device=dev), torch.randn(n, 64, device=dev)
for _ in range(5):
    rbg.score(u, it)
torch.cuda.synchronize()
