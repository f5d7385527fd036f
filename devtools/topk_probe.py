import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ua, it = torch.randn(nu, 64, device=dev), torch.randn(ni, 64, device=dev)
users = torch.randint(1, nu, (B,), device=dev)
for _ in range(5):
    rbg.full_sort_topk(g, ua, it, users, 10)
torch.cuda.synchronize()
