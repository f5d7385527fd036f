"""The fused LightGCN step (Gowalla shape, batch 2048) with option "deterministic" on, 20 steps (for devtools/kstats.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
g = torch.Generator().manual_seed(1)
batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev), "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
         "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
fused = rbg.FusedBPRAdam(model, lr=1e-3)
rbg.set_option("deterministic", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(20):
    fused.step(batch)
torch.cuda.synchronize()
