"""One epoch of driver.fit for one model at the Gowalla shape (device sampler), for rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
name = sys.argv[1]
torch.manual_seed(0); np.random.seed(0)
m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
rbg.driver.fit(m, uid, iid, epochs=int(sys.argv[2]) if len(sys.argv) > 2 else 2, lr=1e-3, device_sampler=True)
torch.cuda.synchronize()
