import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_inter.npz"))
uid, iid, nu, ni = z["uid"], z["iid"], int(z["n_users"]), int(z["n_items"])
ds = rbg.InteractionDataset(uid, iid, nu, ni)
cfg = {"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "reg_weight": 1e-4, "device_sampling": False}
g = torch.Generator().manual_seed(1)
mk = lambda b: {k: torch.randint(1, n, (b,), generator=g).to(dev) for k, n in (("user_id", nu), ("item_id", ni), ("neg_item_id", ni))}
case = sys.argv[1]
torch.manual_seed(1); np.random.seed(7)
m = rbg.SGL(cfg, ds)
m.train()
gs = rbg.GraphedStep(m, mk(64), lr=1e-3)
for _ in range(3):
    gs.step(mk(64))
torch.cuda.synchronize()
if case == "keep_old_views":
    keep = (m.sub_graph1, m.sub_graph2)
if case == "sync_after_train":
    pass
m.train()
if case == "drop_loss":
    gs.loss = None
if case == "keep_old_graph":
    old = gs.graph
if case == "sync_after_train":
    torch.cuda.synchronize()
if case == "thread_local":
    import torch.cuda.graphs as G
    orig = G.graph.__init__
    def init(self, cuda_graph, pool=None, stream=None, capture_error_mode="global"):
        orig(self, cuda_graph, pool=pool, stream=stream, capture_error_mode="thread_local")
    G.graph.__init__ = init
try:
    gs.step(mk(64))
    torch.cuda.synchronize()
    print(case, "ok")
except Exception as e:  # noqa: BLE001
    print(case, "FAILED", str(e).splitlines()[0][:100])
