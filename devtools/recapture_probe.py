"""GraphedStep on SGL: which sequence invalidates the re-capture of a new epoch's views?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "ref_test_inter.npz"))
uid, iid, nu, ni = z["uid"], z["iid"], int(z["n_users"]), int(z["n_items"])
ds = rbg.InteractionDataset(uid, iid, nu, ni)
cfg = {"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": int(os.environ.get("K", "2")), "reg_weight": 1e-4, "device_sampling": False}
g = torch.Generator().manual_seed(1)
mk = lambda b: {k: torch.randint(1, n, (b,), generator=g).to(dev) for k, n in (("user_id", nu), ("item_id", ni), ("neg_item_id", ni))}
BS = int(os.environ.get('BS', '500'))
for case in sys.argv[1:] or ["plain", "eager", "eager_same_size"]:
    torch.manual_seed(1); np.random.seed(7)
    m = rbg.SGL(cfg, ds)
    m.train()
    gs = rbg.GraphedStep(m, mk(BS), lr=1e-3)
    for _ in range(3):
        gs.step(mk(BS))
    if case == "eager":
        gs.eager_step(mk(123))
    if case == "eager_same_size":
        gs.eager_step(mk(BS))
    m.train()
    try:
        gs.step(mk(BS))
        torch.cuda.synchronize()
        print(case, "ok")
    except Exception as e:  # noqa: BLE001
        print(case, "FAILED", str(e).splitlines()[0][:100])
        torch.cuda.synchronize()
