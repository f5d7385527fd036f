import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
name, graphed, dev_s, drop = sys.argv[1], sys.argv[2] == "1", sys.argv[3] == "1", float(sys.argv[4])
torch.manual_seed(0); np.random.seed(0)
m = getattr(rbg, name)({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "message_dropout": drop}, ds)
st = rbg.fused_stepper(m, graphed=graphed)
s = rbg.driver.BPRSampler(uid, iid, ni, batch_size=2048, seed=2020, device="cuda:0" if dev_s else None)
first_bad = None
for ep in range(2):
    m.train()
    for n, b in enumerate(s):
        b = {k: v.to("cuda:0") for k, v in b.items()}
        l = float(st.step(b))
        if not np.isfinite(l) and first_bad is None:
            first_bad = (ep, n, len(b["user_id"]))
            diag = {"loss": l, "sums": st.sums.tolist(), "coef_finite": bool(torch.isfinite(st.coef).all())}
            if hasattr(st, "e"):
                diag["e_finite"] = [bool(torch.isfinite(t).all()) for t in st.e]
                diag["g_finite"] = [bool(torch.isfinite(t).all()) for t in st.g]
                diag["inv_finite"] = [bool(torch.isfinite(t).all()) for t in st.inv]
            with torch.no_grad():
                pass
            m.eval(); m.train()
            diag["autograd_loss_same_batch(after the update)"] = float(m.calculate_loss(b))
            diag["next_step_loss"] = float(st.step(b))
            print(json.dumps(diag))
            break
    if first_bad:
        break
print(json.dumps({"args": sys.argv[1:], "first_bad(epoch, step, batch)": first_bad, "params_finite": bool(all(torch.isfinite(p).all() for p in m.parameters()))}))
