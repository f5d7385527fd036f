import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
def timed(fn, iters=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for shape, d in (("gowalla", 64), ("yelp2018", 64), ("amazon-book", 64), ("gowalla", 128), ("yelp2018", 128), ("amazon-book", 128)):
    uid, iid, nu, ni = rbg.synth.make(shape)
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    n = nu + ni
    x, y0, y1 = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
    uw, iw = x[:nu].contiguous(), x[nu:].contiguous()
    out = {"shape": shape, "d": d, "split_rows": g.bins(d)["n_split_rows"]}
    for cs in (0, 1):
        rbg.set_option("col_split", cs)
        out[f"spmm_us_cs{cs}"] = round(timed(lambda: rbg.ops.spmm_raw(g, x, out=(y1 if cs else y0))), 2)
        o = torch.empty(n, d, device=dev); L = torch.empty(3, n, d, device=dev)
        out[f"prop_us_cs{cs}"] = round(timed(lambda: rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=o, layers=L)), 2)
        if cs == 0: ref = o.clone()
    out["max_abs_diff_spmm"] = float((y0 - y1).abs().max()); out["max_abs_diff_prop"] = float((ref - o).abs().max())
    rp, c, v = g.export_csr()
    import scipy.sparse as sp
    truth = sp.csr_matrix((v.astype(np.float64), c, rp), shape=(n, n)) @ x.double().cpu().numpy()
    out["max_err_vs_f64_cs0"] = float(np.abs(y0.cpu().numpy() - truth).max()); out["max_err_vs_f64_cs1"] = float(np.abs(y1.cpu().numpy() - truth).max())
    rbg.set_option("col_split", 0)
    print(json.dumps(out), flush=True)
