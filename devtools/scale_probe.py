#!/usr/bin/env python
"""SGL config-#5-like scale on ONE GPU: synthetic power-law 10M users / 5M items / 200M interactions, d = 128
(scaled by --frac).  Reports build time, SpMM and propagation time, and a spot parity check of sampled rows
against float64 (a full oracle run at this size is too slow)."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
ap = argparse.ArgumentParser(); ap.add_argument("--frac", type=float, default=0.25); ap.add_argument("--d", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda:0")
nu, ni, e = int(10_000_000 * a.frac) + 1, int(5_000_000 * a.frac) + 1, int(200_000_000 * a.frac)
t0 = time.time(); uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=2020); t_gen = time.time() - t0
t0 = time.time(); g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev); torch.cuda.synchronize(); t_build = time.time() - t0
n, d = nu + ni, a.d
deg = np.bincount(np.concatenate([uid, iid + nu]), minlength=n)
x = torch.randn(n, d, device=dev); y = torch.empty_like(x)
def time_ms(fn, iters=3):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.time() - t) / iters * 1e3
t_spmm = time_ms(lambda: rbg.ops.spmm_raw(g, x, out=y))
rbg.set_option("col_split", 0); t_spmm_full = time_ms(lambda: rbg.ops.spmm_raw(g, x, out=y)); rbg.set_option("col_split", -1); rbg.ops.spmm_raw(g, x, out=y)
uw, iw = x[:nu], x[nu:]
out = torch.empty(n, d, device=dev); layers = torch.empty(3, n, d, device=dev)
t_prop = time_ms(lambda: rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=out, layers=layers))
# spot check: 200 random rows + the 20 heaviest rows of Y = A_hat X against float64 computed from the raw interactions
rng = np.random.default_rng(0)
rows = np.concatenate([rng.integers(0, n, 200), np.argsort(deg)[-20:]])
dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
xc = None
order_u = np.argsort(uid, kind="stable"); ptr_u = np.searchsorted(uid[order_u], np.arange(nu + 1))
order_i = np.argsort(iid, kind="stable"); ptr_i = np.searchsorted(iid[order_i], np.arange(ni + 1))
yh = y.cpu().numpy() if n * d < 6e8 else None
err = 0.0
for r in rows.tolist():
    if r < nu: nb = iid[order_u[ptr_u[r]:ptr_u[r + 1]]] + nu
    else: nb = uid[order_i[ptr_i[r - nu]:ptr_i[r - nu + 1]]]
    ref = (dis[r] * dis[nb])[:, None] * x[torch.from_numpy(nb).to(dev)].double().cpu().numpy()
    got = (yh[r] if yh is not None else y[r].cpu().numpy())
    err = max(err, float(np.abs(ref.sum(0) - got).max()))
bl, bp = rbg.synth.algorithmic_bytes(n, 2 * e, d, 3)
print(json.dumps(dict(kind="scale_probe", frac=a.frac, n=n, nnz=2 * e, d=d, gen_s=round(t_gen, 1), build_s=round(t_build, 2), max_deg=int(deg.max()),
                      bins=g.bins(d), spmm_ms=round(t_spmm, 2), spmm_ms_full_width=round(t_spmm_full, 2), propagation_ms=round(t_prop, 2), prop_per_s=round(1e3 / t_prop, 2),
                      frac_roofline=round(bl / (t_spmm * 1e-3) / 8e12, 4), spot_max_abs_err_vs_f64=err)))
