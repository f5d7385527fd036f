"""r06: rbg_infonce_f32 with gradients — the fp16 two-term form (option lse_f16 = 1: tiles from fp16 plane images by LDS-DMA, 2: fetched and split per workgroup, 3: 1 with the software-pipelined tile loop = default) against the bf16 three-term form (0): us per forward + backward,
and loss / gradient errors of BOTH against a float64 torch reference of sgl.py:191-199 (2048 batch rows, tau 0.2)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
import torch.nn.functional as F
dev = torch.device("cuda:0")


def ref64(t1, t2, idx, tau):
    a = t1.double().detach().requires_grad_(True)
    b = t2.double().detach().requires_grad_(True)
    u1, u2, allv = F.normalize(a[idx], dim=1), F.normalize(b[idx], dim=1), F.normalize(b, dim=1)
    pos = (u1 * u2).sum(1) / tau
    ttl = torch.logsumexp(u1 @ allv.T / tau, dim=1)
    loss = (ttl - pos).sum()
    loss.backward()
    return loss.detach(), a.grad, b.grad


cases = ((29858, 64, 0.2), (40982, 64, 0.2), (40982, 64, 0.05), (40982, 64, 1.0), (91600, 128, 0.2), (5000, 32, 0.2), (40982, 48, 0.2))
for n, d, tau in cases:
    B = 2048
    g = torch.Generator().manual_seed(n + d)
    scale = 0.1 if n != 5000 else 3.0
    t1 = (torch.randn(n, d, generator=g) * scale).to(dev).requires_grad_(True)
    t2 = (torch.randn(n, d, generator=g) * scale + 0.3 * t1.detach().cpu()).to(dev).requires_grad_(True)
    idx = torch.randint(1, n, (B,), generator=g).to(dev)
    l64, g1, g2 = ref64(t1, t2, idx, tau)
    rec = {"n": n, "d": d, "tau": tau}
    for mode in (0, 1, 2, 3, 0, 1, 2, 3):
        rbg.set_option("lse_f16", mode)
        def step():
            t1.grad = t2.grad = None
            loss = rbg.ops.info_nce(t1, t2, idx, tau)
            loss.backward()
            return loss
        for _ in range(3): step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): step()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 20)
        rec.setdefault(f"f16_{mode}_us", []).append(round(sorted(ts)[2], 1))
        loss = step()
        key = f"f16_{mode}"
        rec[key + "_loss_rel"] = float(((loss.double() - l64) / l64).abs())
        rec[key + "_g1_max_rel"] = float((t1.grad.double() - g1).abs().max() / g1.abs().max())
        rec[key + "_g2_max_rel"] = float((t2.grad.double() - g2).abs().max() / g2.abs().max())
        rec[key + "_g2_fro_rel"] = float((t2.grad.double() - g2).norm() / g2.norm())
        rec[key + "_finite"] = bool(torch.isfinite(t1.grad).all() and torch.isfinite(t2.grad).all())
    print(json.dumps(rec), flush=True)
rbg.set_option("lse_f16", 3)
