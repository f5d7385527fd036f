"""InfoNCE (SGL calc_ssl_loss, sgl.py:176-209) forward+backward: torch formula vs the fused lse_rows path."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
F = torch.nn.functional
dev = torch.device("cuda:0")


def time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def torch_nce(a, p, c, tau):
    a, p, c = F.normalize(a, dim=1), F.normalize(p, dim=1), F.normalize(c, dim=1)
    return -torch.log(torch.exp((a * p).sum(1) / tau) / torch.exp(a.matmul(c.T) / tau).sum(1)).sum()


for name, n, d, B in (("gowalla users", 29859, 64, 2048), ("gowalla items", 40982, 64, 2048), ("amazon items d128", 91600, 128, 2048)):
    t1 = torch.randn(n, d, device=dev, requires_grad=True)
    t2 = torch.randn(n, d, device=dev, requires_grad=True)
    idx = torch.randint(1, n, (B,), device=dev)

    def run(fn):
        t1.grad = t2.grad = None
        fn(t1[idx], t2[idx], t2, 0.2).backward()

    us_t = time_us(lambda: run(torch_nce))
    us_f = time_us(lambda: run(rbg.SGL._info_nce))
    def run2():
        t1.grad = t2.grad = None
        rbg.ops.info_nce(t1, t2, idx, 0.2).backward()
    us_f2 = time_us(run2)
    torch.cuda.reset_peak_memory_stats()
    run(torch_nce)
    mem_t = torch.cuda.max_memory_allocated() / 1e6
    torch.cuda.reset_peak_memory_stats()
    run(rbg.SGL._info_nce)
    mem_f = torch.cuda.max_memory_allocated() / 1e6
    q = F.normalize(t1[idx].detach(), dim=1)
    c = F.normalize(t2.detach(), dim=1)
    us_fwd = time_us(lambda: rbg.ops.lse_rows_raw(q, c, 5.0, 5.0))
    print(json.dumps({"kind": "info_nce_fwd_bwd", "case": name, "B": B, "n": n, "d": d, "us_torch": round(us_t, 1),
                      "us_fused_denominator": round(us_f, 1), "us_fused_all": round(us_f2, 1), "peak_MB_torch": round(mem_t), "peak_MB_fused": round(mem_f),
                      "us_lse_forward_only": round(us_fwd, 1),
                      "fwd_TFLOPs": round(2.0 * B * n * d / us_fwd / 1e6, 1)}), flush=True)
