#!/usr/bin/env python
"""rbg_score_f32 at the evaluation batch: the uniform-phase store stream (option score_uniform) against the shuffled one."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")

def time_us(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(out)[1]

for b, n, d in ((4096, 40982, 64), (4096, 91600, 64), (4096, 40982, 128), (4096, 38049, 256), (2048, 40982, 64), (4096, 40960, 64), (4096, 40981, 64)):
    u, it = torch.randn(b, d, device=dev), torch.randn(n, d, device=dev)
    s = torch.empty(b, n, device=dev)
    rec = {"B": b, "n": n, "d": d, "write_floor_us_at_6.3TBps": b * n * 4 / 6.3e12 * 1e6}
    for uni in (1, 0, 1, 0):
        rbg.set_option("score_uniform", uni)
        rec.setdefault(f"us_uniform{uni}", []).append(time_us(lambda: rbg._lib.lib.rbg_score_f32(u.data_ptr(), d, it.data_ptr(), d, s.data_ptr(), b, n, d, torch.cuda.current_stream().cuda_stream)))
    rbg.set_option("score_uniform", 1)
    if d == 64:
        for tiles in (20, 28, 40, 80):
            rbg.set_option("score_tiles", tiles)
            rec[f"us_tiles{tiles}"] = time_us(lambda: rbg._lib.lib.rbg_score_f32(u.data_ptr(), d, it.data_ptr(), d, s.data_ptr(), b, n, d, torch.cuda.current_stream().cuda_stream))
        rbg.set_option("score_tiles", 0)
    rec["fill_us"] = time_us(lambda: s.fill_(1.0))
    print(json.dumps(rec), flush=True)
