"""rbg_score_f32, uniform-phase kernel: sixteen 4-byte stores per lane and tile (option score_quads = 0) against four 16-byte stores
from the transposed product (1; 2 = non-temporal), interleaved; max difference against mode 0 and float64 on sampled rows."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
for (b, n, d) in ((4096, 40982, 64), (4096, 91600, 64), (2048, 40982, 128), (4096, 40960, 64), (1000, 29859, 64)):
    torch.manual_seed(0)
    u, it = torch.randn(b, d, device=dev), torch.randn(n, d, device=dev)
    ref = (u[:64].double() @ it.double().T)
    out = {"B": b, "n": n, "d": d}
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    base = None
    for rep in range(3):
        for mode in (0, 1, 2):
            rbg.set_option("score_quads", mode)
            for _ in range(3):
                s = rbg.score(u, it)
            torch.cuda.synchronize(); a.record()
            for _ in range(20):
                s = rbg.score(u, it)
            e.record(); torch.cuda.synchronize()
            out.setdefault(f"mode{mode}_us", []).append(round(a.elapsed_time(e) * 1e3 / 20, 1))
            if rep == 0:
                if mode == 0:
                    base = s.clone()
                out[f"mode{mode}_err_vs_f64"] = float((s[:64].double() - ref).abs().max())
                out[f"mode{mode}_max_diff_vs_mode0"] = float((s - base).abs().max())
            del s
    rbg.set_option("score_quads", 0)
    print(json.dumps(out), flush=True)
