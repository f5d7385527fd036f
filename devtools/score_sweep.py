import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
def timed(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for B, n, d in ((4096, 40982, 64), (1024, 40982, 64), (4096, 91600, 64), (4096, 38049, 256)):
    u, it = torch.randn(B, d, device=dev), torch.randn(n, d, device=dev)
    row = {"B": B, "n": n, "d": d, "torch": round(timed(lambda: torch.matmul(u, it.T)), 1)}
    for tiles in (0, 8, 16, 27, 32, 41, 54, 81):
        rbg.set_option("score_tiles", tiles)
        row[f"t{tiles}"] = round(timed(lambda: rbg.score(u, it)), 1)
    rbg.set_option("score_tiles", 0)
    print(json.dumps(row), flush=True)
