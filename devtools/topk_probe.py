"""Fused score+mask+top-k timing.  `python devtools/topk_probe.py B` = 5 calls (for rocprofv3);
`python devtools/topk_probe.py sweep` = wall time per call over B x topk_sample."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
uid, iid, nu, ni = rbg.synth.make("gowalla")
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
torch.manual_seed(0)
ua, it = torch.randn(nu, 64, device=dev) * 0.1, torch.randn(ni, 64, device=dev) * 0.1


def timed(users, reps=20):
    for _ in range(3):
        rbg.full_sort_topk(g, ua, it, users, 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rbg.full_sort_topk(g, ua, it, users, 10)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if len(sys.argv) > 1 and sys.argv[1] == "short":  # 24-entry lists (three workgroups per CU) against 48-entry lists, interleaved
    for B in (128, 1024, 4096):
        users = torch.randint(1, nu, (B,), device=dev)
        out = {"B": B}
        res = {}
        for rep in range(3):
            for mode in (1, 0):
                rbg.set_option("topk_short_lists", mode)
                out.setdefault("short_us" if mode else "long_us", []).append(round(timed(users), 1))
                res[mode] = rbg.full_sort_topk(g, ua, it, users, 10)
        out["same_items"] = bool(torch.equal(res[0][1], res[1][1]))
        out["same_scores"] = bool(torch.equal(res[0][0], res[1][0]))
        print(json.dumps(out), flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for B in (128, 1024, 4096):
        users = torch.randint(1, nu, (B,), device=dev)
        for sample in (1024, 2048, 4096, 8192, 16384):
            rbg.set_option("topk_sample", sample)
            print(json.dumps({"B": B, "topk_sample": sample, "us": round(timed(users), 1)}), flush=True)
else:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if len(sys.argv) > 2:
        rbg.set_option("topk_sample", int(sys.argv[2]))
    users = torch.randint(1, nu, (B,), device=dev)
    for _ in range(5):
        rbg.full_sort_topk(g, ua, it, users, 10)
    torch.cuda.synchronize()
