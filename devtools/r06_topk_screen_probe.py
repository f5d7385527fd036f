#!/usr/bin/env python
"""r06: the fused top-k behind the bf16 screen (option "topk_screen", csrc/topk_screen.hip) against the exact passes: time per call
(HIP graph replay of 10 calls), the two results compared.  4096 users x the Gowalla item table, k = 10; fresh LightGCN tables and
tables with a few thousand rows 30x larger than the rest (what overfitting one batch leaves behind)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")


def replay_us(fn, calls=10, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            keep = fn()
    out = []
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / (3 * calls))
    del keep
    return sorted(out)[len(out) // 2]


def main():
    uid, iid, nu, ni = rbg.synth.make("gowalla")
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    for d in (64, 128):
        torch.manual_seed(0)
        model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": 3}, ds)
        with torch.no_grad():
            ue, ie = model.forward()
        ue2, ie2 = ue.clone(), ie.clone()
        g = torch.Generator().manual_seed(5)
        ue2[torch.randint(1, nu, (2048,), generator=g).to(dev)] *= 30.0
        ie2[torch.randint(1, ni, (4096,), generator=g).to(dev)] *= 30.0
        for state, (u, i) in (("fresh", (ue, ie)), ("skewed", (ue2, ie2))):
            for nb in (4096, 1024):
                users = torch.randint(1, nu, (nb,), generator=torch.Generator().manual_seed(1)).to(dev)
                rec = {"d": d, "users": nb, "k": 10, "tables": state}
                res = {}
                for scr in (0, 1, 0, 1):
                    rbg.set_option("topk_screen", scr)
                    v, ix = rbg.full_sort_topk(model.graph, u, i, users, 10)
                    res[scr] = (v.clone(), ix.clone())
                    rec.setdefault(f"screen{scr}_us", []).append(round(replay_us(lambda: rbg.full_sort_topk(model.graph, u, i, users, 10)), 2))
                rec["same_items"] = float((res[0][1] == res[1][1]).all(dim=1).float().mean())
                rec["max_rel_diff"] = float(((res[0][0] - res[1][0]).abs() / res[0][0].abs().clamp_min(1e-30)).max())
                print(json.dumps(rec), flush=True)
    rbg.set_option("topk_screen", 1)


if __name__ == "__main__":
    main()
