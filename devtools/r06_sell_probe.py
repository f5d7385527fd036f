#!/usr/bin/env python
"""r06: A/B of the column-slab kernel.  One JSON line per call: layer / propagation times (HIP graph replay of 20 calls) on the
quoted shapes, the fused shard layer of rank 0 at P = 4 / 8 (Amazon-Book), and SHA-256 of every result, so that two library
builds (RBGNN_LIB=...) can be compared bit for bit.  `python devtools/r06_sell_probe.py --tag base`."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_gnn_amd as rbg  # noqa: E402


def sha(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


def replay_us(fn, calls=20, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    out = []
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / (5 * calls))
    return sorted(out)[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--shapes", default="gowalla,yelp2018,amazon-book")
    ap.add_argument("--no-shards", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="re-plan every graph with this chunk (0: the creation-time plan)")
    ap.add_argument("--options", default="", help="key=value,... passed to rbg.set_option before the graphs are built")
    args = ap.parse_args()
    for kv in filter(None, args.options.split(",")):
        k, v = kv.split("=")
        rbg.set_option(k, int(v))
    dev = torch.device("cuda:0")
    rec = {"tag": args.tag, "lib": os.environ.get("RBGNN_LIB", "product"), "options": args.options, "chunk": args.chunk}
    d, K = 64, 3
    for name in args.shapes.split(","):
        uid, iid, nu, ni = rbg.synth.make(name)
        g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        n = nu + ni
        if args.chunk:
            g.plan_sell(32, args.chunk)
        gen = torch.Generator().manual_seed(7)
        e0 = (torch.randn(n, d, generator=gen) * 0.1).to(dev)
        y = torch.empty(n, d, device=dev)
        out = torch.empty(n, d, device=dev)
        layers = torch.empty(K, n, d, device=dev)
        r = {"status": g.sell_status(), "info": g.sell_info() if g.sell_status() == "planned" else None}
        r["layer_us"] = replay_us(lambda: rbg.ops.spmm_raw(g, e0, out=y))
        r["prop_us"] = replay_us(lambda: rbg.ops.lightgcn_forward_raw(g, e0[:nu], e0[nu:], K, out=out, layers=layers))
        rbg.ops.spmm_raw(g, e0, out=y)
        rbg.ops.lightgcn_forward_raw(g, e0[:nu], e0[nu:], K, out=out, layers=layers)
        torch.cuda.synchronize()
        r["layer_sha"], r["prop_sha"] = sha(y), sha(out)
        # fixed point A sqrt(deg) = sqrt(deg)
        deg = np.bincount(np.concatenate([uid, iid + nu]), minlength=n)
        root = torch.from_numpy(np.sqrt(deg).astype(np.float32)).to(dev)
        xr = root[:, None].expand(n, d).contiguous()
        yr = rbg.ops.spmm_raw(g, xr, out=y)
        r["fixed_point_rel_err"] = float(((yr - xr).abs() / root[:, None].clamp(min=1.0)).max())
        if name == "amazon-book" and not args.no_shards:
            sh = rbg.sharded
            be = sh.HipBackend(dev)
            rowptr, col, val = g.device_csr()
            for world in (4, 8):
                owner = sh.degree_striped_partition(uid, iid, nu, ni, world)
                plan = sh.plan_from_csr(rowptr, col, val, nu, owner, 0, world)
                prop = sh.ShardedPropagation(plan, be, transport="staged", fused=True)
                xc = e0[torch.as_tensor(np.concatenate([plan.owned, plan.halo_ids]), device=dev)].contiguous()
                yo = torch.empty(plan.n_owned, d, device=dev)
                r[f"shard_P{world}_us"] = replay_us(lambda: be.spmm(prop.g_cat, xc, yo, False))
                be.spmm(prop.g_cat, xc, yo, False)
                torch.cuda.synchronize()
                r[f"shard_P{world}_sha"] = sha(yo)
                r[f"shard_P{world}_status"] = prop.kernel_status()["cat"]
                r[f"shard_P{world}_err_vs_full"] = float((yo - rbg.ops.spmm_raw(g, e0)[torch.as_tensor(plan.owned, device=dev)]).abs().max())
        rec[name] = r
        del g
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
