#!/usr/bin/env python
"""Column-sweep SpMM variants vs the binned kernel on one graph shape: parity, time per launch, and (under rocprofv3
--pmc) fabric traffic / L2 hit rate per variant.

  python devtools/sweep_probe.py [--workload gowalla] [--dim 64] [--variants a,b,...] [--iters 200]
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d DIR -o p -- python devtools/sweep_probe.py --pmc-run --iters 12
  python devtools/sweep_probe.py --summarize DIR1 DIR2 ...   (joins the counter CSVs with the manifest the run wrote)

A variant is name=threads:n_wg:range_kib:hot  (range_kib 0 = one range = plain row order; hot = LDS hot rows per class).
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT = ("t1_512=512:512:0:0,r4m_512=512:512:4096:0,r2m_512=512:512:2048:0,r1m_512=512:512:1024:0,r512k_512=512:512:512:0,"
           "t1_1024=1024:256:0:0,r2m_1024=1024:256:2048:0,r1m_1024=1024:256:1024:0,"
           "r2m_1024_h128=1024:256:2048:128,r2m_1024_h256=1024:256:2048:256,r1m_1024_h256=1024:256:1024:256,"
           "r2m_512_h64=512:512:2048:64,r2m_256=256:1024:2048:0")


def parse_variants(text):
    out = []
    for item in text.split(","):
        name, spec = item.split("=")
        f = spec.split(":")
        threads, n_wg, rk, hot = (int(x) for x in f[:4])
        kw = dict(threads=threads, n_wg=n_wg, range_bytes=(rk * 1024 if rk else 1 << 40), hot_rows_per_class=hot)
        if len(f) > 4:
            kw["piece_cost"] = float(f[4])
        if len(f) > 5:
            kw["part_frac"] = float(f[5])
        out.append((name, kw))
    return out


def summarize(dirs):
    man = None
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        mf = os.path.join(d, "manifest.json")
        if os.path.exists(mf):
            man = json.load(open(mf))
    if man is None:
        raise SystemExit("no manifest.json found")
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            recs = [r for r in csv.DictReader(open(f)) if "spmm_" in r.get("Kernel_Name", "")]
            by_counter = collections.defaultdict(list)
            for r in recs:
                by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
            for cname, lst in by_counter.items():
                lst.sort()
                pos = 0
                for name, count in man["blocks"]:
                    blk = lst[pos:pos + count]
                    pos += count
                    vals = [v for _, _, v in blk[len(blk) // 3:]]
                    if vals:
                        rows[name][cname] = sum(vals) / len(vals)
    out = {}
    for name, c in rows.items():
        rec = dict(c)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["traffic_MB"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            rec["l2_hit"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
        out[name] = rec
        print(name, json.dumps(rec))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--variants", default=DEFAULT)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--pmc-run", action="store_true", help="few launches per variant, write the dispatch manifest")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_probe.jsonl"))
    ap.add_argument("--manifest", default=None)
    ap.add_argument("--summarize", nargs="+")
    args = ap.parse_args()
    if args.summarize:
        res = summarize(args.summarize)
        json.dump(res, open(os.path.join(args.summarize[0], "pmc_by_variant.json"), "w"), indent=1)
        return

    import torch
    import recbole_gnn_amd as rbg
    from recbole_gnn_amd import sweep
    from oracle import coracle

    dev = torch.device("cuda:0")
    uid, iid, nu, ni = rbg.synth.make(args.workload)
    n, d = nu + ni, args.dim
    graph = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    rowptr, col, val = graph.export_csr()
    gen = torch.Generator().manual_seed(0)
    x_h = torch.randn(n, d, generator=gen)
    x, y = x_h.to(dev), torch.empty(n, d, device=dev)
    crow, ccol, cval = coracle.build_norm_csr(uid, iid, nu, ni)
    ref = coracle.spmm(crow, ccol, cval, x_h.numpy())
    b_layer, _ = rbg.synth.algorithmic_bytes(n, graph.nnz, d, 3)

    def time_us(fn, iters):
        for _ in range(5):
            fn()
        outs = []
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            outs.append(a.elapsed_time(b) * 1e3 / iters)
        return sorted(outs)[1]

    blocks, results = [], []
    spmm = lambda: rbg.ops.spmm_raw(graph, x, out=y)  # noqa: E731

    def run(name, stats=None):
        spmm()
        torch.cuda.synchronize()
        err = float(np.abs(y.cpu().numpy() - ref).max())
        if args.pmc_run:
            for _ in range(args.iters):
                spmm()
            torch.cuda.synchronize()
            blocks.append((name, args.iters + 1))
            rec = {"variant": name, "err": err}
        else:
            us = time_us(spmm, args.iters)
            rec = {"variant": name, "us": us, "frac": b_layer / (us * 1e-6) / 8e12, "err": err}
        if stats:
            rec.update(stats)
        results.append(rec)
        print(json.dumps(rec), flush=True)

    rbg.set_option("sweep", 0)
    run("binned")
    for key, value, default in (("col_split", 1, -1), ("spmm_unroll", 4, 8), ("nt_store", 0, 1)):
        rbg.set_option(key, value)  # launch-time options of the binned kernel, one at a time
        run(f"binned_{key}={value}")
        rbg.set_option(key, default)
    rbg.set_option("sweep", 1)
    for name, kw in (parse_variants(args.variants) if args.variants != "none" else []):
        t0 = time.time()
        try:
            lds = 150 * 1024 if kw["threads"] == 1024 else (72 * 1024 if kw["threads"] == 512 else 36 * 1024)
            plan = sweep.build_plan(rowptr, col, val, nu, d, lds_bytes=lds, **kw)
            sweep.attach(graph, plan)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"variant": name, "error": str(e)[:300]}), flush=True)
            continue
        st = plan.stats()
        st["plan_s"] = round(time.time() - t0, 2)
        run(name, st)
        if d == 64 and os.environ.get("SWEEP_PLAIN_TOO"):
            rbg.set_option("sweep_lean", 0)
            run(name + "_plain", st)
            rbg.set_option("sweep_lean", 1)
    sweep.detach(graph)
    if args.pmc_run:
        mf = args.manifest or os.path.join(ROOT, "gpurun_out", "manifest.json")
        os.makedirs(os.path.dirname(mf), exist_ok=True)
        json.dump({"blocks": blocks, "workload": args.workload, "dim": d}, open(mf, "w"))
    else:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "a") as f:
            for r in results:
                r["workload"], r["dim"] = args.workload, d
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
