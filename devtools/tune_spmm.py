#!/usr/bin/env python
"""GPU measurement harness (one process, interleaved A/B): SpMM tuning sweep, propagation, scoring GEMM,
BiGNN layer and a copy-bandwidth calibration.  Writes JSON lines to gpurun_out/tune.jsonl.

  python devtools/tune_spmm.py [--shapes gowalla,amazon-book] [--quick] [--big]
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "tune.jsonl"), "a")


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    LOG.write(line + "\n")
    LOG.flush()


def time_us(fn, iters=200, warmup=20, rounds=3):
    for _ in range(warmup):
        fn()
    best = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(best)), float(np.min(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="gowalla,amazon-book")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--big", action="store_true", help="also the 1.3M-node shape (X = 333 MB > Infinity Cache)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    emit(kind="device", name=torch.cuda.get_device_name(0), cus=torch.cuda.get_device_properties(0).multi_processor_count)

    # copy bandwidth calibration (read + write of 1 GiB)
    a = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    med, mn = time_us(lambda: b.copy_(a), iters=20, warmup=3)
    emit(kind="copy_bw", gbps=2 * a.numel() * 4 / (mn * 1e-6) / 1e9, us=mn)
    del a, b

    # locality lever: the same Gowalla-sized graph with 8 communities, default launch plan vs communities pinned to XCDs
    nu, ni, e = rbg.synth.shape("gowalla")
    for p_in, layout in ((0.99, "contiguous"), (0.95, "contiguous"), (0.8, "contiguous"), (0.5, "contiguous"), (0.95, "striped")):
        uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=2020, n_blocks=8, p_in=p_in, layout=layout)
        part = rbg.synth.partition_of(nu, ni, 8, layout=layout)
        x = torch.randn(nu + ni, 64, device=dev)
        y = torch.empty_like(x)
        g0 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        g1 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev, xcd_part=part)
        t0, _ = time_us(lambda: rbg.ops.spmm_raw(g0, x, out=y), iters=100, warmup=10)
        t1, _ = time_us(lambda: rbg.ops.spmm_raw(g1, x, out=y), iters=100, warmup=10)
        bl, _ = rbg.synth.algorithmic_bytes(nu + ni, 2 * e, 64, 3)
        emit(kind="community_xcd", p_in=p_in, layout=layout, us_default_plan=t0, us_communities_on_xcds=t1, frac_default=bl / (t0 * 1e-6) / 8e12,
             frac_pinned=bl / (t1 * 1e-6) / 8e12)
        del g0, g1
    # ... and with the node ids scrambled, communities rediscovered by find_communities (xcd_part="auto")
    for p_in in (0.99, 0.95, 0.8):
        uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=2020, n_blocks=8, p_in=p_in, layout="contiguous")
        rs = np.random.default_rng(1)
        pu = np.concatenate([[0], rs.permutation(nu - 1) + 1])
        pi = np.concatenate([[0], rs.permutation(ni - 1) + 1])
        uid, iid = pu[uid], pi[iid]
        t0 = time.time()
        lab, cut, imb = rbg.find_communities(uid, iid, nu, ni)
        t_lp = time.time() - t0
        x = torch.randn(nu + ni, 64, device=dev)
        y = torch.empty_like(x)
        g0 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        g1 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev, xcd_part=lab)
        ta, _ = time_us(lambda: rbg.ops.spmm_raw(g0, x, out=y), iters=100, warmup=10)
        tb, _ = time_us(lambda: rbg.ops.spmm_raw(g1, x, out=y), iters=100, warmup=10)
        emit(kind="community_auto", p_in=p_in, found_cut=cut, imbalance=imb, label_propagation_s=t_lp, us_default_plan=ta,
             us_found_communities_on_xcds=tb)
        del g0, g1
    shapes = args.shapes.split(",") + (["g-1.3m"] if args.big else [])
    for name in shapes:
        t0 = time.time()
        uid, iid, nu, ni = rbg.synth.make(name)
        n, nnz = nu + ni, 2 * len(uid)
        b_layer, b_prop = rbg.synth.algorithmic_bytes(n, nnz, 64, 3)
        emit(kind="shape", name=name, n=n, nnz=nnz, gen_s=time.time() - t0, b_layer=b_layer)
        for label, flags in (("device", 0), ("host", rbg._lib.GRAPH_BUILD_ON_HOST)):
            ts = []
            for _ in range(3):
                t0 = time.time()
                gb = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev, flags=flags)
                torch.cuda.synchronize()
                ts.append(time.time() - t0)
                del gb
            emit(kind="graph_build", shape=name, builder=label, ms=min(ts) * 1e3, ms_all=[round(t * 1e3, 2) for t in ts])
        x = torch.randn(n, 64, device=dev)
        y = torch.empty_like(x)
        if args.quick:
            grid = [(64, 256, 4096)]
        else:
            grid = [(s, w, l) for s, w, l in itertools.product((32, 64, 128, 256), (256, 1024), (1024, 4096, 16384))
                    if w >= s]
        results = []
        for (s, w, l) in grid:
            rbg.set_tuning(s, w, l)
            for split in (4, 5, 3, 0):
                rbg.set_option("xcd_split", split)
                g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
                for unroll in ((8,) if args.quick else (4, 8)):
                    for nt in ((1,) if args.quick else (1, 0)):
                        rbg.set_option("spmm_unroll", unroll)
                        rbg.set_option("nt_store", nt)
                        med, mn = time_us(lambda: rbg.ops.spmm_raw(g, x, out=y), iters=100, warmup=10)
                        results.append((med, s, w, l, unroll, split, nt))
                        emit(kind="spmm", shape=name, short_max=s, wave_max=w, seg_len=l, unroll=unroll, xcd_split=split,
                             nt_store=nt, us=med, us_min=mn, gbps=b_layer / (med * 1e-6) / 1e9,
                             frac=b_layer / (med * 1e-6) / 8e12, bins=g.bins(64))
                del g
        results.sort()
        emit(kind="best", shape=name, top=results[:5])
        # natural order (no binning) for comparison, with the best tuning otherwise
        med0, s, w, l, unroll, split, nt = results[0]
        rbg.set_tuning(s, w, l)
        rbg.set_option("spmm_unroll", unroll)
        rbg.set_option("xcd_split", split)
        rbg.set_option("nt_store", nt)
        gn = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev, flags=rbg._lib.GRAPH_NATURAL_ORDER)
        med, mn = time_us(lambda: rbg.ops.spmm_raw(gn, x, out=y), iters=50, warmup=5)
        emit(kind="spmm_natural_order", shape=name, us=med)
        del gn
        g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        # torch.sparse CSR on the same GPU (rocSPARSE) as a vendor-library yardstick
        try:
            rp, col, val = g.export_csr()
            a = torch.sparse_csr_tensor(torch.from_numpy(rp).to(dev), torch.from_numpy(col.astype(np.int64)).to(dev),
                                        torch.from_numpy(val).to(dev), size=(n, n))
            med, mn = time_us(lambda: torch.matmul(a, x), iters=30, warmup=5)
            emit(kind="torch_sparse_csr_gpu", shape=name, us=med, gbps=b_layer / (med * 1e-6) / 1e9)
            del a
        except Exception as ex:  # noqa: BLE001
            emit(kind="torch_sparse_csr_gpu", shape=name, error=str(ex)[:200])
        # whole propagation (K = 3, fused mean), d = 64 and 128
        for d in (64, 128):
            uw, iw = torch.randn(nu, d, device=dev), torch.randn(ni, d, device=dev)
            out = torch.empty(n, d, device=dev)
            layers = torch.empty(3, n, d, device=dev)
            med, mn = time_us(lambda: rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=out, layers=layers), iters=100)
            bl, bp = rbg.synth.algorithmic_bytes(n, nnz, d, 3)
            emit(kind="propagation", shape=name, d=d, us=med, prop_per_s=1e6 / med, gbps_prop=bp / (med * 1e-6) / 1e9,
                 frac_layer=3 * bl / (med * 1e-6) / 8e12)
        # one training step of LightGCN (forward + BPR/reg loss + backward + Adam), batch 2048 (RecBole default)
        try:
            ds = rbg.InteractionDataset(uid, iid, nu, ni)
            model = rbg.LightGCN({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3}, ds)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            bu = torch.randint(1, nu, (2048,), device=dev)
            bp = torch.randint(1, ni, (2048,), device=dev)
            bn = torch.randint(1, ni, (2048,), device=dev)
            batch = {"user_id": bu, "item_id": bp, "neg_item_id": bn}

            def train_step():
                opt.zero_grad(set_to_none=True)
                loss = model.calculate_loss(batch)
                loss.backward()
                opt.step()

            med, mn = time_us(train_step, iters=50, warmup=5)
            emit(kind="train_step", shape=name, path="torch autograd + torch Adam", us=med, steps_per_s=1e6 / med)
            model.require_pow = True
            fused = rbg.FusedBPRAdam(model, lr=1e-3)
            med, mn = time_us(lambda: fused.step(batch), iters=50, warmup=5)
            emit(kind="train_step", shape=name, path="fused (5 C-ABI calls)", us=med, steps_per_s=1e6 / med)
            del model, opt
        except Exception as ex:  # noqa: BLE001
            emit(kind="train_step", shape=name, error=str(ex)[:300])
        # model-level numbers through the reference-shaped classes
        try:
            ds = rbg.InteractionDataset(uid, iid, nu, ni)
            ngcf = rbg.NGCF({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "hidden_size_list": [64, 64, 64]}, ds)
            with torch.no_grad():
                med, mn = time_us(lambda: ngcf.forward(), iters=30, warmup=3)
            emit(kind="ngcf_forward", shape=name, us=med, note="3 BiGNN layers + LeakyReLU + L2-norm, fused inference path, [N,256] out")
            users = torch.arange(1, 1 + 1024, device=dev) % nu
            with torch.no_grad():
                ngcf.full_sort_predict({"user_id": users})
                med, mn = time_us(lambda: ngcf.full_sort_predict({"user_id": users}), iters=10, warmup=2)
            emit(kind="ngcf_full_sort", shape=name, users=1024, us=med, users_per_s=1024 / (med * 1e-6))
            del ngcf
            lgc = rbg.LightGCN({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3}, ds)
            with torch.no_grad():
                lgc.full_sort_predict({"user_id": users})
                med, mn = time_us(lambda: lgc.full_sort_predict({"user_id": users}), iters=10, warmup=2)
            emit(kind="lightgcn_full_sort", shape=name, users=1024, us=med, users_per_s=1024 / (med * 1e-6))
            del lgc
            sgl = rbg.SGL({"device": "cuda:0", "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "type": "ED"}, ds)
            np.random.seed(0)
            t0 = time.time()
            sgl.train()
            torch.cuda.synchronize()
            emit(kind="sgl_graph_construction", shape=name, ms=(time.time() - t0) * 1e3, note="2 ED views: numpy sampling + device build")
            with torch.no_grad():
                med, mn = time_us(lambda: sgl.propagate_views(), iters=30, warmup=3)
            emit(kind="sgl_three_propagations", shape=name, us=med)
            del sgl
        except Exception as ex:  # noqa: BLE001
            emit(kind="models", shape=name, error=str(ex)[:300])
        # scoring GEMM
        for bsz in (1, 128, 4096):
            u = torch.randn(bsz, 64, device=dev)
            it = torch.randn(ni, 64, device=dev)
            med, mn = time_us(lambda: rbg.score(u, it), iters=30, warmup=3)
            med_t, _ = time_us(lambda: torch.matmul(u, it.T), iters=30, warmup=3)
            byt = 4 * (bsz * 64 + ni * 64 + bsz * ni)
            emit(kind="score", shape=name, B=bsz, us=med, us_torch=med_t, gbps=byt / (med * 1e-6) / 1e9,
                 tflops=2 * bsz * ni * 64 / (med * 1e-6) / 1e12)
        # NGCF layer 64 -> 64
        w1, w2 = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, 64, device=dev) * 0.1
        b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        yo = torch.empty(n, 64, device=dev)
        med, mn = time_us(lambda: rbg.ops.bignn_conv_raw(g, x, w1, b1, w2, b2, out=yo, leaky_norm=True), iters=50, warmup=5)
        emit(kind="bignn_layer", shape=name, us=med)
        del g


if __name__ == "__main__":
    main()
