#!/usr/bin/env python
"""bench.py — LightGCN propagations/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step is ONE propagation = LightGCN.forward (lightgcn.py:70-81): K = 3 SpMM layers E(k+1) = Â·E(k) plus
the layer mean, 64-d fp32, inputs resident in HBM before the timed region.

N = 1: BASELINE.json configs[1] — the Gowalla-shaped synthetic power-law graph (29,859 users / 40,982
       items / 1,027,370 interactions incl. the two PAD rows).
N > 1: the node-range sharded path (recbole-gnn_amd/sharded.py), trimmed halo exchange per layer (RCCL all_to_all over
       xGMI; single-stream or overlapped on a second stream — both are timed on the real group, the faster is kept).
       `python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run on 127.0.0.1; when the
       box has fewer than N GPUs the ranks share them and exchange through the host: "transport": "staged").
       --scaling strong (default): ONE fixed graph cut into node shards, no planted locality; value = global forwards/s.
         The graph is BASELINE.json's configuration for that GPU count: N = 2 the Gowalla shape (config #2), N = 4 the
         Amazon-Book shape (config #4), N = 8 config #5's shape (10 M users / 5 M items / 200 M interactions, 128-d, the
         LightGCN backbone of SGL; falls back to the 1.3 M-node shape at 64-d, reason printed, when the host cannot hold
         the generator's temporaries).  `same_workload_one_gpu` carries rank 0's single-GPU propagations/s of the same graph.
       --scaling weak: every rank owns one Gowalla-shaped block of a P-times larger graph with PLANTED locality: a
         fraction p_in (printed) of each user's interactions stays inside the rank's block.  value counts
         shard-propagations: one global forward over P shards = P propagations.
       Whichever mode is not the headline is measured too and reported under "other_scaling_mode".

Also reported: "roofline" (algorithmic bytes of one SpMM launch / its average duration from HIP events on
the launch stream, against the 8 TB/s HBM peak) and "cpu_baseline" (the oracle's C restatement of the
reference's CPU path, timed on this host's cores on a bounded sample — a reported baseline, not a target).
"""
import argparse
import json
import os
import sys
import time

# the CPU baseline's OpenMP threads stay where they start (reproducible: r02's figure moved 30-98 propagations/s run to run)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--clock-warmup", type=int, default=0,
                    help="extra untimed propagations in front of the W warm-up steps (counted in `warmup_effective`).  0 = the "
                         "headline is measured at exactly the flags given; the steady-clock figure is reported beside it either way")
    ap.add_argument("--eager", action="store_true", help="N = 1: issue the K timed steps from the host instead of replaying one HIP graph")
    ap.add_argument("--workload", default=None, help="default: gowalla at N = 1; BASELINE.json's configuration for N at N > 1")
    ap.add_argument("--dim", type=int, default=None, help="default 64 (128 for config #5's shape)")
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--p-in", type=float, default=0.95, help="N>1: fraction of interactions inside a rank's block")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--seed", type=int, default=2020)
    ap.add_argument("--no-extras", action="store_true", help="skip the extra (non-headline) measurements at N = 1")
    ap.add_argument("--shard-ceiling-only", action="store_true",
                    help="N = 1: print only the shard_compute_ceiling extra (per-rank layer products of P = 2 / 4 / 8 shards, alone on the GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N>1 headline: strong = ONE fixed graph (BASELINE.json's configuration for N: gowalla / amazon-book / "
                         "config5) cut into node shards, no planted locality (value = global forwards/s); weak = one "
                         "gowalla-shaped block per rank with planted locality p_in (value counts shard-propagations).  The other "
                         "one is reported alongside.")
    ap.add_argument("--partition", choices=["auto", "ranges", "striped"], default="auto",
                    help="strong scaling: contiguous nnz-balanced id ranges, or degree-striped (degree order dealt round-robin); "
                         "auto = the one with the smaller max over ranks of (nnz + halo rows)")
    ap.add_argument("--shard", choices=["rows", "columns", "hybrid"], default="rows",
                    help="N>1 headline: rows = node-range shards + halo exchange (the mode BASELINE.json's north_star names); columns = "
                         "feature-column shards (colsharded.py: the K layers exchange nothing); hybrid = 2 column groups x N/2 node "
                         "shards (hybrid.py: halos of d/2 floats inside a column group).  All are measured in a strong-scaling "
                         "run; this picks which one is `value`, the others are reported under `column_sharding` / `hybrid_sharding` / "
                         "`node_range_sharding`")
    ap.add_argument("--no-train-extra", action="store_true", help="N>1: skip the sharded SGL training-step measurement")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins a gloo group, rank 0 prints {\"launch_check\": world} (no GPU needed)")
    ap.add_argument("--no-secondary", action="store_true", help="N>1: skip the other scaling mode's measurement")
    ap.add_argument("--halo-push", choices=["auto", "on", "off"], default="auto",
                    help="N>1: also time the propagation with the halo PUSHED into the peers' exported tables (no collective; "
                         "sharded.PushExchange).  auto = only when the ranks share GPUs (the configuration the tests cover): between "
                         "different GPUs it has never run, and a faulting peer mapping would cost the run its JSON line")
    ap.add_argument("--transport", choices=["nccl", "staged"], default=None,
                    help="N>1 halo transport: RCCL all_to_all (default) or host-staged gloo send/recv (self-test: lets "
                         "several ranks share one GPU)")
    return ap.parse_args()


def xavier(rows, d, gen):
    bound = float(np.sqrt(6.0 / (rows + d)))
    return (torch.rand(rows, d, generator=gen, dtype=torch.float32) * 2 - 1) * bound


def cpu_baseline(uid, iid, nu, ni, uw, iw, k_layers, budget_s):
    """The oracle's C restatement of torch_sparse's spmm_cpu loop (OpenMP over rows), all host cores."""
    from oracle import coracle
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    n, d = nu + ni, uw.shape[1]
    buffers = (np.empty((k_layers + 1, n, d), dtype=np.float32), np.empty((n, d), dtype=np.float32))  # reused: no page faults in the loop
    rowptr, col, val = (np.ascontiguousarray(rowptr, dtype=np.int64), np.ascontiguousarray(col, dtype=np.int64),
                        np.ascontiguousarray(val, dtype=np.float32))
    coracle.lightgcn_forward(rowptr, col, val, uw, iw, k_layers, buffers=buffers)  # warm-up
    # The row-parallel loop does not scale across this kind of host (2-socket EPYC 9575F, r01: 16 threads 66.7 prop/s,
    # 64 threads 29.1, 128 threads 13.6 — first-touch NUMA placement + dynamic-schedule contention), so the baseline
    # runs at the thread count that is fastest on a short probe, and reports that count as `cores`.
    max_threads = coracle.num_threads()
    probe = {}
    for t in sorted({min(t, max_threads) for t in (4, 8, 16, 24, 32, 48, 64, 128, max_threads)}):
        coracle.set_num_threads(t)
        coracle.lightgcn_forward(rowptr, col, val, uw, iw, k_layers, buffers=buffers)
        reps, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 0.5:  # half a second each: a 3-call burst reads 2-4x the sustained rate on this host
            coracle.lightgcn_forward(rowptr, col, val, uw, iw, k_layers, buffers=buffers)
            reps += 1
        probe[t] = reps / (time.perf_counter() - t1)
    # The same arithmetic with thread-owned row blocks and first-touch placement (oracle/rbg_oracle.c ora_numa_*; effective with
    # OMP_PROC_BIND / OMP_PLACES set before the first OpenMP region of the process — main() sets them when they are unset): what the
    # reference's CPU path could reach on this box if it were NUMA-aware.  Reported beside the plain loop, never instead of it.
    numa_probe = {}
    for t in sorted({min(t, max_threads) for t in (16, 32, 64, 128, max_threads)}):
        coracle.set_num_threads(t)
        try:
            nf = coracle.NumaForward(rowptr, col, val, nu, ni, d, k_layers)
        except MemoryError:
            break
        nf(uw, iw, want_result=False)
        reps, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 0.5:
            nf(uw, iw, want_result=False)
            reps += 1
        numa_probe[t] = reps / (time.perf_counter() - t1)
        nf.close()
    best = max(probe, key=probe.get)
    coracle.set_num_threads(best)
    samples, total_reps, total_s = [], 0, 0.0
    for _ in range(3):  # median of three samples of budget / 3 each
        reps, t0 = 0, time.perf_counter()
        while True:
            coracle.lightgcn_forward(rowptr, col, val, uw, iw, k_layers, buffers=buffers)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s / 3 or reps >= 200:
                break
        samples.append(reps / el)
        total_reps, total_s = total_reps + reps, total_s + el
    coracle.set_num_threads(max_threads)
    return {"value": sorted(samples)[1], "unit": "propagations/s", "cores": best, "kind": "port",
            "cores_note": f"best of the thread probe = {best} of the {os.cpu_count()} logical CPUs of this host: the OpenMP row loop of the "
                          "restatement stops scaling past one CCD (see thread_probe_prop_per_s), so this is the reference's CPU path on "
                          f"{best} cores, not on all of them",
            "samples_prop_per_s": [round(v, 1) for v in samples],
            "thread_probe_prop_per_s": {str(k): round(v, 1) for k, v in probe.items()},
            "numa_aware_variant": {"thread_probe_prop_per_s": {str(k): round(v, 1) for k, v in numa_probe.items()},
                                   "best_prop_per_s": round(max(numa_probe.values()), 1) if numa_probe else None,
                                   "best_threads": max(numa_probe, key=numa_probe.get) if numa_probe else None,
                                   "note": "same arithmetic (bit-identical), static nnz-balanced row blocks per thread, col / val / layers first "
                                           "touched by their thread (oracle/rbg_oracle.c ora_numa_*): context for the baseline, not the baseline"},
            "thread_pinning": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
            "sample": f"median of 3 samples, {total_reps} full propagations of the same workload in {total_s:.1f} s "
                      f"(oracle/rbg_oracle.c, gcc -O3 + AVX2 clone, OpenMP dynamic rows, buffers reused; {os.cpu_count()} logical cpus visible)"}


def cpu_baseline_torch_sparse(uid, iid, nu, ni, uw, iw, k_layers, budget_s):
    """Second stand-in for the reference's CPU path: ``torch.sparse`` CSR ``A @ X`` per
    layer + the stack/mean of lightgcn.py:77-78, torch's own intra-op threads (torch_sparse itself is not installable)."""
    from oracle import coracle
    rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
    n = nu + ni
    a = torch.sparse_csr_tensor(torch.from_numpy(np.asarray(rowptr, dtype=np.int64)), torch.from_numpy(np.asarray(col, dtype=np.int64)),
                                torch.from_numpy(np.asarray(val, dtype=np.float32)), size=(n, n))
    e0 = torch.cat([torch.from_numpy(uw), torch.from_numpy(iw)])

    def prop():
        x, layers = e0, [e0]
        for _ in range(k_layers):
            x = a @ x
            layers.append(x)
        return torch.mean(torch.stack(layers, dim=1), dim=1)

    prop()
    reps, t0 = 0, time.perf_counter()
    while True:
        prop()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    return {"value": reps / el, "unit": "propagations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} full propagations in {el:.1f} s: torch.sparse CSR A @ X x {k_layers} + stack/mean, "
                      f"torch {torch.__version__} CPU, {torch.get_num_threads()} intra-op threads"}


def measured_hbm_peak(dev):
    """what this box's HBM delivers to a plain streaming kernel, beside the 8 TB/s of the data sheet — a device-side
    stream triad a = b + s c and a device-to-device copy over 1 GiB arrays (4x the Infinity Cache), torch's own elementwise
    kernels (plumbing: nothing of the product is measured here), best of 5 after 2 warm-ups, HIP events."""
    n = 1 << 28  # floats: 1 GiB per array
    try:
        a, b, c = torch.empty(n, device=dev), torch.ones(n, device=dev), torch.ones(n, device=dev)
    except RuntimeError as ex:  # noqa: BLE001
        return {"error": str(ex)[:120]}

    def best(fn, nbytes):
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        return nbytes / (min(ts) * 1e-3) / 1e9

    out = {"triad_GBps": best(lambda: torch.add(b, c, alpha=0.5, out=a), 3 * 4 * n), "copy_GBps": best(lambda: a.copy_(b), 2 * 4 * n),
           "array_GiB": 1.0, "how": "torch.add(b, c, alpha, out=a) and a.copy_(b) on 1 GiB fp32 arrays, best of 5, HIP events"}
    del a, b, c
    torch.cuda.empty_cache()
    return out


# The gather path's own ceiling (r05, devtools/microbench/gather_rate.hip -> profiles/r05_gather_rate.jsonl): random 128-byte row
# gathers — the propagation's access pattern with nothing else of it — sustain one 1 KiB wave-load per 17-19 clocks per CU from an
# L2-resident table (55-60 B/clk/CU, ~30 TB/s over the chip at 2.1 GHz), 53 clocks from a 16 MB table, 77 from HBM.  A layer gathers
# nnz x d x 4 bytes through that path whatever the kernel does with them.
GATHER_PATH_GBPS = 30_000.0


def gather_path_cap(n, nnz, d, k_layers):
    b_layer, _ = rbg_algorithmic_bytes(n, nnz, d, k_layers)
    t_min = nnz * d * 4 / (GATHER_PATH_GBPS * 1e9)
    return {"frac": b_layer / t_min / 1e9 / HBM_PEAK_GBPS, "layer_us_min": t_min * 1e6,
            "basis": "a layer gathers nnz x d x 4 B of 128-byte rows through the vector memory path, which sustains ~30 TB/s of random row "
                     "gathers from an L2-resident table (1 KiB per 17-19 clk per CU; profiles/r05_gather_rate.jsonl): the cap of ANY "
                     "gather formulation at this shape; larger tables are capped lower (53 clk at 16 MB, 77 from HBM)"}


def rbg_algorithmic_bytes(n, nnz, d, k_layers):
    b_layer = 4 * (n + 1) + 8 * nnz + 8 * n * d
    return b_layer, k_layers * b_layer + 4 * n * d * (k_layers + 2)


def extras_n1(rbg, graph, uid, iid, nu, ni, d, k_layers, dev):
    """Non-headline measurements on the same graph (each: median of 3 x 50 iterations, HIP events)."""
    def time_us(fn, iters=50, warm=5):
        for _ in range(warm):
            fn()
        out = []
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) * 1e3 / iters)
        return sorted(out)[1]

    n = nu + ni
    x, y = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
    ex = {"spmm_single_layer_us": time_us(lambda: rbg.ops.spmm_raw(graph, x, out=y)),  # rbg_spmm_f32: row-major in, row-major out
          "spmm_single_layer_kernel": graph.spmm_kernel_name(d)}
    b_layer, _ = rbg.synth.algorithmic_bytes(n, 2 * len(uid), d, k_layers)
    ex["spmm_single_layer_roofline_frac"] = b_layer / (ex["spmm_single_layer_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
    # the headline loop re-uses one buffer set (~110 MB: it lives in the 256 MB Infinity Cache).  Rotating
    # over enough independent sets (own graph handle, E0, layer and output buffers) that the footprint exceeds the cache
    # gives the HBM-resident figure.
    set_mb = (4 * (n + 1) + 8 * graph.nnz + 16 * n + (k_layers + 3) * n * d * 4) / 1e6  # CSR + row descriptors + E0, layers, out
    n_sets = max(2, int(np.ceil(320.0 / set_mb)))
    sets = []
    for _ in range(n_sets):
        sets.append((rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev), torch.randn(nu, d, device=dev) * 0.1,
                     torch.randn(ni, d, device=dev) * 0.1, torch.empty(n, d, device=dev),
                     torch.empty(max(k_layers, 1), n, d, device=dev)))
    turn = [0]

    def rotated():
        gq, uq, iq, oq, lq = sets[turn[0] % n_sets]
        turn[0] += 1
        rbg.ops.lightgcn_forward_raw(gq, uq, iq, k_layers, out=oq, layers=lq)

    def hot():
        gq, uq, iq, oq, lq = sets[0]
        rbg.ops.lightgcn_forward_raw(gq, uq, iq, k_layers, out=oq, layers=lq)

    ex["propagation_hot_us"] = time_us(hot, iters=100)
    prev_sell = rbg.get_option("sell")
    try:
        rbg.set_option("sell", 0)
        ex["propagation_binned_kernel_us"] = time_us(hot, iters=100)  # the same propagation with the column-slab path off
    finally:
        rbg.set_option("sell", prev_sell)
    ex["propagation_rotated_us"] = time_us(rotated, iters=100)
    ex["propagation_rotated_sets"] = n_sets
    ex["propagation_rotated_footprint_MB"] = round(n_sets * set_mb, 1)
    del sets
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": k_layers,
                          "require_pow": True}, ds)
    g = torch.Generator().manual_seed(1)
    batch = {"user_id": torch.randint(1, nu, (2048,), generator=g).to(dev),
             "item_id": torch.randint(1, ni, (2048,), generator=g).to(dev),
             "neg_item_id": torch.randint(1, ni, (2048,), generator=g).to(dev)}
    # evaluation first, on the freshly initialised model's propagated tables (the state every top-k / scoring figure of DESIGN.md is
    # quoted on); the training-step extras below then overfit ONE batch for ~ 300 steps, and the top-k is timed once more on those
    # tables (a few thousand rows 30x larger than the rest: more candidates pass the pre-pass threshold)
    users = torch.randint(1, nu, (4096,), generator=g).to(dev)
    with torch.no_grad():
        model.full_sort_topk({"user_id": users}, 10)
        ex["full_sort_topk_us(4096 users, k 10, history masked)"] = time_us(lambda: model.full_sort_topk({"user_id": users}, 10), iters=10, warm=2)
        try:  # the same call replayed from a HIP graph: the four launches without the host's share (workspace allocation, ctypes)
            uall, iall = model.restore_user_e, model.restore_item_e
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rbg.full_sort_topk(model.graph, uall, iall, users, 10)
            torch.cuda.current_stream().wait_stream(side)
            tk_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(tk_graph):
                for _ in range(10):
                    tk_keep = rbg.full_sort_topk(model.graph, uall, iall, users, 10)
            ex["full_sort_topk_graph_replay_us"] = time_us(tk_graph.replay, iters=3, warm=1) / 10
            del tk_graph, tk_keep
        except Exception as e:  # noqa: BLE001  (an extra: never fail the line)
            ex["full_sort_topk_graph_replay_us"] = f"not captured: {type(e).__name__}"
        iw_all = model.restore_item_e if model.restore_item_e is not None else model.forward()[1]
        uq = torch.randn(4096, d, device=dev)
        ex["score_gemm_us(4096 users x all items)"] = time_us(lambda: rbg.score(uq, iw_all), iters=20, warm=3)
        bq = 4096
        sc_bytes = 4 * (bq * d + ni * d + bq * ni)  # the scoring GEMM at d = 64 is bound by writing S
        sc_us = ex["score_gemm_us(4096 users x all items)"]
        ex["score_roofline"] = {"bound": "hbm", "bytes": sc_bytes, "achieved": sc_bytes / (sc_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": sc_bytes / (sc_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                "note": "bytes = 4 (B d + n d + B n): the [B, n] score matrix written once"}
        tk_us = ex["full_sort_topk_us(4096 users, k 10, history masked)"]
        if isinstance(ex.get("full_sort_topk_graph_replay_us"), float):
            tk_us = min(tk_us, ex["full_sort_topk_graph_replay_us"])  # (the device's share is what the roofline prices)
        tk_flop = 2.0 * bq * ni * d  # the product the top-k needs
        screened = bool(rbg.get_option("topk_screen")) and d <= 128
        # r06: behind the screen (csrc/topk_screen.hip) every pair costs ONE bf16 product (+ 1/4 of one for the bound): the peak is the
        # dense bf16 MFMA peak; the exact passes ran three products on split operands per fp32-grade product (peak / 3)
        tk_peak = 2500.0 if screened else 2500.0 / 3
        ex["topk_roofline"] = {"bound": "mfma", "flop": tk_flop, "achieved": tk_flop / (tk_us * 1e-6) / 1e12, "peak": tk_peak, "unit": "TFLOP/s",
                               "frac": tk_flop / (tk_us * 1e-6) / 1e12 / tk_peak, "screened": screened,
                               "note": ("one bf16 x bf16 product per (user, item) pair with a rigorous error bound screens the call, the ~ 75 survivors "
                                        "per user are rescored exactly in fp32 (option topk_screen); flops 2 B n d against the dense bf16 MFMA peak; "
                                        "the call is five launches (image, pre-pass, thresholds, screen, rescoring merge) of which the screen is ~ 40 %"
                                        if screened else
                                        "fp32-equivalent flops 2 B n d against a third of the dense bf16 MFMA peak (each product is three bf16 "
                                        "MFMA products on split operands)") + "; writes nothing but [B, k]: the [B, n] matrix never exists"}
        if screened:  # the exact passes beside it (what r05 measured as the top-k)
            rbg.set_option("topk_screen", 0)
            try:
                ex["full_sort_topk_exact_passes_us"] = time_us(lambda: model.full_sort_topk({"user_id": users}, 10), iters=10, warm=2)
            finally:
                rbg.set_option("topk_screen", 1)
    # r06: one InfoNCE half (sgl.py:191-199: 2048 batch rows against all item rows, value + both table gradients, one C call) — the fp16
    # two-term form of the gradient launches (option lse_f16) and the bf16 three-term form beside it
    try:
        import ctypes as _ct
        from recbole_gnn_amd._lib import lib as _lib_c, check as _check, c_vp as _vp, c_i64 as _i64
        nb_, tau_ = 2048, 0.2
        gen_n = torch.Generator(device=dev).manual_seed(3)
        t1n = torch.randn(ni, d, device=dev, generator=gen_n) * 0.1
        t2n = 0.5 * t1n + torch.randn(ni, d, device=dev, generator=gen_n) * 0.1
        idxn = torch.randint(1, ni, (nb_,), device=dev, generator=gen_n)
        g1n, g2n, lossn = torch.zeros_like(t1n), torch.zeros_like(t2n), torch.zeros((), device=dev)
        nbytes = _i64()
        _check(_lib_c.rbg_infonce_workspace(nb_, ni, d, _ct.byref(nbytes)))
        wsn = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=dev)
        st_ = _vp(torch.cuda.current_stream(dev).cuda_stream)

        def nce_call():
            _check(_lib_c.rbg_infonce_f32(_vp(t1n.data_ptr()), _vp(t2n.data_ptr()), ni, d, _vp(idxn.data_ptr()), nb_, tau_, 1.0,
                                          _vp(lossn.data_ptr()), _vp(g1n.data_ptr()), _vp(g2n.data_ptr()), _vp(wsn.data_ptr()), st_))

        mode0 = int(rbg.get_option("lse_f16"))
        nce_us = time_us(nce_call, iters=20, warm=3)
        ex[f"infonce_us({nb_} x {ni} x {d}, value + both gradients)"] = nce_us
        rbg.set_option("lse_f16", 0)
        try:
            ex["infonce_bf16_three_term_form_us"] = time_us(nce_call, iters=20, warm=3)
        finally:
            rbg.set_option("lse_f16", mode0)
        terms = 3 if mode0 and d % 4 == 0 else 6
        nce_flop = 4 * 2.0 * nb_ * ni * d * terms  # four [B, n] x d products (scores twice, dA, dC), each `terms` matrix-core products
        ex["infonce_roofline"] = {"bound": "mfma", "flop": nce_flop, "achieved": nce_flop / (nce_us * 1e-6) / 1e12, "peak": 2500.0,
                                  "unit": "TFLOP/s", "frac": nce_flop / (nce_us * 1e-6) / 1e12 / 2500.0, "terms_per_product": terms,
                                  "note": "matrix-core flops of the call (4 products of 2 B n d, each as 3 fp16 products on two-term operands — "
                                          "6 bf16 products in the three-term form) against the dense 16-bit MFMA peak; the call is 8 launches, "
                                          "the two lse_tile_kernel launches are ~ 70 % of it (matrix pipe busy 41 % / 34 % in them: "
                                          "profiles/r06_lse_f16_pmc.json)"}
        del t1n, t2n, g1n, g2n, wsn
    except Exception as e:  # noqa: BLE001  (an extra: never fail the line)
        ex["infonce_us"] = f"not measured: {type(e).__name__}: {e}"
    fused = rbg.FusedBPRAdam(model, lr=1e-3)
    ex["train_step_fused_us(batch 2048, fwd + BPR + bwd + Adam)"] = time_us(lambda: fused.step(batch))
    rbg.set_option("deterministic", 1)  # the same step with ordered row scatters and fixed-point sums (bit-stable; csrc/ordered.h)
    try:
        ex["train_step_fused_deterministic_us"] = time_us(lambda: fused.step(batch))
    finally:
        rbg.set_option("deterministic", 0)
    with torch.no_grad():
        model.restore_user_e = model.restore_item_e = None
        model.full_sort_topk({"user_id": users}, 10)
        ex["full_sort_topk_after_overfitting_one_batch_us"] = time_us(lambda: model.full_sort_topk({"user_id": users}, 10), iters=10, warm=2)
    t0 = time.perf_counter()
    rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    torch.cuda.synchronize()
    ex["graph_build_device_ms"] = (time.perf_counter() - t0) * 1e3
    # The other shapes a roofline figure is quoted for, each with its measured fabric traffic and L2 hit rate (committed PMC
    # passes, profiles/traffic.json): the scale the north_star names ("~1.3 M nodes": X = 333 MB is beyond every cache),
    # the Amazon-Book and Yelp2018 shapes, and the Gowalla shape at d = 128 (column-half kernel).
    try:
        ref = {}
        for name, dd in (("g-1.3m", d), ("amazon-book", d), ("yelp2018", d), ("gowalla", 2 * d)):
            gu, gi, gnu, gni = rbg.synth.make(name)
            gg = rbg.GraphHandle.from_interactions(gu, gi, gnu, gni, device=dev)
            gn = gnu + gni
            gx, gy = torch.randn(gn, dd, device=dev), torch.empty(gn, dd, device=dev)
            big = gn > 1_000_000
            us = time_us(lambda: rbg.ops.spmm_raw(gg, gx, out=gy), iters=10 if big else 50, warm=2 if big else 5)
            gb, _ = rbg.synth.algorithmic_bytes(gn, gg.nnz, dd, k_layers)
            kern = gg.spmm_kernel_name(dd)  # (the plain layer measured here: rbg_spmm_f32)
            traffic, l2_hit = traffic_from_profiles(name, dd, kern)
            key = name if dd == d else f"{name}:d{dd}"
            ex[key] = {"nodes": gn, "nnz": gg.nnz, "us": us, "frac": gb / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                       "l2_hit": l2_hit, "algorithmic_bytes_per_layer": gb, "kernel": kern,
                       # what the 0.1-0.2 is read against: the gather path's ceiling at an L2-resident table, and — with the fabric
                       # traffic this kernel has at this shape — the fabric's 7.3 TB/s (profiles/traffic.json, DESIGN results log 6.10)
                       "cap": {"gather_path_frac": gather_path_cap(gn, gg.nnz, dd, k_layers)["frac"],
                               "fabric_frac": (gb / (traffic / 7.3e12) / 1e9 / HBM_PEAK_GBPS) if traffic else None}}
            go, gl = torch.empty(gn, dd, device=dev), torch.empty(max(k_layers, 1), gn, dd, device=dev)
            pus = time_us(lambda: rbg.ops.lightgcn_forward_raw(gg, gx[:gnu], gx[gnu:], k_layers, out=go, layers=gl),
                          iters=5 if big else 20, warm=1 if big else 3)
            ex[key].update(propagation_us=pus, propagation_kernel=gg.propagation_kernel_name(dd), sell_status=gg.sell_status(),
                           propagation_frac=(k_layers * gb) / (pus * 1e-6) / 1e9 / HBM_PEAK_GBPS)
            # a parity figure for every shape that is timed: the fixed point A sqrt(deg) = sqrt(deg) on every
            # row through the same kernel — relative to sqrt(deg); the oracle-checked tests of these
            # shapes are tests/test_gpu_sell_native.py::test_parity_at_scale_through_the_plan
            deg_all = np.bincount(np.concatenate([gu, gi + gnu]), minlength=gn)
            root = torch.from_numpy(np.sqrt(deg_all).astype(np.float32)).to(dev)
            xr = root[:, None].expand(gn, dd).contiguous()
            yr = rbg.ops.spmm_raw(gg, xr, out=gy)
            ex[key]["fixed_point_rel_err"] = float(((yr - xr).abs() / root[:, None].clamp(min=1.0)).max())
            del xr, yr, root
            if name in ("g-1.3m", "amazon-book"):
                ref[name] = 1e6 / pus
            del go, gl
            del gg, gx, gy
        ex["strong_scaling_reference(single-GPU propagations/s of the --scaling strong graphs)"] = ref
    except Exception as e:  # noqa: BLE001
        ex["shapes_error"] = str(e)[:200]
    # the node-range shards' compute ceiling on this one GPU: per-rank layer product alone, P = 2 / 4 / 8
    try:
        ex["shard_compute_ceiling"] = shard_compute_ceiling(rbg, dev, d, time_us)
    except Exception as e:  # noqa: BLE001
        ex["shard_compute_ceiling_error"] = str(e)[:300]
    # the other two model families of the path (NGCF bi-interaction layers, SGL views + InfoNCE), one training step each
    try:
        cfg = {"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": k_layers, "reg_weight": 1e-5,
               "hidden_size_list": [d] * 3, "message_dropout": 0.0}
        ngcf = rbg.NGCF(cfg, ds)
        ngcf.train()
        opt = torch.optim.Adam(ngcf.parameters(), lr=1e-3)

        def ngcf_step():
            opt.zero_grad(set_to_none=True)
            ngcf.calculate_loss(batch).backward()
            opt.step()

        ex["ngcf_train_step_us(3 BiGNN layers, batch 2048)"] = time_us(ngcf_step, iters=20, warm=3)
        with torch.no_grad():
            ex["ngcf_forward_us"] = time_us(lambda: ngcf.forward(), iters=20, warm=3)
        gs = rbg.GraphedStep(ngcf, batch, lr=1e-3)  # the same step captured once into a HIP graph (train.py)
        ex["ngcf_train_step_graphed_us"] = time_us(lambda: gs.step(batch), iters=20, warm=3)
        del opt, gs
        # the same step without autograd: per layer one forward and one backward library call, the loss on the rows of the
        # concatenation by rbg_concat_bpr_*, torch's fused Adam (train.FusedNGCFAdam), eager and replayed from a HIP graph
        fs = rbg.FusedNGCFAdam(ngcf, lr=1e-3)
        ex["ngcf_fused_step_us"] = time_us(lambda: fs.step(batch), iters=20, warm=3)
        fg = rbg.FusedNGCFAdam(ngcf, lr=1e-3, graphed=True)
        ex["ngcf_fused_step_graphed_us"] = time_us(lambda: fg.step(batch), iters=20, warm=4)
        del ngcf, fs, fg
        np.random.seed(0)
        sgl = rbg.SGL({"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": k_layers, "type": "ED",
                       "drop_ratio": 0.1, "ssl_tau": 0.2, "ssl_weight": 0.05, "reg_weight": 1e-4}, ds)
        sgl.train()
        opt = torch.optim.Adam(sgl.parameters(), lr=1e-3)

        def sgl_step():
            opt.zero_grad(set_to_none=True)
            sgl.calculate_loss(batch).backward()
            opt.step()

        ex["sgl_train_step_us(ED views, InfoNCE, batch 2048)"] = time_us(sgl_step, iters=10, warm=2)
        gs = rbg.GraphedStep(sgl, batch, lr=1e-3)
        ex["sgl_train_step_graphed_us"] = time_us(lambda: gs.step(batch), iters=10, warm=2)
        del gs
        # the same step without autograd (train.FusedSGLAdam: three propagations, rbg_concat_bpr_*, rbg_infonce_f32, three
        # backward chains, EmbLoss, fused Adam), eager and replayed from a HIP graph
        fs = rbg.FusedSGLAdam(sgl, lr=1e-3)
        ex["sgl_fused_step_us"] = time_us(lambda: fs.step(batch), iters=10, warm=3)
        fg = rbg.FusedSGLAdam(sgl, lr=1e-3, graphed=True)
        ex["sgl_fused_step_graphed_us"] = time_us(lambda: fg.step(batch), iters=10, warm=4)
        del fs, fg
        gc = {}
        for mode, flag in (("device_sampling", True), ("numpy_sampling(reference calls)", False)):
            sgl.device_sampling = flag
            sgl.graph_construction()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sgl.graph_construction()
            torch.cuda.synchronize()
            gc[mode] = (time.perf_counter() - t0) / 3 * 1e3
        ex["sgl_graph_construction_ms(two ED views)"] = gc
        del sgl
    except Exception as e:  # noqa: BLE001  (diagnostics only: never cost the headline its JSON line)
        ex["model_steps_error"] = str(e)[:200]
    # BASELINE config #3 as named: NGCF on the Yelp2018 shape (forward = ngcf.py:73-104 -> [N, 256]; one training step)
    try:
        yu, yi, ynu, yni = rbg.synth.make("yelp2018")
        yds = rbg.InteractionDataset(yu, yi, ynu, yni)
        torch.manual_seed(0)
        ym = rbg.NGCF({"device": str(dev), "enable_sparse": True, "embedding_size": d, "hidden_size_list": [d] * 3, "message_dropout": 0.0,
                       "reg_weight": 1e-5}, yds)
        gy_ = torch.Generator().manual_seed(1)
        yb = {"user_id": torch.randint(1, ynu, (2048,), generator=gy_).to(dev), "item_id": torch.randint(1, yni, (2048,), generator=gy_).to(dev),
              "neg_item_id": torch.randint(1, yni, (2048,), generator=gy_).to(dev)}
        ym.eval()
        with torch.no_grad():
            fwd_us = time_us(lambda: ym.forward(), iters=20, warm=3)
        ym.train()
        yf = rbg.FusedNGCFAdam(ym, lr=1e-3, graphed=True)
        ex["config3_ngcf_yelp2018_shape"] = {"nodes": ynu + yni, "nnz": 2 * len(yu), "forward_us([N, 256] out)": fwd_us,
                                             "fused_step_graphed_us(batch 2048)": time_us(lambda: yf.step(yb), iters=20, warm=4),
                                             "kernel": ym.graph.spmm_kernel_name(d)}
        del ym, yf, yds
    except Exception as e:  # noqa: BLE001
        ex["config3_error"] = str(e)[:200]
    # whole epochs through the minimal driver (driver.fit: device-side BPR sampler, the models' autograd-free steps, SGL's views
    # re-sampled per epoch): seconds for the SECOND epoch of every model = 502 batches of 2048 at the Gowalla shape
    try:
        ep = {}
        for name in ("LightGCN", "NGCF", "SGL", "SimGCL", "XSimGCL", "NCL", "NCL(prototype term on: after warm_up_step)"):
            torch.manual_seed(0)
            np.random.seed(0)
            cfg = {"device": str(dev), "enable_sparse": True, "embedding_size": d, "n_layers": k_layers, "require_pow": True}
            if name.startswith("NCL("):  # (NCL.yaml's warm_up_step = 20: the epochs timed as "NCL" run WITHOUT ProtoNCE_loss, ncl.py:106-135)
                cfg["warm_up_step"] = 0
            mm = getattr(rbg, name.split("(")[0])(cfg, ds)
            marks = []

            def mark(_msg, marks=marks):
                torch.cuda.synchronize()
                marks.append(time.perf_counter())

            t0 = time.perf_counter()
            rbg.driver.fit(mm, uid, iid, epochs=2, lr=1e-3, log=mark)
            ep[name] = {"epoch_s": round(marks[1] - marks[0], 4), "first_epoch_s(incl. warm-up and capture)": round(marks[0] - t0, 4),
                        "stepper": type(rbg.fused_stepper(mm)).__name__}
            del mm
        ep["batches_per_epoch"] = (len(uid) + 2047) // 2048
        ex["driver_epoch(device sampler; autograd-free fused steps for all six models (train.fused_stepper), replayed from HIP graphs; model defaults: NGCF message_dropout 0.1, SGL ED views)"] = ep
    except Exception as e:  # noqa: BLE001
        ex["driver_epoch_error"] = str(e)[:200]
    return ex


def shard_compute_ceiling(rbg, dev, d, time_us, shapes=("amazon-book", "g-1.3m"), worlds=(2, 4, 8)):
    """the COMPUTE ceiling of the node-range shards on the one GPU there is — per shape and P, rank r's plan
    is cut out of the global CSR (sharded.plan_from_csr, degree-striped partition), its layer product is timed ALONE on the GPU
    (nothing else running, no exchange) in three forms: `fused` = ONE launch over [A_int | A_halo] (r06, the default),
    `two_handles` = interior + halo accumulate, both on the column-slab kernel (r06), `r05_form` = the same with the halo block on
    the binned kernel (option "sell" off for that launch: what r05 ran).  ceiling = T_1 / max_r T_{P,r} with T_1 = the same
    layer (rbg_spmm_f32) on the whole graph.  All ranks at the Amazon-Book shape, ranks 0 and P - 1 at 1.3 M nodes (the
    degree-striped ranks are near-identical by construction; the plan of one rank of that graph takes ~1 s)."""
    sh = rbg.sharded
    be = sh.HipBackend(dev)
    out = {}
    for name in shapes:
        uid, iid, nu, ni = rbg.synth.make(name)
        g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        n = nu + ni
        big = n > 1_000_000
        it, wm = (10, 2) if big else (50, 5)
        x, y = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
        t1 = time_us(lambda: rbg.ops.spmm_raw(g, x, out=y), iters=it, warm=wm)
        rowptr, col, val = g.device_csr()
        rec = {"nodes": n, "nnz": g.nnz, "T1_layer_us": t1, "T1_kernel": g.spmm_kernel_name(d)}
        del x, y
        for world in worlds:
            owner = sh.degree_striped_partition(uid, iid, nu, ni, world)
            ranks = list(range(world)) if not big else sorted({0, world - 1})
            per = []
            for r in ranks:
                plan = sh.plan_from_csr(rowptr, col, val, nu, owner, r, world)
                prop = sh.ShardedPropagation(plan, be, transport="staged", fused=True)
                no, nh = plan.n_owned, plan.n_halo
                xc, yo = torch.randn(no + nh, d, device=dev), torch.empty(no, d, device=dev)
                row = {"rank": r, "owned_rows": no, "halo_rows": nh, "nnz": int(plan.int_csr[0][-1] + plan.halo_csr[0][-1]),
                       "status": prop.kernel_status()["cat"]}
                row["fused_us"] = time_us(lambda: be.spmm(prop.g_cat, xc, yo, False), iters=it, warm=wm)
                gi_, gh_ = prop.g_int, prop.g_halo

                def two():
                    be.spmm(gi_, xc[:no], yo, False)
                    be.spmm(gh_, xc[no:], yo, True)

                row["two_handles_us"] = time_us(two, iters=it, warm=wm)

                def r05():
                    be.spmm(gi_, xc[:no], yo, False)
                    rbg.set_option("sell", 0)
                    be.spmm(gh_, xc[no:], yo, True)
                    rbg.set_option("sell", 1)

                try:
                    row["r05_form_us"] = time_us(r05, iters=it, warm=wm)
                finally:
                    rbg.set_option("sell", 1)
                per.append(row)
                del prop, xc, yo, plan, gi_, gh_
            worst = {k: max(q[k] for q in per) for k in ("fused_us", "two_handles_us", "r05_form_us")}
            rec[f"P{world}"] = {"partition": "degree-striped", "ranks_timed": ranks,
                                "max_rank_us": worst,
                                "ceiling(T1 / max_r T_P,r)": {k[:-3]: t1 / v for k, v in worst.items()},
                                "ceiling_over_P": {k[:-3]: t1 / v / world for k, v in worst.items()},
                                "per_rank": per}
        out[name] = rec
        del g, rowptr, col, val
    return out


def capture_steps(step, steps):
    """The K steps of the timed region as ONE HIP graph (N = 1 only): the library enqueues on torch's current stream and
    never synchronises or allocates, so a step is capturable as it is; a replay then submits the same 3 K kernels with one
    host call instead of K ctypes calls — what train.GraphedStep does for the training steps.  None if capture fails."""
    try:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            for _ in range(steps):
                step()
        return graph
    except Exception:  # noqa: BLE001  (the eager loop is always available)
        return None


def timed_loop(step, steps, warmup, world, gloo_group, graph=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; (wall s, event ms), MAX over ranks.
    With `graph` (the K steps captured by capture_steps) the timed region is one replay of it."""
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    if graph is not None:
        graph.replay()  # untimed: uploads the executable graph
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=gloo_group)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream every kernel is launched on
    if graph is not None:
        graph.replay()
    else:
        for _ in range(steps):
            step()
    ev1.record()
    while not ev1.query():  # the host spins on the closing event: a blocking wait adds its wake-up latency (tens of us)
        pass                # to a region that is 2.6 ms at the driver's K = 20
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=gloo_group)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=gloo_group)
        elapsed, ev_ms = float(t[0]), float(t[1])
    return elapsed, ev_ms


CONFIG5 = (10_000_001, 5_000_001, 200_000_000)  # BASELINE.json configs[4]: users / items / interactions (PAD rows included)


def baseline_workload(world):
    """BASELINE.json's configuration for this GPU count: (workload name, embedding width)."""
    return {2: ("gowalla", 64), 4: ("amazon-book", 64), 8: ("config5", 128)}.get(world, ("amazon-book", 64))


def load_workload(rbg, name, seed, rank, world, gloo_group):
    """(uid, iid, n_users, n_items, note).  config5 (200 M interactions, 3.2 GB of ids) is generated ONCE on the node and
    shared through /dev/shm; when the host cannot hold the generator's temporaries the 1.3 M-node shape stands in, with the
    reason in `note`."""
    import torch.distributed as dist
    note = None
    if name == "config5":
        nu, ni, e = CONFIG5
        path = f"/dev/shm/rbg_config5_devgen_seed{seed}.npy"
        ok = os.path.exists(path)
        reason = None
        if not ok and rank == 0:
            try:
                import psutil
                avail = psutil.virtual_memory().available / 2 ** 30
                shm = psutil.disk_usage("/dev/shm").free / 2 ** 30
            except Exception:  # noqa: BLE001
                avail, shm = 0.0, 0.0
            if avail < 24 or shm < 4:
                reason = f"host has {avail:.0f} GiB available RAM / {shm:.0f} GiB free in /dev/shm (need 24 / 4)"
            else:
                t0 = time.time()
                # the generator's algorithm with torch on this rank's GPU (3 s; the numpy generator needs 181 s and 40 GB
                # of temporaries at this size): same distribution, its own random stream
                uid, iid = rbg.synth.powerlaw_bipartite_device(nu, ni, e, torch.device("cuda", torch.cuda.current_device()), seed=seed)
                np.save(path + ".tmp.npy", np.stack([uid, iid]))
                os.replace(path + ".tmp.npy", path)
                del uid, iid
                print(f"[bench] config5 graph generated in {time.time() - t0:.0f} s -> {path}", file=sys.stderr)
        if world > 1:
            box = [reason]
            dist.broadcast_object_list(box, src=0, group=gloo_group)
            reason = box[0]
        if reason is None and os.path.exists(path):
            both = np.load(path, mmap_mode="r")
            return np.asarray(both[0]), np.asarray(both[1]), nu, ni, None
        note = f"config5 replaced by g-1.3m: {reason or 'shared graph file missing'}"
        name = "g-1.3m"
    uid, iid, nu, ni = rbg.synth.make(name, seed=seed)
    return uid, iid, nu, ni, note


def strong_setup(rbg, sh, name, seed, world, rank, dev, d, gen, transport, gloo_group, overlap, partition="auto"):
    """Strong scaling: ONE fixed graph cut into node shards (users and items each), no planted locality.  The global
    normalized CSR is built on the device and the rank's blocks are cut out of it there (plan_from_csr) — nothing of size
    nnz is sorted on the host."""
    import torch.distributed as dist
    uid, iid, nu, ni, note = load_workload(rbg, name, seed, rank, world, gloo_group)
    owner, part_name, part_stats = sh.choose_partition(uid, iid, nu, ni, world, partition)
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    plan = sh.plan_from_csr(*g.device_csr(), nu, owner, rank, world)
    del g
    torch.cuda.empty_cache()
    prop = sh.ShardedPropagation(plan, sh.HipBackend(dev), group=gloo_group if transport == "staged" else None,
                                 transport=transport, overlap=overlap)
    e0 = xavier(plan.n_owned, d, gen).to(dev)
    mine = (int(plan.n_owned), int(plan.int_csr[0][-1]) + int(plan.halo_csr[0][-1]), int(plan.n_halo))
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, mine, group=gloo_group)
    else:
        per_rank = [mine]
    shape = "config5" if (name == "config5" and note is None) else ("g-1.3m" if note else name)
    gen_note = " (config #5's shape; generated by synth.powerlaw_bipartite_device: the generator's algorithm with torch's RNG)" if shape == "config5" else ""
    desc = (f"{shape}-shape graph{gen_note} ({nu} users / {ni} items / {len(uid)} interactions) cut into {world} node shards per side "
            f"({part_name} partition), no planted locality, trimmed halo all_to_all per layer")
    info = {"partition": part_name,
            "partition_candidates(max over ranks)": {k: {"nnz": max(v["nnz"]), "rows": max(v["rows"]), "min_rows": min(v["rows"])}
                                                      for k, v in part_stats.items()},
            "per_rank(owned rows, nnz, halo rows)": per_rank, "workload_note": note}
    return prop, e0, plan, desc, rbg.synth.algorithmic_bytes(nu + ni, 2 * len(uid), d, 3), (uid, iid, nu, ni, shape, owner), info


def traffic_from_profiles(workload, d, kernel):
    """(HBM-side bytes per launch, L2 hit rate) of `kernel` from the committed PMC passes (profiles/traffic.json; the
    counters need their own rocprofv3 runs — devtools/traffic_session.sh — so they cannot be collected inside this process).
    (None, None) when no pass exists for this workload / width / kernel."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path)).get(f"{workload}:d{d}:{kernel}")
    except Exception:  # noqa: BLE001
        rec = None
    if isinstance(rec, dict):
        return rec.get("traffic"), rec.get("l2_hit")
    return rec, None


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (one per GPU; when the box has fewer
    GPUs the ranks share them through the host-staged transport) and pass their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = list(sys.argv[1:])
    if args.transport is None and not args.launch_check:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        argv += ["--transport", "nccl" if n_dev >= args.gpus else "staged"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    # (for the CPU baseline's OpenMP runtime — the oracle library's own, loaded later: threads pinned one per core, spread over the
    # sockets, so that first-touch placement means something; a caller's own setting wins)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(self_launch(args))  # no launcher around us: start the ranks here
    # stdout carries the ONE JSON line and nothing else: C++ libraries (gloo's "Rank 0 is connected to ..." banner) write to
    # file descriptor 1 directly, so descriptor 1 is pointed at stderr for the run and the line goes to a saved copy
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.launch_check:  # rendezvous only (CPU test of the self-launch path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t) == world
        if rank == 0:
            print(json.dumps({"launch_check": world}), file=json_out, flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    if args.transport is None:
        args.transport = "nccl" if torch.cuda.device_count() >= world else "staged"
    dev_index = local_rank if args.transport == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gloo_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.transport == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            gloo_group = dist.new_group(backend="gloo")  # control plane + fallback transport
        else:
            dist.init_process_group("gloo")

    import recbole_gnn_amd as rbg
    from recbole_gnn_amd import sharded as sh

    k_layers = args.layers
    gen = torch.Generator().manual_seed(args.seed + rank)
    extra = {}

    if world == 1 and args.shard_ceiling_only:
        def _time_us(fn, iters=50, warm=5):
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / iters)
            return sorted(ts)[1]
        print(json.dumps({"shard_compute_ceiling": shard_compute_ceiling(rbg, dev, args.dim or 64, _time_us)}), file=json_out, flush=True)
        return

    if world == 1:
        wl_name, d = args.workload or "gowalla", args.dim or 64
        nu, ni, n_inter = rbg.synth.shape(wl_name)
        n = nu + ni
        b_layer, b_prop = rbg.synth.algorithmic_bytes(n, 2 * n_inter, d, k_layers)
        uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, n_inter, seed=args.seed)
        graph = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
        uw_h, iw_h = xavier(nu, d, gen), xavier(ni, d, gen)
        uw, iw = uw_h.to(dev), iw_h.to(dev)
        out = torch.empty((n, d), device=dev)
        layers = torch.empty((max(k_layers, 1), n, d), device=dev)

        def step():
            rbg.ops.lightgcn_forward_raw(graph, uw, iw, k_layers, out=out, layers=layers)

        # parity gate on the very buffers that get timed (1e-5 fp32, BASELINE.json north_star)
        from oracle import coracle
        step()
        torch.cuda.synchronize()
        rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
        ref = coracle.lightgcn_forward(rowptr, col, val, uw_h.numpy(), iw_h.numpy(), k_layers)
        err = float(np.abs(out.cpu().numpy() - ref).max())
        if not err <= 1e-5:
            raise SystemExit(f"parity gate failed: max|E_hip - E_oracle| = {err:.3e} > 1e-5")
        extra.update(max_abs_err_vs_oracle=err, bins=graph.bins(d), tuning=rbg.get_tuning())
        # the kernel a propagation of this handle launches per layer: the column-slab kernel (csrc/sell.hip; a SELL plan is
        # built by rbg_graph_create itself) or — sell_status says why — the binned SpMM kernel
        kernel_name = graph.propagation_kernel_name(d)
        extra["propagation_path"] = {"kernel": kernel_name, "sell_plan_attached": bool(graph.has_sell(d)), "sell_status": graph.sell_status(),
                                     "planner": "rbg_graph_plan_sell inside rbg_graph_create (csrc/sell_plan.hip)",
                                     "launches_per_propagation": k_layers + (1 if graph.has_sell(d) and not rbg.get_option("sell_rowmajor") else 0),
                                     "note": ("column-slab path: K launches of sell_spmm_kernel — the first gathers E0 where it lies through the "
                                              "valued entries (instantiation <.., false>), the others read the scaled slabs through 4-byte entries "
                                              "(<.., true>, the name quoted); the last one carries the mean")
                                     if graph.has_sell(d) else "K binned SpMM launches (the last one carries the mean)"}
        launches_per_step = k_layers
        units_per_step = 1
        scaling = "weak"  # N = 1: per-GPU work is what it is; the label only matters at N > 1
        workload = (f"{wl_name}-shape synthetic power-law bipartite graph: {nu} users / {ni} items / "
                    f"{n_inter} interactions (PAD rows included), nnz(A_hat) = {2 * n_inter}")
    else:
        transport = args.transport
        scaling = args.scaling
        base_name, base_d = baseline_workload(world)
        strong_name = args.workload or base_name
        d = args.dim or (128 if strong_name == "config5" else 64)
        weak_name = args.workload if (args.workload and args.workload != "config5") else "gowalla"
        nu, ni, n_inter = rbg.synth.shape(weak_name)
        one_gpu = {}

        def weak_setup():
            nu_g, ni_g = (nu - 1) * world + 1, (ni - 1) * world + 1
            wu, wi = rbg.synth.powerlaw_bipartite(nu_g, ni_g, n_inter * world, seed=args.seed, n_blocks=world, p_in=args.p_in)
            owner = sh.striped_partition(nu_g, ni_g, world)
            wplan = sh.build_plans(wu, wi, nu_g, ni_g, world, owner=owner, ranks=[rank])[rank]
            wprop = sh.ShardedPropagation(wplan, sh.HipBackend(dev), group=gloo_group if transport == "staged" else None,
                                          transport=transport)
            desc = (f"{world} x {weak_name}-shape blocks, node-range sharded: {nu_g} users / {ni_g} items / "
                    f"{n_inter * world} interactions, PLANTED locality p_in = {args.p_in} (striped communities = the partition), "
                    f"trimmed halo all_to_all per layer")
            return wprop, xavier(wplan.n_owned, d, gen).to(dev), wplan, desc, rbg.synth.algorithmic_bytes(nu + ni, 2 * n_inter, d, k_layers)

        if scaling == "weak":
            prop, e0, plan, workload, (b_layer, b_prop) = weak_setup()
            strong_graph = None
        else:
            prop, e0, plan, workload, (b_layer, b_prop), strong_graph, pinfo = strong_setup(
                rbg, sh, strong_name, args.seed, world, rank, dev, d, gen, transport, gloo_group, False, args.partition)
            extra.update(pinfo)
        if transport == "nccl":
            # rehearse one exchange; if RCCL cannot run it on this node, every rank falls back to the host-staged
            # gloo transport (slow, but the run still reports a labelled number instead of crashing)
            ok = torch.ones(1, device=dev)
            try:
                prop.forward(e0, 1)
                torch.cuda.synchronize()
            except Exception as ex:  # noqa: BLE001
                ok.zero_()
                extra["nccl_error"] = str(ex)[:300]
            votes = [torch.zeros(1) for _ in range(world)]
            dist.all_gather(votes, ok.cpu(), group=gloo_group)
            if min(float(v) for v in votes) == 0.0:
                transport = "staged"
                prop = sh.ShardedPropagation(plan, sh.HipBackend(dev), group=gloo_group, transport="staged")
        if transport == "nccl":  # pick the stream structure on the real group: both are timed, the faster is kept
            extra["overlap_autotune"] = prop.autotune(e0, k_layers, try_push=(args.halo_push == "on"))

        def step():
            prop.forward(e0, k_layers)

        launches_per_step = k_layers
        units_per_step = world if scaling == "weak" else 1
        kernel_name = (prop.g_cat if prop.fused else prop.g_int).spmm_kernel_name(d)
        extra["layer_form"] = prop.kernel_status()  # r06: "fused" = one planned handle over [owned | halo], one launch per layer
        extra.update(p_in=args.p_in if scaling == "weak" else None, halo_rows_rank0=int(plan.n_halo),
                     owned_rows_rank0=int(plan.n_owned), halo_bytes_per_layer_rank0=int(plan.n_halo) * d * 4,
                     overlap=bool(prop.overlap), transport=transport,
                     rccl_ranks=(dist.get_world_size() if transport == "nccl" else 0))

    # ---- timing ---------------------------------------------------------------------------------------------------------
    # N = 1: the K timed steps are replayed from one HIP graph unless --eager (a 20-step region is 2.6 ms: a single host hiccup
    # between two ctypes calls would otherwise be a visible share of it); N > 1 stays host-issued (collectives on side streams).
    # The headline is measured at EXACTLY the flags given: W warm-up steps (+ --clock-warmup, default 0, counted in
    # `warmup_effective`), K timed steps.  A fresh process needs ~10 ms of load before the GPU runs at its steady clocks, more
    # than the driver's W = 5 gives it, so the steady-clock figure is measured too (150 extra untimed propagations in front of
    # the same region) and reported beside the headline as `steady_clock`.
    clock_warmup = max(0, args.clock_warmup)
    if world == 1:  # first of all, the plain protocol: W host-issued warm-up steps, K host-issued timed steps, nothing else
        el0, ev0 = timed_loop(step, args.steps, args.warmup, world, None)
        extra["value_at_driver_flags_no_clock_warmup"] = {
            "value": args.steps / el0, "ms_per_step": el0 * 1e3 / args.steps, "timed_region": "K host-issued steps",
            "warmup_effective": args.warmup, "avg_launch_us": ev0 * 1e3 / (args.steps * launches_per_step)}
    step_graph = capture_steps(step, args.steps) if (world == 1 and not args.eager) else None
    if world == 1 and step_graph is not None:
        time.sleep(0.05)  # the capture left the GPU idle anyway; make the cold start the same run to run
    elapsed, ev_ms = timed_loop(step, args.steps, args.warmup + clock_warmup, world, gloo_group if world > 1 else None, graph=step_graph)
    # (timed_loop replays the captured graph once, untimed, to upload it: K more propagations of warm-up)
    extra["warmup_effective"] = args.warmup + clock_warmup + (args.steps if step_graph is not None else 0)
    # EVERY propagation this process issued before the headline's timed region: the parity step, the
    # plain-protocol loop (W + K), the capture's warm-up step, then the region's own warm-up (`warmup_effective`)
    if world == 1:
        extra["propagations_before_timed_region"] = {
            "parity_gate": 1, "plain_protocol_loop": args.warmup + args.steps,
            "graph_capture_warmup": 1 if step_graph is not None else 0, "region_warmup": extra["warmup_effective"],
            "total": 1 + args.warmup + args.steps + (1 if step_graph is not None else 0) + extra["warmup_effective"],
            "note": "the plain-protocol figure (value_at_driver_flags_no_clock_warmup: W warm-up steps exactly, K host-issued steps) is "
                    "the one measured with nothing but the parity step in front of it"}
    extra["timed_region"] = "one HIP-graph replay of the K steps" if step_graph is not None else "K host-issued steps"
    if world == 1:
        el_s, ev_s = timed_loop(step, args.steps, args.warmup + 150, world, None, graph=step_graph)
        extra["steady_clock"] = {"value": args.steps / el_s, "ms_per_step": el_s * 1e3 / args.steps, "warmup_effective": args.warmup + 150,
                                 "avg_launch_us": ev_s * 1e3 / (args.steps * launches_per_step),
                                 "roofline_frac": b_layer / (ev_s * 1e-3 / (args.steps * launches_per_step)) / 1e9 / HBM_PEAK_GBPS}
        if step_graph is not None:  # and the host-issued loop beside it
            el_e, _ = timed_loop(step, args.steps, args.warmup, world, None)
            extra["eager_ms_per_step"] = el_e * 1e3 / args.steps

    if world > 1 and not args.no_secondary:
        # the other scaling mode, same process group, bounded: its own short timed loop
        try:
            if scaling == "weak":
                p2, e2, plan2, desc2, _, strong_graph, pinfo2 = strong_setup(rbg, sh, strong_name, args.seed, world, rank, dev, d, gen,
                                                                             transport, gloo_group, prop.overlap, args.partition)
                units2, label2 = 1, "strong"
            else:
                p2, e2, plan2, desc2, _ = weak_setup()
                p2.set_overlap(prop.overlap)
                units2, label2, pinfo2 = world, "weak", {}
            steps2 = max(10, min(args.steps, 100))
            el2, _ = timed_loop(lambda: p2.forward(e2, k_layers), steps2, max(3, min(args.warmup, 10)), world, gloo_group)
            secondary = {"scaling": label2, "value": units2 * steps2 / el2, "unit": "propagations/s", "ms_per_step": el2 * 1e3 / steps2,
                         "steps": steps2, "workload": desc2, "halo_rows_rank0": int(plan2.n_halo), "owned_rows_rank0": int(plan2.n_owned),
                         "p_in": args.p_in if label2 == "weak" else None}
            secondary.update({k: v for k, v in pinfo2.items() if k == "partition"})
            del p2, e2
        except Exception as ex:  # noqa: BLE001  (a deterministic failure is raised on every rank alike)
            secondary = {"error": str(ex)[:300]}
        extra["other_scaling_mode"] = secondary

    if world > 1:
        try:
            extra["halo_per_layer(rank 0)"] = prop.halo_bytes_per_layer(d)
        except Exception:  # noqa: BLE001
            pass
    if world > 1 and strong_graph is not None and not args.no_secondary:
        # The OTHER sharding of the same graph (r04, colsharded.py): every rank holds the whole graph and d / N columns of the
        # tables — the K layers exchange nothing.  Timed with the same loop (barrier + synchronize, max over ranks); a global
        # forward = all N slabs, so value = steps / time.  Reported beside the node-range headline the north_star names.
        try:
            su, si, snu, sni = strong_graph[:4]
            width = d // world if d % world == 0 else 0
            if width in (32, 64, 128):
                from recbole_gnn_amd import colsharded as cs
                gfull = rbg.GraphHandle.from_interactions(su, si, snu, sni, device=dev)
                cprop = cs.ColumnShardedPropagation(gfull, snu, sni, d, sh.HipBackend(dev), rank=rank, world=world, group=gloo_group)
                slab = xavier(snu + sni, width, gen).to(dev)
                stepsc = max(10, min(args.steps, 100))
                elc, evc = timed_loop(lambda: cprop.forward(slab, k_layers), stepsc, max(3, min(args.warmup, 10)), world, gloo_group)
                bl_c, _ = rbg.synth.algorithmic_bytes(snu + sni, gfull.nnz, width, k_layers)
                extra["column_sharding"] = {
                    "value": stepsc / elc, "unit": "propagations/s", "ms_per_step": elc * 1e3 / stepsc, "steps": stepsc,
                    "columns_per_rank": width, "kernel": gfull.propagation_kernel_name(width), "bytes_exchanged_per_layer": 0,
                    "per_rank_roofline_frac": bl_c / (evc * 1e-3 / (stepsc * k_layers)) / 1e9 / HBM_PEAK_GBPS,
                    "note": "feature-column sharding: the graph (its column-slab plan) is replicated, every rank propagates d / N columns "
                            "of the tables; only the loss exchanges anything ([2 B] score partials per training step)"}
                del gfull, cprop, slab
            else:
                extra["column_sharding"] = {"skipped": f"d / N = {d} / {world}: not a width the column-slab plan serves (32, 64, 128)"}
        except Exception as ex:  # noqa: BLE001
            extra["column_sharding"] = {"error": str(ex)[:200]}
        # ... and their product (r05, hybrid.py): 2 column groups x N / 2 node shards — the mode that reaches 4 and 8 GPUs at d = 64
        # with halos half as wide.  Collective construction (process subgroups): every rank runs this block.
        try:
            su, si, snu, sni = strong_graph[:4]
            if world % 2 == 0 and d % 2 == 0 and (d // 2) in (32, 64, 128):
                from recbole_gnn_amd import hybrid as hy
                hprop = hy.HybridShardedPropagation(su, si, snu, sni, d, sh.HipBackend(dev), 2, rank=rank, world=world, transport=transport)
                blk = xavier(hprop.plan.n_owned, hprop.width, gen).to(dev)
                stepsh = max(10, min(args.steps, 100))
                elh, evh = timed_loop(lambda: hprop.forward(blk, k_layers), stepsh, max(3, min(args.warmup, 10)), world, gloo_group)
                extra["hybrid_sharding"] = {
                    "value": stepsh / elh, "unit": "propagations/s", "ms_per_step": elh * 1e3 / stepsh, "steps": stepsh,
                    "grid": f"2 column groups x {world // 2} node shards", "columns_per_rank": hprop.width,
                    "halo_per_layer(rank 0)": hprop.halo_bytes_per_layer(), "interior_kernel": hprop.prop.g_int.propagation_kernel_name(hprop.width)
                    if hasattr(hprop.prop.g_int, "propagation_kernel_name") else None,
                    "note": "hybrid.py: a rank holds its node shard's rows and d / 2 columns; halos travel inside a column group as rows of "
                            "d / 2 floats (half the pure node-range mode's bytes for the same node shards), the loss all-reduces partial "
                            "dots over the two column groups"}
                del hprop, blk
            else:
                extra["hybrid_sharding"] = {"skipped": f"needs an even rank count and d / 2 in (32, 64, 128); world {world}, d {d}"}
        except Exception as ex:  # noqa: BLE001
            extra["hybrid_sharding"] = {"error": str(ex)[:200]}

    if world > 1:
        # phase breakdown of one sharded layer (each phase alone, back to back; rank-0 view) so the scaling
        # number comes with its explanation: halo exchange vs interior SpMM vs halo SpMM
        def phase_us(fn, iters=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            dist.barrier(group=gloo_group)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / iters

        # diagnostic only: a failure here must not cost the measurement above its JSON line (the calls are the ones the
        # timed region already made, so an exception would be a deterministic one raised on every rank alike)
        try:
            halo_buf = torch.empty((max(plan.n_halo, 1), d), device=dev)
            y_buf = torch.empty((plan.n_owned, d), device=dev)
            extra["phase_us"] = {
                "halo_exchange(pack + all_to_all)": phase_us(
                    lambda: (prop._exchange_nccl if transport == "nccl" else prop._exchange_staged)(e0, halo_buf[: plan.n_halo])),
            }
            if prop.fused:  # one launch over the table [owned rows | halo rows]
                cat_buf = torch.empty((plan.n_owned + plan.n_halo, d), device=dev)
                extra["phase_us"]["fused_layer_spmm"] = phase_us(lambda: prop.backend.spmm(prop.g_cat, cat_buf, y_buf, False))
                del cat_buf
            else:
                extra["phase_us"]["interior_spmm"] = phase_us(lambda: prop.backend.spmm(prop.g_int, e0, y_buf, False))
                extra["phase_us"]["halo_spmm"] = phase_us(lambda: prop.backend.spmm(prop.g_halo, halo_buf, y_buf, True)) if prop.g_halo else 0.0
            del halo_buf, y_buf
        except Exception as ex:  # noqa: BLE001
            extra["phase_us_error"] = str(ex)[:200]
        # r06: the same propagation with the halo PUSHED into the peers' exported layer tables instead of a collective
        # (sharded.PushExchange, csrc/ipc.hip): us per propagation (max over ranks) and its exchange alone.  Collective set-up:
        # every rank tries, a failure anywhere is voted on, so nobody waits on a peer that gave up.
        try:
            pp, perr = None, None
            shared_gpus = torch.cuda.device_count() < world
            if args.halo_push == "off" or (args.halo_push == "auto" and not shared_gpus):
                raise RuntimeError("not requested (--halo-push on): untested between different GPUs")
            try:
                if k_layers + 1 > 9 or not prop.fused:
                    raise RuntimeError("push serves the fused layer")
                pp = sh.ShardedPropagation(plan, prop.backend, group=gloo_group, transport="push", push_tables=k_layers + 1, push_timeout_ms=500)
                pp._g_cat = prop.g_cat  # (the same planned handle)
            except Exception as ex:  # noqa: BLE001
                perr = str(ex)[:160]
            ok = torch.tensor([0.0 if perr else 1.0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=gloo_group)
            if float(ok) == 1.0:
                try:
                    pp.forward(e0, k_layers)  # (set-up of the exchange: collective over the gloo group)
                    pp.push.check()           # (flag words that never arrive time out here, once, not in the timed loops)
                except Exception as ex:  # noqa: BLE001
                    perr = str(ex)[:160]
                ok = torch.tensor([0.0 if perr else 1.0])
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=gloo_group)
            if float(ok) == 1.0:
                ref_mean = prop.forward(e0, k_layers).clone()
                same = bool(torch.equal(pp.forward(e0, k_layers), ref_mean))
                t_push = phase_us(lambda: pp.forward(e0, k_layers), iters=10)
                t_ex = phase_us(lambda: (pp.push.exchange(0, torch.cuda.current_stream().cuda_stream), pp.push.consumed(0, torch.cuda.current_stream().cuda_stream)), iters=10)
                pp.push.check()
                tt = torch.tensor([t_push, t_ex])
                dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=gloo_group)
                extra["halo_push"] = {"propagation_us(max over ranks)": float(tt[0]), "exchange_alone_us(push + flags, one layer)": float(tt[1]),
                                      "bit_equal_to_the_timed_transport": same,
                                      "note": "no collective: peers' pack kernels store into this rank's exported tables (hipIpc), flag words order it"
                                              + ("" if torch.cuda.device_count() >= world else "; ranks share one GPU here: functional, not a measurement")}
                pp.push.close()
            else:
                extra["halo_push"] = {"error": perr or "a peer could not set the exchange up"}
        except Exception as ex:  # noqa: BLE001
            extra["halo_push_error"] = str(ex)[:200]
        # BASELINE config #5 is "SGL ... 8 x MI355X": one sharded SGL TRAINING step on the strong graph (sharded_train.py: three
        # propagations over the full graph and two edge-drop views, BPR + reg + InfoNCE with distributed denominators, the
        # transposed chains, Adam on the owned rows), batch 2048 — a few steps, reported beside the propagation figure
        if strong_graph is not None and not args.no_train_extra:
            tr, setup_err = None, None
            try:  # rank-local part (no collectives): view graphs, their plans, the trainer
                from recbole_gnn_amd import sharded_train as st
                su, si, snu, sni, sname, owner = strong_graph
                t_setup = time.perf_counter()
                u_dev, i_dev = torch.from_numpy(np.ascontiguousarray(su)).to(dev), torch.from_numpy(np.ascontiguousarray(si)).to(dev)
                view_plans = []
                for v in range(2):  # the same mask on every rank: same seed, same device type, same Philox stream
                    gk = torch.Generator(device=dev).manual_seed(args.seed + 101 + v)
                    keep = torch.zeros(len(su), dtype=torch.bool, device=dev)
                    keep[torch.randperm(len(su), generator=gk, device=dev)[: int(len(su) * 0.9)]] = True
                    gv = rbg.GraphHandle.from_interactions(u_dev, i_dev, snu, sni, device=dev, keep=keep)
                    view_plans.append(sh.plan_from_csr(*gv.device_csr(), snu, owner, rank, world))
                    del gv, keep
                del u_dev, i_dev
                torch.cuda.empty_cache()
                tr = st.ShardedTrainer(plan, sh.HipBackend(dev), e0, snu, sni, k_layers, view_plans=view_plans,
                                       group=gloo_group if transport == "staged" else None, transport=transport, lr=1e-3,
                                       reg_weight=1e-4, ssl_tau=0.2, ssl_weight=0.05, overlap=extra["overlap"])
                t_setup = time.perf_counter() - t_setup
            except Exception as ex:  # noqa: BLE001
                tr, setup_err = None, str(ex)[:300]
            votes = [None] * world  # every rank built its trainer, or nobody steps (a step is full of collectives)
            dist.all_gather_object(votes, setup_err, group=gloo_group)
            if any(v is not None for v in votes):
                extra["sgl_sharded_train_step"] = {"error": next(v for v in votes if v is not None)}
                tr = None
            try:
                if tr is None:
                    raise StopIteration
                gb = torch.Generator().manual_seed(args.seed + 7)
                bu, bp, bn = (torch.randint(1, hi, (2048,), generator=gb) for hi in (snu, sni, sni))
                losses = [tr.step(bu, bp, bn)]  # untimed: buffers, workspaces
                torch.cuda.synchronize()
                dist.barrier(group=gloo_group)
                n_tr = 3
                t1 = time.perf_counter()
                for _ in range(n_tr):
                    losses.append(tr.step(bu, bp, bn))
                torch.cuda.synchronize()
                dist.barrier(group=gloo_group)
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=gloo_group)
                extra["sgl_sharded_train_step"] = {"ms_per_step": float(tt[0]) * 1e3 / n_tr, "steps": n_tr, "batch": 2048, "workload": sname,
                                                   "views": "2 x ED, drop_ratio 0.1, plans cut on the device", "setup_s": round(t_setup, 1),
                                                   "loss_first_last": [losses[0], losses[-1]],
                                                   "view_halo_rows_rank0": [int(p.n_halo) for p in view_plans]}
                del tr, view_plans
                torch.cuda.empty_cache()
            except StopIteration:
                pass
            except Exception as ex:  # noqa: BLE001  (deterministic failures are raised on every rank alike)
                extra["sgl_sharded_train_step"] = {"error": str(ex)[:300]}
        # the same graph on ONE GPU (rank 0, the others wait): what the sharded number has to be compared with
        if strong_graph is not None:
            if rank == 0:
                try:
                    su, si, snu, sni, sname, _ = strong_graph
                    del prop, e0
                    torch.cuda.empty_cache()
                    g1 = rbg.GraphHandle.from_interactions(su, si, snu, sni, device=dev)
                    x1 = xavier(snu + sni, d, gen).to(dev)
                    o1, l1 = torch.empty(snu + sni, d, device=dev), torch.empty(max(k_layers, 1), snu + sni, d, device=dev)
                    reps = 3 if snu + sni > 4_000_000 else 30
                    f1 = lambda: rbg.ops.lightgcn_forward_raw(g1, x1[:snu], x1[snu:], k_layers, out=o1, layers=l1)  # noqa: E731
                    f1()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        f1()
                    torch.cuda.synchronize()
                    one_gpu = {"workload": sname, "value": reps / (time.perf_counter() - t1), "unit": "propagations/s", "reps": reps,
                               "kernel": g1.spmm_kernel_name(d)}
                    del g1, x1, o1, l1
                except Exception as ex:  # noqa: BLE001
                    one_gpu = {"error": str(ex)[:200]}
            dist.barrier(group=gloo_group)
            extra["same_workload_one_gpu"] = one_gpu

    if rank == 0:
        launch_us = ev_ms * 1e3 / (args.steps * launches_per_step)
        achieved = b_layer / (launch_us * 1e-6) / 1e9
        value = units_per_step * args.steps / elapsed
        if world > 1 and scaling == "strong" and one_gpu.get("value"):
            extra["speedup_vs_one_gpu_same_workload"] = value / one_gpu["value"]
        traffic, l2_hit = traffic_from_profiles(wl_name, d, kernel_name) if world == 1 else (None, None)
        result = {
            "metric": "LightGCN propagations/sec",
            "value": value,
            "unit": "propagations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "embedding_dim": d, "n_layers": k_layers,
                       "algorithmic_bytes_per_layer": b_layer, "algorithmic_bytes_per_propagation": b_prop,
                       "sharding": "none" if world == 1 else
                       f"node shards x{world}, transport={extra['transport']}, "
                       f"{'exchange overlapped on a second stream' if extra['overlap'] else 'single-stream layers'}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # SURVEY 8(d) quotes its 60 % target (19 227 prop/s) on the PER-PROPAGATION bytes B_prop = K B_layer + 4 N d (K + 2):
                         # the same timed region read against that figure
                         "frac_prop": (b_prop / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS) if world == 1 else None,
                         "traffic": traffic, "l2_hit": l2_hit,
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT,TCC_MISS, separate "
                                           "passes, devtools/traffic_session.sh; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per the gfx950 correction)",
                         "kernel": kernel_name, "avg_launch_us": launch_us,
                         "launches_per_step": launches_per_step,
                         "note": "achieved = B_layer (4(N+1) + 8 nnz + 8 N d) / mean layer duration; duration = HIP-event "
                                 "time of the timed region / (steps x K layers), so inter-kernel gaps and (N>1) halo waits count against "
                                 "the kernel; rocprofv3's per-instantiation averages are in profiles/r06_bench_kernel_stats.csv"},
            "cpu_baseline": None,
        }
        if world == 1:
            pk = measured_hbm_peak(dev)
            result["roofline"]["peak_measured"] = pk
            if pk.get("triad_GBps"):
                result["roofline"]["frac_of_measured"] = achieved / pk["triad_GBps"]
            result["roofline"]["cap"] = gather_path_cap(nu + ni, graph.nnz, d, k_layers)
        result.update(extra)
        if world > 1 and args.shard == "hybrid" and isinstance(extra.get("hybrid_sharding"), dict) and extra["hybrid_sharding"].get("value"):
            hs_ = extra["hybrid_sharding"]  # the hybrid propagation becomes the headline, the node-range one is kept beside it
            result["node_range_sharding"] = {"value": result["value"], "ms_per_step": result["ms_per_step"], "roofline_frac": result["roofline"]["frac"],
                                             "sharding": result["config"]["sharding"]}
            result["value"], result["ms_per_step"], result["steps"] = hs_["value"], hs_["ms_per_step"], hs_["steps"]
            result["config"]["sharding"] = f"hybrid: {hs_['grid']}, {hs_['columns_per_rank']} columns per rank, halos of d / 2 floats inside a column group"
            if "speedup_vs_one_gpu_same_workload" in result:
                result["node_range_sharding"]["speedup_vs_one_gpu_same_workload"] = result.pop("speedup_vs_one_gpu_same_workload")
                if one_gpu.get("value"):
                    result["speedup_vs_one_gpu_same_workload"] = hs_["value"] / one_gpu["value"]
        if world > 1 and args.shard == "columns" and isinstance(extra.get("column_sharding"), dict) and extra["column_sharding"].get("value"):
            cs_ = extra["column_sharding"]  # the column-sharded propagation becomes the headline, the node-range one is kept beside it
            result["node_range_sharding"] = {"value": result["value"], "ms_per_step": result["ms_per_step"], "roofline_frac": result["roofline"]["frac"],
                                             "sharding": result["config"]["sharding"]}
            result["value"], result["ms_per_step"], result["steps"] = cs_["value"], cs_["ms_per_step"], cs_["steps"]
            result["config"]["sharding"] = f"feature-column shards x{world}: {cs_['columns_per_rank']} columns per rank, nothing exchanged in the K layers"
            result["config"]["workload"] = (result["config"]["workload"].split(" cut into ")[0] +
                                            f": every rank holds the whole graph and {cs_['columns_per_rank']} of the {d} embedding columns")
            if "speedup_vs_one_gpu_same_workload" in result:  # (that figure belongs to the node-range run)
                result["node_range_sharding"]["speedup_vs_one_gpu_same_workload"] = result.pop("speedup_vs_one_gpu_same_workload")
                if one_gpu.get("value"):
                    result["speedup_vs_one_gpu_same_workload"] = cs_["value"] / one_gpu["value"]
            result["roofline"].update(frac=cs_["per_rank_roofline_frac"], achieved=cs_["per_rank_roofline_frac"] * HBM_PEAK_GBPS, kernel=cs_["kernel"],
                                      note="per-rank: algorithmic bytes of the rank's slab layer / its mean layer duration")
        if world > 1:
            real = extra.get("transport") == "nccl" and torch.cuda.device_count() >= world
            result["measured"] = bool(real)
            if not real:
                result["measured_note"] = ("NOT a measurement of multi-GPU scaling: the ranks share "
                                           f"{torch.cuda.device_count()} GPU(s) and exchange halos through the host (staged gloo transport); "
                                           "a functional run of the N > 1 code path only")
        if world == 1 and not args.no_extras:
            result["extras"] = extras_n1(rbg, graph, uid, iid, nu, ni, d, k_layers, dev)
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(uid, iid, nu, ni, uw_h.numpy(), iw_h.numpy(), k_layers, args.cpu_seconds)
            result["cpu_baseline_torch_sparse"] = cpu_baseline_torch_sparse(uid, iid, nu, ni, uw_h.numpy(), iw_h.numpy(), k_layers,
                                                                            min(args.cpu_seconds, 6.0))
        print(json.dumps(result), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
