#!/usr/bin/env python
"""BiGNN dense layer (rbg_bignn_dense_f32, d_in = d_out = 64): LDS-DMA kernel vs the general kernel, interleaved timing at the
row counts of the BASELINE shapes, and the whole layer (SpMM + dense) at the Gowalla shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg  # noqa: E402

dev = torch.device("cuda:0")
DEFAULT = rbg.get_option("bignn_dma")


def time_us(fn, iters=200, warm=10, per_graph=20):
    """Launches replayed from a HIP graph: the Python call (~25 us of checks + ctypes) would otherwise bound the figure."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(per_graph):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters // per_graph):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters // per_graph * per_graph)


out = open(os.path.join(ROOT, "gpurun_out", "bignn_probe.jsonl"), "a")
for shape in ("gowalla", "yelp2018", "amazon-book"):
    nu, ni, _ = rbg.synth.SHAPES[shape]
    n = nu + ni
    x, p = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
    w1, w2 = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, 64, device=dev) * 0.1
    b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    y = torch.empty(n, 64, device=dev)
    for leaky in (True, False):
        run = lambda: rbg.ops.bignn_dense_raw(p, x, w1, b1, w2, b2, out=y, leaky_norm=leaky)  # noqa: E731
        res = {0: [], 1: []}
        for _ in range(3):
            for v in (1, 0):
                rbg.set_option("bignn_dma", v)
                res[v].append(time_us(run))
        rbg.set_option("bignn_dma", DEFAULT)
        bytes_ = 3 * n * 64 * 4
        rec = dict(kind="bignn_dense", shape=shape, rows=n, leaky_norm=leaky, us_dma=sorted(res[1])[1], us_general=sorted(res[0])[1],
                   hbm_frac_dma=bytes_ / (sorted(res[1])[1] * 1e-6) / 8e12)
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
uid, iid, nu, ni = rbg.synth.make("gowalla")
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
n = nu + ni
x = torch.randn(n, 64, device=dev)
w1, w2 = torch.randn(64, 64, device=dev) * 0.1, torch.randn(64, 64, device=dev) * 0.1
b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
yo, y = torch.empty(n, 64, device=dev), torch.empty(n, 64, device=dev)
layer = lambda: rbg.ops.bignn_conv_raw(g, x, w1, b1, w2, b2, out=yo, leaky_norm=True)  # noqa: E731
res = {0: [], 1: []}
for _ in range(3):
    for v in (1, 0):
        rbg.set_option("bignn_dma", v)
        res[v].append(time_us(layer))
rbg.set_option("bignn_dma", DEFAULT)
rec = dict(kind="bignn_layer", shape="gowalla", us_dma=sorted(res[1])[1], us_general=sorted(res[0])[1],
           us_spmm=time_us(lambda: rbg.ops.spmm_raw(g, x, out=y)))
print(json.dumps(rec), flush=True)
out.write(json.dumps(rec) + "\n")
