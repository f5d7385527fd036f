#!/usr/bin/env python
"""r05: what each part of a resident-round launch costs — the RBG_SELL_TRACE build's what-if switches (results are wrong on
purpose): 1 = gathers fall into an L1-resident 16 KB window, 2 = no epilogue, 4 = entries synthesised (not loaded), 8 = gathers
out of range.  Propagation (K = 3) and plain layer in us.  JSON lines -> gpurun_out/r05_whatif.jsonl"""
import ctypes, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_selltrace.so")
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg
sys.path.insert(0, HERE)

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
d = 64
lib = rbg._lib.lib
log = open(os.path.join(ROOT, "gpurun_out", "r05_whatif.jsonl"), "a")


def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[1]


uid, iid, nu, ni = rbg.synth.make(name)
n = nu + ni
g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
gen = torch.Generator().manual_seed(1)
uwd, iwd = torch.randn(nu, d, generator=gen).to(dev), torch.randn(ni, d, generator=gen).to(dev)
o, L = torch.empty(n, d, device=dev), torch.empty(3, n, d, device=dev)
xx, yy = torch.randn(n, d, device=dev), torch.empty(n, d, device=dev)
fwd = lambda: rbg.ops.lightgcn_forward_raw(g, uwd, iwd, 3, out=o, layers=L)
lay = lambda: rbg.ops.spmm_raw(g, xx, out=yy)
fwd(); lay()
for stream, wgs in ((0, 8), (1, 7), (1, 4)):
    rbg.set_option("sell_stream", stream)
    rbg.set_option("sell_stream_wgs", wgs)
    for bits in ((0,) if not stream else (0, 1, 2, 4, 8, 3, 5, 6, 7, 14)):
        assert lib.mb_sell_debug_set(bits) == 0
        rec = {"what": "sell_whatif", "workload": name, "stream": stream, "wgs": wgs, "bits": bits,
               "prop_us": round(timeit(fwd), 1), "layer_us": round(timeit(lay), 1)}
        print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
lib.mb_sell_debug_set(0)
