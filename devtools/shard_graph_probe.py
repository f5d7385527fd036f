#!/usr/bin/env python
"""HIP-graph capture of the C-ABI sharded propagation (rbg_lightgcn_forward_sharded_f32: library-issued grouped
ncclSend / ncclRecv on the library's comm stream) on a one-rank communicator that exchanges every third row with itself.
Prints one JSON line; run under `timeout` — r02 saw the torch.distributed variant hang at process-group teardown."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from recbole_gnn_amd import synth

sh = rbg.sharded
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
import faulthandler
faulthandler.enable()
name = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
single = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rbg.set_option("shard_single_stream", single)
def stage(msg):
    print(f"[stage] {msg}", file=sys.stderr, flush=True)
d, k_layers = 64, 3
uid, iid, nu, ni = synth.make(name)
plan = sh.self_exchange_plan(sh.build_plans(uid, iid, nu, ni, 1)[0], 3)
n = plan.n_owned
shard = sh.RcclShard(plan, sh.comm_unique_id(), dev, nranks=1, rank=0, d_max=d)
x = torch.randn(n, d, device=dev)
out_e, out_g = torch.empty_like(x), torch.empty_like(x)
layers = torch.empty(k_layers, n, d, device=dev)
shard.forward_into(x, k_layers, out_e, layers)
torch.cuda.synchronize()
ref = out_e.clone()

def timeit(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

res = {"workload": name, "n": n, "halo_rows": int(plan.n_halo), "single_stream": single}
stage("eager ok")
res["eager_us"] = timeit(lambda: shard.forward_into(x, k_layers, out_e, layers))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    shard.forward_into(x, k_layers, out_g, layers)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
stage("side-stream warm-up ok")
t0 = time.time()
with torch.cuda.graph(g):
    shard.forward_into(x, k_layers, out_g, layers)
res["capture_s"] = round(time.time() - t0, 3)
stage("capture ok")
same = []
for _ in range(3):
    out_g.zero_()
    g.replay()
    torch.cuda.synchronize()
    same.append(bool(torch.equal(out_g, ref)))
res["replay_bit_identical"] = same
stage("replays ok")
res["replay_us"] = timeit(g.replay)
# a changed input is picked up (the graph reads x in place)
x.mul_(0.5)
shard.forward_into(x, k_layers, out_e, layers); torch.cuda.synchronize()
ref2 = out_e.clone()
g.replay(); torch.cuda.synchronize()
res["replay_follows_input"] = bool(torch.equal(out_g, ref2))
stage("before teardown")
del g
torch.cuda.synchronize()
shard.close()
res["clean_exit"] = True
print(json.dumps(res), flush=True)
