#!/usr/bin/env python
"""r05: what each part of score_topk_kernel costs on PROPAGATED embeddings (the bench's case) — the RBG_TOPK_TRACE build's what-if
switches (results are wrong on purpose): 1 = no product, 2 = no filter, 4 = no fetch / publish after the first tile, 8 = nothing passes the threshold, 16 = no raised priority between the product and the barrier.
`prof` = five plain calls of the product library for rocprofv3.  -> gpurun_out/r05_topk_whatif.jsonl"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
prof = len(sys.argv) > 1 and sys.argv[1] == "prof"
plain = len(sys.argv) > 1 and sys.argv[1] == "time"  # the product library, timed
if not prof and not plain:
    os.environ["RBGNN_LIB"] = os.path.join(HERE, "microbench", "librbgnn_topktrace.so")
sys.path.insert(0, ROOT)
import torch
import recbole_gnn_amd as rbg

dev = torch.device("cuda:0")
lib = rbg._lib.lib
uid, iid, nu, ni = rbg.synth.make("gowalla")
ds = rbg.InteractionDataset(uid, iid, nu, ni)
torch.manual_seed(0)
model = rbg.LightGCN({"device": str(dev), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "require_pow": True}, ds)
g = torch.Generator().manual_seed(1)
users = torch.randint(1, nu, (4096,), generator=g).to(dev)
with torch.no_grad():
    ua, it = model.forward()
    ua, it = ua.contiguous(), it.contiguous()
call = lambda: rbg.full_sort_topk(model.graph, ua, it, users, 10)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]


if prof:
    for _ in range(5): call()
    torch.cuda.synchronize()
    sys.exit(0)
log = open(os.path.join(ROOT, "gpurun_out", "r05_topk_whatif.jsonl"), "a")
if plain:
    ur, ir = torch.randn(nu, 64, device=dev) * 0.1, torch.randn(ni, 64, device=dev) * 0.1
    rnd = lambda: rbg.full_sort_topk(model.graph, ur, ir, users, 10)
    rec = {"what": "topk product library", "tag": sys.argv[2] if len(sys.argv) > 2 else "", "propagated_us": [round(timeit(call), 1) for _ in range(3)],
           "random_us": [round(timeit(rnd), 1) for _ in range(3)]}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n")
    sys.exit(0)
for bits in (0, 32, 0, 32, 8):
    assert lib.mb_topk_debug_set(bits) == 0
    rec = {"what": "topk_whatif", "bits": bits, "us": round(timeit(call), 1)}
    print(json.dumps(rec), flush=True); log.write(json.dumps(rec) + "\n"); log.flush()
lib.mb_topk_debug_set(0)
