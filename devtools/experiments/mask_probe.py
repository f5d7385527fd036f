"""[EXPERIMENT, needs the temporary exec-mask patch of sell_kernel.h] the propagation with the slab layers' gathers of internal
columns below T EXEC-MASKED (no instruction lanes at all for them; wrong results, timing only): does the address units' cost scale
with the ACTIVE lanes of a gather instruction?  (The r04 skip-hot probe kept the lanes active and pushed their offsets out of
range: 92.5 -> 81.0 us with everything skipped.)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
dev = torch.device("cuda:0")
for shape in ("gowalla", "amazon-book"):
    uid, iid, nu, ni = rbg.synth.make(shape)
    g = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    torch.manual_seed(0)
    uw, iw = torch.randn(nu, 64, device=dev), torch.randn(ni, 64, device=dev)
    out = torch.empty(nu + ni, 64, device=dev); layers = torch.empty(3, nu + ni, 64, device=dev)
    res = {"shape": shape}
    for T in (0, 1, 1024, 4096, 16384, 1 << 22):
        rbg.set_option("sell_nt", (T * 128) << 8)
        for _ in range(3):
            rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=out, layers=layers)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                rbg.ops.lightgcn_forward_raw(g, uw, iw, 3, out=out, layers=layers)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            gr.replay()
        torch.cuda.synchronize(); a.record()
        for _ in range(5):
            gr.replay()
        b.record(); torch.cuda.synchronize()
        res[f"T={T}"] = round(a.elapsed_time(b) * 1e3 / 100, 1)
    rbg.set_option("sell_nt", 0)
    print(json.dumps(res), flush=True)
