#!/usr/bin/env python
"""A/B of the propagation (K = 3, Gowalla / Yelp2018 shapes, d = 64) between the r03 library (devtools/ab/librbgnn_r03.so, plan
attached from sell.py) and the current one (native plan), same process, alternating, HIP-graph replay."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import recbole_gnn_amd as rbg
from recbole_gnn_amd import sell

vp, i64 = ctypes.c_void_p, ctypes.c_int64
old = ctypes.CDLL(os.path.join(ROOT, "devtools", "ab", "librbgnn_r03.so"))
new = rbg._lib.lib
for lib in (old,):
    lib.rbg_last_error.restype = ctypes.c_char_p
    lib.rbg_graph_create.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, ctypes.c_int, ctypes.c_uint32]
    lib.rbg_lightgcn_forward_f32.argtypes = [ctypes.POINTER(vp), ctypes.c_int, i64, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, vp]
    lib.rbg_graph_attach_sell.argtypes = [vp, ctypes.c_int, vp, i64, vp, vp, vp, vp]
    lib.rbg_graph_sell_set_factors.argtypes = [vp, vp]
    lib.rbg_graph_device_arrays.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
dev = torch.device("cuda:0")


def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters): fn()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[2]


for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["gowalla", "yelp2018"]):
    uid, iid, nu, ni = rbg.synth.make(name)
    n, d, K = nu + ni, 64, 3
    uid, iid = np.ascontiguousarray(uid), np.ascontiguousarray(iid)
    gn = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=dev)
    go = vp()
    assert old.rbg_graph_create(ctypes.byref(go), nu, ni, len(uid), uid.ctypes.data, iid.ctypes.data, 0, 0) == 0
    plan = sell.build_plan(*gn.device_csr(), nu, ni, W=32)
    ub, nun = (ctypes.c_int32 * 2)(*plan["unit_base"]), (ctypes.c_int32 * 2)(*plan["n_units"])
    assert old.rbg_graph_attach_sell(go, 32, plan["ent"].data_ptr(), plan["n_ent"], plan["head"].data_ptr(), ub, nun, plan["orig"].data_ptr()) == 0, old.rbg_last_error()
    assert old.rbg_graph_sell_set_factors(go, plan["factors"].data_ptr()) == 0
    uw, iw = torch.randn(nu, d, device=dev), torch.randn(ni, d, device=dev)
    o1, o2, L = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(K, n, d, device=dev)
    arr_o, arr_n = (vp * 1)(go), (vp * 1)(gn.ptr)

    def f_old():
        old.rbg_lightgcn_forward_f32(arr_o, 1, nu, uw.data_ptr(), iw.data_ptr(), o1.data_ptr(), L.data_ptr(), d, K, 2, torch.cuda.current_stream().cuda_stream)

    def f_new():
        new.rbg_lightgcn_forward_f32(arr_n, 1, nu, vp(uw.data_ptr()), vp(iw.data_ptr()), vp(o2.data_ptr()), vp(L.data_ptr()), d, K, 2,
                                     vp(torch.cuda.current_stream().cuda_stream))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f_old(); f_new(); torch.cuda.synchronize()
        rec = {"workload": name, "equal": bool(torch.equal(o1, o2)), "maxdiff": float((o1 - o2).abs().max())}
        rec["us"] = [[timeit(f_old), timeit(f_new)] for _ in range(3)]
    print(json.dumps(rec), flush=True)
