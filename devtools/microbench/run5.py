#!/usr/bin/env python
"""Does workgroup b run on XCD b % 8?  (observed dispatch rule the SpMM's XCD pinning relies on for SPEED only)"""
import ctypes, json, os
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(HERE, "libmb.so"))
lib.mb_xcc_census.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
for blocks in (64, 1024, 6472, 65536):
    out = torch.full((blocks,), -1, dtype=torch.int32, device=dev)
    lib.mb_xcc_census(ctypes.c_void_p(out.data_ptr()), blocks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    x = out.cpu().numpy()
    match = float(np.mean(x == (np.arange(blocks) % 8)))
    print(json.dumps(dict(kind="xcc_census", blocks=blocks, fraction_on_xcd_b_mod_8=match, xcc_ids_seen=sorted(set(x.tolist())),
                          per_xcd=np.bincount(x, minlength=8).tolist())))
