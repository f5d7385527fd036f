#!/usr/bin/env python
"""fp32 MFMA instruction rates: 16 instructions per loop trip, TFLOP/s over the chip."""
import ctypes, json, os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfma_rate.so"))
vp = ctypes.c_void_p
dev = torch.device("cuda:0")
out = torch.zeros(16, device=dev)
for shape, flops in ((32, 4096), (16, 2048)):
    for nacc in ((1, 2, 4) if shape == 32 else (1, 2, 4, 8)):
        for wps in (1, 2):
            blocks, iters = 256 * wps, 4000
            lib.mb_mfma_rate(vp(out.data_ptr()), blocks, iters, shape, nacc, vp(0)); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); lib.mb_mfma_rate(vp(out.data_ptr()), blocks, iters, shape, nacc, vp(0)); b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            n_inst = blocks * 4 * iters * 16
            print(json.dumps(dict(kind="mfma_rate", shape=f"{shape}x{shape}", accumulators=nacc, waves_per_simd=wps, ms=round(ms, 3),
                                  TFLOPs=round(n_inst * flops / ms / 1e9, 1), ns_per_inst_per_simd=round(ms * 1e6 / (iters * 16 * wps), 2))), flush=True)
