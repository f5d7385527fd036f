#!/usr/bin/env python
"""fp32 MFMA (v_mfma_f32_32x32x2_f32) issue rate vs waves per SIMD and independent accumulator chains."""
import ctypes, json, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmb.so")
lib = ctypes.CDLL(so)
lib.mb_mfma_rate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(16, device=dev)
iters = 2000
for waves_per_simd in (1, 2, 3, 4):
    for chains in (1, 2, 4):
        blocks, threads = 256 * waves_per_simd, 256  # one 4-wave block per CU per resident wave
        def go():
            lib.mb_mfma_rate(ctypes.c_void_p(out.data_ptr()), blocks, threads, iters, chains, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        go(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); go(); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        n_mfma = blocks * 4 * iters * 32
        tflops = n_mfma * 4096 / ms / 1e9
        per_simd_cycles = ms * 1e-3 * 2.4e9 / (n_mfma / 1024)
        print(json.dumps(dict(kind="mfma_f32_rate", waves_per_simd=waves_per_simd, chains=chains, ms=round(ms, 3),
                              TFLOPs=round(tflops, 1), cycles_per_mfma_at_2p4GHz=round(per_simd_cycles, 1))), flush=True)
