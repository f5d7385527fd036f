#!/usr/bin/env python
"""3 x bf16 split GEMM on the bf16 matrix cores vs the exact-fp32 MFMA: accuracy against float64 and instruction rate."""
import ctypes, json, os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libbf16x3.so"))
vp = ctypes.c_void_p
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, n, d, scale) in [(128, 256, 64, 1.0), (128, 256, 256, 1.0), (64, 128, 64, 0.01), (64, 96, 128, 100.0)]:
    U = (torch.randn(B, d, generator=g) * scale).to(dev)
    I = (torch.randn(n, d, generator=g) * scale).to(dev)
    ref = U.double().cpu() @ I.double().cpu().T
    res = {}
    for terms in (0, 1, 3, 6):
        S = torch.zeros(B, n, device=dev)
        lib.mb_bf16x3_gemm(vp(U.data_ptr()), vp(I.data_ptr()), vp(S.data_ptr()), B, n, d, terms, vp(0))
        torch.cuda.synchronize()
        res[terms] = float((S.double().cpu() - ref).abs().max() / ref.abs().max())
    t32 = float(((U @ I.T).double().cpu() - ref).abs().max() / ref.abs().max())
    print(json.dumps(dict(kind="bf16x3_accuracy", B=B, n=n, d=d, scale=scale, rel_err_fp32_mfma=res[0], rel_err_1term=res[1],
                          rel_err_3terms=res[3], rel_err_6terms=res[6], rel_err_torch_matmul=t32)), flush=True)
out = torch.zeros(16, device=dev)
for mode, name, flops_per_iter in ((0, "fp32 mfma 32x32x2 x8 (16 k)", 8 * 4096), (1, "bf16 mfma 32x32x16 x6 (16 k, fp32-equivalent)", 8 * 4096)):
    for wps in (1, 2, 4):
        blocks, iters = 256 * wps, 4000
        lib.mb_bf16x3_rate(vp(out.data_ptr()), blocks, iters, mode, vp(0)); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.mb_bf16x3_rate(vp(out.data_ptr()), blocks, iters, mode, vp(0)); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        tf = blocks * 4 * iters * flops_per_iter / ms / 1e9
        print(json.dumps(dict(kind="bf16x3_rate", mode=name, waves_per_simd=wps, ms=round(ms, 3), fp32_equivalent_TFLOPs=round(tf, 1))), flush=True)
