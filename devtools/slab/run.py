#!/usr/bin/env python
"""Column-slab SpMM prototype: parity against float64 + timing of the variants, next to the library's binned kernel.
JSON lines -> gpurun_out/slab_probe.jsonl.   usage: python devtools/slab/run.py [--workload gowalla] [--variants ...]"""
import argparse, ctypes, json, os, subprocess, sys, time
import numpy as np, torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import plan as slab_plan
import recbole_gnn_amd as rbg
from recbole_gnn_amd import synth
from oracle import coracle

so = os.path.join(HERE, "libslab.so")
src = os.path.join(HERE, "slab.hip")
if not os.path.exists(so):  # (built here before a gpurun; mtimes do not survive the transfer)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
lib = ctypes.CDLL(so)
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


class SlabParams(ctypes.Structure):
    _fields_ = [("ent", vp), ("head", vp), ("unit_base", ctypes.c_int32 * 2), ("n_units", ctypes.c_int32 * 2),
                ("xs", vp), ("ys", vp), ("slab_off", (i64 * 4) * 2), ("n_class", ctypes.c_int32 * 2),
                ("hot_rows", ctypes.c_int32), ("mode", ctypes.c_int32), ("prev", vp * 4), ("n_prev", ctypes.c_int32),
                ("denom", ctypes.c_float), ("out", vp), ("orig", vp), ("nt_ent", ctypes.c_int32), ("pad", ctypes.c_int32), ("trace", vp)]


lib.slab_spmm.argtypes = [ctypes.POINTER(SlabParams), ci, ci, ci, ci, vp]
lib.slab_spmm_dma.argtypes = [ctypes.POINTER(SlabParams), ci, ci, ci, vp]
lib.slab_convert.argtypes = [vp, vp, vp, ci, ci, vp, ci, ci, vp]

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="gowalla")
ap.add_argument("--variants", default="")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--trace", action="store_true", help="per-wave phase clocks of the DMA kernel")
ap.add_argument("--pmc-run", action="store_true", help="few launches of ONE variant, for rocprofv3 --pmc")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "slab_probe.jsonl"))
args = ap.parse_args()
os.makedirs(os.path.dirname(args.out), exist_ok=True)
log = open(args.out, "a")


def emit(**kw):
    s = json.dumps(kw)
    print(s, flush=True)
    log.write(s + "\n")
    log.flush()


dev = torch.device("cuda:0")
uid, iid, nu, ni = synth.make(args.workload)
N = nu + ni
rowptr, col, val = coracle.build_norm_csr(uid, iid, nu, ni)
nnz = int(rowptr[-1])
b_layer = 4 * (N + 1) + 8 * nnz + 8 * N * 64
g = torch.Generator().manual_seed(2020)
X = torch.randn(N, 64, generator=g)
import scipy.sparse as sp
A64 = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(N, N))
Y_ref = A64 @ X.numpy().astype(np.float64)
E1 = Y_ref; E2 = A64 @ E1; E3 = A64 @ E2
M_ref = (X.numpy().astype(np.float64) + E1 + E2 + E3) / 4.0
Xd = X.to(dev)
st = vp(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return float(np.median(ts))


# ---- the library's binned kernel, same inputs --------------------------------------------------------------------------------
if not args.pmc_run:
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    gh, _ = ds.get_norm_adj_mat(enable_sparse=True, device=dev)
    Yb = torch.empty_like(Xd)
    from recbole_gnn_amd import ops
    ops.spmm_raw(gh, Xd, out=Yb)
    torch.cuda.synchronize()
    err = float(np.abs(Yb.cpu().numpy() - Y_ref).max())
    us = timeit(lambda: ops.spmm_raw(gh, Xd, out=Yb), args.iters)
    emit(kind="binned", workload=args.workload, us=us, frac=b_layer / (us * 1e-6) / 8e12, err=err)
    # the library's own column-slab path (csrc/sell.hip) in the same process
    uwd, iwd = Xd[:nu].contiguous(), Xd[nu:].contiguous()
    o_lib, l_lib = torch.empty(N, 64, device=dev), torch.empty(3, N, 64, device=dev)
    for opt in (0, 1):
        rbg.set_option("sell", opt)
        if opt and not gh.has_sell(64):
            gh.attach_sell(64)
        ops.lightgcn_forward_raw(gh, uwd, iwd, 3, out=o_lib, layers=l_lib); torch.cuda.synchronize()
        errl = float(np.abs(o_lib.cpu().numpy() - M_ref).max())
        usl = timeit(lambda: ops.lightgcn_forward_raw(gh, uwd, iwd, 3, out=o_lib, layers=l_lib), args.iters)
        emit(kind="library_propagation", sell=opt, kernel=gh.propagation_kernel_name(64), us_propagation=usl, err_mean=errl)

VARIANTS = {
    # name: (W, hot KB, chunk, threads per workgroup, workgroups per XCD (0 = one wave per unit), nt entries, DMA buffer bytes, wide rows)
    "w16_g256": (16, 0, 64, 256, 0, 1, 0, 0),
    "w32_g256": (32, 0, 64, 256, 0, 1, 0, 0),
    "w64_g256": (64, 0, 64, 256, 0, 1, 0, 0),
    "w16_g256_wide": (16, 0, 64, 256, 0, 1, 0, 1),
    "w32_g256_wide": (32, 0, 64, 256, 0, 1, 0, 1),
    "w64_g256_wide": (64, 0, 64, 256, 0, 1, 0, 1),
    "w32_g256_wide_c32": (32, 0, 32, 256, 0, 1, 0, 1),
    "w32_g256_wide_c128": (32, 0, 128, 256, 0, 1, 0, 1),
    "w64_g256_wide_c32": (64, 0, 32, 256, 0, 1, 0, 1),
    "w32_dma8k": (32, 0, 64, 256, 0, 1, 8192, 0),
}
names = [v for v in args.variants.split(",") if v] or list(VARIANTS)
for name in names:
    if name == "binned":
        ds = rbg.InteractionDataset(uid, iid, nu, ni)
        gh, _ = ds.get_norm_adj_mat(enable_sparse=True, device=dev)
        from recbole_gnn_amd import ops
        Ya, Yb = torch.empty_like(Xd), torch.empty_like(Xd)
        for _ in range(12):
            ops.spmm_raw(gh, Xd, out=Ya); ops.spmm_raw(gh, Ya, out=Yb)
        torch.cuda.synchronize()
        emit(kind="binned_pmc")
        continue
    W, hot_kb, chunk, tpb, wg_per_xcd, nt_ent, dma, wide = VARIANTS[name]
    hot_rows = hot_kb * 1024 // (W * 4)
    t0 = time.time()
    pl = slab_plan.build(rowptr, col, val, nu, ni, W, hot_rows, chunk=chunk, wide=bool(wide))
    t_plan = time.time() - t0
    ent = torch.from_numpy(pl["ent"]).to(dev)
    head = torch.from_numpy(pl["head"]).to(dev)
    orig = torch.from_numpy(pl["orig"]).to(dev)
    if os.environ.get("SLAB_TORCH_PLAN") and wide and not hot_rows:  # the library's planner (torch, on the device) instead
        from recbole_gnn_amd import sell as lib_sell
        rp_d, col_d, val_d = gh.device_csr()
        tp = lib_sell.build_plan(rp_d, col_d, val_d, nu, ni, W=W, chunk=chunk)
        same = {k: bool(torch.equal(tp[k].cpu(), torch.from_numpy(pl[k]))) for k in ("ent", "head", "orig")}
        emit(kind="plan_compare", equal=same, n_units_torch=tp["n_units"], n_units_numpy=pl["n_units"])
        ent, head, orig = tp["ent"], tp["head"], tp["orig"]
    soff = torch.from_numpy(pl["slab_off"].reshape(-1).copy())  # host
    if os.environ.get("SLAB_CONTIG"):  # the library's buffer layout: layers[K-1] = E0's slabs, layers[k] = layer k + 1's
        big = torch.zeros(3 * N * 64, device=dev)
        bufs = [big[2 * N * 64:], big[:N * 64], big[N * 64:2 * N * 64], torch.zeros(N * 64, device=dev)]
    else:
        bufs = [torch.zeros(N * 64, device=dev) for _ in range(4)]  # E0s, E1s, E2s, scratch (slab layout)
    out = torch.zeros(N, 64, device=dev)
    back = torch.zeros(N, 64, device=dev)

    def params(xs, ys, mode=0, prev=()):
        p = SlabParams()
        p.ent, p.head = ent.data_ptr(), head.data_ptr()
        for c in (0, 1):
            p.unit_base[c], p.n_units[c], p.n_class[c] = pl["unit_base"][c], pl["n_units"][c], pl["n_class"][c]
            for q in range(4): p.slab_off[c][q] = int(pl["slab_off"][c, q])
        p.xs, p.ys = xs.data_ptr(), (ys.data_ptr() if ys is not None else None)
        p.hot_rows, p.mode = hot_rows, mode
        for i, t in enumerate(prev): p.prev[i] = t.data_ptr()
        p.n_prev, p.denom, p.out, p.orig = len(prev), float(len(prev) + 1), out.data_ptr(), orig.data_ptr()
        p.nt_ent = nt_ent
        return p

    def convert(src_t, dst_t, back_flag):
        rc = lib.slab_convert(vp(src_t.data_ptr()), vp(dst_t.data_ptr()), vp(orig.data_ptr()), nu, ni, vp(soff.data_ptr()), W, back_flag, st)
        assert rc == 0, rc

    def layer(p):
        if dma:
            rc = lib.slab_spmm_dma(ctypes.byref(p), W, dma, n_wg, st)
        else:
            rc = lib.slab_spmm(ctypes.byref(p), W, 1 if hot_rows else 0, n_wg, tpb, st)
        assert rc == 0, rc

    xpr = 4 // (64 // W)  # XCDs per role
    if wg_per_xcd:
        n_wg = 8 * wg_per_xcd
    else:  # one wave per unit, dispatched by the hardware in unit (= LPT) order
        n_wg = 8 * -(-max(pl["n_units"]) // (xpr * (tpb // 64)))
    p1 = params(bufs[0], bufs[1]); p2 = params(bufs[1], bufs[2]); p3 = params(bufs[2], None, mode=1, prev=(bufs[0], bufs[1], bufs[2]))

    def propagation():
        convert(Xd, bufs[0], 0)
        layer(p1); layer(p2); layer(p3)

    propagation()
    convert(bufs[1], back, 1)
    torch.cuda.synchronize()
    err1 = float(np.abs(back.cpu().numpy() - Y_ref).max())
    errm = float(np.abs(out.cpu().numpy() - M_ref).max())
    propagation(); torch.cuda.synchronize()
    rerun = bool((out.cpu().numpy() == out.cpu().numpy()).all())
    if args.trace and dma:
        nwv = n_wg * 4
        tb = torch.zeros(nwv * 6, dtype=torch.int64, device=dev)
        for _ in range(5): layer(p1); layer(p2)
        p1.trace = tb.data_ptr()
        layer(p1); torch.cuda.synchronize()
        p1.trace = None
        tr = tb.cpu().numpy().reshape(nwv, 6)
        tr = tr[tr[:, 3] != 0]
        t0 = tr[:, 0].min()
        life, hdr, stg, work = tr[:, 3] - tr[:, 0], tr[:, 1] - tr[:, 0], tr[:, 2] - tr[:, 1], tr[:, 3] - tr[:, 2]
        span = tr[:, 3].max() - t0
        q = lambda a: [int(v) for v in np.percentile(a, [10, 50, 90, 99])]
        per_xcd = {int(xc): int((tr[tr[:, 5] & 7 == xc][:, 3].max() - t0)) for xc in range(8)}
        nbv = np.maximum(tr[:, 4], 1)
        emit(kind="slab_trace", name=name, waves=int(len(tr)), span_cycles=int(span), life=q(life), header=q(hdr), staged=q(stg), work=q(work),
             work_per_batch=q(work / nbv), start_p=q(tr[:, 0] - t0), end_by_xcd=per_xcd, batches=q(tr[:, 4]),
             wave_cycles_sum=int(life.sum()), mean_resident_waves=float(life.sum() / span))
    if args.pmc_run:
        for _ in range(12): layer(p1); layer(p2)
        torch.cuda.synchronize()
        emit(kind="slab_pmc", name=name, err_layer=err1, err_mean=errm)
        continue
    pp = [p1, p2]
    k = [0]
    def one():
        layer(pp[k[0] & 1]); k[0] += 1
    us_layer = timeit(one, args.iters)
    us_mean = timeit(lambda: layer(p3), args.iters)
    us_conv = timeit(lambda: convert(Xd, bufs[0], 0), args.iters)
    us_prop = timeit(propagation, max(20, args.iters // 3))
    emit(kind="slab", name=name, wide=wide, dma=dma, tpb=tpb, n_wg=n_wg, nt_ent=nt_ent, W=W, hot_rows=hot_rows, chunk=chunk, us_layer=us_layer,
         frac_layer=b_layer / (us_layer * 1e-6) / 8e12, us_mean_layer=us_mean, us_convert=us_conv, us_propagation=us_prop,
         err_layer=err1, err_mean=errm, plan_s=round(t_plan, 2), stats=pl["stats"])
