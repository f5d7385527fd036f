"""r04: the column-slab plan as a property of the C ABI (csrc/sell_plan.hip: rbg_graph_plan_sell inside rbg_graph_create*).

* the native planner's arrays equal the executable specification's (tests/sell_spec.py) bit for bit;
* a caller that binds librbgnn.so with raw ctypes only (rbg_graph_create -> rbg_lightgcn_forward_f32,
  the sites layers.py:19-20 / lightgcn.py:74-76) runs sell_spmm_kernel and matches the oracle;
* every caller of the plain product goes through the plan: the NGCF layer (layers.py:54-58) contiguous and as a column block
  of the concatenated buffer (ngcf.py:100), the SimGCL noise epilogue (simgcl.py:29-33), re-weighted views (ngcf.py:74-90),
  per-layer graph lists (sgl.py:89-91), d = 32;
* every form of the propagation over the plan (slab chain, row-major chain, plain layer, accumulate, noise) against float64, bit-stable;
* every allocation of the plan code can fail without leaving a dangling pointer (fault injection: option "fail_alloc_after");
* parity at the north_star's named scale (1.3 M nodes) and at the single-GPU Amazon-Book shape through the plan.
"""
import ctypes
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

import sell_spec as sell
from conftest import ROOT
from oracle import coracle as C
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5


def close(got, ref, tol=TOL):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(1.0, float(np.abs(ref).max()) if ref.size else 1.0)
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) if ref.size else 0.0
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e}"
    return err


def randn(shape, seed, dev):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float32).to(dev)


def hub_graph(rbg, seed=5, nu=3001, ni=2201, e=60_000):
    """power-law graph + hubs (user 1: 1 500 items, user 7: 700, item 3: 900 users), empty rows, the PAD rows"""
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=seed)
    hub_u = np.concatenate([uid, np.full(1500, 1), np.full(700, 7), (np.arange(900) * 3 % (nu - 1) + 1)]).astype(np.int64)
    hub_i = np.concatenate([iid, (np.arange(1500) % (ni - 1) + 1), (np.arange(700) * 3 % (ni - 1) + 1), np.full(900, 3)]).astype(np.int64)
    key = np.unique(hub_u * ni + hub_i)
    return key // ni, key % ni, nu, ni


def graphs(rbg, name):
    if name == "hubs":
        return hub_graph(rbg)
    if name == "duplicates":  # duplicated interactions stay separate entries (equal keys in the planner's sort)
        rng = np.random.default_rng(3)
        nu, ni = 60, 90
        uid, iid = rng.integers(1, nu, 2500), rng.integers(1, ni, 2500)
        return np.concatenate([uid, uid[:400]]), np.concatenate([iid, iid[:400]]), nu, ni
    return rbg.synth.make(name)


# ---- the planner ------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("W,chunk", [(32, 0), (32, 4), (32, 16), (64, 0), (64, 8)])
@pytest.mark.parametrize("name", ["toy", "ml-100k", "hubs", "duplicates"])
def test_native_plan_equals_the_specification(rbg, cuda, name, W, chunk):
    """rbg_graph_plan_sell (rocPRIM sorts / scans + one-pass kernels) against sell.build_plan (torch ops) on the handle's own
    device CSR: entries (offsets AND value bits), unit headers, row numbering — bit for bit; the factors to one ulp (float64
    pow(-0.5) there, 1 / sqrt in double here); the slot -> CSR position map reproduces every entry."""
    import sell_spec as sell
    uid, iid, nu, ni = graphs(rbg, name)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    assert h.sell_status() == "planned"
    info = h.plan_sell(W=W, chunk=chunk)
    spec = sell.build_plan(*h.device_csr(), nu, ni, W=W, chunk=chunk or sell.CHUNK)
    arr = h.sell_arrays()
    assert info["n_ent"] == spec["n_ent"] and info["n_units"] == spec["n_units"] and info["W"] == W
    assert torch.equal(arr["head"], spec["head"]), "unit headers differ"
    assert torch.equal(arr["orig"], spec["orig"]), "row numbering differs"
    assert torch.equal(arr["ent"], spec["ent"][: spec["n_ent"]]), "entries differ"
    assert info["factored"] and spec["factors"] is not None
    f_nat, f_spec = arr["factors"].view(torch.int32), spec["factors"].view(torch.int32)
    assert int((f_nat - f_spec).abs().max()) <= 1
    # src: the CSR position of every slot
    rowptr, col, val = h.device_csr()
    src = arr["src"].long()
    real = src >= 0
    assert int(real.sum()) == h.nnz and torch.equal(real, arr["ent"][:, 0] != sell.K_PAST)
    assert torch.equal(arr["ent"][real, 1], val[src[real]].view(torch.int32))
    assert torch.equal(torch.sort(src[real]).values, torch.arange(h.nnz, device=cuda))
    # and the product through it
    x = randn((nu + ni, {32: 64, 64: 128}[W]), 4, cuda)
    rp, cl, vl = C.build_norm_csr(uid, iid, nu, ni)
    assert h.spmm_kernel_name(x.shape[1]).startswith(f"sell_spmm_kernel<{W}, {x.shape[1] // W}, false")
    close(rbg.ops.spmm_raw(h, x), O.conv_csr_f64(x.cpu().numpy().astype(np.float64), rp, cl, vl))


def test_plan_reasons_are_visible(rbg, cuda):
    """r06: a hub row of any length is planned (U units; r03-r05 refused a hub beyond 4 LGW x max(512, nnz / 8192) entries and
    kept the binned kernel).  A graph outside the plan's reach keeps the binned kernel and says why (rbg_graph_sell_status):
    a CSR without a user / item boundary; a host graph; planning switched off."""
    lgw = 8
    hub = 512 * 4 * lgw + 1000
    nu, ni = 400, hub + 10
    u = np.concatenate([np.full(hub, 1), np.arange(1000) % (nu - 2) + 2]).astype(np.int64)
    i = np.concatenate([np.arange(hub) + 1, (np.arange(1000) * 7919) % (ni - 1) + 1]).astype(np.int64)
    key = np.unique(u * ni + i)
    u, i = key // ni, key % ni
    h = rbg.GraphHandle.from_interactions(u, i, nu, ni, device=cuda)
    assert h.has_sell(64) and h.sell_status() == "planned" and "sell_spmm_kernel" in h.propagation_kernel_name(64)
    x = randn((nu + ni, 64), 1, cuda)
    rp, cl, vl = C.build_norm_csr(u, i, nu, ni)
    x64 = x.cpu().numpy().astype(np.float64)  # (float64: the hub row is a sum of 17 384 terms, the fp32 restatement itself is 1e-4 off)
    l1 = O.conv_csr_f64(x64, rp, cl, vl)
    want = (x64 + l1 + O.conv_csr_f64(l1, rp, cl, vl)) / 3
    got = rbg.ops.lightgcn_forward_raw(h, x[:nu].contiguous(), x[nu:].contiguous(), 2)[0]
    close(got, want)
    for _ in range(3):  # the hub's 17 units arrive in any order: the last one adds them in unit order — the same bits every time
        assert torch.equal(rbg.ops.lightgcn_forward_raw(h, x[:nu].contiguous(), x[nu:].contiguous(), 2)[0], got)
    close(rbg.ops.spmm_raw(h, x), l1)
    # a square CSR / COO without stated classes: the bipartite boundary is detected (r04)
    hc = rbg.GraphHandle.from_csr(rp, cl, vl, nu + ni, device=cuda)
    assert hc.has_sell(64) and hc.sell_status() == "planned"
    close(rbg.ops.spmm_raw(hc, x), l1)
    # ... and a graph that is NOT bipartite (a triangle among the users) has no boundary
    tri_rp = np.array([0, 2, 4, 6, 6], dtype=np.int64)
    tri = rbg.GraphHandle.from_csr(tri_rp, np.array([1, 2, 0, 2, 0, 1], dtype=np.int32), np.full(6, 0.5, dtype=np.float32), 4, device=cuda)
    assert not tri.has_sell(64) and "boundary" in tri.sell_status()
    assert rbg.GraphHandle.from_interactions(u, i, nu, ni).sell_status() == "host graph"
    rbg.set_option("sell_auto", 0)
    try:
        assert "disabled" in rbg.GraphHandle.from_interactions(u[:50], i[:50], nu, ni, device=cuda).sell_status()
    finally:
        rbg.set_option("sell_auto", 1)
    # a CSR graph WITH the two row classes and a bipartite structure (a shard's interior block) is planned
    hb = rbg.GraphHandle.from_csr(*C.build_norm_csr(u[:3000] % 300 + 1, i[:3000] % 500 + 1, 400, 600), 1000, device=cuda, n_class0_rows=400)
    assert hb.sell_status() == "planned" and hb.spmm_kernel_name(64).startswith("sell_spmm_kernel")


RAW_CALLER = r'''
import ctypes, sys, json
import numpy as np, torch
lib = ctypes.CDLL(sys.argv[1])
z = np.load(sys.argv[2])
uid, iid, nu, ni = np.ascontiguousarray(z["uid"]), np.ascontiguousarray(z["iid"]), int(z["n_users"]), int(z["n_items"])
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.rbg_last_error.restype = ctypes.c_char_p
lib.rbg_graph_create.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, ctypes.c_int, ctypes.c_uint32]
lib.rbg_lightgcn_forward_f32.argtypes = [ctypes.POINTER(vp), ctypes.c_int, i64, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, vp]
lib.rbg_lightgcn_forward_kernel_name.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int]
lib.rbg_spmm_kernel_name.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
lib.rbg_spmm_f32.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
lib.rbg_graph_destroy.argtypes = [vp]
g = vp()
rc = lib.rbg_graph_create(ctypes.byref(g), nu, ni, len(uid), uid.ctypes.data, iid.ctypes.data, 0, 0)
assert rc == 0, lib.rbg_last_error()
d, K = 64, 3
gen = torch.Generator().manual_seed(7)
uw = torch.randn(nu, d, generator=gen).cuda(); iw = torch.randn(ni, d, generator=gen).cuda()
out = torch.empty(nu + ni, d, device="cuda"); layers = torch.empty(K, nu + ni, d, device="cuda")
buf = ctypes.create_string_buffer(128); buf2 = ctypes.create_string_buffer(128)
assert lib.rbg_lightgcn_forward_kernel_name(g, d, 2, buf, 128) == 0 and lib.rbg_spmm_kernel_name(g, d, buf2, 128) == 0
arr = (vp * 1)(g)
s = torch.cuda.current_stream().cuda_stream
rc = lib.rbg_lightgcn_forward_f32(arr, 1, nu, uw.data_ptr(), iw.data_ptr(), out.data_ptr(), layers.data_ptr(), d, K, 2, s)
assert rc == 0, lib.rbg_last_error()
x = torch.cat([uw, iw]); y = torch.empty_like(x)
rc = lib.rbg_spmm_f32(g, x.data_ptr(), y.data_ptr(), d, 0, s)
assert rc == 0, lib.rbg_last_error()
torch.cuda.synchronize()
np.savez(sys.argv[3], out=out.cpu().numpy(), y=y.cpu().numpy(), uw=uw.cpu().numpy(), iw=iw.cpu().numpy())
lib.rbg_graph_destroy(g)
print(json.dumps({"fwd": buf.value.decode(), "spmm": buf2.value.decode(), "modules": sorted(m for m in sys.modules if "recbole" in m or "oracle" in m)}))
'''


def test_raw_ctypes_caller_runs_the_column_slab_kernel(tmp_path):
    """INTEGRATION.md's stub, literally: ctypes.CDLL on librbgnn.so, rbg_graph_create with host id arrays, rbg_lightgcn_forward_f32
    and rbg_spmm_f32 on torch-owned HBM — no recbole_gnn_amd import in that process.  The kernel the library reports is the
    column-slab one and the numbers match the oracle."""
    import json
    script = tmp_path / "raw_caller.py"
    script.write_text(textwrap.dedent(RAW_CALLER))
    out = tmp_path / "out.npz"
    lib = os.path.join(ROOT, "recbole-gnn_amd", "librbgnn.so")
    fix = os.path.join(ROOT, "tests", "golden", "ref_test_inter.npz")
    r = subprocess.run([sys.executable, str(script), lib, fix, str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["modules"] == [], rep
    assert rep["fwd"].startswith("sell_spmm_kernel<32, 2, true") and rep["spmm"].startswith("sell_spmm_kernel<32, 2, false"), rep
    z, f = np.load(out), np.load(fix)
    rp, cl, vl = C.build_norm_csr(f["uid"], f["iid"], int(f["n_users"]), int(f["n_items"]))
    close(z["out"], C.lightgcn_forward(rp, cl, vl, z["uw"], z["iw"], 3))
    close(z["y"], C.spmm(rp, cl, vl, np.concatenate([z["uw"], z["iw"]])))


def test_default_branch_pair_gets_the_plan(rbg, cuda, golden):
    """The reference's DEFAULT branch (enable_sparse unset: get_norm_adj_mat returns (edge_index, edge_weight), dataset.py:77-79;
    LightGCNConv.forward(x, edge_index, edge_weight), layers.py:13-17) hands the adjacency over as a pair without a user / item
    boundary: rbg_graph_create_coo detects the boundary of the bipartite structure, so that handle is planned as well — and a
    model built without enable_sparse runs the column-slab kernel layer by layer."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    ei, ew = rbg.norm_edges(g["uid"], g["iid"], nu, ni)
    h = rbg.ops.graph_from_pair(ei, ew, nu + ni, cuda)
    assert h.sell_status() == "planned" and h.spmm_kernel_name(64).startswith("sell_spmm_kernel<32, 2, false")
    x = randn((nu + ni, 64), 3, cuda)
    close(rbg.ops.spmm_raw(h, x), O.conv_csr_f64(x.cpu().numpy().astype(np.float64), g["rowptr"], g["col"].astype(np.int64), g["val"]))
    ds = rbg.InteractionDataset(g["uid"], g["iid"], nu, ni)
    torch.manual_seed(0)
    model = rbg.LightGCN({"device": str(cuda), "embedding_size": 64, "n_layers": 3}, ds)  # enable_sparse: None, the reference default
    with torch.no_grad():
        u, i = model.forward()
    ref = C.lightgcn_forward(g["rowptr"], g["col"].astype(np.int64), g["val"], model.user_embedding.weight.detach().cpu().numpy(),
                             model.item_embedding.weight.detach().cpu().numpy(), 3)
    close(torch.cat([u, i]), ref)


# ---- every caller of the product --------------------------------------------------------------------------------------------

def _truth_layers(x64, rowptr, col, val, k):
    out, cur = [], x64
    for _ in range(k):
        cur = O.conv_csr_f64(cur, rowptr, col, val)
        out.append(cur)
    return out


@pytest.mark.parametrize("d", [32, 64, 128])
def test_every_form_of_the_propagation_over_the_plan(rbg, cuda, d):
    """The propagation (K = 1..3, forward and backward), the row-major chain, the plain layer, Y += A X and the noise epilogue on
    a graph with split and wide rows: equal to float64 within 1e-5 and bit-stable run to run (the plan fixes the summation
    order); the unfactored chain (option "sell_factored" = 0) too.  (r04's launch-form options — two batches in flight, one launch per row class — were measured slower and left the
    product in r05.)"""
    uid, iid, nu, ni = hub_graph(rbg)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    x = randn((nu + ni, d), 11, cuda)
    gout = randn((nu + ni, d), 12, cuda)
    noise = torch.rand(nu + ni, d, generator=torch.Generator().manual_seed(6)).to(cuda)
    x64 = x.cpu().numpy().astype(np.float64)

    def run():
        res = []
        for k in (1, 2, 3):
            xg = x.clone().requires_grad_(True)
            out = rbg.ops.lightgcn_forward(h, xg[:nu], xg[nu:], k)
            out.backward(gout)
            res += [out.detach().clone(), xg.grad.clone()]
            res.append(rbg.ops.lightgcn_forward_raw(h, x[:nu].contiguous(), x[nu:].contiguous(), k, keep_layers=True)[0].clone())
        res.append(rbg.ops.spmm_raw(h, x).clone())
        acc = x.clone()
        rbg.ops.spmm_raw(h, x, out=acc, accumulate=True)
        res.append(acc)
        res.append(rbg.ops.spmm_noise_raw(h, x, noise, 0.1).clone())
        return res

    assert h.propagation_kernel_name(d) == f"sell_spmm_kernel<32, {d // 32}, true>"
    base = run()
    lay = _truth_layers(x64, rowptr, col, val, 3)
    close(base[6], (x64 + lay[0] + lay[1] + lay[2]) / 4)
    close(base[8], (x64 + lay[0] + lay[1] + lay[2]) / 4)
    close(base[9], lay[0])
    close(base[10], x64 + lay[0])
    g64 = gout.cpu().numpy().astype(np.float64)
    gl = _truth_layers(g64, rowptr, col, val, 3)
    close(base[7], (g64 + gl[0] + gl[1] + gl[2]) / 4)
    for a, b in zip(run(), base):
        assert torch.equal(a, b)
    try:  # r05: the compact launches read 16-bit slab-row numbers (both classes < 65 536 rows); 32-bit offsets give the same bits
        assert rbg.get_option("sell_c16") == 1
        rbg.set_option("sell_c16", 0)
        for a, b in zip(run(), base):
            assert torch.equal(a, b)
    finally:
        rbg.set_option("sell_c16", 1)
    try:  # the unfactored chain differs by rounding only
        rbg.set_option("sell_factored", 0)
        assert h.propagation_kernel_name(d) == f"sell_spmm_kernel<32, {d // 32}, false>"
        close(run()[6], (x64 + lay[0] + lay[1] + lay[2]) / 4)
    finally:
        rbg.set_option("sell_factored", 1)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_noise_epilogue_over_the_plan(rbg, cuda, golden, d):
    """rbg_spmm_noise_f32 (simgcl.py:29-33) on a planned handle runs the column-slab kernel: the noise row's norm spans all the
    slabs of the row; against the torch expression in float64 and against the binned kernel."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    n = nu + ni
    x = randn((n, d), 5, cuda)
    noise = torch.rand(n, d, generator=torch.Generator().manual_seed(6)).to(cuda)
    assert h.spmm_kernel_name(d).startswith("sell_spmm_kernel")
    y = rbg.ops.spmm_noise_raw(h, x, noise, 0.1)
    ax = torch.from_numpy(O.conv_csr_f64(x.cpu().numpy().astype(np.float64), g["rowptr"], g["col"].astype(np.int64), g["val"]))
    ref = ax + torch.sign(ax) * torch.nn.functional.normalize(noise.cpu().double(), dim=-1) * 0.1
    mask = ax.abs() > 1e-6
    assert float((y.cpu().double() - ref)[mask].abs().max()) <= 1e-5
    assert torch.all(y[0] == 0) and torch.all(y[nu] == 0)
    rbg.set_option("sell", 0)
    try:
        yb = rbg.ops.spmm_noise_raw(h, x, noise, 0.1)
    finally:
        rbg.set_option("sell", 1)
    assert float((y - yb)[mask.to(cuda)].abs().max()) <= 2e-6


def test_plain_layer_of_a_plan_without_row_major_entries(rbg, cuda, golden):
    """A plan without the row-major twin of its entries (tables beyond 32-bit byte offsets: the config-#5 shape; here: option
    "sell_rowmajor" = 0) still serves rbg_spmm_f32 / rbg_spmm_noise_f32: X is converted into the handle's slab scratch and the
    launch writes row-major.  The scratch grows when a wider table follows a narrower one on the same handle."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    n = nu + ni
    rbg.set_option("sell_rowmajor", 0)
    try:
        for d in (32, 64, 128, 64):
            x = randn((n, d), 11 + d, cuda)
            assert h.spmm_kernel_name(d).startswith("sell_spmm_kernel"), h.spmm_kernel_name(d)
            y = rbg.ops.spmm_raw(h, x)
            ref = O.conv_csr_f64(x.cpu().numpy().astype(np.float64), g["rowptr"], g["col"].astype(np.int64), g["val"])
            assert close(y, ref) <= 1e-5
            y2 = y.clone()
            rbg.ops.spmm_raw(h, x, out=y2, accumulate=True)  # Y += A X reads Y where it lies
            assert close(y2, 2 * ref) <= 2e-5
            rbg.set_option("sell_rowmajor", 1)
            assert torch.equal(rbg.ops.spmm_raw(h, x), y)  # the two operand layouts of one kernel: the same sums in the same order
            rbg.set_option("sell_rowmajor", 0)
        noise = torch.rand(n, 64, generator=torch.Generator().manual_seed(6)).to(cuda)
        yn = rbg.ops.spmm_noise_raw(h, x, noise, 0.1)
        rbg.set_option("sell_rowmajor", 1)
        assert torch.equal(rbg.ops.spmm_noise_raw(h, x, noise, 0.1), yn)
    finally:
        rbg.set_option("sell_rowmajor", 1)


def test_ngcf_layer_over_the_plan(rbg, cuda):
    """BiGNNConv (layers.py:54-58): the product inside rbg_bignn_conv_f32 / rbg_bignn_layer_f32 / rbg_bignn_backward_f32 runs
    over the plan — also when X is a 64-wide column block of the [N, 256] concatenated buffer (ngcf.py:100; the row stride
    scales the row-major entries' offsets) — and equals the binned path within rounding and float64 within 1e-5."""
    uid, iid, nu, ni = hub_graph(rbg, seed=9)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    n, d = nu + ni, 64
    gen = torch.Generator().manual_seed(3)
    w1, w2 = (torch.randn(d, d, generator=gen) * 0.2).to(cuda), (torch.randn(d, d, generator=gen) * 0.2).to(cuda)
    b1, b2 = torch.randn(d, generator=gen).to(cuda) * 0.1, torch.randn(d, generator=gen).to(cuda) * 0.1
    wide = randn((n, 256), 8, cuda)
    for x in (wide[:, 64:128], wide[:, :64].contiguous()):
        x64 = x.cpu().numpy().astype(np.float64)
        p64 = O.conv_csr_f64(x64, rowptr, col, val)
        ref = (p64 + x64) @ w1.cpu().numpy().astype(np.float64).T + b1.cpu().numpy() + (p64 * x64) @ w2.cpu().numpy().astype(np.float64).T + b2.cpu().numpy()
        y, _ = rbg.ops.bignn_conv_raw(h, x, w1, b1, w2, b2)
        close(y, ref, tol=2e-5)
        rbg.set_option("sell", 0)
        try:
            yb, _ = rbg.ops.bignn_conv_raw(h, x, w1, b1, w2, b2)
        finally:
            rbg.set_option("sell", 1)
        close(y, yb, tol=5e-6)
    # the training layer and its backward
    xg = wide[:, :64].contiguous().requires_grad_(True)
    res = {}
    for sell in (1, 0):
        rbg.set_option("sell", sell)
        try:
            xg.grad = None
            out = rbg.ops.bignn_layer(xg, w1, b1, w2, b2, h, 0.2)
            out.backward(torch.ones_like(out))
            res[sell] = (out.detach().clone(), xg.grad.clone())
        finally:
            rbg.set_option("sell", 1)
    close(res[1][0], res[0][0], tol=5e-6)
    close(res[1][1], res[0][1], tol=2e-5)


def test_reweighted_view_runs_the_plan_after_a_refresh(rbg, cuda):
    """NGCF's edge dropout (ngcf.py:74-90): a re-weighted view of a planned graph borrows the plan and owns a copy of the valued
    entries; rbg_graph_refresh_values rewrites the copy from the caller's array through the slot -> CSR position map.  Weights
    that are NOT symmetric (every directed edge dropped independently), duplicated interactions whose copies get different
    weights, a second rewrite; until the first refresh the view stays on the binned kernel, which reads the array itself."""
    uid, iid, nu, ni = graphs(rbg, "duplicates")
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    n = nu + ni
    x = randn((n, 64), 2, cuda)
    x64 = x.cpu().numpy().astype(np.float64)
    vals = h.values()
    buf = vals.clone()
    view = h.reweighted(buf)
    assert "binned" in view.spmm_kernel_name(64)
    for trial in range(3):
        w = torch.rand(h.nnz, generator=torch.Generator().manual_seed(trial)).to(cuda)
        buf.copy_(vals * (w >= 0.3) * (0.5 + w))
        if trial:
            view.refresh_values()
            assert view.spmm_kernel_name(64).startswith("sell_spmm_kernel<32, 2, false")
        ref = O.conv_csr_f64(x64, rowptr, col, buf.cpu().numpy())
        close(rbg.ops.spmm_raw(view, x), ref)
        mean, layers = rbg.ops.lightgcn_forward_raw(view, x[:nu].contiguous(), x[nu:].contiguous(), 2, keep_layers=True)
        l1 = ref
        l2 = O.conv_csr_f64(l1, rowptr, col, buf.cpu().numpy())
        close(layers[0], l1)
        close(mean, (x64 + l1 + l2) / 3)
    close(rbg.ops.spmm_raw(h, x), O.conv_csr_f64(x64, rowptr, col, val))  # the base graph's own values are untouched
    view.destroy()
    h.destroy()


def test_a_borrowed_plan_outlives_its_views(rbg, cuda):
    """a re-weighted view holds raw pointers into its base handle's plan.  While views live the base's plan is
    neither detached nor replaced (RBG_EUNSUPPORTED, nothing freed); the views keep computing; once they are gone the base
    re-plans."""
    RBG_EUNSUPPORTED, RbgError = rbg._lib.RBG_EUNSUPPORTED, rbg.RbgError
    uid, iid, nu, ni = hub_graph(rbg, seed=3)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    vals = h.values() * 0.5
    v1, v2 = h.reweighted(vals, symmetric=True), h.reweighted(vals, symmetric=True)
    v1.refresh_values(), v2.refresh_values()
    x = randn((nu + ni, 64), 2, cuda)
    want = O.conv_csr_f64(x.cpu().numpy().astype(np.float64), rowptr, col, val * 0.5)
    assert v1.spmm_kernel_name(64).startswith("sell_")
    close(rbg.ops.spmm_raw(v1, x), want)
    for call in (h.detach_sell, lambda: h.plan_sell(W=64), lambda: sell.attach(h, W=32)):
        with pytest.raises(RbgError) as ei:
            call()
        assert ei.value.code == RBG_EUNSUPPORTED and "view" in str(ei.value)
        assert h.sell_status() == "planned"
        close(rbg.ops.spmm_raw(v2, x), want)  # (the borrowed arrays are still there)
    v1.destroy()
    with pytest.raises(RbgError):
        h.detach_sell()  # one view left
    v2.update_values(h.values() * 0.25)  # write + refresh in one call
    close(rbg.ops.spmm_raw(v2, x), want / 2)
    v2.destroy()
    assert h.plan_sell(W=64)["W"] == 64
    h.detach_sell()
    assert not h.has_sell(64)


def test_per_layer_graphs_over_their_own_plans(rbg, cuda, golden):
    """SGL's RW views (sgl.py:89-91): one graph per layer, each with its own plan and row numbering — the layers pass row-major
    between them (rbg_lightgcn_forward_f32 with n_graphs = K)."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    rng = np.random.default_rng(8)
    hs, csrs = [], []
    for k in range(3):
        keep = (rng.random(len(g["uid"])) < 0.8).astype(np.uint8)
        hs.append(rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, keep=keep))
        csrs.append(C.build_norm_csr(g["uid"], g["iid"], nu, ni, keep=keep))
        assert hs[-1].sell_status() == "planned"
    x = randn((nu + ni, 64), 4, cuda)
    cur = x.cpu().numpy().astype(np.float64)
    acc = cur.copy()
    for rp, cl, vl in csrs:
        cur = O.conv_csr_f64(cur, rp, cl, vl)
        acc += cur
    out = rbg.ops.lightgcn_forward_raw(hs, x[:nu].contiguous(), x[nu:].contiguous(), 3)[0]
    close(out, acc / 4)
    rbg.set_option("sell", 0)
    try:
        close(out, rbg.ops.lightgcn_forward_raw(hs, x[:nu].contiguous(), x[nu:].contiguous(), 3)[0], tol=2e-6)
    finally:
        rbg.set_option("sell", 1)


# ---- error paths ------------------------------------------------------------------------------------------------------------

def test_every_allocation_of_the_plan_code_may_fail(rbg, cuda):
    """Fault injection (option "fail_alloc_after" = n: the (n + 1)-th device allocation of the plan code fails once): walk
    rbg_graph_plan_sell, rbg_graph_attach_sell + rbg_graph_sell_set_factors, the view constructor and the backward's lazy
    scratch one allocation at a time; after every failure the handle propagates correctly (with the plan it still has, or the
    binned kernel) and is destroyed cleanly — r03's attach freed entc / rs without clearing them when the row-major twin failed
    (ADVICE r03, medium)."""
    import sell_spec as sell
    uid, iid, nu, ni = rbg.synth.make("toy")
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    x = randn((nu + ni, 64), 3, cuda)
    uw, iw = x[:nu].contiguous(), x[nu:].contiguous()
    ref = C.lightgcn_forward(rowptr, col, val, uw.cpu().numpy(), iw.cpu().numpy(), 3)
    gref = None

    def check(h):
        nonlocal gref
        close(rbg.ops.lightgcn_forward_raw(h, uw, iw, 3)[0], ref)
        xg = x.clone().requires_grad_(True)
        rbg.ops.lightgcn_forward(h, xg[:nu], xg[nu:], 3).sum().backward()
        if gref is None:
            gref = xg.grad.clone()
        close(xg.grad, gref, tol=2e-6)
        close(rbg.ops.spmm_raw(h, x), C.spmm(rowptr, col, val, x.cpu().numpy()))

    rbg.set_option("sell_auto", 0)
    try:
        outcomes = []
        for n in range(0, 64):
            h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
            rbg.set_option("fail_alloc_after", n)
            try:
                h.plan_sell()
                failed = False
            except rbg.RbgError as ex:
                failed = True
                assert ex.code in (rbg._lib.RBG_ENOMEM, rbg._lib.RBG_EHIP), ex
            consumed = rbg.get_option("fail_alloc_after") < 0
            rbg.set_option("fail_alloc_after", -1)
            outcomes.append((failed, consumed, h.has_sell(64)))
            check(h)
            h.destroy()
            if not consumed:
                break
        assert not outcomes[-1][0] and outcomes[-1][2] and len(outcomes) > 8  # the walk reached the end of the planner
        assert any(f for f, _, _ in outcomes) and any((not f) and c and p for f, c, p in outcomes)  # hard failures AND optional arrays
        # the external attach + factors + lazy backward scratch
        for n in range(0, 24):
            h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
            rbg.set_option("fail_alloc_after", n)
            try:
                sell.attach(h, W=32)
            except rbg.RbgError:
                pass
            rbg.set_option("sell_rowmajor", 0)  # (the backward then needs the per-handle slab scratch)
            try:
                check(h)
            finally:
                rbg.set_option("sell_rowmajor", 1)
            consumed = rbg.get_option("fail_alloc_after") < 0
            rbg.set_option("fail_alloc_after", -1)
            check(h)
            h.destroy()
            if not consumed:
                break
        # the view
        h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
        h.plan_sell()
        for n in range(0, 6):
            rbg.set_option("fail_alloc_after", n)
            v = h.reweighted(h.values())
            consumed = rbg.get_option("fail_alloc_after") < 0
            rbg.set_option("fail_alloc_after", -1)
            v.refresh_values()
            close(rbg.ops.spmm_raw(v, x), C.spmm(rowptr, col, val, x.cpu().numpy()))
            v.destroy()
            if not consumed:
                break
        torch.cuda.synchronize()
    finally:
        rbg.set_option("fail_alloc_after", -1)
        rbg.set_option("sell_auto", 1)
        rbg.set_option("sell_rowmajor", 1)


# ---- the north_star's named scale -------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["amazon-book", "g-1.3m"])
def test_parity_at_scale_through_the_plan(rbg, cuda, name):
    """Single-GPU Amazon-Book shape (52 644 / 91 600 / 2 984 108) and the north_star's "1.3 M nodes" shape (550 000 / 750 000 /
    18 850 000), d = 64, K = 3, through the column-slab plan (both launch forms: one launch per layer, one per row class):
    every layer on sampled rows, the heaviest hubs and the PAD rows against float64 (sums over the rows' neighbour lists of
    the previous float64 layer — which needs THAT layer on every row: computed with scipy in float64 here), and the linearity /
    fixed-point properties on every row."""
    import scipy.sparse as sp
    uid, iid, nu, ni = rbg.synth.make(name)
    n, d = nu + ni, 64
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    assert h.sell_status() == "planned" and h.propagation_kernel_name(d).startswith("sell_spmm_kernel<32, 2, true")
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    got = h.export_csr()
    assert np.array_equal(got[0], rowptr) and np.array_equal(got[1], col) and np.array_equal(got[2], val)
    a = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(n, n))
    gen = torch.Generator().manual_seed(2020)
    uw, iw = O.xavier_uniform(nu, d, gen), O.xavier_uniform(ni, d, gen)
    e0 = np.concatenate([uw.numpy(), iw.numpy()]).astype(np.float64)
    acc, cur = e0.copy(), e0
    for _ in range(3):
        cur = a @ cur
        acc += cur
    ref = acc / 4
    deg = np.diff(rowptr)
    rows = np.unique(np.concatenate([np.random.default_rng(0).integers(0, n, 4000), np.argsort(deg)[-50:], [0, nu, n - 1]]))
    scale = float(np.abs(ref).max())
    mean, _ = rbg.ops.lightgcn_forward_raw(h, uw.to(cuda), iw.to(cuda), 3)
    err = float(np.abs(mean.cpu().numpy().astype(np.float64) - ref).max())
    assert err <= 1e-5 and err <= 1e-5 * scale, (err, scale)  # absolute AND normalized
    y = rbg.ops.spmm_raw(h, torch.from_numpy(e0.astype(np.float32)).to(cuda))
    r1 = a[rows] @ e0
    e1 = float(np.abs(y[torch.from_numpy(rows).to(cuda)].cpu().numpy() - r1).max())
    assert e1 <= 1e-5 * max(float(np.abs(r1).max()), 1e-30) * 4 and e1 <= 1e-5, e1
    # fixed point: A sqrt(deg) = sqrt(deg) on every non-isolated row, N(0, 1)-scale check of the gather
    root = torch.from_numpy(np.sqrt(deg.astype(np.float64)).astype(np.float32)).to(cuda)
    xr = root[:, None].expand(n, d).contiguous()
    yr = rbg.ops.spmm_raw(h, xr)
    rel = ((yr - xr).abs() / root[:, None].clamp(min=1.0)).max()
    assert float(rel) <= 5e-5 and float(yr[root == 0].abs().max()) == 0.0


# ---- scoring: the uniform-phase store stream (csrc/score.hip::score_uni_kernel) ------------------------------------------------

def test_score_uniform_phase_store_stream(rbg, cuda):
    """r04: workgroups of rbg_score_f32 (lightgcn.py:123-133) take their 128 users from ONE alignment class u = r mod q
    (q = 32 / gcd(n mod 32, 32)) and shift their item tiles to that class's line boundary, so every store is a whole aligned
    line straight from the accumulator.  Every q (n mod 32 = 0, 16, 8, 4, 2, odd), user counts that leave classes ragged,
    output bases at several 128-byte phases, d = 64 / 20 / 128 / 256, forced walk lengths: bit-identical to the shuffled store
    stream (option "score_uniform" = 0: the same products in the same order), float64-checked, and nothing outside the
    [B, n] block is written."""
    import ctypes
    lib, c_vp = rbg._lib.lib, ctypes.c_void_p
    gen = torch.Generator().manual_seed(78)
    pad = 96
    try:
        for tiles in (0, 1, 3):
            rbg.set_option("score_tiles", tiles)
            for d in (64, 20, 128, 256):
                for n in (1024, 1040, 1000, 996, 994, 993, 65, 33):
                    for b in (2048, 2051, 4096 + 7, 70 * 32):
                        if (tiles or d != 64) and (n, b) not in ((994, 2051), (993, 70 * 32), (1040, 2048)):
                            continue
                        u = torch.randn(b, d, generator=gen).to(cuda)
                        it = torch.randn(n, d, generator=gen).to(cuda)
                        for off in (0, 1, 21):
                            res = []
                            for uni in (1, 0):
                                rbg.set_option("score_uniform", uni)
                                buf = torch.full((pad + off + b * n + pad,), -777.0, device=cuda)
                                out = buf[pad + off: pad + off + b * n]
                                rc = lib.rbg_score_f32(c_vp(u.data_ptr()), d, c_vp(it.data_ptr()), d, c_vp(out.data_ptr()), b, n, d,
                                                       c_vp(torch.cuda.current_stream().cuda_stream))
                                assert rc == 0, lib.rbg_last_error()
                                assert bool((buf[: pad + off] == -777.0).all()) and bool((buf[pad + off + b * n:] == -777.0).all()), \
                                    (tiles, d, b, n, off, uni, "wrote outside the output")
                                res.append(out.view(b, n).clone())
                            assert torch.equal(res[0], res[1]), (tiles, d, b, n, off)
                        ref = u[:64].cpu().double() @ it.cpu().double().T
                        close(res[0][:64], ref, tol=2e-6 * max(1, d / 64))
    finally:
        rbg.set_option("score_tiles", 0)
        rbg.set_option("score_uniform", 1)


def test_attached_plan_with_wide_rows_is_validated_and_runs(rbg, cuda):
    """r06 header format: a wide row is U consecutive units at the front of its class, unit j of U.  rbg_graph_attach_sell (the
    specification's arrays) counts them, checks every unit's (j, U) against its row's first unit and the class's wide range, and
    the attached plan then runs like the native one; a plan with a wrong index / count / a wide unit behind a narrow one is refused."""
    import ctypes
    import sell_spec as sell
    hub = 5000
    nu, ni = 300, hub + 50
    u = np.concatenate([np.full(hub, 1), np.full(2500, 2), np.arange(3000) % (nu - 3) + 3]).astype(np.int64)
    i = np.concatenate([np.arange(hub) + 1, np.arange(2500) * 2 + 1, (np.arange(3000) * 7919) % (ni - 1) + 1]).astype(np.int64)
    key = np.unique(u * ni + i)
    u, i = key // ni, key % ni
    rbg.set_option("sell_auto", 0)
    try:
        h = rbg.GraphHandle.from_interactions(u, i, nu, ni, device=cuda)
    finally:
        rbg.set_option("sell_auto", 1)
    lib, vp = rbg._lib.lib, ctypes.c_void_p
    plan = sell.build_plan(*h.device_csr(), nu, ni, W=32)
    head = plan["head"]
    wide = ((head[:, 3] >> 16) & 1).bool()
    assert int(wide.sum()) >= 5 + 3          # user 1: ceil(5000 / 1024) = 5 units, user 2: 3

    def attach(hd):
        ub, nun = (ctypes.c_int32 * 2)(*plan["unit_base"]), (ctypes.c_int32 * 2)(*plan["n_units"])
        return lib.rbg_graph_attach_sell(h.ptr, 32, vp(plan["ent"].data_ptr()), plan["n_ent"], vp(hd.data_ptr()), ub, nun, vp(plan["orig"].data_ptr()))

    for field, delta in ((2, 1), (3, 1 << 17)):      # unit 1 of the first row: a wrong index j, a wrong count U
        bad = head.clone()
        bad[1, field] += delta
        assert attach(bad) == rbg._lib.RBG_EINVAL and b"unit" in lib.rbg_last_error()
    bad = head.clone()
    bad[int(wide.sum()), 3] |= (1 << 16) | (1 << 17)  # a "wide" unit behind the class's wide range
    assert attach(bad) == rbg._lib.RBG_EINVAL
    bad = head.clone()
    bad[0, 3] &= ~(1 << 16)                           # the first unit of a wide row loses its flag: the row's other units point at it
    assert attach(bad) == rbg._lib.RBG_EINVAL
    assert attach(head) == 0 and h.has_sell(64) and h.sell_status() == "attached"
    x = randn((nu + ni, 64), 3, cuda)
    rp, cl, vl = C.build_norm_csr(u, i, nu, ni)
    close(rbg.ops.spmm_raw(h, x), O.conv_csr_f64(x.cpu().numpy().astype(np.float64), rp, cl, vl))
    native = rbg.GraphHandle.from_interactions(u, i, nu, ni, device=cuda)
    assert torch.equal(rbg.ops.spmm_raw(h, x), rbg.ops.spmm_raw(native, x))  # the same plan, the same bits
