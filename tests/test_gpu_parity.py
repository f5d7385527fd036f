"""Parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs, against the committed golden vectors, and — at BASELINE.json's full sizes — through
size-independent properties.  Tolerance: 1e-5 fp32 (BASELINE.json north_star), applied as
|hip - oracle| <= 1e-5 * max(1, max|oracle|); graph construction is bit-exact.
Needs a real MI355X: run with `pytest -m gpu`.
"""
import os

import numpy as np
import pytest
import torch

from conftest import known_graphs
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


# ---- graph construction on the device -------------------------------------------------------

@pytest.mark.parametrize("builder", ["device", "host"])
def test_device_graph_is_bit_exact(rbg, cuda, golden, builder):
    """Both builders (HBM-side sort/scan/weights, and host C++ + upload) give the oracle's CSR bit for bit."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    flags = rbg._lib.GRAPH_BUILD_ON_HOST if builder == "host" else 0
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, flags=flags)
    assert h.is_device
    rowptr, col, val = h.export_csr()
    assert np.array_equal(rowptr, g["rowptr"]) and np.array_equal(col, g["col"]) and np.array_equal(val, g["val"])
    v = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, keep=g["sgl_keep"], flags=flags)
    vrp, vcol, vval = v.export_csr()
    assert np.array_equal(vrp, g["sgl_rowptr"]) and np.array_equal(vcol, g["sgl_col"]) and np.array_equal(vval, g["sgl_val"])
    hk = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, flags=flags | rbg._lib.GRAPH_KEEP_HOST)
    for a, b in zip(hk.export_csr(), (g["rowptr"], g["col"], g["val"])):
        assert np.array_equal(a, b)


def test_device_builder_random_graphs(rbg, cuda):
    """Duplicated interactions, masks, isolated nodes, empty graphs and bad ids through the device builder."""
    rng = np.random.default_rng(12)
    for trial in range(12):
        nu, ni = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        e = int(rng.integers(0, 400))
        uid, iid = rng.integers(0, nu, e), rng.integers(0, ni, e)
        keep = (rng.random(e) < 0.7).astype(np.uint8)
        for k in (None, keep):
            h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, keep=k)
            ref = C.build_norm_csr(uid, iid, nu, ni, keep=k)
            for a, b in zip(h.export_csr(), ref):
                assert np.array_equal(a, b)
            x = randn((nu + ni, 32), trial, cuda)
            close(rbg.ops.spmm_raw(h, x), C.spmm(ref[0], ref[1], ref[2], x.cpu().numpy()))
    h = rbg.GraphHandle.from_interactions([], [], 3, 4, device=cuda)
    assert h.nnz == 0 and h.n_rows == 7
    with pytest.raises(rbg.RbgError) as ei:
        rbg.GraphHandle.from_interactions([1, 9], [1, 1], 3, 3, device=cuda)
    assert ei.value.code == rbg._lib.RBG_EINVAL


# ---- SpMM -----------------------------------------------------------------------------------

@pytest.mark.parametrize("name", list(known_graphs()))
def test_spmm_known_answers(rbg, cuda, name):
    kg = known_graphs()[name]
    n = kg["n_users"] + kg["n_items"]
    h = rbg.GraphHandle.from_interactions(kg["uid"], kg["iid"], kg["n_users"], kg["n_items"], device=cuda)
    for d in (4, 64):
        x = randn((n, d), 3, cuda)
        y = rbg.ops.spmm_raw(h, x)
        close(y, kg["dense"] @ x.cpu().numpy().astype(np.float64))
        assert torch.all(y[0] == 0) and torch.all(y[kg["n_users"]] == 0)  # PAD rows are written as zeros


@pytest.mark.parametrize("d", [64, 32, 128, 256, 8, 100, 1, 67])
def test_spmm_vs_oracle(rbg, cuda, golden, d):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    x = randn((nu + ni, d), 11 + d, cuda)
    x_poison = torch.full_like(x, float("nan"))
    y = rbg.ops.spmm_raw(h, x, out=x_poison)  # every output element must be overwritten
    ref = C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], x.cpu().numpy())
    close(y, ref)
    y2 = rbg.ops.spmm_raw(h, x, out=y.clone(), accumulate=True)
    close(y2, 2 * ref)
    assert torch.equal(rbg.ops.spmm_raw(h, x), y)  # bit-stable run to run (no float atomics)


@pytest.mark.parametrize("d", [64, 128, 100, 8])
def test_spmm_add_vs_oracle(rbg, cuda, golden, d):
    """r06, rbg_spmm_add_f32: Y = Z + Â·X in one launch (the column-slab kernel's mean epilogue with one addend and no division; the
    binned kernel at the widths without a plan, and with option "sell" off) against the C oracle's product + Z; X and Z may be the
    same table (a Horner chain's first step on the gradient itself); every output element is written."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    x, z = randn((nu + ni, d), 3 + d, cuda), randn((nu + ni, d), 5 + d, cuda)
    ref = C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], x.cpu().numpy())
    for sell in (1, 0):
        rbg.set_option("sell", sell)
        try:
            y = rbg.ops.spmm_add_raw(h, x, z, out=torch.full_like(x, float("nan")))
            close(y, ref + z.cpu().numpy())
            close(rbg.ops.spmm_add_raw(h, x, x), ref + x.cpu().numpy())
            assert torch.equal(rbg.ops.spmm_add_raw(h, x, z), y)  # bit-stable
        finally:
            rbg.set_option("sell", 1)
    with pytest.raises(rbg.RbgError):
        rbg.ops.spmm_add_raw(h, x, z, out=x)


def test_spmm_golden_layers(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    x = torch.from_numpy(g["e0_d16"]).to(cuda)
    for k in (1, 2, 3):
        x = rbg.ops.spmm_raw(h, x)
        close(x, g[f"e{k}_d16"])


@pytest.mark.parametrize("tuning", [(0, 0, 64), (2, 4, 64), (4, 16, 128), (1000, 1000, 4096)])
@pytest.mark.parametrize("xcd_split", [4, 0, 5, 1])
def test_spmm_every_bin_and_split_rows(rbg, cuda, tuning, xcd_split):
    """Force rows through each mapping: lane-group, wavefront, workgroup, and split workgroup rows whose
    partial sums are combined by the last-arriving segment."""
    old = rbg.get_tuning()
    try:
        rbg.set_tuning(*tuning)
        rbg.set_option("xcd_split", xcd_split)
        rng = np.random.default_rng(5)
        nu, ni = 40, 3000
        # user 1 is a hub with 2500 items (degree >> seg_len), user 2 has 700, the rest are short
        uid = np.concatenate([np.full(2500, 1), np.full(700, 2), rng.integers(3, nu, 4000)])
        iid = np.concatenate([rng.permutation(ni - 1)[:2500] + 1, rng.permutation(ni - 1)[:700] + 1,
                              rng.integers(1, ni, 4000)])
        h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
        bins = h.bins(64)
        if tuning[2] == 64 and tuning[1] < 1000:
            assert bins["n_split_rows"] >= 2
        rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
        for d in (64, 128, 32, 256):
            rbg.set_option("spmm_unroll", 8 if d == 128 else 4)
            rbg.set_option("nt_store", 0 if d == 32 else 1)
            x = randn((nu + ni, d), d, cuda)
            ref = C.spmm(rowptr, col, val, x.cpu().numpy())
            for _ in range(3):  # the split-row counters must be back at zero after every launch
                close(rbg.ops.spmm_raw(h, x), ref)
        rbg.set_option("spmm_unroll", 4)
        rbg.set_option("nt_store", 1)
        hn = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, flags=rbg._lib.GRAPH_NATURAL_ORDER)
        x = randn((nu + ni, 64), 1, cuda)
        close(rbg.ops.spmm_raw(hn, x), C.spmm(rowptr, col, val, x.cpu().numpy()))
    finally:
        rbg.set_tuning(**old)
        rbg.set_option("xcd_split", 4)


def test_hub_rows_at_scale(rbg, cuda):
    """Config-#5-like hubs: rows of degree 250 000 / 100 000 (dozens of split segments each, default tuning),
    next to 300 k short rows; d = 64 and 128; repeated launches must leave the arrival counters at zero."""
    rng = np.random.default_rng(8)
    nu, ni = 2000, 300_001
    hub1 = rng.permutation(ni - 1)[:250_000] + 1
    hub2 = rng.permutation(ni - 1)[:100_000] + 1
    uid = np.concatenate([np.full(len(hub1), 1), np.full(len(hub2), 2), rng.integers(3, nu, 200_000)])
    iid = np.concatenate([hub1, hub2, rng.integers(1, ni, 200_000)])
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    bins = h.bins(64)
    assert bins["n_split_rows"] >= 2 and bins["n_block_tasks"] >= 80
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    got = h.export_csr()
    assert np.array_equal(got[0], rowptr) and np.array_equal(got[1], col) and np.array_equal(got[2], val)
    # A 250 000-term fp32 sum accumulated strictly left to right (the reference's CPU loop, = the C oracle) is itself
    # ~1e-4 away from the exact value, so for these rows the bar is float64 truth: the HIP result (segmented sums)
    # must be within 1e-5 of it, and no further from the fp32 oracle than the oracle is from the truth.
    for d in (64, 128):
        x = randn((nu + ni, d), d, cuda)
        truth = O.conv_csr_f64(x.cpu().numpy(), rowptr, col, val)
        ref32 = C.spmm(rowptr, col, val, x.cpu().numpy())
        y0 = rbg.ops.spmm_raw(h, x)
        err_hip = close(y0, truth)
        err_ref = float(np.abs(ref32 - truth).max())
        assert err_hip <= err_ref + 1e-6
        short = np.diff(rowptr) <= 4096  # every ordinary row still matches the fp32 oracle itself
        close(y0[torch.from_numpy(short).to(cuda)], ref32[short])
        for _ in range(3):
            assert torch.equal(rbg.ops.spmm_raw(h, x), y0)  # bit-stable, counters self-clean
    uw, iw = randn((nu, 64), 1, cuda), randn((ni, 64), 2, cuda)
    mean, _ = rbg.ops.lightgcn_forward_raw(h, uw, iw, 3)
    e64 = np.concatenate([uw.cpu().numpy(), iw.cpu().numpy()]).astype(np.float64)
    acc, cur = e64.copy(), e64
    for _ in range(3):
        cur = O.conv_csr_f64(cur, rowptr, col, val)
        acc = acc + cur
    close(mean, acc / 4.0)


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_column_half_mode(rbg, cuda, d):
    """"col_split": even / odd XCDs own the lower / upper half of the columns (auto at d = 128).  Same result as the
    full-width mode up to the summation grouping (both against float64), through every epilogue (plain, accumulate,
    fused layer mean, backward chain), including rows split into segments."""
    nu, ni, e = 3001, 2201, 60_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=5)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    assert h.bins(d)["n_split_rows"] == 0
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    n = nu + ni
    x = randn((n, d), 3, cuda)
    truth = O.conv_csr_f64(x.cpu().numpy().astype(np.float64), rowptr, col, val)
    res = {}
    try:
        for mode in (0, 1):
            rbg.set_option("col_split", mode)
            y = rbg.ops.spmm_raw(h, x)
            close(y, truth)
            acc = torch.ones(n, d, device=cuda)
            rbg.ops.spmm_raw(h, x, out=acc, accumulate=True)
            close(acc, truth + 1.0)
            uw, iw = x[:nu].contiguous(), x[nu:].contiguous()
            mean, layers = rbg.ops.lightgcn_forward_raw(h, uw, iw, 3, keep_layers=True)
            xg = x.clone().requires_grad_(True)
            out = rbg.ops._LightGCNForward.apply(xg[:nu], xg[nu:], 3, h)
            out.backward(torch.ones_like(out))
            res[mode] = (y, mean, layers.clone(), xg.grad.clone())
        for a, b in zip(res[0], res[1]):
            close(a, b, tol=2e-6)
        # hub rows longer than seg_len -> split rows: each column half collects its own partial sums and arrivals
        hub_u = np.concatenate([uid, np.full(5000, 1, dtype=np.int64), np.full(3000, 7, dtype=np.int64)])
        hub_i = np.concatenate([iid, (np.arange(5000) % (ni - 1) + 1).astype(np.int64), (np.arange(3000) * 3 % (ni - 1) + 1).astype(np.int64)])
        rbg.set_tuning(64, 256, 1024)
        hh = rbg.GraphHandle.from_interactions(hub_u, hub_i, nu, ni, device=cuda)
        assert hh.bins(d)["n_split_rows"] > 0
        hr, hc, hv = C.build_norm_csr(hub_u, hub_i, nu, ni)
        hub_truth = O.conv_csr_f64(x.cpu().numpy().astype(np.float64), hr, hc, hv)
        for mode in (1, 0, 1):
            rbg.set_option("col_split", mode)
            for _ in range(3):  # counters are self-cleaning: repeated launches stay correct
                close(rbg.ops.spmm_raw(hh, x), hub_truth)
    finally:
        rbg.set_option("col_split", -1)
        rbg.set_tuning(64, 256, 4096)


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("k_layers", [1, 2, 3])
def test_lightgcn_forward_slab_option(rbg, cuda, d, k_layers):
    """Option "slab" (r03 experiment, off by default — it measured slower): rbg_lightgcn_forward_f32 keeps the
    layers as two column slabs and runs the column-half kernel over contiguous half rows; E0 converted once, the mean written
    row-major by the last epilogue.  Same values as the default path (both against float64), split rows included, and the
    autograd path (which ignores `layers`) unchanged."""
    nu, ni, e = 3001, 2201, 60_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=5)
    hub_u = np.concatenate([uid, np.full(5000, 1, dtype=np.int64)])
    hub_i = np.concatenate([iid, (np.arange(5000) % (ni - 1) + 1).astype(np.int64)])
    rbg.set_tuning(64, 256, 1024)
    try:
        for (u, i) in ((uid, iid), (hub_u, hub_i)):
            h = rbg.GraphHandle.from_interactions(u, i, nu, ni, device=cuda)
            rowptr, col, val = C.build_norm_csr(u, i, nu, ni)
            x = randn((nu + ni, d), 11, cuda)
            cur = x.cpu().numpy().astype(np.float64)
            acc = cur.copy()
            for _ in range(k_layers):
                cur = O.conv_csr_f64(cur, rowptr, col, val)
                acc = acc + cur
            truth = acc / (k_layers + 1)
            got = {}
            for slab in (0, 1):
                rbg.set_option("slab", slab)
                for _ in range(2):  # split-row counters are self-cleaning
                    mean, _ = rbg.ops.lightgcn_forward_raw(h, x[:nu].contiguous(), x[nu:].contiguous(), k_layers)
                close(mean, truth)
                got[slab] = mean.clone()
            close(got[0], got[1], tol=2e-6)
    finally:
        rbg.set_option("slab", 0)
        rbg.set_tuning(64, 256, 4096)


@pytest.mark.parametrize("d", [64, 128])
def test_lightgcn_forward_column_slab_path(rbg, cuda, d):
    """csrc/sell.hip behind rbg_lightgcn_forward_f32 (lightgcn.py:70-81): E0 re-laid out as two column slabs in the plan's
    numbering, K slab layers over the SELL plan, the mean written row-major in the reference's numbering.  A graph with hub
    rows (cut over the lane-groups of a wave, and over the four waves of a workgroup), empty rows and the PAD rows; K = 1..3
    against float64; equal to the binned path within rounding; bit-stable; the option, detach and the kernel-name query."""
    nu, ni, e = 3001, 2201, 60_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=5)
    # hubs: user 1 with 1 500 items, user 7 with 700, item 3 with 900 users (d = 64: wide above 512 entries; d = 128: above 256)
    hub_u = np.concatenate([uid, np.full(1500, 1), np.full(700, 7), (np.arange(900) * 3 % (nu - 1) + 1)]).astype(np.int64)
    hub_i = np.concatenate([iid, (np.arange(1500) % (ni - 1) + 1), (np.arange(700) * 3 % (ni - 1) + 1), np.full(900, 3)]).astype(np.int64)
    key = np.unique(hub_u * ni + hub_i)
    hub_u, hub_i = key // ni, key % ni
    h = rbg.GraphHandle.from_interactions(hub_u, hub_i, nu, ni, device=cuda)
    rowptr, col, val = C.build_norm_csr(hub_u, hub_i, nu, ni)
    assert np.diff(rowptr).max() > 1024 and np.diff(rowptr).min() == 0
    x = randn((nu + ni, d), 11, cuda)
    uw, iw = x[:nu].contiguous(), x[nu:].contiguous()
    truth, cur = [], x.cpu().numpy().astype(np.float64)
    acc = cur.copy()
    for _ in range(3):
        cur = O.conv_csr_f64(cur, rowptr, col, val)
        acc = acc + cur
        truth.append(acc / (len(truth) + 2))
    try:
        assert h.has_sell(d) and h.sell_status() == "planned"  # rbg_graph_create planned the handle (option "sell_auto")
        rbg.set_option("sell", 0)
        binned = [rbg.ops.lightgcn_forward_raw(h, uw, iw, k)[0].clone() for k in (1, 2, 3)]
        assert "binned" in h.propagation_kernel_name(d)
        rbg.set_option("sell", 1)
        h.detach_sell()
        assert not h.has_sell(d)
        info = h.attach_sell(d)
        assert h.has_sell(64) and h.has_sell(128) and h.has_sell(32) and info["padding"] < 1.2  # (W = 32 serves the three widths)
        # val_ij = r_i r_j (the symmetric normalisation): the chain's launches after the first read 4-byte entries
        assert info["factored"] and h.propagation_kernel_name(d) == f"sell_spmm_kernel<32, {d // 32}, true>"
        # a caller that reads the layers gets them row-major: the same kernel gathering / writing the reference's layout
        assert h.propagation_kernel_name(d, scratch_layers=False) == f"sell_spmm_kernel<32, {d // 32}, false>" == h.spmm_kernel_name(d)
        rbg.set_option("sell_rowmajor", 0)
        # without the row-major entries the per-layer outputs come from the binned kernel; the plain layer converts X to slabs
        assert "binned" in h.propagation_kernel_name(d, scratch_layers=False)
        assert h.spmm_kernel_name(d) == f"sell_spmm_kernel<32, {d // 32}, false>"
        rbg.set_option("sell_rowmajor", 1)
        for k in (1, 2, 3):
            out = torch.full((nu + ni, d), 7.0, device=cuda)
            rbg.ops.lightgcn_forward_raw(h, uw, iw, k, out=out)
            close(out, truth[k - 1])
            close(out, binned[k - 1], tol=2e-6)
            again = rbg.ops.lightgcn_forward_raw(h, uw, iw, k)[0]
            assert torch.equal(out, again)  # the plan fixes the summation order
            assert float(out[0].abs().max()) == float((x[0].abs() / (k + 1)).max()) and float(out[nu].abs().max()) == float((x[nu].abs() / (k + 1)).max())
        # keep_layers (NCL reads every layer): every layer row-major through the plan; equal to the slab chain bit for bit
        x64 = x.cpu().numpy().astype(np.float64)
        for k in (1, 2, 3):
            mean, layers = rbg.ops.lightgcn_forward_raw(h, uw, iw, k, keep_layers=True)
            close(mean, truth[k - 1])
            close(mean, rbg.ops.lightgcn_forward_raw(h, uw, iw, k)[0], tol=2e-6)  # (the slab chain runs factored)
            rbg.set_option("sell_factored", 0)
            assert torch.equal(mean, rbg.ops.lightgcn_forward_raw(h, uw, iw, k)[0])
            assert h.propagation_kernel_name(d) == f"sell_spmm_kernel<32, {d // 32}, false>"
            rbg.set_option("sell_factored", 1)
            cur = x64
            for j in range(k):
                cur = O.conv_csr_f64(cur, rowptr, col, val)
                close(layers[j], cur)
        # the plain layer (rbg_spmm_f32: NGCF, SimGCL) over the plan, Y = A X and Y += A X
        y = rbg.ops.spmm_raw(h, x)
        close(y, O.conv_csr_f64(x64, rowptr, col, val))
        y2 = x.clone()
        rbg.ops.spmm_raw(h, x, out=y2, accumulate=True)
        close(y2, x64 + O.conv_csr_f64(x64, rowptr, col, val))
        rbg.set_option("sell_rowmajor", 0)
        assert torch.equal(rbg.ops.spmm_raw(h, x), y)  # X through the slab scratch: the same sums in the same order
        rbg.set_option("sell_rowmajor", 1)
        # autograd: forward over the slabs, backward = the Horner chain of the binned kernel
        xg = x.clone().requires_grad_(True)
        out = rbg.ops.lightgcn_forward(h, xg[:nu], xg[nu:], 3)
        out.backward(torch.ones_like(out))
        ones = np.ones((nu + ni, d))
        g_acc, g_cur = ones.copy(), ones.copy()
        for _ in range(3):
            g_cur = O.conv_csr_f64(g_cur, rowptr, col, val)
            g_acc = g_acc + g_cur
        close(xg.grad, g_acc / 4.0)
        # E0 / the incoming gradient gathered row-major where they lie (option "sell_rowmajor", default) == converted to slabs
        # first: the same sums in the same order, bit for bit — forward and backward, K = 1..3, a non-uniform gradient
        # ... with the factored chain (compact entries, scaled slabs) and without
        gout = randn((nu + ni, d), 23, cuda)
        res = {}
        for fac in (1, 0):
            rbg.set_option("sell_factored", fac)
            for rm in (1, 0):
                rbg.set_option("sell_rowmajor", rm)
                for k in (1, 2, 3):
                    xg = x.clone().requires_grad_(True)
                    out = rbg.ops.lightgcn_forward(h, xg[:nu], xg[nu:], k)
                    out.backward(gout)
                    res[fac, rm, k] = (out.detach().clone(), xg.grad.clone())
        rbg.set_option("sell_rowmajor", 1)
        rbg.set_option("sell_factored", 1)
        gt = gout.cpu().numpy().astype(np.float64)
        for k in (1, 2, 3):
            for fac in (1, 0):
                assert torch.equal(res[fac, 1, k][0], res[fac, 0, k][0]) and torch.equal(res[fac, 1, k][1], res[fac, 0, k][1])
            close(res[1, 1, k][0], res[0, 1, k][0], tol=2e-6)
            close(res[1, 1, k][1], res[0, 1, k][1], tol=2e-6)
            res[1, k] = res[1, 1, k]
            close(res[1, k][0], truth[k - 1])
            g_acc, g_cur = gt.copy(), gt.copy()
            for _ in range(k):
                g_cur = O.conv_csr_f64(g_cur, rowptr, col, val)
                g_acc = g_acc + g_cur
            close(res[1, k][1], g_acc / (k + 1))
        rbg.set_option("sell", 0)
        assert "binned" in h.propagation_kernel_name(d)
        close(rbg.ops.lightgcn_forward_raw(h, uw, iw, 3)[0], binned[2], tol=0)
        rbg.set_option("sell", 1)
        if d == 128:  # the two-slab form of d = 128 (256-byte slab rows): a plan of W = 64
            h.attach_sell(128, W=64)
            assert h.has_sell(128) and not h.has_sell(64) and h.propagation_kernel_name(128) == "sell_spmm_kernel<64, 2, true>"
            close(rbg.ops.lightgcn_forward_raw(h, uw, iw, 3)[0], truth[2])
            close(rbg.ops.spmm_raw(h, x), O.conv_csr_f64(x64, rowptr, col, val))
        h.detach_sell()
        assert not h.has_sell(d)
        close(rbg.ops.lightgcn_forward_raw(h, uw, iw, 3)[0], truth[2])
    finally:
        rbg.set_option("sell", 1)
        rbg.set_option("sell_rowmajor", 1)
        rbg.set_option("sell_factored", 1)


def test_sell_plan_is_range_checked_and_auto_attached(rbg, cuda, golden):
    """rbg_graph_attach_sell validates on the device every index the kernel would dereference; ops.lightgcn_forward attaches
    a plan on the first propagation of an eligible handle (option "sell"), never to a re-weighted view or a host graph."""
    import ctypes
    import sell_spec as sell
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    rbg.set_option("sell_auto", 0)  # (this handle is planned from outside: rbg_graph_attach_sell, the specification's arrays)
    try:
        h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    finally:
        rbg.set_option("sell_auto", 1)
    assert not h.has_sell(64) and "disabled" in h.sell_status()
    lib, vp = rbg._lib.lib, ctypes.c_void_p
    plan = sell.build_plan(*h.device_csr(), nu, ni, W=32)

    def attach(ent=None, head=None, orig=None, n_ent=None):
        ent, head, orig = (plan[k] if t is None else t for k, t in (("ent", ent), ("head", head), ("orig", orig)))
        ub, nun = (ctypes.c_int32 * 2)(*plan["unit_base"]), (ctypes.c_int32 * 2)(*plan["n_units"])
        return lib.rbg_graph_attach_sell(h.ptr, 32, vp(ent.data_ptr()), plan["n_ent"] if n_ent is None else n_ent, vp(head.data_ptr()), ub, nun,
                                         vp(orig.data_ptr()))

    bad_ent = plan["ent"].clone()
    bad_ent[5, 0] = 32 * 4 * (max(nu, ni) + 3)   # a column past the table that is not the padding marker
    assert attach(ent=bad_ent) == rbg._lib.RBG_EINVAL and b"column offset" in lib.rbg_last_error()
    bad_head = plan["head"].clone()
    bad_head[3, 0] = plan["n_ent"]               # a unit whose entries run past the array
    assert attach(head=bad_head) == rbg._lib.RBG_EINVAL and b"unit" in lib.rbg_last_error()
    bad_head = plan["head"].clone()
    bad_head[0, 1] = nu                          # first row beyond the class
    assert attach(head=bad_head) == rbg._lib.RBG_EINVAL
    bad_orig = plan["orig"].clone()
    bad_orig[0] = nu + ni                        # a node id out of range
    assert attach(orig=bad_orig) == rbg._lib.RBG_EINVAL and b"orig" in lib.rbg_last_error()
    assert not h.has_sell(64)
    assert attach() == 0 and h.has_sell(64)
    host = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni)
    assert lib.rbg_graph_attach_sell(host.ptr, 32, None, 0, None, None, None, None) == rbg._lib.RBG_ENODEV
    # a handle created with planning off gets its plan on the first propagation (ops._auto_sell -> rbg_graph_plan_sell);
    # created normally it carries one from rbg_graph_create on
    uw, iw = randn((nu, 64), 1, cuda), randn((ni, 64), 2, cuda)
    rbg.set_option("sell_auto", 0)
    try:
        h3 = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    finally:
        rbg.set_option("sell_auto", 1)
    assert not h3.has_sell(64)
    rbg.ops.lightgcn_forward(h3, uw, iw, 2)
    assert h3.has_sell(64) and h3.sell_status() == "planned"
    h2 = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    assert h2.has_sell(64) and h2.sell_status() == "planned"
    # a re-weighted view borrows the plan but runs it only once its values have been refreshed (the plan holds a copy)
    vals = h2.values()
    view = h2.reweighted(vals)
    assert not view.has_sell(64) and not view.sell_eligible(64) and "binned" in view.spmm_kernel_name(64)
    close(rbg.ops.lightgcn_forward(view, uw, iw, 2), rbg.ops.lightgcn_forward(h2, uw, iw, 2), tol=2e-6)
    view.refresh_values()
    assert view.has_sell(64) and view.spmm_kernel_name(64).startswith("sell_spmm_kernel<32, 2, false")
    close(rbg.ops.lightgcn_forward(view, uw, iw, 2), rbg.ops.lightgcn_forward(h2, uw, iw, 2), tol=2e-6)


@pytest.mark.parametrize("n_parts", [2, 4, 8])
def test_community_partition_changes_only_the_launch_plan(rbg, cuda, n_parts):
    """rbg_graph_create_partitioned: pinning communities to XCDs must give bit-identical results."""
    # the column-half mode (auto at d = 128) groups a wave row's entries differently: bit-identity across launch plans
    # is a property within one summation scheme, so it is pinned off here (test_spmm_column_half_mode covers it)
    rbg.set_option("col_split", 0)
    try:
        _community_partition_body(rbg, cuda, n_parts)
    finally:
        rbg.set_option("col_split", -1)


def _community_partition_body(rbg, cuda, n_parts):
    nu, ni, e = 1201, 2401, 40_000
    uid, iid = rbg.synth.powerlaw_bipartite(nu, ni, e, seed=9, n_blocks=n_parts, p_in=0.9)
    part = rbg.sharded.striped_partition(nu, ni, n_parts)
    h0 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    h1 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, xcd_part=part)
    for a, b in zip(h0.export_csr(), h1.export_csr()):
        assert np.array_equal(a, b)
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    for d in (64, 128):
        x = randn((nu + ni, d), d, cuda)
        y0, y1 = rbg.ops.spmm_raw(h0, x), rbg.ops.spmm_raw(h1, x)
        assert torch.equal(y0, y1)
        close(y1, C.spmm(rowptr, col, val, x.cpu().numpy()))
    uw, iw = randn((nu, 64), 1, cuda), randn((ni, 64), 2, cuda)
    m0, _ = rbg.ops.lightgcn_forward_raw(h0, uw, iw, 3)
    m1, _ = rbg.ops.lightgcn_forward_raw(h1, uw, iw, 3)
    assert torch.equal(m0, m1)
    # communities rediscovered under scrambled ids ("auto") and passed through the model config
    if n_parts == 8:
        rs = np.random.default_rng(2)
        pu = np.concatenate([[0], rs.permutation(nu - 1) + 1])
        pi = np.concatenate([[0], rs.permutation(ni - 1) + 1])
        su, si = pu[uid], pi[iid]
        ds = rbg.InteractionDataset(su, si, nu, ni)
        torch.manual_seed(1)
        ma = rbg.LightGCN({"device": str(cuda), "enable_sparse": True, "xcd_partition": "auto", "n_layers": 2}, ds)
        torch.manual_seed(1)
        mb = rbg.LightGCN({"device": str(cuda), "enable_sparse": True, "n_layers": 2}, ds)
        with torch.no_grad():
            assert all(torch.equal(a, b) for a, b in zip(ma.forward(), mb.forward()))
    # an unbalanced / partly empty partition is still correct
    lop = np.zeros(nu + ni, dtype=np.int32)
    lop[: (nu + ni) // 10] = 1
    h2 = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, xcd_part=lop)
    assert torch.equal(rbg.ops.spmm_raw(h2, x), y0)


def test_spmm_empty_and_rectangular(rbg, cuda):
    h = rbg.GraphHandle.from_interactions([], [], 3, 4, device=cuda)
    y = rbg.ops.spmm_raw(h, randn((7, 64), 0, cuda))
    assert y.shape == (7, 64) and torch.all(y == 0)
    rowptr = np.array([0, 2, 2, 3], dtype=np.int64)
    col = np.array([0, 4, 1], dtype=np.int32)
    val = np.array([0.5, 2.0, -1.0], dtype=np.float32)
    hr = rbg.GraphHandle.from_csr(rowptr, col, val, 5, device=cuda)
    x = randn((5, 64), 2, cuda)
    y = rbg.ops.spmm_raw(hr, x)
    close(y, torch.stack([0.5 * x[0] + 2.0 * x[4], torch.zeros(64, device=cuda), -x[1]]))
    with pytest.raises(ValueError):
        rbg.ops.spmm_raw(hr, randn((4, 64), 2, cuda))
    with pytest.raises(RuntimeError):
        rbg.ops.spmm_raw(hr, torch.zeros(5, 64))  # CPU tensor: no CPU path


def test_dense_pair_branch_runs_the_same_kernel(rbg, cuda, golden):
    """LightGCNConv.forward(x, edge_index, edge_weight) with the reference's default (non-sparse) pair."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    ei, ew = rbg.norm_edges(g["uid"], g["iid"], nu, ni)
    conv = rbg.LightGCNConv(64)
    x = randn((nu + ni, 64), 9, cuda)
    y_pair = conv(x, ei.to(cuda), ew.to(cuda))
    y_ref = O.conv_dense(x.cpu(), ei, ew)
    close(y_pair, y_ref)
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    close(conv(x, h, None), y_ref)
    # the pair -> handle cache follows tensor identity: a different graph at a recycled device address, or an
    # in-place edit of the weights, must not be served by a stale handle
    ei_d, ew_d = ei.to(cuda), ew.to(cuda)
    close(conv(x, ei_d, ew_d), y_ref)
    ew_d.mul_(2.0)
    close(conv(x, ei_d, ew_d), 2 * y_ref)
    del ei_d, ew_d
    half = len(g["uid"]) // 2
    ei2, ew2 = rbg.norm_edges(g["uid"][:half], g["iid"][:half], nu, ni)
    pad = torch.zeros(2, ei.shape[1] - ei2.shape[1], dtype=torch.int64)
    ei2p = torch.cat([ei2, pad], dim=1).to(cuda)  # same shape (and very likely the same address) as the freed pair
    ew2p = torch.cat([ew2, torch.zeros(ei.shape[1] - ei2.shape[1])]).to(cuda)
    close(conv(x, ei2p, ew2p), O.conv_dense(x.cpu(), ei2, ew2))


# ---- LightGCN forward -----------------------------------------------------------------------

@pytest.mark.parametrize("k_layers", [0, 1, 2, 3, 4, 10])
def test_lightgcn_forward_vs_oracle(rbg, cuda, golden, k_layers):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    e0 = torch.from_numpy(g["e0_d64"]).to(cuda)
    mean, layers = rbg.ops.lightgcn_forward_raw(h, e0[:nu], e0[nu:], k_layers, keep_layers=True)
    ref, ref_layers = C.lightgcn_forward(g["rowptr"], g["col"].astype(np.int64), g["val"], g["e0_d64"][:nu],
                                         g["e0_d64"][nu:], k_layers, return_layers=True)
    close(mean, ref)
    for k in range(k_layers):
        close(layers[k], ref_layers[k + 1])
    if k_layers == 3:
        close(mean, g["mean_k3_d64"])
    # PAD rows: mean[pad] = E0[pad] / (K+1)
    close(mean[0], e0[0] / (k_layers + 1))
    close(mean[nu], e0[nu] / (k_layers + 1))


def test_lightgcn_forward_golden_d16(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    e0 = torch.from_numpy(g["e0_d16"]).to(cuda)
    for k in (1, 2, 3):
        mean, _ = rbg.ops.lightgcn_forward_raw(h, e0[:nu], e0[nu:], k)
        close(mean, g[f"mean_k{k}_d16"])
    err64 = np.abs(mean.cpu().numpy() - g["mean_k3_d16_f64"]).max()
    assert err64 <= TOL


def test_sgl_per_layer_graphs(rbg, cuda, golden):
    """SGL.forward(graph=[view]*K or K different views), sgl.py:136-139."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    rng = np.random.default_rng(3)
    masks = [(rng.random(len(g["uid"])) < 0.9).astype(np.uint8) for _ in range(3)]
    views = [rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, keep=m) for m in masks]
    csrs = [C.build_norm_csr(g["uid"], g["iid"], nu, ni, keep=m) for m in masks]
    e0 = torch.from_numpy(g["e0_d64"])
    convs = [lambda t, c=c: torch.from_numpy(C.spmm(c[0], c[1], c[2], t.numpy())) for c in csrs]
    u_ref, i_ref = O.lightgcn_forward(e0[:nu], e0[nu:], convs, 3)
    mean, _ = rbg.ops.lightgcn_forward_raw(views, e0[:nu].to(cuda), e0[nu:].to(cuda), 3)
    close(mean, torch.cat([u_ref, i_ref]))


@pytest.mark.parametrize("seed", range(6))
def test_random_graphs_widths_and_depths(rbg, cuda, seed):
    """Randomised sweep: small graphs with duplicated interactions, masked views and isolated nodes; widths that hit
    the binned kernels (32/64/128/256) and the generic one; depths 0..5; forward, kept layers and backward."""
    rng = np.random.default_rng(100 + seed)
    for _ in range(8):
        nu, ni = int(rng.integers(1, 50)), int(rng.integers(1, 80))
        e = int(rng.integers(0, 600))
        uid, iid = rng.integers(0, nu, e), rng.integers(0, ni, e)
        keep = (rng.random(e) < 0.8).astype(np.uint8) if rng.random() < 0.5 else None
        d = int(rng.choice([1, 3, 8, 32, 64, 100, 128, 256]))
        k_layers = int(rng.integers(0, 6))
        h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, keep=keep,
                                              flags=rbg._lib.GRAPH_BUILD_ON_HOST if rng.random() < 0.3 else 0)
        rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni, keep=keep)
        got = h.export_csr()
        assert all(np.array_equal(a, b) for a, b in zip(got, (rowptr, col, val)))
        uw = randn((nu, d), seed, cuda).requires_grad_(True)
        iw = randn((ni, d), seed + 1, cuda).requires_grad_(True)
        ref, ref_layers = C.lightgcn_forward(rowptr, col, val, uw.detach().cpu().numpy(), iw.detach().cpu().numpy(),
                                             k_layers, return_layers=True)
        mean, layers = rbg.ops.lightgcn_forward_raw(h, uw.detach(), iw.detach(), k_layers, keep_layers=True)
        close(mean, ref)
        for k in range(k_layers):
            close(layers[k], ref_layers[k + 1])
        # backward: d(sum(out * w))/dE0 = mean_k(A^k) w  (A symmetric) -> the same oracle applied to w
        w = randn((nu + ni, d), seed + 2, cuda)
        (rbg.lightgcn_forward(h, uw, iw, k_layers) * w).sum().backward()
        gref = C.lightgcn_forward(rowptr, col, val, w[:nu].cpu().numpy(), w[nu:].cpu().numpy(), k_layers)
        close(torch.cat([uw.grad, iw.grad]), gref)


# ---- models: the reference's interface ------------------------------------------------------

def make_model(rbg, cls, cuda, golden, **cfg):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    torch.manual_seed(4)
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 3}
    if cls.__name__ == "NGCF":  # the mirror defaults to NGCF.yaml's 0.1; value parity is defined at 0
        config["message_dropout"] = 0.0
    if cls.__name__ == "SGL":  # the reference's numpy sampling calls: the views are then reproducible from np.random.seed
        config["device_sampling"] = False
    config.update(cfg)
    return cls(config, ds), ds


@pytest.mark.parametrize("enable_sparse", [True, None, False])
def test_lightgcn_model_forward_and_full_sort(rbg, cuda, golden, enable_sparse):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=enable_sparse)
    assert model.use_sparse == bool(enable_sparse)
    if not enable_sparse:
        assert model.edge_index.shape == (2, 2 * len(g["uid"])) and model.edge_weight.is_cuda
    uw = model.user_embedding.weight.detach().cpu().numpy()
    iw = model.item_embedding.weight.detach().cpu().numpy()
    ref = C.lightgcn_forward(g["rowptr"], g["col"].astype(np.int64), g["val"], uw, iw, 3)
    with torch.no_grad():
        user_all, item_all = model.forward()
    assert user_all.shape == (nu, 64) and item_all.shape == (ni, 64)
    close(torch.cat([user_all, item_all]), ref)
    model.fused = False  # the reference's op-by-op structure over the same kernel
    with torch.no_grad():
        u2, i2 = model.forward()
    close(torch.cat([u2, i2]), ref)
    model.fused = True
    users = torch.tensor([1, 2, nu - 1], device=cuda)
    scores = model.full_sort_predict({"user_id": users})
    assert scores.shape == (3 * ni,)
    ref_scores = O.full_sort_predict(torch.from_numpy(ref[:nu]), torch.from_numpy(ref[nu:]), users.cpu())
    close(scores, ref_scores)
    # cache semantics (lightgcn.py:85-86,125-126)
    assert model.restore_user_e is not None
    cached = model.restore_user_e
    model.full_sort_predict({"user_id": users})
    assert model.restore_user_e is cached
    batch = {"user_id": torch.tensor([1, 2, 3], device=cuda), "item_id": torch.tensor([1, 2, 3], device=cuda),
             "neg_item_id": torch.tensor([4, 5, 6], device=cuda)}
    model.calculate_loss(batch)
    assert model.restore_user_e is None and model.restore_item_e is None


def test_lightgcn_training_gradients(rbg, cuda, golden):
    """calculate_loss + backward through the fused op == torch autograd through the oracle's dense branch."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=True)
    batch = {"user_id": torch.tensor([1, 2, 3, 7], device=cuda), "item_id": torch.tensor([1, 2, 3, 9], device=cuda),
             "neg_item_id": torch.tensor([4, 5, 6, 11], device=cuda)}
    loss = model.calculate_loss(batch)
    loss.backward()
    uw = model.user_embedding.weight.detach().cpu().clone().requires_grad_(True)
    iw = model.item_embedding.weight.detach().cpu().clone().requires_grad_(True)
    ei, ew = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=False)
    u_all, i_all = O.lightgcn_forward(uw, iw, lambda t: O.conv_dense(t, ei, ew), 3)
    u, p, q = batch["user_id"].cpu(), batch["item_id"].cpu(), batch["neg_item_id"].cpu()
    pos = (u_all[u] * i_all[p]).sum(1)
    neg = (u_all[u] * i_all[q]).sum(1)
    mf = -torch.log(1e-10 + torch.sigmoid(pos - neg)).mean()
    reg = (uw[u].norm() ** 2 + iw[p].norm() ** 2 + iw[q].norm() ** 2) / 4 / 2
    ref_loss = mf + 1e-5 * reg
    ref_loss.backward()
    close(loss.reshape(()), ref_loss.reshape(()))
    close(model.user_embedding.weight.grad, uw.grad, tol=1e-6)
    close(model.item_embedding.weight.grad, iw.grad, tol=1e-6)
    # unfused structure gives the same gradients
    model.zero_grad()
    model.fused = False
    model.calculate_loss(batch).backward()
    close(model.user_embedding.weight.grad, uw.grad, tol=1e-6)


@pytest.mark.parametrize("k_layers,per_layer", [(0, False), (1, False), (2, False), (3, False), (4, True), (3, True)])
def test_fused_backward_vs_torch_autograd(rbg, cuda, golden, k_layers, per_layer):
    """rbg_lightgcn_backward_f32 (Horner chain, fused '+ g' epilogue) against torch autograd through the oracle's
    dense-branch formulation; also SGL's per-layer graphs (sgl.py:136-139)."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    rng = np.random.default_rng(21)
    n_graphs = k_layers if per_layer else 1
    masks = [None] * max(n_graphs, 1) if not per_layer else [(rng.random(len(g["uid"])) < 0.9).astype(np.uint8) for _ in range(n_graphs)]
    handles = [rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, keep=m) for m in masks]
    pairs = []
    for m in masks:
        keep = np.ones(len(g["uid"]), dtype=bool) if m is None else m.astype(bool)
        pairs.append(O.get_norm_adj_mat(g["uid"][keep], g["iid"][keep], nu, ni, enable_sparse=False))
    uw = randn((nu, 64), 5, cuda).requires_grad_(True)
    iw = randn((ni, 64), 6, cuda).requires_grad_(True)
    w = randn((nu + ni, 64), 7, cuda)
    out = rbg.lightgcn_forward(handles if per_layer else handles[0], uw, iw, k_layers)
    (out * w).sum().backward()
    uw_r = uw.detach().cpu().clone().requires_grad_(True)
    iw_r = iw.detach().cpu().clone().requires_grad_(True)
    convs = [(lambda t, p=p: O.conv_dense(t, p[0], p[1])) for p in pairs]
    u_all, i_all = O.lightgcn_forward(uw_r, iw_r, convs if per_layer else convs[0], k_layers)
    (torch.cat([u_all, i_all]) * w.cpu()).sum().backward()
    close(uw.grad, uw_r.grad)
    close(iw.grad, iw_r.grad)


@pytest.mark.parametrize("require_pow", [True, False])
def test_fused_training_step_matches_torch_adam(rbg, cuda, golden, require_pow):
    """FusedBPRAdam.step == calculate_loss + backward + torch.optim.Adam.step, checked against the same three steps
    done by torch autograd on the CPU through the oracle's dense-branch propagation; both forms of EmbLoss
    (require_pow = True: LightGCN.yaml; False: RecBole's default, the 2-norm of each gathered block)."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=require_pow, reg_weight=1e-2)
    uw = model.user_embedding.weight.detach().cpu().clone().requires_grad_(True)
    iw = model.item_embedding.weight.detach().cpu().clone().requires_grad_(True)
    ref_opt = torch.optim.Adam([uw, iw], lr=1e-2)
    ei, ew = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=False)
    fused = rbg.FusedBPRAdam(model, lr=1e-2)
    rng = np.random.default_rng(4)
    for step in range(3):
        u = torch.from_numpy(rng.integers(1, nu, 64))   # duplicates inside the batch are likely and intended
        p = torch.from_numpy(rng.integers(1, ni, 64))
        q = torch.from_numpy(rng.integers(1, ni, 64))
        loss = fused.step({"user_id": u.to(cuda), "item_id": p.to(cuda), "neg_item_id": q.to(cuda)})
        ref_opt.zero_grad()
        u_all, i_all = O.lightgcn_forward(uw, iw, lambda t: O.conv_dense(t, ei, ew), 3)
        pos = (u_all[u] * i_all[p]).sum(1)
        neg = (u_all[u] * i_all[q]).sum(1)
        mf = -torch.log(1e-10 + torch.sigmoid(pos - neg)).mean()
        if require_pow:
            reg = (uw[u].norm() ** 2 + iw[p].norm() ** 2 + iw[q].norm() ** 2) / 64 / 2
        else:
            reg = (uw[u].norm() + iw[p].norm() + iw[q].norm()) / 64
        ref_loss = mf + 1e-2 * reg
        ref_loss.backward()
        ref_opt.step()
        close(loss.reshape(()), ref_loss.detach().reshape(()))
        close(model.user_embedding.weight, uw.detach(), tol=2e-5)
        close(model.item_embedding.weight, iw.detach(), tol=2e-5)
    # the torch-autograd path of the model and the fused step agree on the very same batch, too
    model2, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=require_pow, reg_weight=1e-2)
    model3, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=require_pow, reg_weight=1e-2)
    batch = {"user_id": u.to(cuda), "item_id": p.to(cuda), "neg_item_id": q.to(cuda)}
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-2)
    model2.calculate_loss(batch).backward()
    opt2.step()
    rbg.FusedBPRAdam(model3, lr=1e-2).step(batch)
    close(model3.user_embedding.weight, model2.user_embedding.weight, tol=2e-5)
    close(model3.item_embedding.weight, model2.item_embedding.weight, tol=2e-5)


def test_lean_lightgcn_step_equals_the_separate_calls(rbg, cuda, golden):
    """r06: FusedBPRAdam's lean form (rbg_lightgcn_step_head_f32 / _tail_f32: BPR + the regulariser's value + node occurrences in
    one launch, Adam + the regulariser's gradient + the clean-up in another) against its separate calls (rbg_bpr_grad_f32,
    rbg_emb_reg_grad_f32, rbg_adam_step_dev_f32) over six steps with repeated ids — incl. switching between the two forms on ONE
    stepper (the lean form's state: grad_mean all-zero, the occurrence tables alternating with the step's parity, the device step
    count) — and the running loss total the driver reads once per epoch."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    ma, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=True, reg_weight=1e-2)
    mb, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True, require_pow=True, reg_weight=1e-2)
    mb.load_state_dict(ma.state_dict())
    lean, sep = rbg.FusedBPRAdam(ma, lr=1e-2), rbg.FusedBPRAdam(mb, lr=1e-2)
    sep.lean = False
    rng = np.random.default_rng(7)
    total = 0.0
    for step in range(6):
        batch = {"user_id": torch.from_numpy(rng.integers(1, nu, 96)).to(cuda), "item_id": torch.from_numpy(rng.integers(1, min(ni, 40), 96)).to(cuda),
                 "neg_item_id": torch.from_numpy(rng.integers(1, ni, 96)).to(cuda)}
        lean.lean = step not in (2, 3)  # steps 2 and 3 in the separate form on the same stepper, then back
        la, lb = lean.step(batch), sep.step(batch)
        close(la.reshape(()), lb.reshape(()), tol=1e-6)
        total += float(la)
        close(ma.user_embedding.weight, mb.user_embedding.weight, tol=1e-6)
        close(ma.item_embedding.weight, mb.item_embedding.weight, tol=1e-6)
    assert abs(float(lean.loss_total) - total) <= 1e-5 * max(1.0, abs(total))
    assert int(lean.step_dev) == 6 and int(sep.step_dev) == 6
    assert float(lean.grad_mean.abs().max()) == 0.0 and int(lean.row_count[int(lean.step_dev) & 1].abs().max()) == 0
    rbg.set_option("deterministic", 1)  # the lean form adds repeated rows with float atomics: the stepper takes the separate calls
    try:
        lean.lean = True
        l1, l2 = lean.step(batch), sep.step(batch)
        close(l1.reshape(()), l2.reshape(()), tol=1e-6)
        close(ma.user_embedding.weight, mb.user_embedding.weight, tol=1e-6)
    finally:
        rbg.set_option("deterministic", 0)


def test_sgl_model_views(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    for aug in ("ED", "RW", "ND"):
        model, _ = make_model(rbg, rbg.SGL, cuda, golden, enable_sparse=True, type=aug, drop_ratio=0.1)
        np.random.seed(11)
        model.train()
        assert len(model.sub_graph1) == 3
        if aug == "RW":
            assert model.sub_graph1[0][0] is not model.sub_graph1[1][0]
        else:
            assert model.sub_graph1[0][0] is model.sub_graph1[1][0]
        # replay the reference's sampling with the same global numpy stream
        np.random.seed(11)
        n_views = 6 if aug == "RW" else 2
        refs = []
        for _ in range(n_views):
            if aug == "ND":
                du = np.random.choice(np.arange(nu), size=int(nu * 0.1), replace=False)
                di = np.random.choice(np.arange(ni), size=int(ni * 0.1), replace=False)
                keep = ~(np.isin(g["uid"], du) | np.isin(g["iid"], di))
            else:
                idx = np.random.choice(np.arange(len(g["uid"])), size=int(len(g["uid"]) * 0.9), replace=False)
                keep = np.zeros(len(g["uid"]), dtype=bool)
                keep[idx] = True
            refs.append(C.build_norm_csr(g["uid"], g["iid"], nu, ni, keep=keep.astype(np.uint8)))
        got = model.sub_graph1[0][0].export_csr()
        assert all(np.array_equal(a, b) for a, b in zip(got, refs[0]))
        with torch.no_grad():
            outs = model.propagate_views()
        uw = model.user_embedding.weight.detach().cpu()
        iw = model.item_embedding.weight.detach().cpu()
        sub1 = refs[:3] if aug == "RW" else [refs[0]] * 3
        convs = [lambda t, c=c: torch.from_numpy(C.spmm(c[0], c[1], c[2], t.numpy())) for c in sub1]
        u_ref, i_ref = O.lightgcn_forward(uw, iw, convs, 3)
        close(torch.cat(outs[1]), torch.cat([u_ref, i_ref]))
        s = model.full_sort_predict({"user_id": torch.tensor([1, 5], device=cuda)})
        assert s.shape == (2, ni)  # SGL returns the un-flattened matrix (sgl.py:240)
        # ... whose values are the FULL graph's propagation (sgl.py:236-237: self.forward() without a view) scored as :240
        full = lambda t: torch.from_numpy(C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], t.numpy()))  # noqa: E731
        uf, itf = O.lightgcn_forward(uw, iw, full, 3)
        close(s, O.full_sort_predict(uf, itf, [1, 5]).view(2, ni))
        close(model.predict({"user_id": torch.tensor([1, 5], device=cuda), "item_id": torch.tensor([2, 7], device=cuda)}),
              (uf[[1, 5]] * itf[[2, 7]]).sum(1))


def test_sgl_training_loss_and_gradients(rbg, cuda, golden):
    """SGL.calculate_loss (sgl.py:211-233) and its gradients against the same formulas on oracle-propagated
    embeddings with torch autograd on the CPU (views replayed from the model's own keep masks)."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.SGL, cuda, golden, enable_sparse=True, type="ED", ssl_tau=0.5, ssl_weight=0.05,
                          reg_weight=1e-4)
    np.random.seed(5)
    model.train()
    batch = {"user_id": torch.tensor([1, 2, 3, 9, 2], device=cuda), "item_id": torch.tensor([1, 4, 3, 7, 8], device=cuda),
             "neg_item_id": torch.tensor([5, 6, 2, 11, 30], device=cuda)}
    loss = model.calculate_loss(batch)
    loss.backward()
    uw = model.user_embedding.weight.detach().cpu().clone().requires_grad_(True)
    iw = model.item_embedding.weight.detach().cpu().clone().requires_grad_(True)

    def dense_conv(handle):
        rp, c, v = handle.export_csr()
        rows = torch.from_numpy(np.repeat(np.arange(len(rp) - 1), np.diff(rp)))
        ei = torch.stack([torch.from_numpy(c.astype(np.int64)), rows])  # source = column, target = row
        return lambda t: O.conv_dense(t, ei, torch.from_numpy(v))

    props = []
    for handle in (model.graph, model.sub_graph1[0][0], model.sub_graph2[0][0]):
        props.append(O.lightgcn_forward(uw, iw, dense_conv(handle), 3))
    (ua, ia), (u1, i1), (u2, i2) = props
    u, p, q = (batch[k].cpu() for k in ("user_id", "item_id", "neg_item_id"))
    nrm = torch.nn.functional.normalize
    bpr = -torch.nn.functional.logsigmoid((ua[u] * ia[p]).sum(1) - (ua[u] * ia[q]).sum(1)).sum()
    reg = (uw[u].norm() + iw[p].norm() + iw[q].norm()) / 5
    def nce(a, b, allb):
        a, b, allb = nrm(a, dim=1), nrm(b, dim=1), nrm(allb, dim=1)
        return -torch.log(torch.exp((a * b).sum(1) / 0.5) / torch.exp(a @ allb.T / 0.5).sum(1)).sum()
    ref = bpr + 1e-4 * reg + 0.05 * (nce(u1[u], u2[u], u2) + nce(i1[p], i2[p], i2))
    ref.backward()
    close(loss.reshape(()), ref.detach().reshape(()), tol=2e-5)
    close(model.user_embedding.weight.grad, uw.grad, tol=2e-5)
    close(model.item_embedding.weight.grad, iw.grad, tol=2e-5)


# ---- SimGCL / XSimGCL: noise-perturbed propagation (simgcl.py:24-38, xsimgcl.py:28-48) ---------------

@pytest.mark.parametrize("d", [64, 128, 20, 256])
def test_spmm_noise_epilogue(rbg, cuda, golden, d):
    """rbg_spmm_noise_f32: Y = AX + sign(AX) * normalize(noise) * eps against the torch expression in float64; zero rows
    (PAD) keep sign 0; the backward is the plain transposed product."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    n = nu + ni
    x = randn((n, d), 5, cuda)
    noise = torch.rand(n, d, generator=torch.Generator().manual_seed(6)).to(cuda)
    y = rbg.ops.spmm_noise_raw(h, x, noise, 0.1)
    ax = torch.from_numpy(O.conv_csr_f64(x.cpu().numpy().astype(np.float64), g["rowptr"], g["col"].astype(np.int64), g["val"]))
    ref = ax + torch.sign(ax) * torch.nn.functional.normalize(noise.cpu().double(), dim=-1) * 0.1
    # sign() flips where AX is within rounding of zero: compare where |AX| is clearly non-zero, and PAD rows exactly
    mask = ax.abs() > 1e-6
    assert float((y.cpu().double() - ref)[mask].abs().max()) <= 1e-5
    assert torch.all(y[0] == 0) and torch.all(y[nu] == 0)
    xg = x.clone().requires_grad_(True)
    up = randn((n, d), 7, cuda)
    (rbg.ops.spmm_noise(h, xg, noise, 0.1) * up).sum().backward()
    close(xg.grad, rbg.ops.spmm_raw(h, up))


@pytest.mark.parametrize("d", [64, 128, 20, 32])
def test_sign_noise_is_the_noise_epilogue_alone(rbg, cuda, golden, d):
    """rbg_sign_noise_f32 (r06): out = Y + sign(Y) * normalize(noise) * eps on a product that exists already == rbg_spmm_noise_f32 of the
    same product (SimGCL's three passes share A E_0), to the last bits of the row norm's summation order; exact against the float64
    expression; zero rows stay zero; in place."""
    from recbole_gnn_amd._lib import lib, check, c_vp
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    n = nu + ni
    x = randn((n, d), 5, cuda)
    noise = torch.rand(n, d, generator=torch.Generator().manual_seed(6)).to(cuda)
    fused = rbg.ops.spmm_noise_raw(h, x, noise, 0.1)
    y = rbg.ops.spmm_raw(h, x)
    out = torch.empty_like(y)
    st = c_vp(torch.cuda.current_stream(cuda).cuda_stream)
    check(lib.rbg_sign_noise_f32(c_vp(y.data_ptr()), c_vp(noise.data_ptr()), n, d, 0.1, c_vp(out.data_ptr()), st))
    ref = y.cpu().double() + torch.sign(y.cpu().double()) * torch.nn.functional.normalize(noise.cpu().double(), dim=-1) * 0.1
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-6
    assert float((out - fused).abs().max()) <= 1e-6  # (the same product bit for bit; the norms are summed in another order)
    assert torch.all(out[0] == 0) and torch.all(out[nu] == 0)
    check(lib.rbg_sign_noise_f32(c_vp(y.data_ptr()), c_vp(noise.data_ptr()), n, d, 0.1, c_vp(y.data_ptr()), st))
    assert torch.equal(y, out)
    assert lib.rbg_sign_noise_f32(c_vp(y.data_ptr()), c_vp(noise.data_ptr()), n, 129, 0.1, c_vp(y.data_ptr()), st) != 0  # d > 128


@pytest.mark.parametrize("name", ["SimGCL", "XSimGCL"])
def test_simgcl_models(rbg, cuda, golden, name):
    """Model mirrors: clean forward = mean of layers 1..K (no E0); perturbed forward reproduces the reference's expression
    on the same torch.rand_like draws; calculate_loss and its gradients equal the reference formulas (oracle) evaluated
    with torch autograd on the same noise."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    extra = {"layer_cl": 1, "require_pow": True} if name == "XSimGCL" else {}
    model, _ = make_model(rbg, getattr(rbg, name), cuda, golden, enable_sparse=True, n_layers=2, eps=0.1, temperature=0.2,
                          reg_weight=1e-4, **{"lambda": 0.5}, **extra)
    model.train()
    uw = model.user_embedding.weight.detach().cpu().double()
    iw = model.item_embedding.weight.detach().cpu().double()
    rp, col, val = g["rowptr"], g["col"].astype(np.int64), g["val"]
    conv64 = lambda t: torch.from_numpy(O.conv_csr_f64(t.detach().numpy(), rp, col, val))
    with torch.no_grad():
        got = model.forward()
    ref = O.simgcl_forward(uw, iw, conv64, 2)
    close(torch.cat(got), torch.cat(ref).float())
    # perturbed forward on the same random draws
    n, d = nu + ni, 64
    torch.manual_seed(123)
    with torch.no_grad():
        pert = model.forward(perturbed=True)
    torch.manual_seed(123)
    noises = [torch.rand(n, d, device=cuda).cpu().double() for _ in range(2)]
    refp = O.simgcl_forward(uw, iw, conv64, 2, noises=noises, eps=0.1, layer_cl=1 if name == "XSimGCL" else None)
    for a, b in zip(pert, refp):
        assert float((a.cpu().double() - b).abs().max()) <= 2e-5  # (sign flips need |AX| ~ 1e-7: not on these embeddings)
    # loss + gradients against the oracle formulas with autograd (dense operator), same noise stream
    batch = {"user_id": torch.tensor([1, 2, 3, 9, 2], device=cuda), "item_id": torch.tensor([1, 4, 3, 7, 8], device=cuda),
             "neg_item_id": torch.tensor([5, 6, 2, 11, 30], device=cuda)}
    torch.manual_seed(321)
    loss = model.calculate_loss(batch)
    loss = sum(loss) if isinstance(loss, tuple) else loss
    loss.backward()
    torch.manual_seed(321)
    n_draws = 2 if name == "XSimGCL" else 4
    draws = [torch.rand(n, d, device=cuda).cpu().double() for _ in range(n_draws)]
    uwr, iwr = uw.clone().requires_grad_(True), iw.clone().requires_grad_(True)
    dense = torch.zeros(n, n, dtype=torch.float64)
    rows = np.repeat(np.arange(n), np.diff(rp))
    dense.index_put_((torch.from_numpy(rows), torch.from_numpy(col)), torch.from_numpy(val.astype(np.float64)), accumulate=True)
    convd = lambda t: dense @ t
    u, p, q = (batch[k].cpu() for k in ("user_id", "item_id", "neg_item_id"))
    uu, pu = torch.unique(u), torch.unique(p)
    if name == "SimGCL":
        ua, ia = O.simgcl_forward(uwr, iwr, convd, 2)
        bpr = -torch.log(1e-10 + torch.sigmoid((ua[u] * ia[p]).sum(1) - (ua[u] * ia[q]).sum(1))).mean()
        reg = (uwr[u].norm() + iwr[p].norm() + iwr[q].norm()) / 5
        u1, i1 = O.simgcl_forward(uwr, iwr, convd, 2, noises=draws[0:2], eps=0.1)
        u2, i2 = O.simgcl_forward(uwr, iwr, convd, 2, noises=draws[2:4], eps=0.1)
        cl = O.simgcl_cl_loss(u1[uu], u2[uu], 0.2) + O.simgcl_cl_loss(i1[pu], i2[pu], 0.2)
        ref_loss = bpr + 1e-4 * reg + 0.5 * cl
    else:
        ua, ia, uc, ic = O.simgcl_forward(uwr, iwr, convd, 2, noises=draws, eps=0.1, layer_cl=1)
        bpr = -torch.log(1e-10 + torch.sigmoid((ua[u] * ia[p]).sum(1) - (ua[u] * ia[q]).sum(1))).mean()
        reg = (uwr[u].pow(2).sum() + iwr[p].pow(2).sum() + iwr[q].pow(2).sum()) / 2 / 5  # EmbLoss(require_pow=True)
        cl = O.simgcl_cl_loss(ua[uu], uc[uu], 0.2, "mean") + O.simgcl_cl_loss(ia[pu], ic[pu], 0.2, "mean")
        ref_loss = bpr + 1e-4 * reg + 0.5 * cl
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * max(1.0, abs(float(ref_loss.detach())))
    close(model.user_embedding.weight.grad, uwr.grad.float(), tol=2e-5)
    close(model.item_embedding.weight.grad, iwr.grad.float(), tol=2e-5)


# ---- InfoNCE denominator (sgl.py:195-198) ------------------------------------------------------

@pytest.mark.parametrize("b,n,d", [(1, 1, 4), (5, 7, 8), (33, 65, 16), (64, 1000, 64), (100, 333, 100), (257, 2049, 128),
                                   (40, 31, 3)])
def test_lse_rows_forward_backward(rbg, cuda, b, n, d):
    """rbg_lse_rows_f32 / _backward_f32 against float64 autograd of the reference expression
    log(sum(exp(q @ c.T / tau), dim=1)) — ragged B, n and d (tile edges), unit and non-unit rows."""
    gen = torch.Generator().manual_seed(b * 1000 + n + d)
    q = torch.nn.functional.normalize(torch.randn(b, d, generator=gen), dim=1)
    c = torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1)
    w = torch.randn(b, generator=gen)
    for scale, shift in ((5.0, 5.0), (5.0, 0.0), (2.0, 1.0)):
        q64, c64 = q.double().requires_grad_(True), c.double().requires_grad_(True)
        ref = O.lse_rows(q64, c64, scale)
        (ref * w.double()).sum().backward()
        qg, cg = q.to(cuda).requires_grad_(True), c.to(cuda).requires_grad_(True)
        out = rbg.ops.lse_rows(qg, cg, scale, shift)
        (out * w.to(cuda)).sum().backward()
        close(out, ref.detach().float(), tol=1e-5)
        close(qg.grad, q64.grad.float(), tol=1e-5)
        close(cg.grad, c64.grad.float(), tol=1e-5)


def test_lse_rows_batch_shape_and_determinism(rbg, cuda):
    """SGL's training shape (B = 2048 against the Gowalla-sized item table): value and both gradients against torch's
    own matmul/exp/sum on the GPU, and bit-identical across calls (no atomics in the reductions)."""
    gen = torch.Generator().manual_seed(3)
    q = torch.nn.functional.normalize(torch.randn(2048, 64, generator=gen), dim=1).to(cuda)
    c = torch.nn.functional.normalize(torch.randn(40982, 64, generator=gen), dim=1).to(cuda)
    tau = 0.2
    outs = []
    for _ in range(2):
        qg, cg = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
        out = rbg.ops.lse_rows(qg, cg, 1 / tau, 1 / tau)
        out.sum().backward()
        outs.append((out.detach(), qg.grad, cg.grad))
    for a, b2 in zip(*outs):
        assert torch.equal(a, b2)
    qr, cr = q.double().requires_grad_(True), c.double().requires_grad_(True)
    ref = torch.log(torch.exp(qr @ cr.T / tau).sum(1))
    ref.sum().backward()
    close(outs[0][0], ref.detach().float(), tol=1e-5)
    close(outs[0][1], qr.grad.float(), tol=1e-5)
    close(outs[0][2], cr.grad.float(), tol=1e-5)


def test_sgl_ssl_loss_matches_reference_formula(rbg, cuda):
    """SGL._info_nce x 2 == calc_ssl_loss (sgl.py:176-209) on random view embeddings."""
    gen = torch.Generator().manual_seed(11)
    nu, ni, d, tau = 300, 500, 32, 0.3
    u1, u2 = torch.randn(nu, d, generator=gen), torch.randn(nu, d, generator=gen)
    i1, i2 = torch.randn(ni, d, generator=gen), torch.randn(ni, d, generator=gen)
    users, pos = torch.randint(1, nu, (70,), generator=gen), torch.randint(1, ni, (70,), generator=gen)
    ref = O.calc_ssl_loss(users, pos, u1.double(), u2.double(), i1.double(), i2.double(), tau, 1.0)
    U1, U2, I1, I2 = (t.to(cuda) for t in (u1, u2, i1, i2))
    got = rbg.SGL._info_nce(U1[users.to(cuda)], U2[users.to(cuda)], U2, tau) + \
        rbg.SGL._info_nce(I1[pos.to(cuda)], I2[pos.to(cuda)], I2, tau)
    assert abs(float(got) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("n,d,b,tau", [(40, 8, 7, 0.5), (300, 32, 70, 0.2), (1000, 64, 257, 0.2), (500, 100, 64, 0.3), (65, 128, 33, 1.0)])
def test_info_nce_value_and_gradients(rbg, cuda, n, d, b, tau):
    """rbg_infonce_f32 == one half of calc_ssl_loss (sgl.py:191-199) in float64 autograd: loss, d/dT1, d/dT2; the batch
    repeats rows (scatter with duplicates) and contains the PAD row."""
    gen = torch.Generator().manual_seed(n + d + b)
    t1, t2 = torch.randn(n, d, generator=gen), torch.randn(n, d, generator=gen)
    idx = torch.randint(0, n, (b,), generator=gen)
    idx[: min(4, b)] = idx[0]  # duplicates
    w = 0.37
    a64, b64 = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
    nrm = torch.nn.functional.normalize
    u1, u2, allu = nrm(a64[idx], dim=1), nrm(b64[idx], dim=1), nrm(b64, dim=1)
    ref = -torch.sum(torch.log(torch.exp(torch.sum(u1 * u2, dim=1) / tau) / torch.sum(torch.exp(u1.matmul(allu.T) / tau), dim=1)))
    (ref * w).backward()
    g1, g2 = t1.to(cuda).requires_grad_(True), t2.to(cuda).requires_grad_(True)
    out = rbg.ops.info_nce(g1, g2, idx.to(cuda), tau)
    (out * w).backward()
    assert abs(float(out.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    close(g1.grad, a64.grad.float(), tol=1e-5)
    close(g2.grad, b64.grad.float(), tol=1e-5)
    # value-only call (no gradient requested) and a table that does not require grad
    with torch.no_grad():
        assert abs(float(rbg.ops.info_nce(t1.to(cuda), t2.to(cuda), idx.to(cuda), tau)) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    h1 = t1.to(cuda).requires_grad_(True)
    rbg.ops.info_nce(h1, t2.to(cuda), idx.to(cuda), tau).backward()
    close(h1.grad, a64.grad.float() / w, tol=1e-5)


def test_info_nce_zero_row(rbg, cuda):
    """A zero row takes F.normalize's clamp branch (x / eps): finite loss, gradients equal to float64 autograd."""
    gen = torch.Generator().manual_seed(2)
    t1, t2 = torch.randn(50, 16, generator=gen), torch.randn(50, 16, generator=gen)
    t1[3] = 0
    t2[5] = 0
    idx = torch.tensor([3, 5, 7, 3])
    a64, b64 = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
    ref = O.calc_ssl_loss(idx, idx, a64, b64, a64, b64, 0.5, 1.0) / 2  # both halves identical -> one half
    ref.backward()
    g1, g2 = t1.to(cuda).requires_grad_(True), t2.to(cuda).requires_grad_(True)
    out = rbg.ops.info_nce(g1, g2, idx.to(cuda), 0.5)
    out.backward()
    assert torch.isfinite(out.detach())
    assert abs(float(out.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    # calc_ssl_loss used (a64, b64) for both halves: each table's gradient there is twice one half's... divided by 2 above
    close(g1.grad, a64.grad.float(), tol=2e-5)
    close(g2.grad, b64.grad.float(), tol=2e-5)


@pytest.mark.parametrize("f16", [1, 2, 3, 0])
@pytest.mark.parametrize("n,d,b,tau,w,kind", [
    (3000, 64, 300, 0.2, 1.0, "normal"), (3000, 64, 300, 0.05, 1e-7, "heavy"), (2000, 128, 257, 0.2, -3.0, "normal"),
    (1500, 64, 100, 1.0, 250.0, "sparse"), (700, 36, 65, 0.1, 0.05, "heavy"), (900, 62, 40, 0.2, 1.0, "normal")])
def test_info_nce_fp16_form(rbg, cuda, f16, n, d, b, tau, w, kind):
    """Option "lse_f16" (r06, default 3; 1 = tiles from fp16 plane images the row kernels write, 2 = fetched and split per workgroup, 3 = 1 with the tile loop software-pipelined): the gradient passes of the unweighted rbg_infonce_f32 split unit rows and weights in
    [0, 1] into TWO fp16 terms (three products on v_mfma_f32_32x32x16_f16) instead of three bf16 terms (six products).  Both forms
    against float64 autograd of sgl.py:191-199 at the SAME tolerance — rows with heavy tails (elements 1e-6 .. 1 of the row's norm:
    fp16's subnormal range after the 2^8 scale), one-hot-like rows, a weight as small as NCL's ssl_reg and a negative one (the
    weight is divided out of the second product's operand and multiplied back), d = 62 (unaligned rows keep the bf16 kernels)."""
    gen = torch.Generator().manual_seed(n + d + b)
    t1, t2 = torch.randn(n, d, generator=gen), torch.randn(n, d, generator=gen)
    if kind == "heavy":
        t1 = t1 * torch.exp(4.0 * torch.randn(n, d, generator=gen))
        t2 = t2 * torch.exp(4.0 * torch.randn(n, d, generator=gen))
    elif kind == "sparse":
        t1 = t1 * (torch.rand(n, d, generator=gen) < 0.05)
        t2 = t2 * (torch.rand(n, d, generator=gen) < 0.05) + 1e-6 * torch.randn(n, d, generator=gen)
    idx = torch.randint(0, n, (b,), generator=gen)
    idx[:3] = idx[0]
    a64, b64 = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
    nrm = torch.nn.functional.normalize
    u1, u2, allu = nrm(a64[idx], dim=1), nrm(b64[idx], dim=1), nrm(b64, dim=1)
    ref = torch.sum(torch.logsumexp(u1.matmul(allu.T) / tau, dim=1) - torch.sum(u1 * u2, dim=1) / tau)
    (ref * w).backward()
    old = rbg.get_option("lse_f16")
    rbg.set_option("lse_f16", f16)
    try:
        g1, g2 = t1.to(cuda).requires_grad_(True), t2.to(cuda).requires_grad_(True)
        out = rbg.ops.info_nce(g1, g2, idx.to(cuda), tau)
        (out * w).backward()
    finally:
        rbg.set_option("lse_f16", old)
    assert abs(float(out.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
    for got, want in ((g1.grad, a64.grad), (g2.grad, b64.grad)):  # relative to the gradient's own size (w = 1e-7 makes it tiny)
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        assert torch.isfinite(got).all() and err <= 1e-5, f"max err / max |grad| = {err:.3e}"


def test_sgl_device_sampling(rbg, cuda, golden):
    """SGL views sampled on the GPU (device_sampling, the default): a view built from device-resident interactions and a
    device mask equals the host-built view of the same mask bit for bit; ED keeps exactly int(E (1 - ratio)) interactions
    (sgl.py:107-109), ND drops whole nodes (sgl.py:97-106), successive draws differ."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    uid, iid = g["uid"], g["iid"]
    e = len(uid)
    u_dev, i_dev = torch.from_numpy(uid.astype(np.int64)).to(cuda), torch.from_numpy(iid.astype(np.int64)).to(cuda)
    keep = torch.rand(e, generator=torch.Generator().manual_seed(2)) < 0.8
    a = rbg.GraphHandle.from_interactions(u_dev, i_dev, nu, ni, device=cuda, keep=keep.to(cuda))
    b = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, keep=keep.numpy())
    for x, y in zip(a.export_csr(), b.export_csr()):
        assert np.array_equal(x, y)
    full = rbg.GraphHandle.from_interactions(u_dev, i_dev, nu, ni, device=cuda)
    for x, y in zip(full.export_csr(), (g["rowptr"], g["col"], g["val"])):
        assert np.array_equal(x, y)
    model, _ = make_model(rbg, rbg.SGL, cuda, golden, enable_sparse=True, type="ED", drop_ratio=0.1, device_sampling=True)
    v1, _ = model.random_graph_augment()
    v2, _ = model.random_graph_augment()
    assert v1.nnz == v2.nnz == 2 * int(e * (1 - 0.1))
    c1, c2 = v1.export_csr(), v2.export_csr()
    assert not (np.array_equal(c1[0], c2[0]) and np.array_equal(c1[1], c2[1]))
    rp, col, val = c1
    rows = np.repeat(np.arange(nu + ni), np.diff(rp))
    deg = np.diff(rp).astype(np.float32)
    with np.errstate(divide="ignore"):
        dis = np.where(deg > 0, np.float32(1.0) / np.sqrt(deg), np.float32(0.0)).astype(np.float32)
    assert np.array_equal(val, (dis[rows] * np.float32(1.0)) * dis[col])  # re-normalized on the view's own degrees
    full_pairs = set(zip(g["col"].tolist(), np.repeat(np.arange(nu + ni), np.diff(g["rowptr"])).tolist()))
    assert set(zip(col.tolist(), rows.tolist())) <= full_pairs
    model_nd, _ = make_model(rbg, rbg.SGL, cuda, golden, enable_sparse=True, type="ND", drop_ratio=0.2, device_sampling=True)
    vn, _ = model_nd.random_graph_augment()
    d_nd = np.diff(vn.export_csr()[0])
    d_full = np.diff(g["rowptr"])
    dropped_u = int(((d_nd[:nu] == 0) & (d_full[:nu] > 0)).sum())
    assert dropped_u >= int(nu * 0.2) * 0.5  # the sampled users (those with interactions) lost every edge
    model.train()  # sgl.py:82-91: train() re-samples both views
    u1, i1 = model.forward(model.sub_graph1)
    assert torch.isfinite(u1).all() and u1.shape == (nu, 64)


# ---- NCL (ncl.py) -------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,d,k", [(1472, 64, 40), (5000, 64, 100), (700, 16, 7), (3000, 128, 33)])
def test_device_kmeans(rbg, cuda, n, d, k):
    """ops.nearest_centroid (fused MFMA scoring + top-1, no [n, k] matrix) against the float64 distance matrix, and
    ops.kmeans against Lloyd's algorithm in float64 from the same start (the algorithm faiss.Kmeans runs, ncl.py:69-74)."""
    gen = torch.Generator().manual_seed(n + k)
    centers = torch.randn(k, d, generator=gen) * 3.0  # clustered data: a clear optimum, few near-ties
    x = (centers[torch.randint(0, k, (n,), generator=gen)] + torch.randn(n, d, generator=gen)).to(cuda)
    c0 = x[torch.randperm(n, generator=gen)[:k].to(cuda)].clone()
    a = rbg.ops.nearest_centroid(x, c0).cpu().numpy()
    x64, c64 = x.double().cpu().numpy(), c0.double().cpu().numpy()
    d2 = (x64 * x64).sum(1)[:, None] - 2 * x64 @ c64.T + (c64 * c64).sum(1)[None, :]
    best = d2.min(1)
    assert np.all(d2[np.arange(n), a] <= best + 1e-4 * np.maximum(1.0, np.abs(best)))  # nearest up to fp32 near-ties
    assert np.mean(a == d2.argmin(1)) > 0.999
    # one round from the same start is the same round (up to fp32 near-ties in the assignment)
    cent1, _ = rbg.ops.kmeans(x, k, init=c0, niter=1)
    c_ref1, _, _ = O.kmeans_lloyd(x64, c64, niter=1)
    close(cent1, c_ref1.astype(np.float32), tol=2e-3)
    # 25 rounds: trajectories may part at a near-tie, so the runs are compared by what k-means optimises
    cent, assign = rbg.ops.kmeans(x, k, init=c0)
    c_ref, a_ref, obj = O.kmeans_lloyd(x64, c64)
    assert all(b <= a_ + 1e-6 * a_ for a_, b in zip(obj, obj[1:]))  # Lloyd's objective never increases
    cd, ad = cent.double().cpu().numpy(), assign.cpu().numpy()
    obj_dev = float(((x64 - cd[ad]) ** 2).sum())
    obj_ref = float(((x64 - c_ref[a_ref]) ** 2).sum())
    # not worse than the float64 run by more than 5 % (it may be better: an emptied cluster is re-seeded here, kept dead there)
    assert obj_dev <= obj[0] and obj_dev <= 1.05 * obj_ref, (obj_dev, obj_ref, obj[0])
    d2f = (x64 * x64).sum(1)[:, None] - 2 * x64 @ cd.T + (cd * cd).sum(1)[None, :]
    assert np.mean(ad == d2f.argmin(1)) > 0.999  # the returned assignment is the nearest-centroid assignment
    # random start from the fixed seed 1234 (faiss's default): the same start every call; the centroid sums are float
    # atomics (index_add_), so two runs agree to rounding, not bit for bit
    cent2, assign2 = rbg.ops.kmeans(x, k)
    cent3, assign3 = rbg.ops.kmeans(x, k)
    o2 = float(((x64 - cent2.double().cpu().numpy()[assign2.cpu().numpy()]) ** 2).sum())
    o3 = float(((x64 - cent3.double().cpu().numpy()[assign3.cpu().numpy()]) ** 2).sum())
    assert abs(o2 - o3) <= 0.02 * o2 and o2 <= obj[0]
    assert int(assign2.min()) >= 0 and int(assign2.max()) < k
    with pytest.raises(ValueError):
        rbg.ops.kmeans(x[: k - 1], k)


def test_ncl_model(rbg, cuda, golden):
    """NCL (ncl.py:93-184): forward with every layer kept, the 3-term loss (BPR + reg, structure contrast, prototype
    contrast) and its gradients against torch autograd through the restated formulas, with the model's own prototypes."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.NCL, cuda, golden, enable_sparse=True, num_clusters=16, ssl_reg=1e-3, proto_reg=1e-3, hyper_layers=1)
    conv = lambda t: torch.from_numpy(C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], t.detach().numpy()))  # noqa: E731
    uw, iw = model.user_embedding.weight.detach().cpu(), model.item_embedding.weight.detach().cpu()
    u_ref, i_ref, embs = O.ncl_forward(uw, iw, conv, 3, 1)
    with torch.no_grad():
        u, i, lst = model.forward()
    assert len(lst) == 4
    close(torch.cat([u, i]), torch.cat([u_ref, i_ref]))
    close(lst[2], embs[2])
    with pytest.raises(RuntimeError):
        model.calculate_loss({"user_id": torch.tensor([1], device=cuda), "item_id": torch.tensor([1], device=cuda),
                              "neg_item_id": torch.tensor([2], device=cuda)})
    model.e_step()
    assert model.user_centroids.shape == (16, 64) and model.item_2cluster.shape == (ni,)
    close(model.user_centroids.norm(dim=1), torch.ones(16))
    batch = {"user_id": torch.tensor([1, 2, 3, 9, 2]), "item_id": torch.tensor([1, 4, 3, 7, 8]), "neg_item_id": torch.tensor([5, 6, 2, 11, 30])}
    model.train()
    losses = model.calculate_loss({k_: v.to(cuda) for k_, v in batch.items()})
    assert isinstance(losses, tuple) and len(losses) == 3
    sum(losses).backward()
    ei, ew = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=False)
    ul, il = uw.clone().requires_grad_(True), iw.clone().requires_grad_(True)
    ur, ir, er = O.ncl_forward(ul, il, lambda t: O.conv_dense(t, ei, ew), 3, 1)
    ue, pe, ne = ur[batch["user_id"]], ir[batch["item_id"]], ir[batch["neg_item_id"]]
    mf = -torch.log(1e-10 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean()
    reg = (ul[batch["user_id"]].norm(p=2) + il[batch["item_id"]].norm(p=2) + il[batch["neg_item_id"]].norm(p=2)) / 5
    ssl = O.ncl_ssl_layer_loss(er[2], er[0], nu, batch["user_id"], batch["item_id"], model.ssl_temp, model.ssl_reg, model.alpha)
    proto = O.ncl_proto_nce_loss(er[0], nu, batch["user_id"], batch["item_id"], model.user_centroids.cpu(), model.user_2cluster.cpu(),
                                 model.item_centroids.cpu(), model.item_2cluster.cpu(), model.ssl_temp, model.proto_reg)
    ref = (mf + model.reg_weight * reg, ssl, proto)
    for got, want in zip(losses, ref):
        close(got.detach().reshape(()), want.detach().reshape(()), tol=2e-5)
    sum(ref).backward()
    close(model.user_embedding.weight.grad, ul.grad, tol=2e-5)
    close(model.item_embedding.weight.grad, il.grad, tol=2e-5)
    scores = model.full_sort_predict({"user_id": torch.tensor([3, 4], device=cuda)})
    close(scores, O.full_sort_predict(u_ref, i_ref, [3, 4]))


# ---- NGCF -----------------------------------------------------------------------------------

def test_bignn_conv_golden(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    e0 = torch.from_numpy(g["e0_d16"]).to(cuda)
    p0 = [torch.from_numpy(g[f"ngcf_{nm}_0"]).to(cuda) for nm in ("w1", "b1", "w2", "b2")]
    out, p = rbg.ops.bignn_conv_raw(h, e0, *p0)
    close(out, g["bignn_conv0"])
    close(p, g["e1_d16"])
    # whole forward: 16 -> 16 -> 8, fused LeakyReLU + L2 normalize, written into the concat buffer
    n = nu + ni
    buf = torch.empty((n, 40), device=cuda)
    buf[:, :16] = e0
    p1 = [torch.from_numpy(g[f"ngcf_{nm}_1"]).to(cuda) for nm in ("w1", "b1", "w2", "b2")]
    rbg.ops.bignn_conv_raw(h, buf[:, :16], *p0, out=buf[:, 16:32], leaky_norm=True)
    rbg.ops.bignn_conv_raw(h, buf[:, 16:32], *p1, out=buf[:, 32:40], leaky_norm=True)
    close(buf, g["ngcf_out"])


@pytest.mark.parametrize("dims", [(64, 64), (64, 32), (32, 128), (128, 64), (20, 12), (64, 200)])
def test_bignn_conv_shapes(rbg, cuda, golden, dims):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    d_in, d_out = dims
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    x = randn((nu + ni, d_in), 21, cuda)
    w1 = randn((d_out, d_in), 22, cuda) * (2.0 / (d_in + d_out)) ** 0.5
    w2 = randn((d_out, d_in), 23, cuda) * (2.0 / (d_in + d_out)) ** 0.5
    b1, b2 = randn((d_out,), 24, cuda) * 0.1, randn((d_out,), 25, cuda) * 0.1
    conv = lambda t: torch.from_numpy(C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], t.numpy()))  # noqa: E731
    ref = O.bignn_conv(x.cpu(), conv, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu())
    out, _ = rbg.ops.bignn_conv_raw(h, x, w1, b1, w2, b2)
    close(out, ref)
    ref_n = torch.nn.functional.normalize(torch.nn.functional.leaky_relu(ref, 0.2), p=2, dim=1)
    out_n, _ = rbg.ops.bignn_conv_raw(h, x, w1, b1, w2, b2, leaky_norm=True)
    close(out_n, ref_n)


def test_ngcf_model(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.NGCF, cuda, golden, enable_sparse=True, hidden_size_list=[64, 64, 64])
    for layer in model.GNNlayers:  # non-zero biases exercise the bias path
        torch.nn.init.normal_(layer.lin1.bias, std=0.05)
        torch.nn.init.normal_(layer.lin2.bias, std=0.05)
    conv = lambda t: torch.from_numpy(C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], t.numpy()))  # noqa: E731
    params = [(l.lin1.weight.detach().cpu(), l.lin1.bias.detach().cpu(), l.lin2.weight.detach().cpu(),
               l.lin2.bias.detach().cpu()) for l in model.GNNlayers]
    u_ref, i_ref = O.ngcf_forward(model.user_embedding.weight.detach().cpu(), model.item_embedding.weight.detach().cpu(),
                                  conv, params)
    with torch.no_grad():
        u, i = model.forward()  # fused inference path
    assert u.shape == (nu, 256) and i.shape == (ni, 256)
    close(torch.cat([u, i]), torch.cat([u_ref, i_ref]))
    u2, i2 = model.forward()  # autograd path (BiGNNConv modules)
    close(torch.cat([u2, i2]), torch.cat([u_ref, i_ref]))
    scores = model.full_sort_predict({"user_id": torch.tensor([3, 4], device=cuda)})
    close(scores, O.full_sort_predict(u_ref, i_ref, [3, 4]))
    # gradients of the BiGNNConv autograd function against torch autograd through the oracle
    x = randn((nu + ni, 64), 31, cuda).requires_grad_(True)
    layer = model.GNNlayers[0]
    y = layer(x, model.graph, None)
    y.square().sum().backward()
    xr = x.detach().cpu().clone().requires_grad_(True)
    ei, ew = O.get_norm_adj_mat(g["uid"], g["iid"], nu, ni, enable_sparse=False)
    w1 = layer.lin1.weight.detach().cpu().clone().requires_grad_(True)
    w2 = layer.lin2.weight.detach().cpu().clone().requires_grad_(True)
    yr = O.bignn_conv(xr, lambda t: O.conv_dense(t, ei, ew), w1, layer.lin1.bias.detach().cpu(), w2,
                      layer.lin2.bias.detach().cpu())
    yr.square().sum().backward()
    close(x.grad, xr.grad, tol=2e-5)
    close(layer.lin1.weight.grad, w1.grad, tol=2e-5)
    close(layer.lin2.weight.grad, w2.grad, tol=2e-5)


def test_ngcf_node_dropout(rbg, cuda, golden):
    """NGCF with node_dropout > 0 (ngcf.py:74-90): per-forward Bernoulli drop of DIRECTED edges, surviving weights kept
    (PyG dropout_adj).  With the mask fixed, the training forward, the loss and every gradient — the backward runs on the
    transposed (non-symmetric) view — match torch autograd through the restated formulas; eval mode uses the full graph."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.NGCF, cuda, golden, enable_sparse=True, node_dropout=0.3, reg_weight=1e-4)
    rowptr, col, val = model.graph.export_csr()
    nnz = len(col)
    keep = torch.rand(nnz, generator=torch.Generator().manual_seed(11)) >= 0.3
    model._draw_edge_keep = lambda n_edges: keep.to(cuda)
    rows = np.repeat(np.arange(nu + ni), np.diff(rowptr))
    ei = torch.from_numpy(np.stack([col.astype(np.int64), rows]))  # source = column, target = row of adj_t
    ei_k, ew_k = O.dropout_adj(ei, torch.from_numpy(val), keep)
    params = [tuple(t.detach().cpu().clone().requires_grad_(True) for t in (l.lin1.weight, l.lin1.bias, l.lin2.weight, l.lin2.bias))
              for l in model.GNNlayers]
    uw = model.user_embedding.weight.detach().cpu().clone().requires_grad_(True)
    iw = model.item_embedding.weight.detach().cpu().clone().requires_grad_(True)
    batch = {"user_id": torch.tensor([1, 2, 3, 9, 2]), "item_id": torch.tensor([1, 4, 3, 7, 8]), "neg_item_id": torch.tensor([5, 6, 2, 11, 30])}
    ur, ir = O.ngcf_forward(uw, iw, lambda t: O.conv_dense(t, ei_k, ew_k), params)
    ue, pe, ne = ur[batch["user_id"]], ir[batch["item_id"]], ir[batch["neg_item_id"]]
    ref_loss = (-torch.log(1e-10 + torch.sigmoid((ue * pe).sum(1) - (ue * ne).sum(1))).mean()
                + 1e-4 * (ue.norm(p=2) + pe.norm(p=2) + ne.norm(p=2)) / 5)
    ref_loss.backward()
    for fused in (True, False):
        model.fused = fused
        model.zero_grad(set_to_none=True)
        model.train()
        u, i = model.forward()
        close(torch.cat([u, i]), torch.cat([ur, ir]).detach())
        loss = model.calculate_loss({k: v.to(cuda) for k, v in batch.items()})
        loss.backward()
        close(loss.detach().reshape(()), ref_loss.detach().reshape(()))
        close(model.user_embedding.weight.grad, uw.grad, tol=2e-5)
        close(model.item_embedding.weight.grad, iw.grad, tol=2e-5)
        for layer, (w1, b1, w2, b2) in zip(model.GNNlayers, params):
            close(layer.lin1.weight.grad, w1.grad, tol=2e-5)
            close(layer.lin2.weight.grad, w2.grad, tol=2e-5)
    # the drawn mask has the requested rate and differs between forwards
    del model._draw_edge_keep
    m1, m2 = model._draw_edge_keep(nnz), model._draw_edge_keep(nnz)
    assert abs(float(m1.float().mean()) - 0.7) < 0.03 and not torch.equal(m1, m2)
    # eval: dropout_adj(training=False) is the identity
    model.eval()
    conv = lambda t: torch.from_numpy(C.spmm(g["rowptr"], g["col"].astype(np.int64), g["val"], t.numpy()))  # noqa: E731
    p0 = [tuple(t.detach() for t in p) for p in params]
    u_ref, i_ref = O.ngcf_forward(uw.detach(), iw.detach(), conv, p0)
    with torch.no_grad():
        u, i = model.forward()
    close(torch.cat([u, i]), torch.cat([u_ref, i_ref]))


def test_reweighted_view_and_transpose_map(rbg, cuda, golden):
    """GraphHandle.reweighted / transpose_map (rbg_graph_create_reweighted, rbg_graph_transpose_map): a view's product
    uses the caller's values at launch time; the transpose map pairs (r, c) with (c, r), duplicates included."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    rowptr, col, val = h.export_csr()
    tmap = h.transpose_map().cpu().numpy()
    rows = np.repeat(np.arange(nu + ni), np.diff(rowptr))
    assert np.array_equal(col[tmap], rows) and np.array_equal(rows[tmap], col) and np.array_equal(np.sort(tmap), np.arange(len(col)))
    w = torch.rand(len(col), generator=torch.Generator().manual_seed(3)).to(cuda)
    view = h.reweighted(w)
    x = randn((nu + ni, 64), 5, cuda)
    ref = C.spmm(rowptr, col.astype(np.int64), w.cpu().numpy(), x.cpu().numpy())
    close(rbg.ops.spmm_raw(view, x), ref)
    w.mul_(0.5)  # the view reads the values at launch time
    close(rbg.ops.spmm_raw(view, x), 0.5 * ref)
    close(rbg.ops.spmm_raw(h, x), C.spmm(rowptr, col.astype(np.int64), val, x.cpu().numpy()))  # the base is untouched
    # a duplicated interaction and a non-symmetric structure
    hd = rbg.GraphHandle.from_interactions([1, 1, 1, 2], [1, 1, 2, 1], 3, 3, device=cuda)
    rp, cc, _ = hd.export_csr()
    tm = hd.transpose_map().cpu().numpy()
    rr = np.repeat(np.arange(6), np.diff(rp))
    assert np.array_equal(cc[tm], rr) and np.array_equal(np.sort(tm), np.arange(len(cc)))
    bad = rbg.GraphHandle.from_csr(np.array([0, 1, 1]), np.array([1], dtype=np.int32), np.array([1.0], dtype=np.float32), 2, device=cuda)
    with pytest.raises(rbg.RbgError):
        bad.transpose_map()


@pytest.mark.parametrize("n,d_in,d_out", [(1, 4, 4), (31, 16, 8), (33, 64, 64), (1000, 64, 64), (70841, 64, 64), (257, 20, 50),
                                           (500, 128, 96), (100, 64, 128), (64, 7, 5)])
def test_bignn_wgrad(rbg, cuda, n, d_in, d_out):
    """rbg_bignn_wgrad_f32: G^T (P + X), G^T (P * X), column sums of G against float64; strided G and X (views into wider
    buffers, as NGCF's concat buffer gives them); bit-identical across calls (fixed-order split-K)."""
    gen = torch.Generator().manual_seed(n + d_in * 7 + d_out)
    gbuf = torch.randn(n, d_out + 8, generator=gen).to(cuda)
    xbuf = torch.randn(n, d_in + 12, generator=gen).to(cuda)
    g, x = gbuf[:, 4:4 + d_out], xbuf[:, 8:8 + d_in]
    p = torch.randn(n, d_in, generator=gen).to(cuda)
    w1, w2, b = rbg.ops.bignn_wgrad_raw(g, p, x)
    g64, x64, p64 = g.double().cpu(), x.double().cpu(), p.double().cpu()
    scale = max(1.0, float(n) ** 0.5)
    close(w1, (g64.T @ (p64 + x64)).float(), tol=2e-6 * scale)
    close(w2, (g64.T @ (p64 * x64)).float(), tol=2e-6 * scale)
    close(b, g64.sum(0).float(), tol=2e-6 * scale)
    w1b, w2b, bb = rbg.ops.bignn_wgrad_raw(g, p, x)
    assert torch.equal(w1, w1b) and torch.equal(w2, w2b) and torch.equal(b, bb)


# ---- whole training step as a HIP graph --------------------------------------------------------

@pytest.mark.parametrize("name", ["LightGCN", "NGCF", "SGL"])
def test_graphed_step_matches_eager(rbg, cuda, golden, name):
    """train.GraphedStep (zero_grad + calculate_loss + backward + Adam captured once, replayed per batch) takes the same
    optimisation steps as the eager loop: losses and parameters after three different batches agree, and the warm-up
    inside the constructor leaves parameters and optimiser state untouched."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    cfg = dict(enable_sparse=True, reg_weight=1e-4, require_pow=False)
    if name == "SGL":
        cfg.update(type="ED", drop_ratio=0.1, ssl_tau=0.5, ssl_weight=0.05)
    np.random.seed(3)
    model, _ = make_model(rbg, getattr(rbg, name), cuda, golden, **cfg)
    twin, _ = make_model(rbg, getattr(rbg, name), cuda, golden, **cfg)
    twin.load_state_dict(model.state_dict())
    model.train()
    twin.train()  # SGL.train() samples the augmented views (sgl.py:94-98) ...
    if name == "SGL":  # ... and both models must train on the very same ones
        twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2
    gen = torch.Generator().manual_seed(9)
    batches = [{"user_id": torch.randint(1, nu, (64,), generator=gen).to(cuda), "item_id": torch.randint(1, ni, (64,), generator=gen).to(cuda),
                "neg_item_id": torch.randint(1, ni, (64,), generator=gen).to(cuda)} for _ in range(3)]
    before = [p.detach().clone() for p in model.parameters()]
    stepper = rbg.GraphedStep(model, batches[0], lr=1e-3)
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)  # the warm-up left no trace
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    for step_no, batch in enumerate(batches):
        lg = float(stepper.step(batch).detach())
        opt.zero_grad(set_to_none=True)
        le = twin.calculate_loss(batch)
        le.backward()
        opt.step()
        # same parameters in the first step; afterwards the Adam noise described below feeds back into the loss
        assert abs(lg - float(le.detach())) <= (2e-5 if step_no == 0 else 2e-4) * max(1.0, abs(float(le.detach())))
    # Adam divides by |g|: where a gradient is ~0 the order of the float atomics in the row scatters decides the sign of a
    # step of size lr, so parameters are compared at a fraction of 3 lr, the losses above at 2e-5
    for pg, pe in zip(model.parameters(), twin.parameters()):
        close(pg, pe.detach(), tol=1e-4)
    with pytest.raises(ValueError):
        stepper.step({k: v[:10] for k, v in batches[0].items()})
    stepper.eager_step({k: v[:10] for k, v in batches[0].items()})  # odd-sized batch: eager path
    if name == "SGL":  # a new epoch samples new views: the step must re-capture, not replay the old handles
        old = stepper.graph
        model.train()
        twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2
        twin.load_state_dict(model.state_dict())
        lg = float(stepper.step(batches[1]).detach())
        assert stepper.graph is not old
        with torch.no_grad():
            pass
        le = float(twin.calculate_loss(batches[1]).detach())
        # the graphed loss was computed BEFORE its Adam update, on the parameters twin holds now
        assert abs(lg - le) <= 2e-4 * max(1.0, abs(le))


@pytest.mark.parametrize("n,d_out", [(1, 64), (15, 64), (16, 32), (17, 48), (1000, 64), (4099, 16), (70841, 64), (200003, 64), (33000, 32)])
def test_bignn_dense_dma_kernel(rbg, cuda, n, d_out):
    """rbg_bignn_dense_f32 at the NGCF width (d_in = 64, d_out a multiple of 16 up to 64): the LDS-DMA kernel (rows by
    global_load_lds, weights in registers, 16x16x4 MFMA on Y^T, one software-pipelined wave per SIMD; `bignn_dma` option)
    against float64 of layers.py:56-58 [+ ngcf.py:96,98] and against the general kernel, plain and fused-tail, with x / out
    as column slices of wider buffers (NGCF's concat buffer) and every pipeline depth (one tile per wave up to several,
    ragged last tile whose missing rows are clamped copies)."""
    gen = torch.Generator().manual_seed(n * 7 + d_out)
    xbuf = torch.randn(n, 64 + 8, generator=gen).to(cuda)
    x = xbuf[:, 4:68]
    p = torch.randn(n, 64, generator=gen).to(cuda)
    w1, w2 = (torch.randn(d_out, 64, generator=gen) * 0.2).to(cuda), (torch.randn(d_out, 64, generator=gen) * 0.2).to(cuda)
    b1, b2 = (torch.randn(d_out, generator=gen) * 0.1).to(cuda), (torch.randn(d_out, generator=gen) * 0.1).to(cuda)
    p64, x64 = p.double().cpu(), x.double().cpu()
    z = (p64 + x64) @ w1.double().cpu().T + b1.double().cpu() + (p64 * x64) @ w2.double().cpu().T + b2.double().cpu()
    zn = torch.nn.functional.normalize(torch.nn.functional.leaky_relu(z, 0.2), p=2, dim=1)
    default = rbg.get_option("bignn_dma")
    assert default != 0
    try:
        for leaky, ref in ((False, z), (True, zn)):
            ybuf = torch.full((n, d_out + 8), 7.0, device=cuda)
            y = ybuf[:, 4:4 + d_out]
            rbg.set_option("bignn_dma", default)
            rbg.ops.bignn_dense_raw(p, x, w1, b1, w2, b2, out=y, leaky_norm=leaky)
            close(y, ref.float())
            assert torch.all(ybuf[:, :4] == 7.0) and torch.all(ybuf[:, 4 + d_out:] == 7.0)  # nothing outside the slice
            rbg.set_option("bignn_dma", 0)
            y0 = rbg.ops.bignn_dense_raw(p, x, w1, b1, w2, b2, leaky_norm=leaky)
            close(y, y0.cpu(), tol=5e-6)
    finally:
        rbg.set_option("bignn_dma", default)


@pytest.mark.parametrize("n,d_in,d_out", [(1, 8, 8), (33, 16, 24), (500, 64, 64), (257, 20, 50), (300, 128, 64), (129, 64, 128)])
def test_bignn_layer_forward_backward(rbg, cuda, n, d_in, d_out):
    """ops.bignn_layer = normalize(LeakyReLU(BiGNNConv(x))) (layers.py:54-58, ngcf.py:96,98): value and all five gradients
    against float64 autograd of the same expression on the oracle's dense operator."""
    rng = np.random.default_rng(n + d_in)
    nu = max(1, n // 2)
    ni = n - nu + 1 if n > 1 else 1
    nu = n - ni if n > 1 else 0
    if n == 1:
        uid, iid, nu, ni = np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64), 0, 1
    else:
        e = 6 * n
        uid, iid = rng.integers(0, nu, e), rng.integers(0, ni, e)
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    rp, c, v = h.export_csr()
    dense = torch.zeros(n, n, dtype=torch.float64)
    rows = np.repeat(np.arange(n), np.diff(rp))
    dense.index_put_((torch.from_numpy(rows), torch.from_numpy(c.astype(np.int64))), torch.from_numpy(v.astype(np.float64)), accumulate=True)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, d_in, generator=gen)
    w1, w2 = torch.randn(d_out, d_in, generator=gen) * 0.3, torch.randn(d_out, d_in, generator=gen) * 0.3
    b1, b2 = torch.randn(d_out, generator=gen) * 0.1, torch.randn(d_out, generator=gen) * 0.1
    up = torch.randn(n, d_out, generator=gen)
    for p_drop in (0.0, 0.3):  # ngcf.py:96-98: LeakyReLU -> Dropout -> normalize, here with an explicit scaled keep mask
        mask = None if p_drop == 0 else (torch.rand(n, d_out, generator=gen) >= p_drop).float() / (1 - p_drop)
        ref_in = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        xr, w1r, b1r, w2r, b2r = ref_in
        pr = dense @ xr
        zr = (pr + xr) @ w1r.T + b1r + (pr * xr) @ w2r.T + b2r
        ar = torch.nn.functional.leaky_relu(zr, 0.2)
        if mask is not None:
            ar = ar * mask.double()
        yr = torch.nn.functional.normalize(ar, p=2, dim=1)
        (yr * up.double()).sum().backward()
        got_in = [t.to(cuda).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = rbg.ops.bignn_layer(*got_in, h, 0.2, mask=mask.to(cuda) if mask is not None else None)
        (y * up.to(cuda)).sum().backward()
        close(y, yr.detach().float(), tol=2e-5)
        for got, ref, name in zip(got_in, ref_in, ("x", "w1", "b1", "w2", "b2")):
            scale = max(1.0, float(ref.grad.abs().max()))
            err = float((got.grad.cpu().double() - ref.grad).abs().max())
            assert err <= 3e-5 * scale, (name, p_drop, err, scale)
    # p_drop draws its own mask: about that fraction of the layer output is exactly zero
    with torch.no_grad():
        yd = rbg.ops.bignn_layer(*[t.to(cuda) for t in (x, w1, b1, w2, b2)], h, 0.2, p_drop=0.25)
    if n * d_out >= 4000:
        assert abs(float((yd == 0).float().mean()) - 0.25) < 0.05


@pytest.mark.parametrize("n", [17, 4099, 70841, 150001])
@pytest.mark.parametrize("variant", ["plain", "tail", "tail+mask"])
def test_bignn_backward_dma_kernel(rbg, cuda, n, variant):
    """rbg_bignn_backward_f32 at d_in = d_out = 64: the fused input + weight gradient LDS-DMA kernel (`bignn_dma`; every
    pipeline depth, ragged last tile whose missing rows must not enter the weight gradients) against the general kernels — dX, dW1, dW2, db — and, at the small sizes, against float64 autograd of
    layers.py:54-58 [+ ngcf.py:96-98]; gy / x are column slices of wider buffers."""
    rng = np.random.default_rng(n)
    nu = n // 2
    ni = n - nu
    e = 4 * n
    h = rbg.GraphHandle.from_interactions(rng.integers(0, nu, e), rng.integers(0, ni, e), nu, ni, device=cuda)
    gen = torch.Generator().manual_seed(n + 3)
    xbuf, gbuf = torch.randn(n, 72, generator=gen).to(cuda), torch.randn(n, 80, generator=gen).to(cuda)
    x, gy = xbuf[:, 4:68], gbuf[:, 8:72]
    w1, w2 = (torch.randn(64, 64, generator=gen) * 0.2).to(cuda), (torch.randn(64, 64, generator=gen) * 0.2).to(cuda)
    b1, b2 = (torch.randn(64, generator=gen) * 0.1).to(cuda), (torch.randn(64, generator=gen) * 0.1).to(cuda)
    mask = ((torch.rand(n, 64, generator=gen) >= 0.3).float() / 0.7).to(cuda) if variant == "tail+mask" else None
    p = rbg.ops.spmm_raw(h, x.contiguous())
    if variant == "plain":
        y, inv = None, None
    else:  # the forward of the fused layer gives y and the rows' 1 / norm
        z = (p + x) @ w1.T + b1 + (p * x) @ w2.T + b2
        a = torch.nn.functional.leaky_relu(z, 0.2)
        if mask is not None:
            a = a * mask
        norm = a.norm(dim=1)
        inv = 1.0 / norm.clamp_min(1e-12)
        y = a * inv[:, None]
    out = {}
    default = rbg.get_option("bignn_dma")
    try:
        for opt in (1, 0):  # 1: input + weight gradients in one LDS-DMA kernel, 0: the general kernels
            rbg.set_option("bignn_dma", opt)
            out[opt] = rbg.ops.bignn_backward_raw(h.transpose(), gy, y, inv, mask, x, p, w1, w2, 0.2)
        rbg.set_option("bignn_dma", 1)
        again = rbg.ops.bignn_backward_raw(h.transpose(), gy, y, inv, mask, x, p, w1, w2, 0.2)
    finally:
        rbg.set_option("bignn_dma", default)
    for got, ref, name in zip(out[1], out[0], ("dX", "dW1", "dW2", "db")):
        scale = max(1.0, float(ref.abs().max()))
        err = float((got - ref).abs().max())
        assert err <= 1e-5 * scale, (name, err, scale)
    for a_, b_ in zip(out[1], again):  # fixed-order partial sums: bit-identical across calls
        assert torch.equal(a_, b_)
    if n <= 5000:  # float64 autograd of the reference expression on the dense operator
        rp, c, v = h.export_csr()
        dense = torch.zeros(n, n, dtype=torch.float64)
        rows = np.repeat(np.arange(n), np.diff(rp))
        dense.index_put_((torch.from_numpy(rows), torch.from_numpy(c.astype(np.int64))), torch.from_numpy(v.astype(np.float64)), accumulate=True)
        xr, w1r, b1r, w2r, b2r = [t.detach().double().cpu().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        pr = dense @ xr
        zr = (pr + xr) @ w1r.T + b1r + (pr * xr) @ w2r.T + b2r
        if variant != "plain":
            zr = torch.nn.functional.leaky_relu(zr, 0.2)
            if mask is not None:
                zr = zr * mask.double().cpu()
            zr = torch.nn.functional.normalize(zr, p=2, dim=1)
        (zr * gy.double().cpu()).sum().backward()
        for got, ref, name in zip(out[1], (xr.grad, w1r.grad, w2r.grad, b1r.grad), ("dX", "dW1", "dW2", "db")):
            scale = max(1.0, float(ref.abs().max()))
            err = float((got.cpu().double() - ref).abs().max())
            assert err <= 3e-5 * scale, (name, err, scale)


def test_ngcf_fused_training_path_matches_op_by_op(rbg, cuda, golden):
    """NGCF.calculate_loss through the fused layers (default) and through the op-by-op structure of ngcf.py:92-104
    (fused_forward=False): same loss, same gradients."""
    ma, _ = make_model(rbg, rbg.NGCF, cuda, golden, enable_sparse=True, reg_weight=1e-4)
    mb, _ = make_model(rbg, rbg.NGCF, cuda, golden, enable_sparse=True, reg_weight=1e-4, fused_forward=False)
    mb.load_state_dict(ma.state_dict())
    ma.train(), mb.train()
    batch = {"user_id": torch.tensor([1, 2, 3, 9, 2], device=cuda), "item_id": torch.tensor([1, 4, 3, 7, 8], device=cuda),
             "neg_item_id": torch.tensor([5, 6, 2, 11, 30], device=cuda)}
    la, lb = ma.calculate_loss(batch), mb.calculate_loss(batch)
    la.backward(), lb.backward()
    close(la.detach().reshape(()), lb.detach().reshape(()), tol=1e-5)
    for (na, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        close(pa.grad, pb.grad, tol=2e-5)


# ---- scoring GEMM ---------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(1, 1125, 64), (3, 1125, 64), (130, 1000, 64), (33, 70, 256), (5, 40, 128),
                                   (7, 33, 20), (4, 50, 300), (64, 64, 66)])
def test_score_vs_fp64(rbg, cuda, shape):
    b, n, d = shape
    u = randn((b, d), 41, cuda)
    it = randn((n, d), 42, cuda)
    s = rbg.score(u, it)
    ref = u.cpu().double() @ it.cpu().double().T
    close(s, ref, tol=2e-6 * max(1, d / 64))
    # transpose-detecting: asymmetric operands, compare a single known element too
    assert abs(float(s[b - 1, 0]) - float(ref[b - 1, 0])) < 1e-4
    idx = torch.tensor([0, b - 1], device=cuda)
    assert torch.equal(rbg.gather_rows(u, idx), u[idx])


def test_score_edges_alignment_and_canaries(rbg, cuda):
    """The shifted line-aligned store stream of score.hip at its edges: item counts around the 32 / 64 boundaries, user
    counts around a tile, output bases at every kind of 128-byte phase, forced walk lengths of 1-3 tiles and auto; the
    result is checked against float64 and the floats just before / after the [B, n] block must stay untouched."""
    import ctypes
    lib, c_vp = rbg._lib.lib, ctypes.c_void_p
    gen = torch.Generator().manual_seed(77)
    pad = 64
    try:
        for tiles in (0, 1, 2, 3):
            rbg.set_option("score_tiles", tiles)
            for d in (64, 20, 128):
                for b in (1, 31, 33, 130):
                    for n in (1, 31, 32, 33, 63, 64, 65, 97, 129, 1000):
                        if tiles and (b, d) not in ((33, 64), (130, 20), (1, 128)):
                            continue  # the walk-length sweep runs on a subset
                        u = torch.randn(b, d, generator=gen).to(cuda)
                        it = torch.randn(n, d, generator=gen).to(cuda)
                        ref = (u.cpu().double() @ it.cpu().double().T).float()
                        for off in (0, 1, 13, 31):
                            buf = torch.full((pad + off + b * n + pad,), -777.0, device=cuda)
                            out = buf[pad + off: pad + off + b * n]
                            rc = lib.rbg_score_f32(c_vp(u.data_ptr()), d, c_vp(it.data_ptr()), d, c_vp(out.data_ptr()), b, n, d,
                                                   c_vp(torch.cuda.current_stream().cuda_stream))
                            assert rc == 0, rbg._lib.lib.rbg_last_error()
                            got = out.view(b, n).cpu()
                            err = float((got - ref).abs().max())
                            assert err <= 2e-5 * max(1, d / 64), (tiles, d, b, n, off, err)
                            assert bool((buf[: pad + off] == -777.0).all()) and bool((buf[pad + off + b * n:] == -777.0).all()), \
                                (tiles, d, b, n, off, "wrote outside the output")
    finally:
        rbg.set_option("score_tiles", 0)


# ---- fused full-sort evaluation (score + mask + top-k) ---------------------------------------

def reference_topk(user_all, item_all, users, k, uid, iid):
    """RecBole's _full_sort_batch_eval on the oracle side: scores, scores[:,0] = -inf, history = -inf, topk."""
    scores = user_all[users].double() @ item_all.double().T
    scores[:, 0] = -np.inf
    hist = {}
    for u, i in zip(uid.tolist(), iid.tolist()):
        hist.setdefault(u, []).append(i)
    for r, u in enumerate(users.tolist()):
        if u in hist:
            scores[r, hist[u]] = -np.inf
    return scores, torch.topk(scores, k, dim=1)


@pytest.mark.parametrize("bk", [(1, 10), (3, 1), (64, 10), (200, 32), (33, 5)])
@pytest.mark.parametrize("d", [64, 256, 20])
def test_full_sort_topk(rbg, cuda, golden, bk, d):
    g = golden
    b, k = bk
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    user_all, item_all = randn((nu, d), 71, cuda), randn((ni, d), 72, cuda)
    users = torch.from_numpy(np.random.default_rng(b).integers(0, nu, b))
    vals, idx = rbg.full_sort_topk(h, user_all, item_all, users.to(cuda), k)
    scores, (rv, ri) = reference_topk(user_all.cpu(), item_all.cpu(), users, k, g["uid"], g["iid"])
    assert vals.shape == (b, k) and idx.shape == (b, k) and idx.dtype == torch.int64
    close(vals, rv.float(), tol=1e-5 * max(1, d / 64))
    vals_c, idx_c = vals.cpu(), idx.cpu()
    for r in range(b):
        assert len(set(idx_c[r].tolist())) == k            # no duplicates
        assert 0 not in idx_c[r].tolist()                  # PAD item masked
        got = scores[r, idx_c[r]]                          # every returned id carries the score we report ...
        assert torch.all(torch.isfinite(got))              # ... and none of them is a history item
        assert (got.float() - vals_c[r]).abs().max() <= 1e-4
        assert torch.all(vals_c[r, :-1] >= vals_c[r, 1:])  # sorted, best first
    # exact index agreement wherever the k-th and (k+1)-th reference scores are clearly apart
    srt = torch.sort(scores, dim=1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 1e-3 if ni > k else torch.ones(b, dtype=torch.bool)
    for r in torch.nonzero(clear).flatten().tolist():
        assert set(idx_c[r].tolist()) == set(ri[r].tolist())
    # no history graph: only the PAD item is masked
    v2, i2 = rbg.full_sort_topk(None, user_all, item_all, users.to(cuda), k)
    s2 = user_all.cpu()[users].double() @ item_all.cpu().double().T
    s2[:, 0] = -np.inf
    close(v2, torch.topk(s2, k, dim=1).values.float(), tol=1e-5 * max(1, d / 64))


def test_score_and_topk_at_the_evaluation_batch(rbg, cuda):
    """the B = 4096 x 40 982-item shape (the multi-walk path of score.hip and the fused top-k's large
    batch) was only benchmarked.  rbg_score_f32 against float64 on 64 sampled user rows (every column) and on 64 sampled
    columns (every user row); the fused top-10 (PAD masked, no history) against float64 top-k on the sampled rows."""
    b, n, d, k = 4096, 40_982, 64, 10
    u, it = randn((b, d), 91, cuda), randn((n, d), 92, cuda)
    s = rbg.score(u, it)
    rows = torch.from_numpy(np.random.default_rng(0).choice(b, 64, replace=False)).sort().values
    cols = torch.from_numpy(np.random.default_rng(1).choice(n, 64, replace=False)).sort().values
    ud, itd = u.cpu().double(), it.cpu().double()
    ref_rows = ud[rows] @ itd.T
    ref_cols = ud @ itd[cols].T
    close(s[rows.to(cuda)], ref_rows, tol=2e-6)
    close(s[:, cols.to(cuda)], ref_cols, tol=2e-6)
    assert torch.isfinite(s).all()
    vals, idx = rbg.full_sort_topk(None, u, it, torch.arange(b, device=cuda), k)
    ref_rows[:, 0] = -np.inf  # PAD item
    rv, ri = torch.topk(ref_rows, k, dim=1)
    close(vals[rows.to(cuda)], rv.float(), tol=1e-5)
    idx_c = idx.cpu()[rows]
    srt = torch.sort(ref_rows, dim=1, descending=True).values
    clear = (srt[:, k - 1] - srt[:, k]) > 1e-3
    for r in torch.nonzero(clear).flatten().tolist():
        assert set(idx_c[r].tolist()) == set(ri[r].tolist())
    assert int(clear.sum()) > 32


def test_full_sort_topk_short_lists(rbg, cuda, golden):
    """The 24-entry candidate lists of the large-batch top-k (option "topk_short_lists"): identical output to the 48-entry
    lists — also WITHOUT a pre-pass bound (a small item set: the thresholds start at -inf, every tile brings 32 arrivals per
    user, so every append goes through the fill / prune / go-on rounds) and with history masks and repeated users."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    cases = [(h, randn((nu, 64), 5, cuda), randn((ni, 64), 6, cuda), torch.from_numpy(np.random.default_rng(1).integers(0, nu, 2100)).to(cuda), 10),
             (None, randn((3000, 64), 7, cuda), randn((40_982, 64), 8, cuda), torch.arange(2999, -1, -1, device=cuda), 12),
             (None, randn((300, 33), 9, cuda), randn((5000, 33), 10, cuda), torch.arange(300, device=cuda).repeat(8), 1)]
    try:
        for hist, ua, it, users, k in cases:
            out = {}
            for mode in (0, 2):
                rbg.set_option("topk_short_lists", mode)
                out[mode] = rbg.full_sort_topk(hist, ua, it, users, k)
            assert torch.equal(out[0][1], out[2][1]) and torch.equal(out[0][0], out[2][0])
        hist, ua, it, users, k = cases[0]
        scores, (rv, ri) = reference_topk(ua.cpu(), it.cpu(), users.cpu(), k, g["uid"], g["iid"])
        rbg.set_option("topk_short_lists", 2)
        close(rbg.full_sort_topk(hist, ua, it, users, k)[0], rv.float(), tol=1e-5)
    finally:
        rbg.set_option("topk_short_lists", 1)
    with pytest.raises(rbg.RbgError):
        rbg.set_option("topk_short_lists", 3)


def test_full_sort_topk_plane_image(rbg, cuda, golden):
    """r06, option "topk_image": the item table split once per call into bf16 planes in the LDS layout, tiles taken by LDS-DMA in
    the pre-pass and the main pass — the same planes and products as the per-workgroup fetch + split: identical output bit for
    bit (d = 128 by default from 1024 users; forced at d = 64 and d = 33 (ragged rows) with option value 2; history masks,
    repeated users, an item count that is not a multiple of the tile), and equal to the reference top-k."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    cases = [(h, randn((nu, 128), 5, cuda), randn((ni, 128), 6, cuda), torch.from_numpy(np.random.default_rng(1).integers(0, nu, 1100)).to(cuda), 10),
             (None, randn((3000, 64), 7, cuda), randn((40_991, 64), 8, cuda), torch.arange(2999, -1, -1, device=cuda), 12),
             (None, randn((300, 33), 9, cuda), randn((5003, 33), 10, cuda), torch.arange(300, device=cuda).repeat(8), 20),
             (None, randn((1500, 100), 11, cuda), randn((20_000, 100), 12, cuda), torch.arange(1500, device=cuda), 3)]
    try:
        for hist, ua, it, users, k in cases:
            out = {}
            for mode in (0, 2):
                rbg.set_option("topk_image", mode)
                out[mode] = rbg.full_sort_topk(hist, ua, it, users, k)
            assert torch.equal(out[0][1], out[2][1]) and torch.equal(out[0][0], out[2][0])
        hist, ua, it, users, k = cases[0]
        scores, (rv, ri) = reference_topk(ua.cpu(), it.cpu(), users.cpu(), k, g["uid"], g["iid"])
        rbg.set_option("topk_image", 1)  # the default: on at d = 128
        close(rbg.full_sort_topk(hist, ua, it, users, k)[0], rv.float(), tol=2e-5)
    finally:
        rbg.set_option("topk_image", 1)


def _random_history(n_users, n_items, per_user, seed, hub=None):
    rng = np.random.default_rng(seed)
    uid = np.repeat(np.arange(1, n_users), per_user)
    iid = rng.integers(1, n_items, uid.shape[0])
    if hub is not None:
        u, items = hub
        uid = np.concatenate([uid, np.full(len(items), u)])
        iid = np.concatenate([iid, items])
    pairs = np.unique(np.stack([uid, iid], 1), axis=0)
    return pairs[:, 0], pairs[:, 1]


def _check_topk_against_reference(vals, idx, user_all, item_all, users, k, uid, iid, tol):
    scores, (rv, ri) = reference_topk(user_all.cpu(), item_all.cpu(), users.cpu(), k, uid, iid)
    close(vals, rv.float(), tol=tol)
    idx_c, vals_c = idx.cpu(), vals.cpu()
    srt = torch.sort(scores, dim=1, descending=True).values
    gap = (srt[:, k - 1] - srt[:, k]) > 1e-3 * max(1.0, float(srt[:, 0].abs().max()))
    for r in range(users.shape[0]):
        row = idx_c[r].tolist()
        assert len(set(row)) == k and 0 not in row
        assert torch.all(torch.isfinite(scores[r, idx_c[r]]))      # no history item, no PAD
        assert torch.all(vals_c[r, :-1] >= vals_c[r, 1:])
        if gap[r]:
            assert set(row) == set(ri[r].tolist())
    assert int(gap.sum()) > users.shape[0] // 2


def test_full_sort_topk_screen(rbg, cuda):
    """r06, option "topk_screen" (csrc/topk_screen.hip): ONE bf16 product per (user, item) with a rigorous error bound on the matrix
    core screens the call, the survivors are rescored exactly in fp32 — the same top-k as the exact passes (option 0) and as the
    float64 reference (lightgcn.py:123-133 + _full_sort_batch_eval): history masks incl. a hub user whose 2 500 history items all
    score high (its slot maxima in the pre-pass sample are history items: the threshold kernel walks the history), repeated users,
    ragged rows (d = 33: scalar loads), d = 100 / 128 (eight product fragments), an item count that is not a multiple of the tile,
    item ids ordered by popularity (the candidates of every user in the same few item tiles); and the two cases in which candidate regions overflow and the merge takes every pair of the
    region's chunk instead: all scores equal (every pair passes) and a user whose history covers the whole pre-pass sample (no
    threshold: every item is its candidate)."""
    nu, ni = 3000, 9003
    hub_items = np.random.default_rng(3).choice(np.arange(1, ni), 2500, replace=False)
    uid, iid = _random_history(nu, ni, 20, 1, hub=(7, hub_items))
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    ua, it = randn((nu, 64), 5, cuda), randn((ni, 64), 6, cuda)
    it[torch.from_numpy(hub_items).to(cuda)] += 0.5 * ua[7]     # the hub's history items are its best items: all must be dropped
    users_h = torch.cat([torch.tensor([7, 7, 1, nu - 1]), torch.from_numpy(np.random.default_rng(1).integers(1, nu, 2100))]).to(cuda)
    cases = [("history", h, ua, it, users_h, 10, uid, iid),
             ("d128", None, randn((1500, 128), 11, cuda), randn((20_000, 128), 12, cuda), torch.arange(1500, device=cuda), 3, [], []),
             ("d33", None, randn((300, 33), 9, cuda), randn((5003, 33), 10, cuda), torch.arange(300, device=cuda).repeat(8), 20, [], []),
             ("d100k32", h, randn((nu, 100), 13, cuda), randn((ni, 100), 14, cuda), torch.arange(1, 1300, device=cuda), 32, uid, iid)]
    # item ids ordered by popularity (the bench's synthetic graphs, many real ones): the first few hundred items are most users' best
    # items — every user's candidates would meet in the regions of the first item chunk if the chunks were contiguous ranges
    pu, pi = randn((2048, 64), 21, cuda), randn((30_011, 64), 22, cuda)
    mean_dir = pu.mean(dim=0) / pu.mean(dim=0).norm()
    pu += 2.0 * mean_dir
    pi[:700] += 3.0 * mean_dir
    cases.append(("popular", None, pu, pi, torch.arange(2048, device=cuda), 10, [], []))
    rbg.set_option("topk_sample", 1024)
    try:
        for name, hist, u, i, users, k, hu, hi in cases:
            out = {}
            for mode in (0, 2):
                rbg.set_option("topk_screen", mode)
                out[mode] = rbg.full_sort_topk(hist, u, i, users, k)
            close(out[2][0], out[0][0], tol=2e-6)
            same = (out[0][1] == out[2][1]).all(dim=1)
            assert float(same.float().mean()) > 0.99, name      # (a last-bit tie may swap two neighbours)
            _check_topk_against_reference(out[2][0], out[2][1], u, i, users, k, np.asarray(hu), np.asarray(hi), tol=1e-5 * max(1, u.shape[1] / 64))
        # every score equal: the screen passes every pair, every region overflows
        flat_u, flat_i = torch.ones((1200, 64), device=cuda), torch.ones((5000, 64), device=cuda)
        users = torch.arange(1200, device=cuda)
        res = {}
        for mode in (0, 2):
            rbg.set_option("topk_screen", mode)
            res[mode] = rbg.full_sort_topk(None, flat_u, flat_i, users, 10)
        assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
        assert torch.equal(res[2][1][0].cpu(), torch.arange(1, 11))  # ties by item id, PAD masked
        # a user whose history is the whole sample: no bound for it
        cover = np.arange(1, 1100)
        uid2, iid2 = _random_history(nu, ni, 20, 2, hub=(9, cover))
        h2 = rbg.GraphHandle.from_interactions(uid2, iid2, nu, ni, device=cuda)
        users = torch.arange(1, 1201, device=cuda)
        for mode in (0, 2):
            rbg.set_option("topk_screen", mode)
            res[mode] = rbg.full_sort_topk(h2, ua, it, users, 10)
        close(res[2][0], res[0][0], tol=2e-6)
        assert torch.equal(res[0][1][8], res[2][1][8])  # (user 9; its candidates are every item: they fit the regions at this size)
        _check_topk_against_reference(res[2][0], res[2][1], ua, it, users, 10, uid2, iid2, tol=1e-5)
    finally:
        rbg.set_option("topk_screen", 1)
        rbg.set_option("topk_sample", 8192)
    with pytest.raises(rbg.RbgError):
        rbg.set_option("topk_nonexistent", 1)


def test_full_sort_topk_screen_fuzz(cuda):
    """25 seconds of devtools/r06_topk_screen_fuzz.py: random shapes, widths (8 .. 128, incl. ragged), k, batch sizes from 1, history
    graphs incl. hub users, and value distributions (normal, heavy-tailed, popularity-ordered, rank 3, 12 decades of row scales)
    through the screened top-k against the float64 reference: values to 2e-5 of the largest score, the same item sets wherever the
    k-th and (k + 1)-th scores are apart.  (A run of 150 s: 485 cases, 247 495 rows, none bad — profiles/r06_topk_screen_fuzz.json.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "devtools", "r06_topk_screen_fuzz.py"), "11", "25"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["cases"] >= 10 and rec["n_bad"] == 0, rec


def test_full_sort_topk_model_and_few_candidates(rbg, cuda, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    model, _ = make_model(rbg, rbg.LightGCN, cuda, golden, enable_sparse=True)
    users = torch.tensor([1, 2, nu - 1], device=cuda)
    vals, idx = model.full_sort_topk({"user_id": users}, 10)
    flat = model.full_sort_predict({"user_id": users}).view(3, ni).cpu().double()
    scores, (rv, ri) = reference_topk(model.restore_user_e.cpu(), model.restore_item_e.cpu(), users.cpu(), 10, g["uid"], g["iid"])
    close(vals, rv.float())
    assert (flat - (model.restore_user_e.cpu()[users.cpu()].double() @ model.restore_item_e.cpu().double().T)).abs().max() < 1e-5
    # a user whose history covers everything but 3 items: the tail of the top-k is (-inf, -1)
    nu2, ni2 = 3, 40
    uid2 = np.concatenate([np.full(36, 1), [2]])
    iid2 = np.concatenate([np.arange(4, 40), [5]])
    h2 = rbg.GraphHandle.from_interactions(uid2, iid2, nu2, ni2, device=cuda)
    ua, ia = randn((nu2, 64), 1, cuda), randn((ni2, 64), 2, cuda)
    v, i = rbg.full_sort_topk(h2, ua, ia, torch.tensor([1], device=cuda), 8)
    assert set(i[0, :3].tolist()) == {1, 2, 3} and torch.all(i[0, 3:] == -1) and torch.all(torch.isinf(v[0, 3:]))


def test_no_device_memory_leak(rbg, cuda, golden):
    """Graph handles (both builders, views, partitioned, transposes) free everything they allocate."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    x = randn((nu + ni, 64), 3, cuda)

    def cycle(n):
        for i in range(n):
            flags = rbg._lib.GRAPH_BUILD_ON_HOST if i % 2 else 0
            h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda, flags=flags,
                                                  keep=g["sgl_keep"] if i % 3 == 0 else None)
            rbg.ops.spmm_raw(h, x)
            h.destroy()
        torch.cuda.synchronize()

    cycle(5)
    free0, _ = torch.cuda.mem_get_info()
    cycle(60)
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"leaked {(free0 - free1) >> 20} MiB over 60 create/destroy cycles"


# ---- full-size properties (BASELINE.json config #2 shape) -----------------------------------

@pytest.fixture(scope="module")
def gowalla(rbg, cuda):
    uid, iid, nu, ni = rbg.synth.make("gowalla")
    h = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda)
    return uid, iid, nu, ni, h


def test_full_size_vs_c_oracle(rbg, cuda, gowalla):
    uid, iid, nu, ni, h = gowalla
    rowptr, col, val = C.build_norm_csr(uid, iid, nu, ni)
    got = h.export_csr()  # built by the device builder
    assert np.array_equal(got[0], rowptr) and np.array_equal(got[1], col) and np.array_equal(got[2], val)
    hh = rbg.GraphHandle.from_interactions(uid, iid, nu, ni, device=cuda, flags=rbg._lib.GRAPH_BUILD_ON_HOST)
    for a, b in zip(hh.export_csr(), got):
        assert np.array_equal(a, b)
    x = randn((nu + ni, 64), 51, cuda)
    close(rbg.ops.spmm_raw(h, x), C.spmm(rowptr, col, val, x.cpu().numpy()))
    gen = torch.Generator().manual_seed(2020)
    uw, iw = O.xavier_uniform(nu, 64, gen), O.xavier_uniform(ni, 64, gen)
    mean, _ = rbg.ops.lightgcn_forward_raw(h, uw.to(cuda), iw.to(cuda), 3)
    ref = C.lightgcn_forward(rowptr, col, val, uw.numpy(), iw.numpy(), 3)
    err = np.abs(mean.cpu().numpy() - ref).max()
    assert err <= 1e-5 and err <= 1e-5 * np.abs(ref).max()  # absolute AND normalized


def test_full_size_properties(rbg, cuda, gowalla):
    uid, iid, nu, ni, h = gowalla
    n = nu + ni
    deg = np.bincount(np.concatenate([uid, iid + nu]), minlength=n).astype(np.float64)
    # Â (D^1/2 1) = D^1/2 1 on non-isolated nodes, 0 on isolated ones
    s = torch.from_numpy(np.sqrt(deg)).float().to(cuda)
    x = s[:, None].repeat(1, 64).contiguous()
    y = rbg.ops.spmm_raw(h, x)
    close(y / s.clamp(min=1)[:, None], (x > 0).float() * 1.0, tol=2e-5)
    # linearity and determinism
    a, b = randn((n, 64), 61, cuda), randn((n, 64), 62, cuda)
    ya, yb = rbg.ops.spmm_raw(h, a), rbg.ops.spmm_raw(h, b)
    close(rbg.ops.spmm_raw(h, 2 * a - b), 2 * ya - yb, tol=2e-5)
    assert torch.equal(rbg.ops.spmm_raw(h, a), ya)
    # symmetry: <Âa, b> = <a, Âb>
    lhs, rhs = (ya.double() * b.double()).sum(), (a.double() * yb.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * max(1.0, abs(float(lhs)))
    # bipartite: user outputs depend only on item inputs
    a2 = a.clone()
    a2[:nu] += 1.0
    assert torch.equal(rbg.ops.spmm_raw(h, a2)[:nu], ya[:nu])


def test_sgl_views_on_concurrent_streams(rbg, cuda):
    """SGL's three propagations of a step (sgl.py:219-221) issued on three HIP streams (ops.lightgcn_forward_views) give the
    propagations of the three sequential calls bit for bit, the same loss and gradients — eagerly and replayed from a HIP graph
    (GraphedStep captures the fork / join)."""
    uid, iid, nu, ni = rbg.synth.make("ml-100k")
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    cfg = {"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 3, "type": "ED", "drop_ratio": 0.1,
           "ssl_tau": 0.5, "ssl_weight": 0.05, "reg_weight": 1e-4, "require_pow": True}
    torch.manual_seed(3)
    model = rbg.SGL(cfg, ds)
    model.train()
    gen = torch.Generator().manual_seed(1)
    batch = {"user_id": torch.randint(1, nu, (256,), generator=gen).to(cuda), "item_id": torch.randint(1, ni, (256,), generator=gen).to(cuda),
             "neg_item_id": torch.randint(1, ni, (256,), generator=gen).to(cuda)}
    res = {}
    for conc in (True, False):
        model.concurrent_views = conc
        model.zero_grad(set_to_none=True)
        loss = model.calculate_loss(batch)
        loss.backward()
        torch.cuda.synchronize()
        res[conc] = (loss.detach().clone(), model.user_embedding.weight.grad.clone(), model.item_embedding.weight.grad.clone())
    # (the batch-row scatters of the loss use float atomics: their order, not the streams, decides the last bits)
    assert abs(float(res[True][0]) - float(res[False][0])) <= 1e-6 * abs(float(res[False][0]))
    close(res[True][1], res[False][1], tol=1e-6)
    close(res[True][2], res[False][2], tol=1e-6)
    assert float(res[True][1].abs().max()) > 0
    views = model.propagate_views()  # (sequential: the loop above ended with concurrent_views = False, the default)
    model.concurrent_views = True
    for (ua, ia), (ub, ib) in zip(model.propagate_views(), views):  # the propagations themselves: bit for bit
        assert torch.equal(ua, ub) and torch.equal(ia, ib)
    # the captured step: same parameters after two replays with and without the concurrent issue
    finals = {}
    for conc in (True, False):
        torch.manual_seed(3)
        m = rbg.SGL(cfg, ds)
        m.sub_graph1, m.sub_graph2 = model.sub_graph1, model.sub_graph2  # the same views
        m.train()
        m.concurrent_views = conc
        step = rbg.GraphedStep(m, batch, lr=1e-3)
        for _ in range(2):
            step.step(batch)
        torch.cuda.synchronize()
        finals[conc] = m.user_embedding.weight.detach().clone()
    close(finals[True], finals[False], tol=1e-4)  # (Adam on ~0 gradients: see test_graphed_step above)
