"""train.FusedNGCFAdam / train.FusedSGLAdam: the training steps of NGCF (ngcf.py:106-126) and SGL (sgl.py:211-233) + backward +
Adam as library calls, against the autograd path
of the model mirror (whose forward / loss / gradients test_gpu_parity.py checks against the reference formulas) on the same
parameters, batches and dropout draws; and rbg_concat_bpr_{begin,scatter}_f32 alone against torch in float64."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(rbg, cuda, golden, **cfg):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    torch.manual_seed(4)
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 3, "hidden_size_list": [64, 32, 64], "message_dropout": 0.0,
              "node_dropout": 0.0, "reg_weight": 1e-3}
    config.update(cfg)
    return rbg.NGCF(config, ds)


def _batches(golden, cuda, n, b, seed=9):
    nu, ni = int(golden["n_users"]), int(golden["n_items"])
    gen = torch.Generator().manual_seed(seed)
    return [{"user_id": torch.randint(1, nu, (b,), generator=gen).to(cuda), "item_id": torch.randint(1, ni, (b,), generator=gen).to(cuda),
             "neg_item_id": torch.randint(1, ni, (b,), generator=gen).to(cuda)} for _ in range(n)]


@pytest.mark.parametrize("p_drop,node_drop", [(0.0, 0.0), (0.1, 0.0), (0.0, 0.2)])
def test_fused_step_takes_the_autograd_step(rbg, cuda, golden, p_drop, node_drop):
    model = _model(rbg, cuda, golden, message_dropout=p_drop, node_dropout=node_drop)
    twin = _model(rbg, cuda, golden, message_dropout=p_drop, node_dropout=node_drop)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    stepper = rbg.FusedNGCFAdam(model, lr=1e-3)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    for step_no, batch in enumerate(_batches(golden, cuda, 3, 96)):
        torch.manual_seed(100 + step_no)  # the same edge / message dropout draws on both sides
        lf = float(stepper.step(batch))
        torch.manual_seed(100 + step_no)
        opt.zero_grad(set_to_none=True)
        le = twin.calculate_loss(batch)
        le.backward()
        if step_no == 0:  # same parameters: same loss, same gradients
            assert abs(lf - float(le.detach())) <= 2e-6 * max(1.0, abs(float(le.detach())))
            for (name, pf), pe in zip(model.named_parameters(), twin.parameters()):
                scale = max(float(pe.grad.abs().max()), 1e-12)
                assert float((pf.grad - pe.grad).abs().max()) <= 2e-5 * scale, name
        opt.step()
        assert abs(lf - float(le.detach())) <= 2e-4 * max(1.0, abs(float(le.detach())))
    # (Adam divides by |g|: where a gradient is ~0 the order of the float atomics decides the sign of a step of size lr)
    for pf, pe in zip(model.parameters(), twin.parameters()):
        assert float((pf.detach() - pe.detach()).abs().max()) <= 1e-4 * max(1.0, float(pe.detach().abs().max()))


def test_fused_step_replayed_from_a_hip_graph(rbg, cuda, golden):
    model, twin = _model(rbg, cuda, golden, message_dropout=0.1), _model(rbg, cuda, golden, message_dropout=0.1)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    a, b = rbg.FusedNGCFAdam(model, lr=1e-3, graphed=True), rbg.FusedNGCFAdam(twin, lr=1e-3)
    losses = []
    for batch in _batches(golden, cuda, 6, 64):
        la, lb = float(a.step(batch)), float(b.step(batch))  # (different dropout draws: the trajectories agree statistically)
        losses.append((la, lb))
    assert a._graph is not None
    assert all(np.isfinite(x) and np.isfinite(y) and abs(x - y) < 0.25 for x, y in losses), losses
    short = {k: v[:10] for k, v in _batches(golden, cuda, 1, 64)[0].items()}  # an epoch's last batch: enqueued, not replayed
    assert np.isfinite(float(a.step(short))) and np.isfinite(float(b.step(short)))
    assert np.isfinite(float(a.step(_batches(golden, cuda, 1, 64)[0])))  # ... and the captured size still replays
    # without dropout a replayed step IS the eager step
    model, twin = _model(rbg, cuda, golden), _model(rbg, cuda, golden)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    a, b = rbg.FusedNGCFAdam(model, lr=1e-3, graphed=True), rbg.FusedNGCFAdam(twin, lr=1e-3)
    for n, batch in enumerate(_batches(golden, cuda, 6, 64)):
        if n == 4:
            batch = {k: v[:23] for k, v in batch.items()}  # a shorter batch between replays shares the optimizer state
        la, lb = float(a.step(batch)), float(b.step(batch))
        assert abs(la - lb) <= 2e-4 * max(1.0, abs(lb))
    for pa, pb in zip(model.parameters(), twin.parameters()):
        assert float((pa.detach() - pb.detach()).abs().max()) <= 1e-4 * max(1.0, float(pb.detach().abs().max()))
    with torch.no_grad():  # the model's own parameters were trained: evaluation sees them
        s = model.full_sort_predict({"user_id": torch.arange(1, 5, device=cuda)})
    assert bool(torch.isfinite(s).all())


def test_fused_step_refuses_what_it_does_not_compute(rbg, cuda, golden):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    light = rbg.LightGCN({"device": str(cuda), "embedding_size": 64, "n_layers": 2, "enable_sparse": True}, ds)
    with pytest.raises(TypeError):
        rbg.FusedNGCFAdam(light)
    with pytest.raises(TypeError):
        rbg.FusedNGCFAdam(_model(rbg, cuda, golden, fused_forward=False))


# ---- SGL ---------------------------------------------------------------------------------------------------------------------

def _sgl(rbg, cuda, golden, aug="ED", **cfg):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    torch.manual_seed(4)
    np.random.seed(3)
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 3, "enable_sparse": True, "type": aug, "drop_ratio": 0.1, "ssl_tau": 0.5,
              "ssl_weight": 0.05, "reg_weight": 1e-3, "device_sampling": False}
    config.update(cfg)
    return rbg.SGL(config, ds)


@pytest.mark.parametrize("aug", ["ED", "ND", "RW"])
def test_fused_sgl_step_takes_the_autograd_step(rbg, cuda, golden, aug):
    model, twin = _sgl(rbg, cuda, golden, aug), _sgl(rbg, cuda, golden, aug)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2  # both train on the very same views
    stepper = rbg.FusedSGLAdam(model, lr=1e-3)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    for step_no, batch in enumerate(_batches(golden, cuda, 3, 96)):
        lf = float(stepper.step(batch))
        opt.zero_grad(set_to_none=True)
        le = twin.calculate_loss(batch)
        le.backward()
        ref = float(le.detach())
        if step_no == 0:
            assert abs(lf - ref) <= 5e-6 * max(1.0, abs(ref))
            for (name, pf), pe in zip(model.named_parameters(), twin.parameters()):
                scale = max(float(pe.grad.abs().max()), 1e-12)
                assert float((pf.grad - pe.grad).abs().max()) <= 2e-5 * scale, name
        opt.step()
        assert abs(lf - ref) <= 2e-4 * max(1.0, abs(ref))
    for pf, pe in zip(model.parameters(), twin.parameters()):
        assert float((pf.detach() - pe.detach()).abs().max()) <= 1e-4 * max(1.0, float(pe.detach().abs().max()))


def test_fused_sgl_step_replayed_and_recaptured(rbg, cuda, golden):
    model, twin = _sgl(rbg, cuda, golden), _sgl(rbg, cuda, golden)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2
    a, b = rbg.FusedSGLAdam(model, lr=1e-3, graphed=True), rbg.FusedSGLAdam(twin, lr=1e-3)
    batches = _batches(golden, cuda, 9, 64)
    for batch in batches[:5]:
        la, lb = float(a.step(batch)), float(b.step(batch))
        assert abs(la - lb) <= 2e-4 * max(1.0, abs(lb))
    first = a._graph
    assert first is not None
    model.train()  # a new epoch samples new views (sgl.py:94-98): the replay must not keep reading the old handles
    twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2
    for batch in batches[5:]:
        la, lb = float(a.step(batch)), float(b.step(batch))
        assert abs(la - lb) <= 2e-4 * max(1.0, abs(lb))
    assert a._graph is not None and a._graph is not first
    for pa, pb in zip(model.parameters(), twin.parameters()):
        assert float((pa.detach() - pb.detach()).abs().max()) <= 1e-4 * max(1.0, float(pb.detach().abs().max()))
    with pytest.raises(TypeError):
        rbg.FusedSGLAdam(_model(rbg, cuda, golden))


@pytest.mark.parametrize("n,b,d", [(40_982, 2048, 64), (1500, 257, 64), (3000, 96, 128), (700, 33, 20)])
def test_infonce_one_pass_form(rbg, cuda, scatter_mode, n, b, d):
    """rbg_infonce_f32 with gradients (option "lse_onepass", default): denominators and the batch rows' gradient out of one
    pass over the table — against the three-launch form and against sgl.py:191-199 in float64."""
    gen = torch.Generator().manual_seed(n)
    t1, t2 = torch.randn(n, d, generator=gen).to(cuda), torch.randn(n, d, generator=gen).to(cuda)
    idx = torch.randint(0, n, (b,), generator=gen).to(cuda)
    tau = 0.2
    res = {}
    for mode in (1, 0):
        rbg.set_option("lse_onepass", mode)
        try:
            a, c = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
            loss = rbg.ops.info_nce(a, c, idx, tau)
            loss.backward()
            res[mode] = (float(loss.detach()), a.grad.clone(), c.grad.clone())
        finally:
            rbg.set_option("lse_onepass", 1)
    a, c = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
    na, nc = torch.nn.functional.normalize(a[idx], dim=1), torch.nn.functional.normalize(c, dim=1)
    ref = -torch.log(torch.exp((na * nc[idx]).sum(1) / tau) / torch.exp(na @ nc.T / tau).sum(1)).sum()
    ref.backward()
    for mode in (1, 0):
        loss, g1, g2 = res[mode]
        assert abs(loss - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
        for got, want in ((g1, a.grad), (g2, c.grad)):
            assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert abs(res[1][0] - res[0][0]) <= 1e-6 * abs(res[0][0])
    for x, y in zip(res[1][1:], res[0][1:]):
        assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max())


def test_adam_with_the_step_count_on_the_device(rbg, cuda):
    """rbg_adam_step_dev_f32 == torch.optim.Adam over the two tables, also when ONE captured call is replayed: the bias
    corrections advance with the device-side count (a host-side count would repeat the captured step's)."""
    c_vp, check, lib = rbg._lib.c_vp, rbg._lib.check, rbg._lib.lib
    nu, ni, d = 37, 53, 64
    gen = torch.Generator().manual_seed(0)
    uw, iw = torch.randn(nu, d, generator=gen).to(cuda), torch.randn(ni, d, generator=gen).to(cuda)
    ref = [uw.clone().requires_grad_(True), iw.clone().requires_grad_(True)]
    opt = torch.optim.Adam(ref, lr=1e-2)
    grad = torch.empty(nu + ni, d, device=cuda)
    m, v = torch.zeros_like(grad), torch.zeros_like(grad)
    step, fac = torch.zeros((), dtype=torch.int64, device=cuda), torch.zeros(2, device=cuda)

    def call():
        check(lib.rbg_adam_step_dev_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), nu, ni, d, c_vp(grad.data_ptr()), c_vp(m.data_ptr()),
                                        c_vp(v.data_ptr()), c_vp(step.data_ptr()), c_vp(fac.data_ptr()), 1e-2, 0.9, 0.999, 1e-8,
                                        c_vp(torch.cuda.current_stream(cuda).cuda_stream)))

    graph = None
    for it in range(6):
        g = torch.randn(nu + ni, d, generator=gen).to(cuda)
        grad.copy_(g)
        ref[0].grad, ref[1].grad = g[:nu].clone(), g[nu:].clone()
        opt.step()
        if it < 2:
            call()
        else:
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream(device=cuda)
                side.wait_stream(torch.cuda.current_stream(cuda))
                with torch.cuda.stream(side):
                    pass
                with torch.cuda.graph(graph):
                    call()
            graph.replay()
        torch.cuda.synchronize()
        assert int(step) == it + 1
        assert float((uw - ref[0].detach()).abs().max()) <= 2e-6 and float((iw - ref[1].detach()).abs().max()) <= 2e-6
    assert lib.rbg_adam_step_dev_f32(c_vp(uw.data_ptr()), c_vp(iw.data_ptr()), nu, ni, 6, c_vp(grad.data_ptr()), c_vp(m.data_ptr()),
                                     c_vp(v.data_ptr()), c_vp(step.data_ptr()), c_vp(fac.data_ptr()), 1e-2, 0.9, 0.999, 1e-8, None) != 0


@pytest.mark.parametrize("name", ["NGCF", "SGL"])
def test_many_replays_of_a_captured_step(rbg, cuda, golden, name):
    """80 steps of the captured step against the eagerly enqueued one on a twin (the hipMemsetAsync hazard showed at the 51st
    replay): finite, and the two trajectories stay together."""
    mk = (lambda: _model(rbg, cuda, golden)) if name == "NGCF" else (lambda: _sgl(rbg, cuda, golden))
    model, twin = mk(), mk()
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    if name == "SGL":
        twin.sub_graph1, twin.sub_graph2 = model.sub_graph1, model.sub_graph2
    cls = rbg.FusedNGCFAdam if name == "NGCF" else rbg.FusedSGLAdam
    a, b = cls(model, lr=1e-3, graphed=True), cls(twin, lr=1e-3)
    batches = _batches(golden, cuda, 8, 64)
    for n in range(80):
        la, lb = float(a.step(batches[n % 8])), float(b.step(batches[n % 8]))
        assert np.isfinite(la) and np.isfinite(lb)
        assert abs(la - lb) <= 1e-2 * max(1.0, abs(lb)), (n, la, lb)  # (float atomics: the order differs run to run)
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())


def test_zero_fill_kernel_alignment_and_bounds(rbg, cuda):
    """The library zeroes through a fill kernel (csrc/train.hip zero_async — a captured hipMemsetAsync node writes garbage on later
    replays): any 4-byte alignment, any word count, nothing outside the range — and the same values on every replay of a graph."""
    c_vp, check, lib = rbg._lib.c_vp, rbg._lib.check, rbg._lib.lib
    tab = torch.zeros(4, 64, device=cuda)
    ptrs, wid = (c_vp * 1)(tab.data_ptr()), (ctypes.c_int * 1)(64)
    idx = torch.zeros(1, dtype=torch.int64, device=cuda)
    st = lambda: c_vp(torch.cuda.current_stream(cuda).cuda_stream)  # noqa: E731
    buf = torch.empty(64, device=cuda)
    for off in range(5):  # rbg_concat_bpr_begin_f32 with B = 0: zero sums[3] and loss[1], nothing else
        buf.fill_(7.0)
        check(lib.rbg_concat_bpr_begin_f32(ptrs, wid, 1, 2, 2, c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), 0, 0,
                                           c_vp(buf.data_ptr()), c_vp(buf[off:].data_ptr()), c_vp(buf[40 + off:].data_ptr()), st()))
        want = torch.full((64,), 7.0)
        want[off:off + 3] = 0
        want[40 + off] = 0
        assert torch.equal(buf.cpu(), want), off
    for (nu, ni, d, off) in ((3, 4, 5, 1), (100, 57, 64, 3), (1, 1, 1, 2), (1000, 999, 33, 0)):  # rbg_bpr_grad_f32 with B = 0: grad_mean [N, d]
        big = torch.full(((nu + ni) * d + 16,), 7.0, device=cuda)
        loss = torch.full((1,), 7.0, device=cuda)
        check(lib.rbg_bpr_grad_f32(c_vp(big.data_ptr()), nu, ni, c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), 0, d,
                                   c_vp(big[off:].data_ptr()), c_vp(loss.data_ptr()), st()))
        want = torch.full_like(big, 7.0).cpu()
        want[off:off + (nu + ni) * d] = 0
        assert torch.equal(big.cpu(), want) and float(loss) == 0.0, (nu, ni, d, off)
    sums, loss = torch.ones(3, device=cuda), torch.ones((), device=cuda)
    side = torch.cuda.Stream(device=cuda)
    with torch.cuda.stream(side):
        pass
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        check(lib.rbg_concat_bpr_begin_f32(ptrs, wid, 1, 2, 2, c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), c_vp(idx.data_ptr()), 0, 0,
                                           c_vp(buf.data_ptr()), c_vp(sums.data_ptr()), c_vp(loss.data_ptr()), st()))
    for _ in range(60):
        sums.fill_(1.0), loss.fill_(1.0)
        g.replay()
        assert float(sums.abs().sum()) == 0.0 and float(loss) == 0.0


@pytest.mark.parametrize("n,b,d", [(2048, 2048, 64), (300, 300, 64), (1000, 77, 128), (50, 200, 20)])
def test_infonce_with_row_and_column_weights(rbg, cuda, scatter_mode, n, b, d):
    """rbg_infonce_masked_f32: weight * sum_b row_w[b] (log sum_j col_w[j] exp(<a_b, c_j> / tau) - <a_b, p_b> / tau) and its
    gradients against float64 — with 0 / 1 masks (SimGCL's contrast over the distinct ids of a batch) and with real weights;
    value-only calls; rows whose own column is masked."""
    gen = torch.Generator().manual_seed(n + b)
    t1, t2 = torch.randn(n, d, generator=gen).to(cuda), torch.randn(n, d, generator=gen).to(cuda)
    idx = torch.randint(0, n, (b,), generator=gen).to(cuda)
    tau = 0.2
    for kind in ("mask", "weights", "rows_only", "cols_only"):
        row_w = (torch.rand(b, generator=gen) < 0.6).float() if kind == "mask" else torch.rand(b, generator=gen)
        col_w = (torch.rand(n, generator=gen) < 0.7).float() if kind == "mask" else torch.rand(n, generator=gen) + 0.1
        col_w[0] = 1.0  # (never an empty candidate set)
        rw = None if kind == "cols_only" else row_w.to(cuda)
        cw = None if kind == "rows_only" else col_w.to(cuda)
        a, c = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
        loss = rbg.ops.info_nce(a, c, idx, tau, row_w=rw, col_w=cw)
        loss.backward()
        with torch.no_grad():
            value_only = rbg.ops.info_nce(t1, t2, idx, tau, row_w=rw, col_w=cw)
        a64, c64 = t1.double().requires_grad_(True), t2.double().requires_grad_(True)
        na, nc = torch.nn.functional.normalize(a64[idx], dim=1), torch.nn.functional.normalize(c64, dim=1)
        r64 = torch.ones(b, dtype=torch.float64, device=cuda) if rw is None else rw.double()
        w64 = torch.ones(n, dtype=torch.float64, device=cuda) if cw is None else cw.double()
        ref = (r64 * (torch.log((torch.exp(na @ nc.T / tau) * w64[None, :]).sum(1)) - (na * nc[idx]).sum(1) / tau)).sum()
        ref.backward()
        scale = max(1.0, abs(float(ref)))
        assert abs(float(loss.detach()) - float(ref)) <= 1e-5 * scale, kind
        assert abs(float(value_only) - float(ref)) <= 1e-5 * scale, kind
        for got, want in ((a.grad, a64.grad), (c.grad, c64.grad)):
            assert float((got.double() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1e-12), kind
    with pytest.raises(ValueError):
        rbg.ops.info_nce(t1, t2, idx, tau, row_w=torch.ones(b + 1, device=cuda))


def test_a_graph_handle_dying_inside_a_capture_does_not_invalidate_it(rbg, cuda, golden):
    """rbg_graph_destroy frees HBM; Python's cyclic collector can run it at any moment — also inside somebody's stream capture
    (handles do sit in cycles: NGCF's two edge-dropout views point at each other).  A handle that dies there is parked and
    destroyed at the next destroy / create outside a capture."""
    import gc
    from recbole_gnn_amd import graph as G
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    x = torch.ones(16, device=cuda)
    gc.collect()  # (earlier tests' dead cycles must not die inside THIS capture and be counted below)
    G.flush_parked()
    assert len(G._PARKED) == 0
    a = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    b = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    a._transpose, b._transpose = b, a  # a cycle: only the collector frees the pair
    del a, b
    gc.disable()
    try:
        side = torch.cuda.Stream(device=cuda)
        with torch.cuda.stream(side):
            pass
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            y = x * 2
            assert gc.collect() >= 2  # the two handles die here
            assert len(G._PARKED) == 2
            z = y + 1
        gr.replay()
        torch.cuda.synchronize()
        assert float(z.sum()) == 48.0
    finally:
        gc.enable()
    h = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)  # the next create flushes
    assert len(G._PARKED) == 0 and h.nnz == 2 * len(g["uid"])


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("require_pow", [False, True])
@pytest.mark.parametrize("widths", [[64], [64, 32, 16, 128], [8, 100]])
def test_concat_bpr_against_torch(rbg, cuda, scatter_mode, widths, require_pow, form):
    """BPRLoss + reg_weight * EmbLoss on the rows of cat(tables) (ngcf.py:113-126) and their gradient w.r.t. every table."""
    c_vp, check, lib = rbg._lib.c_vp, rbg._lib.check, rbg._lib.lib
    nu, ni, b, reg = 50, 70, 333, 0.37
    gen = torch.Generator().manual_seed(1)
    tabs = [torch.randn(nu + ni, w, generator=gen).to(cuda) for w in widths]
    user = torch.randint(0, nu, (b,), generator=gen).to(cuda)
    pos, neg = torch.randint(0, ni, (b,), generator=gen).to(cuda), torch.randint(0, ni, (b,), generator=gen).to(cuda)
    # float64 reference of ngcf.py:113-126 (recbole BPRLoss / EmbLoss)
    t64 = [t.double().requires_grad_(True) for t in tabs]
    allc = torch.cat(t64, dim=1)
    u, p, n = allc[user], allc[nu + pos], allc[nu + neg]
    x = (u * p).sum(1) - (u * n).sum(1)
    bpr = -torch.log(1e-10 + torch.sigmoid(x)).mean() if form == 0 else -torch.nn.functional.logsigmoid(x).sum()  # (sgl.py:147-162)
    emb = sum(torch.norm(x, p=2).pow(2) for x in (u, p, n)) / b / 2 if require_pow else sum(torch.norm(x, p=2) for x in (u, p, n)) / b
    loss_ref = bpr + reg * emb
    loss_ref.backward()
    coef, sums, loss = torch.empty(b, device=cuda), torch.empty(3, device=cuda), torch.empty((), device=cuda)
    ptrs = (c_vp * len(tabs))(*[t.data_ptr() for t in tabs])
    wid = (ctypes.c_int * len(tabs))(*widths)
    st = c_vp(torch.cuda.current_stream(cuda).cuda_stream)
    check(lib.rbg_concat_bpr_begin_f32(ptrs, wid, len(tabs), nu, ni, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, form,
                                       c_vp(coef.data_ptr()), c_vp(sums.data_ptr()), c_vp(loss.data_ptr()), st))
    for i, (t, w) in enumerate(zip(tabs, widths)):
        base = torch.randn(nu + ni, w, generator=gen).to(cuda)  # "what the layer above wrote": the scatter adds onto it
        grad = base.clone()
        check(lib.rbg_concat_bpr_scatter_f32(c_vp(t.data_ptr()), w, nu, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, reg,
                                             int(require_pow), c_vp(coef.data_ptr()), c_vp(sums.data_ptr()), c_vp(grad.data_ptr()),
                                             c_vp(loss.data_ptr()) if i == 1 % len(tabs) else None, st))
        got = (grad - base).double().cpu()
        ref = t64[i].grad.cpu()
        assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref)))
    # argument checks
    assert lib.rbg_concat_bpr_begin_f32(ptrs, wid, 9, nu, ni, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, 0,
                                        c_vp(coef.data_ptr()), c_vp(sums.data_ptr()), c_vp(loss.data_ptr()), st) != 0
    assert lib.rbg_concat_bpr_begin_f32(ptrs, wid, len(tabs), nu, ni, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, 2,
                                        c_vp(coef.data_ptr()), c_vp(sums.data_ptr()), c_vp(loss.data_ptr()), st) != 0
    assert lib.rbg_concat_bpr_scatter_f32(None, 64, nu, c_vp(user.data_ptr()), c_vp(pos.data_ptr()), c_vp(neg.data_ptr()), b, reg, 0,
                                          c_vp(coef.data_ptr()), c_vp(sums.data_ptr()), c_vp(coef.data_ptr()), None, st) != 0


def test_a_graph_handle_dying_while_another_stream_captures(rbg, cuda, golden):
    """the capture is open on ANOTHER stream than the one the dying handle's thread is on (torch.cuda.graph
    captures in the global mode: every "unsafe" call of the process — hipFree — would invalidate it).  Parking cannot see that
    capture (torch only answers for the current stream); rbg_graph_destroy itself relaxes the thread's capture mode around its
    frees.  The handle is destroyed at once, the capture survives and replays."""
    from recbole_gnn_amd import graph as G
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    x = torch.ones(16, device=cuda)
    a = rbg.GraphHandle.from_interactions(g["uid"], g["iid"], nu, ni, device=cuda)
    other = torch.cuda.Stream(device=cuda)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = x * 2
        with torch.cuda.stream(other):  # the current stream is not the capturing one
            assert not torch.cuda.is_current_stream_capturing()
            a.destroy()  # hipFree x N, now
            assert len(G._PARKED) == 0
        z = y + 1
    gr.replay()
    torch.cuda.synchronize()
    assert float(z.sum()) == 48.0


# ---- SimGCL / XSimGCL (r05) --------------------------------------------------------------------------------------------------

def _contrastive(rbg, cuda, golden, kind, **cfg):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    torch.manual_seed(4)
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 3, "enable_sparse": True, "reg_weight": 1e-3, "lambda": 0.3,
              "eps": 0.15, "temperature": 0.2}
    config.update(cfg)
    return getattr(rbg, kind)(config, ds)


@pytest.mark.parametrize("kind,cfg", [("SimGCL", {}), ("SimGCL", {"n_layers": 1, "require_pow": True}), ("XSimGCL", {"layer_cl": 1}),
                                      ("XSimGCL", {"layer_cl": 3}), ("XSimGCL", {"layer_cl": 0, "n_layers": 2})])
def test_fused_contrastive_step_takes_the_autograd_step(rbg, cuda, golden, kind, cfg):
    """train.FusedSimGCLAdam / FusedXSimGCLAdam (simgcl.py:45-61, xsimgcl.py:56-90 + backward + Adam as library calls: the
    perturbed passes by rbg_spmm_noise_f32, the contrasts by rbg_infonce_masked_f32 on the batch's rows, ONE backward chain for
    all passes) against the model mirror's autograd step on the same parameters, batches (with repeated ids) and noise draws."""
    model, twin = _contrastive(rbg, cuda, golden, kind, **cfg), _contrastive(rbg, cuda, golden, kind, **cfg)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    stepper = rbg.fused_stepper(model, lr=1e-3, graphed=False)
    assert type(stepper).__name__ == f"Fused{kind}Adam"
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    total = lambda l: sum(l) if isinstance(l, tuple) else l  # noqa: E731  (XSimGCL returns its three terms, xsimgcl.py:90)
    for step_no, batch in enumerate(_batches(golden, cuda, 3, 96)):
        batch["user_id"][:7] = batch["user_id"][7:14]  # repeated ids: the one-occurrence mask matters
        batch["item_id"][:5] = batch["item_id"][5:10]
        torch.manual_seed(200 + step_no)  # the same noise draws on both sides (simgcl.py:31)
        lf = float(stepper.step(batch))
        torch.manual_seed(200 + step_no)
        opt.zero_grad(set_to_none=True)
        le = total(twin.calculate_loss(batch))
        le.backward()
        ref = float(le.detach())
        if step_no == 0:
            assert abs(lf - ref) <= 5e-6 * max(1.0, abs(ref)), (lf, ref)
            for (name, pf), pe in zip(model.named_parameters(), twin.parameters()):
                scale = max(float(pe.grad.abs().max()), 1e-12)
                assert float((pf.grad - pe.grad).abs().max()) <= 2e-5 * scale, name
        opt.step()
        assert abs(lf - ref) <= 2e-4 * max(1.0, abs(ref)), (step_no, lf, ref)
    for pf, pe in zip(model.parameters(), twin.parameters()):
        assert float((pf.detach() - pe.detach()).abs().max()) <= 1e-4 * max(1.0, float(pe.detach().abs().max()))


@pytest.mark.parametrize("kind", ["SimGCL", "XSimGCL"])
def test_fused_contrastive_step_replayed_from_a_hip_graph(rbg, cuda, golden, kind):
    """graphed=True: the step is captured on its third full batch and replayed; the noise advances with every replay (the
    generator's offset is part of the captured graph), a short batch runs eagerly and shares the optimizer state."""
    model = _contrastive(rbg, cuda, golden, kind)
    model.train()
    a = rbg.fused_stepper(model, lr=1e-3, graphed=True)
    losses = []
    for n, batch in enumerate(_batches(golden, cuda, 8, 64)):
        if n == 5:
            batch = {k: v[:23] for k, v in batch.items()}
        losses.append(float(a.step(batch)))
    assert a._graph is not None and all(np.isfinite(x) for x in losses)
    same = [float(a.step(_batches(golden, cuda, 1, 64)[0])) for _ in range(2)]
    assert same[0] != same[1]  # (new noise and new parameters every replay)
    with pytest.raises(TypeError):
        rbg.FusedSimGCLAdam(_contrastive(rbg, cuda, golden, "XSimGCL"))
    with pytest.raises(TypeError):
        rbg.FusedXSimGCLAdam(_contrastive(rbg, cuda, golden, "SimGCL"))
    assert rbg.fused_stepper(_contrastive(rbg, cuda, golden, "SimGCL", static_unique=False)) is None


@pytest.mark.parametrize("cfg,with_proto", [({}, True), ({}, False), ({"n_layers": 2, "hyper_layers": 2}, True), ({"n_layers": 3, "hyper_layers": 1, "alpha": 1.5}, True)])
def test_fused_ncl_step_takes_the_autograd_step(rbg, cuda, golden, cfg, with_proto):
    """train.FusedNCLAdam (ncl.py:167-199 + trainer.py:130-133): one chain call for the L layers, rbg_infonce_f32 for the
    structure contrast, one Horner chain for the mean's and the context layer's gradients, the prototype term by autograd.grad on
    the batch's rows — against the model mirror's autograd step (the trainer's sum of the loss tuple, without the prototype term
    during warm-up) on the same parameters, prototypes and batches."""
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 3, "enable_sparse": True, "reg_weight": 1e-3, "ssl_reg": 1e-3, "proto_reg": 1e-3,
              "num_clusters": 7, "ssl_temp": 0.2}
    config.update(cfg)
    torch.manual_seed(4)
    model, twin = rbg.NCL(config, ds), rbg.NCL(config, ds)
    twin.load_state_dict(model.state_dict())
    model.train(), twin.train()
    model.e_step()
    for name in ("user_centroids", "user_2cluster", "item_centroids", "item_2cluster"):
        setattr(twin, name, getattr(model, name).clone())
    stepper = rbg.fused_stepper(model, lr=1e-3, graphed=False)
    assert isinstance(stepper, rbg.FusedNCLAdam)
    stepper.with_proto = with_proto
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    for step_no, batch in enumerate(_batches(golden, cuda, 3, 96)):
        lf = float(stepper.step(batch))
        opt.zero_grad(set_to_none=True)
        terms = twin.calculate_loss(batch)
        le = sum(terms if with_proto else terms[:-1])
        le.backward()
        ref = float(le.detach())
        if step_no == 0:
            assert abs(lf - ref) <= 5e-6 * max(1.0, abs(ref)), (lf, ref)
            for (name, pf), pe in zip(model.named_parameters(), twin.parameters()):
                scale = max(float(pe.grad.abs().max()), 1e-12)
                assert float((pf.grad - pe.grad).abs().max()) <= 2e-5 * scale, name
        opt.step()
        assert abs(lf - ref) <= 2e-4 * max(1.0, abs(ref)), (step_no, lf, ref)
    for pf, pe in zip(model.parameters(), twin.parameters()):
        assert float((pf.detach() - pe.detach()).abs().max()) <= 1e-4 * max(1.0, float(pe.detach().abs().max()))


def test_fused_ncl_step_replayed_and_recaptured_when_the_prototype_term_joins(rbg, cuda, golden):
    g = golden
    ds = rbg.InteractionDataset(g["uid"], g["iid"], int(g["n_users"]), int(g["n_items"]))
    torch.manual_seed(4)
    model = rbg.NCL({"device": str(cuda), "embedding_size": 64, "n_layers": 3, "enable_sparse": True, "num_clusters": 7}, ds)
    model.train()
    model.e_step()
    a = rbg.fused_stepper(model, lr=1e-3, graphed=True)
    a.with_proto = False
    batches = _batches(golden, cuda, 10, 64)
    for batch in batches[:5]:
        assert np.isfinite(float(a.step(batch)))
    first = a._graph
    assert first is not None
    model.e_step()  # new prototypes, same storage: the captured step reads them
    a.with_proto = True
    for batch in batches[5:]:
        assert np.isfinite(float(a.step(batch)))
    assert a._graph is not None and a._graph is not first


def test_once_mask_on_the_device_keeps_one_occurrence_per_id(rbg, cuda):
    """models._once_mask with an id bound on a GPU (scatter + compare instead of a sort): exactly one position per distinct id."""
    from recbole_gnn_amd import models
    gen = torch.Generator().manual_seed(3)
    for n_ids, b in ((50, 400), (30_000, 2048), (7, 7)):
        ids = torch.randint(0, n_ids, (b,), generator=gen).to(cuda)
        m = models._once_mask(ids, n_ids)
        kept = ids[m]
        assert m.dtype == torch.bool and kept.numel() == torch.unique(ids).numel() and torch.equal(torch.sort(kept).values, torch.unique(ids))


# ---- option "deterministic" (csrc/ordered.h): row scatters by owner wavefronts in batch order, sums in fixed point ---------------
@pytest.fixture(params=["atomic", "ordered"])
def scatter_mode(request, rbg):
    rbg.set_option("deterministic", 1 if request.param == "ordered" else 0)
    yield request.param
    rbg.set_option("deterministic", 0)


def _any_model(rbg, cuda, golden, name):
    ds = rbg.InteractionDataset(golden["uid"], golden["iid"], int(golden["n_users"]), int(golden["n_items"]))
    torch.manual_seed(4)
    config = {"device": str(cuda), "embedding_size": 64, "n_layers": 2, "enable_sparse": True, "reg_weight": 1e-3, "require_pow": name == "LightGCN",
              "hidden_size_list": [64, 32], "message_dropout": 0.1, "node_dropout": 0.0, "type": "ED", "drop_ratio": 0.1, "ssl_tau": 0.5,
              "ssl_weight": 0.05, "lambda": 0.3, "eps": 0.15, "temperature": 0.2, "layer_cl": 1}
    return getattr(rbg, name)(config, ds)


@pytest.mark.parametrize("name", ["LightGCN", "NGCF", "SGL", "SimGCL", "XSimGCL"])
def test_fused_steps_are_bit_stable_in_deterministic_mode(rbg, cuda, golden, name):
    """With option "deterministic" two runs of the autograd-free step — same parameters, same random draws, batches in which most ids
    repeat (the tables are a few dozen rows) — end in bit-identical losses and parameters; and they stay within the usual
    tolerance of the default (float-atomic) run."""
    def run():
        model = _any_model(rbg, cuda, golden, name)
        model.train()
        stepper = rbg.fused_stepper(model, lr=1e-2, graphed=False)
        assert stepper is not None
        losses = []
        for step_no, batch in enumerate(_batches(golden, cuda, 5, 256)):
            torch.manual_seed(300 + step_no)
            losses.append(float(stepper.step(batch)))
        torch.cuda.synchronize()
        return losses, [p.detach().clone() for p in model.parameters()]
    default = run()
    rbg.set_option("deterministic", 1)
    try:
        assert rbg.get_option("deterministic") == 1
        a, b = run(), run()
    finally:
        rbg.set_option("deterministic", 0)
    assert a[0] == b[0], (a[0], b[0])
    for pa, pb in zip(a[1], b[1]):
        assert torch.equal(pa, pb)
    for x, y in zip(a[0], default[0]):
        assert abs(x - y) <= 2e-4 * max(1.0, abs(y)), (a[0], default[0])


@pytest.mark.parametrize("require_pow", [True, False])
@pytest.mark.parametrize("b,d", [(2048, 64), (333, 20), (96, 128)])
def test_bpr_and_embloss_scatters_in_both_modes(rbg, cuda, scatter_mode, b, d, require_pow):
    """rbg_bpr_grad_f32 / rbg_emb_reg_grad[_nopow]_f32 on a batch whose ids repeat many times, against torch in float64 — with float
    atomics and with the ordered scatters."""
    lib, vp = rbg._lib.lib, ctypes.c_void_p
    nu, ni = 37, 53
    gen = torch.Generator().manual_seed(b + d)
    mean = torch.randn(nu + ni, d, generator=gen).to(cuda)
    user, pos, neg = (torch.randint(0, n, (b,), generator=gen).to(cuda) for n in (nu, ni, ni))
    st = vp(torch.cuda.current_stream().cuda_stream)
    grad, loss = torch.full((nu + ni, d), 7.0, device=cuda), torch.full((1,), 7.0, device=cuda)  # (the call zeroes both)
    rbg._lib.check(lib.rbg_bpr_grad_f32(vp(mean.data_ptr()), nu, ni, vp(user.data_ptr()), vp(pos.data_ptr()), vp(neg.data_ptr()), b, d,
                                        vp(grad.data_ptr()), vp(loss.data_ptr()), st))
    m64 = mean.double().requires_grad_(True)
    u, p, n = m64[user], m64[nu + pos], m64[nu + neg]
    ref = -torch.log(1e-10 + torch.sigmoid((u * p).sum(1) - (u * n).sum(1))).mean()
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert float((grad.double() - m64.grad).abs().max()) <= 2e-5 * max(float(m64.grad.abs().max()), 1e-12)
    # EmbLoss on the ego tables, added onto an existing gradient and loss
    uw, iw = mean[:nu].contiguous(), mean[nu:].contiguous()
    g0 = torch.randn(nu + ni, d, generator=gen).to(cuda)
    ge, le = g0.clone(), torch.full((1,), 0.25, device=cuda)
    ws = torch.zeros(4, device=cuda)
    args = (vp(uw.data_ptr()), vp(iw.data_ptr()), nu, vp(user.data_ptr()), vp(pos.data_ptr()), vp(neg.data_ptr()), b, d, ctypes.c_float(1e-2),
            vp(ge.data_ptr()), vp(le.data_ptr()))
    rbg._lib.check(lib.rbg_emb_reg_grad_f32(*args, st) if require_pow else lib.rbg_emb_reg_grad_nopow_f32(*args, vp(ws.data_ptr()), st))
    u64, i64_ = uw.double().requires_grad_(True), iw.double().requires_grad_(True)
    blocks = (u64[user], i64_[pos], i64_[neg])
    reg = sum(x.pow(2).sum() for x in blocks) / b / 2 if require_pow else sum(x.norm() for x in blocks) / b
    (1e-2 * reg).backward()
    assert abs(float(le) - 0.25 - 1e-2 * float(reg)) <= 2e-6 * max(1.0, abs(float(reg)))
    want = g0.double() + torch.cat([u64.grad, i64_.grad])
    assert float((ge.double() - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1e-12)  # (dozens of fp32 adds onto rows of size ~ 3)
    if scatter_mode == "ordered":  # and a second run gives the same bits
        ge2, le2 = g0.clone(), torch.full((1,), 0.25, device=cuda)
        args2 = args[:-2] + (vp(ge2.data_ptr()), vp(le2.data_ptr()))
        rbg._lib.check(lib.rbg_emb_reg_grad_f32(*args2, st) if require_pow else lib.rbg_emb_reg_grad_nopow_f32(*args2, vp(ws.data_ptr()), st))
        assert torch.equal(ge, ge2) and torch.equal(le, le2)


@pytest.mark.parametrize("b,n_ids", [(2048, 29859), (7, 5), (5000, 300), (1, 1)])
def test_once_mask_in_one_launch(rbg, cuda, b, n_ids):
    """rbg_once_mask_f32 (r06) == models._once_mask + the row weights of the masked InfoNCE: exactly one position of every distinct id
    carries 1; first_occurrence = 1 keeps the FIRST position (what option "deterministic" asks for); mean_form divides by the number
    of distinct ids; the slot scratch needs no reset between calls."""
    from recbole_gnn_amd._lib import lib, check, c_vp
    gen = torch.Generator().manual_seed(b + n_ids)
    st = c_vp(torch.cuda.current_stream(cuda).cuda_stream)
    slot = torch.full((n_ids,), -7, dtype=torch.int64, device=cuda)  # (garbage on purpose)
    for trial in range(3):
        ids = torch.randint(0, n_ids, (b,), generator=gen).to(cuda)
        distinct = int(torch.unique(ids).numel())
        for first in (0, 1):
            for mean_form in (0, 1):
                once, row_w = torch.empty(b, device=cuda), torch.empty(b, device=cuda)
                check(lib.rbg_once_mask_f32(c_vp(ids.data_ptr()), b, n_ids, c_vp(slot.data_ptr()), first, mean_form, c_vp(once.data_ptr()),
                                            c_vp(row_w.data_ptr()), st))
                assert set(once.unique().tolist()) <= {0.0, 1.0} and int(once.sum()) == distinct
                kept = ids[once > 0]
                assert kept.unique().numel() == distinct  # one position per distinct id
                if first:
                    pos = torch.nonzero(once > 0).flatten().cpu()
                    want = torch.tensor([int((ids.cpu() == v).nonzero()[0]) for v in ids.cpu()[pos]])
                    assert torch.equal(pos, want)
                ref_w = once / distinct if mean_form else once
                assert torch.equal(row_w, ref_w) or float((row_w - ref_w).abs().max()) <= 1e-9


@pytest.mark.parametrize("det", [0, 1])
@pytest.mark.parametrize("rows,b,d,mean_form", [(3000, 300, 64, False), (500, 700, 64, True), (2000, 257, 128, True), (64, 33, 36, False)])
def test_infonce_batch_form_equals_the_gathered_form(rbg, cuda, det, rows, b, d, mean_form):
    """rbg_infonce_batch_f32 (r06: the contrast among the batch's rows straight on the tables) == rbg_infonce_masked_f32 on the gathered
    rows followed by index_add_ of the row gradients (what train._contrast did): the same loss, the same gradient tables (ids repeat:
    the masked positions add zeros); also in "deterministic" mode (ordered row scatters)."""
    from recbole_gnn_amd._lib import lib, check, c_vp, c_i64
    import ctypes
    gen = torch.Generator().manual_seed(rows + b + d)
    ta, tb = torch.randn(rows, d, generator=gen).to(cuda), torch.randn(rows, d, generator=gen).to(cuda)
    ids = torch.randint(0, rows, (b,), generator=gen).to(cuda)
    st = c_vp(torch.cuda.current_stream(cuda).cuda_stream)
    p = lambda t: c_vp(t.data_ptr())  # noqa: E731
    old = rbg.get_option("deterministic")
    rbg.set_option("deterministic", det)
    try:
        slot = torch.empty(rows, dtype=torch.int64, device=cuda)
        once, row_w = torch.empty(b, device=cuda), torch.empty(b, device=cuda)
        check(lib.rbg_once_mask_f32(p(ids), b, rows, p(slot), 1, int(mean_form), p(once), p(row_w), st))
        nbytes = c_i64()
        check(lib.rbg_infonce_workspace(b, b, d, ctypes.byref(nbytes)))
        work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=cuda)
        # the gathered form
        xa, xb = ta.index_select(0, ids), tb.index_select(0, ids)
        gxa, gxb, loss0 = torch.zeros_like(xa), torch.zeros_like(xb), torch.zeros((), device=cuda)
        ar = torch.arange(b, device=cuda)
        check(lib.rbg_infonce_masked_f32(p(xa), p(xb), b, d, p(ar), b, 0.2, 0.37, p(row_w), p(once), p(loss0), p(gxa), p(gxb), p(work), st))
        ga0, gb0 = torch.zeros_like(ta).index_add_(0, ids, gxa), torch.zeros_like(tb).index_add_(0, ids, gxb)
        # the batch form
        ga1, gb1, loss1 = torch.zeros_like(ta), torch.zeros_like(tb), torch.zeros((), device=cuda)
        check(lib.rbg_infonce_batch_f32(p(ta), p(tb), d, p(ids), b, 0.2, 0.37, p(row_w), p(once), p(loss1), p(ga1), p(gb1), p(work), st))
        # value only
        loss2 = torch.zeros((), device=cuda)
        check(lib.rbg_infonce_batch_f32(p(ta), p(tb), d, p(ids), b, 0.2, 0.37, p(row_w), p(once), p(loss2), None, None, p(work), st))
    finally:
        rbg.set_option("deterministic", old)
    assert torch.isfinite(loss1) and abs(float(loss1) - float(loss0)) <= 1e-6 * max(1.0, abs(float(loss0)))
    assert abs(float(loss2) - float(loss0)) <= 1e-6 * max(1.0, abs(float(loss0)))
    for got, want in ((ga1, ga0), (gb1, gb0)):
        assert float((got - want).abs().max()) <= 1e-6 * max(1e-6, float(want.abs().max()))
    assert lib.rbg_infonce_batch_f32(p(ta), p(tb), d, p(ids), b, 0.2, 0.37, p(row_w), p(once), p(loss1), p(ga1), None, p(work), st) != 0


@pytest.mark.parametrize("n1,k,b,d,tau", [(3000, 100, 300, 64, 0.1), (500, 1000, 257, 64, 0.05), (2000, 37, 64, 128, 0.2), (90, 5, 33, 36, 1.0)])
def test_infonce_against_mapped_positives(rbg, cuda, n1, k, b, d, tau):
    """rbg_infonce_map_f32 (r06) == NCL's prototype contrast of one side (ncl.py:106-123) in float64 autograd: rows of a table against
    ALL k (unit) centroids, the positive of a row = its cluster's centroid; value, the rows' gradient (ids repeat), no table gradient;
    with a table gradient too (the general form)."""
    from recbole_gnn_amd._lib import lib, check, c_vp, c_i64
    import ctypes
    gen = torch.Generator().manual_seed(n1 + k + b)
    t1 = torch.randn(n1, d, generator=gen)
    cent = torch.nn.functional.normalize(torch.randn(k, d, generator=gen), dim=1)
    node2c = torch.randint(0, k, (n1,), generator=gen)
    idx = torch.randint(0, n1, (b,), generator=gen)
    idx[:3] = idx[0]
    w = 0.37
    a64, c64 = t1.double().requires_grad_(True), cent.double().requires_grad_(True)
    a = torch.nn.functional.normalize(a64[idx], dim=1)
    cn = torch.nn.functional.normalize(c64, dim=1)
    pos = (a * cn[node2c[idx]]).sum(1) / tau
    ref = w * (torch.logsumexp(a @ cn.T / tau, dim=1) - pos).sum()
    ref.backward()
    T1, C, N2C, IDX = t1.to(cuda), cent.to(cuda), node2c.to(cuda), idx.to(cuda)
    st = c_vp(torch.cuda.current_stream(cuda).cuda_stream)
    p = lambda t: c_vp(t.data_ptr())  # noqa: E731
    nbytes = c_i64()
    check(lib.rbg_infonce_workspace(b, k, d, ctypes.byref(nbytes)))
    work = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=cuda)
    g1, loss = torch.zeros_like(T1), torch.zeros((), device=cuda)
    check(lib.rbg_infonce_map_f32(p(T1), p(C), k, d, p(IDX), p(N2C), b, tau, w, p(loss), p(g1), None, p(work), st))
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float((g1.cpu().double() - a64.grad).abs().max()) <= 1e-5 * max(1e-3, float(a64.grad.abs().max()))
    g1b, g2b, loss_b = torch.zeros_like(T1), torch.zeros_like(C), torch.zeros((), device=cuda)
    check(lib.rbg_infonce_map_f32(p(T1), p(C), k, d, p(IDX), p(N2C), b, tau, w, p(loss_b), p(g1b), p(g2b), p(work), st))
    assert abs(float(loss_b) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float((g1b.cpu().double() - a64.grad).abs().max()) <= 1e-5 * max(1e-3, float(a64.grad.abs().max()))
    assert float((g2b.cpu().double() - c64.grad).abs().max()) <= 1e-5 * max(1e-3, float(c64.grad.abs().max()))
    loss_v = torch.zeros((), device=cuda)
    check(lib.rbg_infonce_map_f32(p(T1), p(C), k, d, p(IDX), p(N2C), b, tau, w, p(loss_v), None, None, p(work), st))
    assert abs(float(loss_v) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
