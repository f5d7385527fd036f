"""The minimal RecBole-free driver: data pipeline and metrics on CPU; one-epoch runs (the reference's own test
strategy, tests/test_model.py: train and do not crash) plus a learning check on the GPU."""
import numpy as np
import pytest
import torch


def test_load_inter_and_remap(rbg, tmp_path):
    p = tmp_path / "toy.inter"
    p.write_text("user_id:token\titem_id:token\trating:float\ttimestamp:float\n"
                 "196\t242\t3\t881250949\n186\t302\t3\t891717742\n196\t377\t1\t878887116\n22\t242\t1\t1\n")
    uid, iid, nu, ni, utok, itok = rbg.driver.load_inter(str(p))
    assert uid.tolist() == [1, 2, 1, 3] and iid.tolist() == [1, 2, 3, 1]
    assert (nu, ni) == (4, 4) and utok[0] == "[PAD]" and utok[1] == "196" and itok[2] == "302"


def test_split_and_sampler(rbg, ref_inter):
    uid, iid, nu, ni = ref_inter
    (tr_u, tr_i), (va_u, va_i), (te_u, te_i) = rbg.driver.split_by_user(uid, iid, seed=1)
    assert len(tr_u) + len(va_u) + len(te_u) == len(uid)
    keys = lambda u, i: set((u * ni + i).tolist())  # noqa: E731
    assert keys(tr_u, tr_i) | keys(va_u, va_i) | keys(te_u, te_i) == keys(uid, iid)
    assert not (keys(tr_u, tr_i) & keys(te_u, te_i))
    deg = np.bincount(uid, minlength=nu)
    tr_deg = np.bincount(tr_u, minlength=nu)
    assert np.all(tr_deg[deg > 0] >= 1)                       # every user keeps a training interaction
    assert abs(len(tr_u) / len(uid) - 0.8) < 0.08
    sampler = rbg.driver.BPRSampler(tr_u, tr_i, ni, batch_size=512, seed=3)
    seen = 0
    pos = keys(tr_u, tr_i)
    for batch in sampler:
        u, p, n = (batch[k].numpy() for k in ("user_id", "item_id", "neg_item_id"))
        assert np.all(n >= 1) and not (keys(u, n) & pos)      # negatives are never training positives (nor PAD)
        assert keys(u, p) <= pos
        seen += len(u)
    assert seen == len(tr_u) and len(sampler) == (len(tr_u) + 511) // 512


def test_sampler_gives_up_on_a_user_without_negatives(rbg):
    """a user who interacted with EVERY item has no negative; the redraw loop is bounded and raises."""
    ni = 6
    uid = np.array([1] * (ni - 1) + [2], dtype=np.int64)
    iid = np.array(list(range(1, ni)) + [3], dtype=np.int64)
    sampler = rbg.driver.BPRSampler(uid, iid, ni, batch_size=4, seed=0)
    with pytest.raises(RuntimeError, match="no negative"):
        for _ in sampler:
            pass


def test_metrics_known_answers(rbg):
    topk = np.array([[5, 7, 9, 2], [1, 2, 3, 4], [8, 1, 2, 3]])
    truth = [{7, 2, 100}, {9}, {8}]
    m = rbg.driver.topk_metrics(topk, truth, 4)
    np.testing.assert_allclose(m["recall"], [2 / 3, 0, 1])
    np.testing.assert_allclose(m["precision"], [0.5, 0, 0.25])
    np.testing.assert_allclose(m["hit"], [1, 0, 1])
    np.testing.assert_allclose(m["mrr"], [0.5, 0, 1])
    d = 1 / np.log2(np.arange(2, 6))
    np.testing.assert_allclose(m["ndcg"][0], (d[1] + d[3]) / (d[0] + d[1] + d[2]))
    np.testing.assert_allclose(m["ndcg"][1], 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["LightGCN", "NGCF", "SGL", "SimGCL", "XSimGCL", "NCL"])
def test_one_epoch_runs(rbg, cuda, ref_inter, name):
    """tests/test_model.py:27-37 of the reference: build, train one epoch, evaluate — and the values are finite."""
    uid, iid, nu, ni = ref_inter
    cfg = {"device": str(cuda), "n_layers": 2, "reg_weight": 1e-5, "hidden_size_list": [64, 64]}
    if name == "NCL":  # the shipped 1000 clusters exceed this dataset's node counts; no warm-up so the prototype term trains
        cfg.update(num_clusters=8, warm_up_step=0, m_step=1)
    if name == "NGCF":
        cfg.update(node_dropout=0.1)  # the edge-dropout path (ngcf.py:74-90) trains too
    out = rbg.driver.run(getattr(rbg, name), uid, iid, nu, ni, config=cfg, epochs=1)
    assert np.isfinite(out["train_loss"][0])
    for split in ("valid", "test"):
        assert set(out[split]) == {"recall@10", "precision@10", "hit@10", "ndcg@10", "mrr@10"}
        assert all(0.0 <= v <= 1.0 for v in out[split].values())


@pytest.mark.gpu
def test_training_improves_ranking(rbg, cuda, ref_inter):
    uid, iid, nu, ni = ref_inter
    (tr_u, tr_i), (va_u, va_i), _ = rbg.driver.split_by_user(uid, iid, seed=2020)
    ds = rbg.InteractionDataset(tr_u, tr_i, nu, ni)
    torch.manual_seed(0)
    model = rbg.LightGCN({"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2,
                          "require_pow": True, "reg_weight": 1e-5}, ds)
    before = rbg.driver.evaluate(model, va_u, va_i, k=10)
    losses = rbg.driver.fit(model, tr_u, tr_i, epochs=30, lr=5e-3, batch_size=1024)
    after = rbg.driver.evaluate(model, va_u, va_i, k=10)
    assert losses[-1] < losses[0]
    assert after["recall@10"] > before["recall@10"] + 0.02 and after["ndcg@10"] > before["ndcg@10"]
    # the torch-autograd training path reaches the same quality region as the fused one
    torch.manual_seed(0)
    model2 = rbg.LightGCN({"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2,
                           "require_pow": True, "reg_weight": 1e-5}, ds)
    rbg.driver.fit(model2, tr_u, tr_i, epochs=30, lr=5e-3, batch_size=1024, fused=False)
    after2 = rbg.driver.evaluate(model2, va_u, va_i, k=10)
    assert abs(after2["recall@10"] - after["recall@10"]) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SimGCL", "XSimGCL"])
def test_fit_trains_the_subclass_objective(rbg, cuda, ref_inter, name):
    """driver.fit on a LightGCN SUBCLASS (SimGCL / XSimGCL: contrastive terms, perturbed passes, a layer mean without E0)
    must train that model's own calculate_loss — not LightGCN's fused BPR step.  Since r05 fit() picks their own autograd-free
    steps (train.FusedSimGCLAdam / FusedXSimGCLAdam): one epoch through fit() equals the eager zero_grad / calculate_loss /
    backward / Adam loop on a twin model with the same seeds up to summation order, and so does the autograd path (fused=False)."""
    uid, iid, nu, ni = ref_inter
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    cfg = {"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2, "require_pow": True}
    torch.manual_seed(1)
    a = getattr(rbg, name)(cfg, ds)
    torch.manual_seed(1)
    b = getattr(rbg, name)(cfg, ds)
    assert not rbg.train.fused_step_applies(a) and a.graph_capturable  # (static_unique: the same loss with static shapes)
    with pytest.raises(TypeError):
        rbg.FusedBPRAdam(a)
    assert type(rbg.fused_stepper(a)).__name__ == f"Fused{name}Adam"
    torch.manual_seed(1)
    c = getattr(rbg, name)(cfg, ds)
    torch.manual_seed(77)  # (graphed=False: a replayed step draws its noise from the graph's own generator state)
    la = rbg.driver.fit(a, uid, iid, epochs=1, lr=1e-3, batch_size=512, seed=5, device_sampler=False, graphed=False)
    torch.manual_seed(77)
    lc = rbg.driver.fit(c, uid, iid, epochs=1, lr=1e-3, batch_size=512, seed=5, device_sampler=False, graphed=False, fused=False)
    torch.manual_seed(77)
    opt = torch.optim.Adam(b.parameters(), lr=1e-3)
    b.train()
    total = 0.0
    for batch in rbg.driver.BPRSampler(uid, iid, ni, batch_size=512, seed=5):
        batch = {k: v.to(cuda) for k, v in batch.items()}
        opt.zero_grad(set_to_none=True)
        loss = b.calculate_loss(batch)
        loss = sum(loss) if isinstance(loss, tuple) else loss
        loss.backward()
        opt.step()
        total += float(loss)
    assert abs(lc[0] - total) <= 1e-4 * max(1.0, abs(total))
    assert abs(la[0] - total) <= 1e-3 * max(1.0, abs(total))
    for pa, pb, pc in zip(a.parameters(), b.parameters(), c.parameters()):
        assert float((pc - pb).abs().max()) <= 5e-6                                             # the autograd path: the same launches (torch float-atomic index_add_: 1.3e-6 seen once)
        assert float((pa - pb).abs().max()) <= 2e-3 * max(1.0, float(pb.abs().max()))           # the fused step: rounding order


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["NGCF", "SGL"])
def test_fit_uses_the_autograd_free_step(rbg, cuda, ref_inter, name):
    """driver.fit picks train.FusedNGCFAdam / FusedSGLAdam for the plain models (replayed from a HIP graph, the epoch's shorter
    last batch enqueued eagerly): two epochs give the losses of the autograd path (``fused=False``) on a twin model."""
    uid, iid, nu, ni = ref_inter
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    cfg = {"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2, "hidden_size_list": [64, 64], "message_dropout": 0.0,
           "node_dropout": 0.0, "reg_weight": 1e-4, "device_sampling": False}
    out = []
    for fused in (None, False):
        torch.manual_seed(1)
        np.random.seed(7)  # (SGL's views, sampled by model.train() at the start of every epoch)
        m = getattr(rbg, name)(cfg, ds)
        assert isinstance(rbg.fused_stepper(m), rbg.FusedNGCFAdam if name == "NGCF" else rbg.FusedSGLAdam)
        out.append((rbg.driver.fit(m, uid, iid, epochs=2, lr=1e-3, batch_size=500, seed=5, fused=fused), m))
    (la, ma), (lb, mb) = out
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), (la, lb)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert float((pa - pb).abs().max()) <= 2e-3 * max(1.0, float(pb.abs().max()))
    assert isinstance(rbg.fused_stepper(rbg.SimGCL({"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2}, ds)),
                      rbg.FusedSimGCLAdam)


@pytest.mark.gpu
def test_device_sampler(rbg, cuda, ref_inter):
    """BPRSampler on the GPU: every interaction once per epoch, negatives never training positives (nor PAD), reproducible from
    the seed, reshuffled and redrawn every epoch — the host sampler's contract with no host work per batch."""
    uid, iid, nu, ni = ref_inter
    pos = set((uid * ni + iid).tolist())

    def epoch(s):
        out = [tuple(b[k].cpu().numpy() for k in ("user_id", "item_id", "neg_item_id")) for b in s]
        assert all(b["user_id"].is_cuda for b in [next(iter(s))])
        return [np.concatenate(c) for c in zip(*out)]

    s = rbg.driver.BPRSampler(uid, iid, ni, batch_size=300, seed=11, device=cuda)
    u1, p1, n1 = epoch(s)
    u2, p2, n2 = epoch(s)
    for u, p, n in ((u1, p1, n1), (u2, p2, n2)):
        assert len(u) == len(uid) and sorted((u * ni + p).tolist()) == sorted((uid * ni + iid).tolist())
        assert n.min() >= 1 and n.max() < ni and not (set((u * ni + n).tolist()) & pos)
    assert not np.array_equal(u1, u2) and len(s) == (len(uid) + 299) // 300
    u3, p3, n3 = epoch(rbg.driver.BPRSampler(uid, iid, ni, batch_size=300, seed=11, device=cuda))
    assert np.array_equal(u1, u3) and np.array_equal(p1, p3) and np.array_equal(n1, n3)
    # a user who interacted with all items but one: the only legal negative is found
    small = 40
    dense_u = np.ones(small - 2, dtype=np.int64)
    dense_i = np.arange(1, small - 1, dtype=np.int64)
    s = rbg.driver.BPRSampler(dense_u, dense_i, small, batch_size=16, seed=0, device=cuda)
    s.ROUNDS = 400  # (a draw is legal with probability 1/39 here: 400 redraw rounds leave 3e-5 of the draws on a positive)
    negs = np.concatenate([b["neg_item_id"].cpu().numpy() for b in s])
    assert np.all(negs == small - 1)


def test_metric_sums_as_tensor_ops(rbg):
    """driver.topk_metric_sums (what evaluate() runs next to the top-k lists) == driver.topk_metrics summed, incl. -1 slots,
    users without hits and a ground truth longer than k."""
    rng = np.random.default_rng(0)
    n_items, k, b = 50, 7, 40
    truth = [set(rng.choice(np.arange(1, n_items), size=rng.integers(1, 12), replace=False).tolist()) for _ in range(b)]
    topk = np.stack([rng.choice(np.arange(1, n_items), size=k, replace=False) for _ in range(b)])
    topk[3, 4:] = -1
    topk[7] = -1
    ref = rbg.driver.topk_metrics(topk, truth, k)
    stride = n_items + 1
    keys = torch.tensor(sorted(r * stride + i for r, t in enumerate(truth) for i in t))
    n_truth = torch.tensor([len(t) for t in truth])
    got = rbg.driver.topk_metric_sums(torch.from_numpy(topk), torch.arange(b), keys, n_truth, stride, k)
    for name, v in zip(rbg.driver.METRIC_NAMES, got.tolist()):
        assert abs(v - float(ref[name].sum())) <= 1e-9, name
    half = rbg.driver.topk_metric_sums(torch.from_numpy(topk[20:]), torch.arange(20, b), keys, n_truth, stride, k)  # a later batch
    assert abs(float(half[0]) - float(ref["recall"][20:].sum())) <= 1e-9


@pytest.mark.gpu
def test_evaluate_on_the_device_equals_the_numpy_path(rbg, cuda, ref_inter):
    uid, iid, nu, ni = ref_inter
    (tr_u, tr_i), (va_u, va_i), (te_u, te_i) = rbg.driver.split_by_user(uid, iid, seed=2020)
    ds = rbg.InteractionDataset(tr_u, tr_i, nu, ni)
    torch.manual_seed(0)
    model = rbg.LightGCN({"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2}, ds)
    dup_u, dup_i = np.concatenate([te_u, te_u[:5]]), np.concatenate([te_i, te_i[:5]])  # repeated pairs count once
    for eu, ei in ((va_u, va_i), (dup_u, dup_i)):
        a = rbg.driver.evaluate(model, eu, ei, k=10, batch_users=100)
        b = rbg.driver.evaluate(model, eu, ei, k=10, batch_users=100, device_metrics=False)
        assert set(a) == set(b)
        for name in a:
            assert abs(a[name] - b[name]) <= 1e-12, name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SimGCL", "XSimGCL"])
def test_static_unique_is_the_unique_loss(rbg, cuda, ref_inter, name):
    """simgcl.py:52-53 / xsimgcl.py:86-87 contrast the ``torch.unique`` ids of the batch; the default ``static_unique`` form keeps one
    occurrence per id by a mask over rows and columns: the same value and gradients on batches full of repeats (same noise
    draws), and the step can be captured — fit() replays it from a HIP graph for more than one epoch."""
    uid, iid, nu, ni = ref_inter
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    cfg = {"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2, "require_pow": True}
    torch.manual_seed(1)
    a = getattr(rbg, name)(cfg, ds)
    torch.manual_seed(1)
    b = getattr(rbg, name)(dict(cfg, static_unique=False), ds)
    assert a.graph_capturable and not b.graph_capturable
    gen = torch.Generator().manual_seed(3)
    batch = {"user_id": torch.randint(1, 40, (256,), generator=gen).to(cuda), "item_id": torch.randint(1, 60, (256,), generator=gen).to(cuda),
             "neg_item_id": torch.randint(1, ni, (256,), generator=gen).to(cuda)}
    out = []
    for m in (a, b):
        m.train()
        torch.manual_seed(9)
        loss = m.calculate_loss(batch)
        loss = sum(loss) if isinstance(loss, tuple) else loss
        loss.backward()
        out.append((float(loss.detach()), [p.grad.clone() for p in m.parameters()]))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * max(1.0, abs(out[1][0]))
    for ga, gb in zip(out[0][1], out[1][1]):
        assert float((ga - gb).abs().max()) <= 1e-5 * max(1e-6, float(gb.abs().max()))
    with pytest.raises(RuntimeError):
        rbg.GraphedStep(b, batch)
    hist = rbg.driver.fit(a, uid, iid, epochs=3, lr=1e-3, batch_size=500, seed=5)  # HIP-graph replay + an odd-sized last batch per epoch
    assert all(np.isfinite(h) for h in hist) and hist[-1] < hist[0]


@pytest.mark.gpu
def test_ncl_trains_through_a_replayed_step(rbg, cuda, ref_inter):
    """NCL through driver.fit: the autograd-free step (train.FusedNCLAdam, r05), captured and eager, and the autograd step
    captured by GraphedStep (fused=False) — e_step rewrites the prototypes in place, the prototype term joins after warm_up_step
    epochs (one re-capture) — give the same losses."""
    uid, iid, nu, ni = ref_inter
    ds = rbg.InteractionDataset(uid, iid, nu, ni)
    cfg = {"device": str(cuda), "enable_sparse": True, "embedding_size": 64, "n_layers": 2, "num_clusters": 8, "warm_up_step": 1, "m_step": 1,
           "proto_reg": 1e-3, "ssl_reg": 1e-5}
    out = []
    for graphed, fused in ((True, None), (False, None), (True, False)):
        torch.manual_seed(1)
        m = rbg.NCL(cfg, ds)
        assert m.graph_capturable and isinstance(rbg.fused_stepper(m), rbg.FusedNCLAdam)
        hist = rbg.driver.fit(m, uid, iid, epochs=3, lr=1e-3, batch_size=500, seed=5, graphed=graphed, fused=fused)
        out.append((hist, m))
        cent = m.user_centroids
        m.e_step()
        assert m.user_centroids is cent  # rewritten in place
    (ha, ma), (hb, mb), (hc, mc) = out
    for x, y, z in zip(ha, hb, hc):
        assert abs(x - z) <= 1e-2 * max(1.0, abs(z)) and abs(y - z) <= 1e-2 * max(1.0, abs(z)), (ha, hb, hc)
    assert ha[1] > 0 and np.isfinite(ha).all()
