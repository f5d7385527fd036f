"""A minimal RecBole-free driver around the engine: just enough of what RecBole does
around the path [recbole==1.1.1] to train and evaluate a model end to end — atomic ``.inter`` reader, token -> id
remap (id 0 = [PAD]), per-user random split, uniform negative sampling, full-sort evaluation with the standard
top-k metrics.  It is NOT a re-implementation of RecBole's config / dataloader / trainer stack (out of scope).

Reference behaviour followed: ``tests/test_model.yaml`` (field names, RS split 0.8/0.1/0.1 grouped by user, full-sort
mode, topk 10), ``Trainer._train_epoch`` / ``_full_sort_batch_eval`` (per-batch loop, history + PAD masking) and
RecBole's metric definitions (Recall, MRR, NDCG, Hit, Precision @k).
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import InteractionDataset
from .graph import GraphHandle
from .train import GraphedStep, fused_stepper, total_without_last


def load_inter(path, user_field="user_id", item_field="item_id", sep="\t"):
    """Atomic file: first line ``name:type`` columns (tests/test_data/test/test.inter:1). Returns
    (uid, iid, n_users, n_items, user_tokens, item_tokens); ids are dense, 0 = [PAD], in order of first appearance."""
    with open(path) as f:
        header = [h.split(":")[0] for h in f.readline().rstrip("\n").split(sep)]
        ucol, icol = header.index(user_field), header.index(item_field)
        users, items = [], []
        for line in f:
            parts = line.rstrip("\n").split(sep)
            if len(parts) > max(ucol, icol):
                users.append(parts[ucol])
                items.append(parts[icol])
    return remap_tokens(users, items)


def remap_tokens(users, items):
    def factorize(tokens):
        table, ids = {}, np.empty(len(tokens), dtype=np.int64)
        for n, t in enumerate(tokens):
            ids[n] = table.setdefault(t, len(table) + 1)
        return ids, ["[PAD]"] + list(table)

    uid, utok = factorize(users)
    iid, itok = factorize(items)
    return uid, iid, len(utok), len(itok), utok, itok


def split_by_user(uid, iid, ratios=(0.8, 0.1, 0.1), seed=2020):
    """``eval_args: {split: {RS: [0.8,0.1,0.1]}, group_by: user, order: RO}``: each user's interactions are shuffled
    and cut by the ratios (RecBole rounds the later parts down and gives the remainder to the first)."""
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(uid))
    order = order[np.argsort(uid[order], kind="stable")]
    parts = [[], [], []]
    bounds = np.flatnonzero(np.diff(uid[order])) + 1
    for grp in np.split(order, bounds):
        n = len(grp)
        n_valid, n_test = int(n * ratios[1]), int(n * ratios[2])
        n_train = n - n_valid - n_test
        parts[0].append(grp[:n_train])
        parts[1].append(grp[n_train:n_train + n_valid])
        parts[2].append(grp[n_train + n_valid:])
    idx = [np.concatenate(p) if p else np.empty(0, dtype=np.int64) for p in parts]
    return [(uid[i], iid[i]) for i in idx]


class BPRSampler:
    """Pair-wise training batches: every training interaction with one uniformly sampled item the user has not
    interacted with in the training set (RecBole ``neg_sampling: {uniform: 1}``), reshuffled every epoch.

    ``device`` = a GPU: the interactions, the shuffle, the draws and the membership test (a binary search in the sorted
    (user, item) keys) stay in HBM, done once per epoch for all interactions; a batch is three slices of the epoch's arrays — no
    host or device work per batch (the host path costs 0.3 ms per batch; its first version, ``np.isin`` against a million
    keys, cost 52 ms per batch next to a 0.23 ms training step).  A
    draw that hits a training positive is redrawn in ``ROUNDS`` unconditional rounds (no host round trip to ask "any left?"):
    at the Gowalla shape a draw hits with probability 8e-4, so a positive survives 8 rounds with probability 2e-25."""

    ROUNDS = 8
    MAX_REDRAWS = 64  # checked redraws after the unconditional rounds; then the sampler raises instead of spinning

    def __init__(self, uid, iid, n_items, batch_size=2048, seed=2020, device=None):
        self.n_items, self.batch_size = int(n_items), int(batch_size)
        self.device = torch.device(device) if device is not None and torch.device(device).type == "cuda" else None
        uid, iid = np.asarray(uid, dtype=np.int64), np.asarray(iid, dtype=np.int64)
        if self.device is None:
            self.uid, self.iid = uid, iid
            self.rng = np.random.default_rng(seed)
            self.pos_keys = np.unique(uid * self.n_items + iid)
        else:
            self.uid, self.iid = torch.from_numpy(uid).to(self.device), torch.from_numpy(iid).to(self.device)
            self.gen = torch.Generator(device=self.device).manual_seed(int(seed))
            self.pos_keys = torch.unique(self.uid * self.n_items + self.iid)  # (sorted)

    def _is_positive(self, users, items):
        keys = users * self.n_items + items
        if self.device is None:
            at = np.minimum(np.searchsorted(self.pos_keys, keys), len(self.pos_keys) - 1)
        else:
            at = torch.searchsorted(self.pos_keys, keys).clamp_(max=self.pos_keys.numel() - 1)
        return self.pos_keys[at] == keys

    def _negatives(self, users):
        if self.device is None:
            neg = self.rng.integers(1, self.n_items, len(users))
            for _ in range(self.MAX_REDRAWS):
                bad = self._is_positive(users, neg)
                if not bad.any():
                    return neg
                neg[bad] = self.rng.integers(1, self.n_items, int(bad.sum()))
            raise RuntimeError(f"BPRSampler: draws still positive after {self.MAX_REDRAWS} redraws (a user who interacted with every item has no negative item)")
        draw = lambda: torch.randint(1, self.n_items, users.shape, generator=self.gen, device=self.device)  # noqa: E731
        neg = draw()
        for _ in range(self.ROUNDS):
            neg = torch.where(self._is_positive(users, neg), draw(), neg)
        # ROUNDS fixed redraws leave a positive with probability (degree / n_items)^(ROUNDS + 1) per draw: ~1e-25 at Gowalla's density,
        # but not on a dense dataset with heavy users.  One check per epoch (one host sync) and a loop for what is left: like the host
        # path and the reference's sampler, every returned item is a true negative.
        bad = self._is_positive(users, neg)
        for _ in range(self.MAX_REDRAWS):  # (bounded: a user who has interacted with EVERY item has no negative — ADVICE r05)
            if not bool(bad.any()):
                return neg
            idx = bad.nonzero(as_tuple=True)[0]
            neg[idx] = torch.randint(1, self.n_items, idx.shape, generator=self.gen, device=self.device)
            bad = self._is_positive(users, neg)
        if bool(bad.any()):
            raise RuntimeError(f"BPRSampler: {int(bad.sum())} draws still positive after {self.MAX_REDRAWS} redraws "
                               "(a user who interacted with every item has no negative item)")
        return neg

    def __iter__(self):
        if self.device is None:
            perm = self.rng.permutation(len(self.uid))
            for s in range(0, len(perm), self.batch_size):
                b = perm[s:s + self.batch_size]
                yield {"user_id": torch.from_numpy(self.uid[b]), "item_id": torch.from_numpy(self.iid[b]),
                       "neg_item_id": torch.from_numpy(self._negatives(self.uid[b]))}
            return
        # the whole epoch at once, like RecBole's sampler: one shuffle, one vectorised draw — a batch is three slices
        perm = torch.randperm(self.uid.numel(), generator=self.gen, device=self.device)
        users, items = self.uid[perm], self.iid[perm]
        neg = self._negatives(users)
        # (r06: the three columns in ONE [3, E] block — a batch's triples are one strided view of it, "_triples", which the captured steps
        #  copy into their static batch with one launch instead of three)
        trip = torch.stack([users, items, neg])
        for s in range(0, perm.numel(), self.batch_size):
            e = s + self.batch_size
            yield {"user_id": trip[0, s:e], "item_id": trip[1, s:e], "neg_item_id": trip[2, s:e], "_triples": trip[:, s:e]}

    def __len__(self):
        n = len(self.uid) if self.device is None else self.uid.numel()
        return (n + self.batch_size - 1) // self.batch_size


def topk_metrics(topk_idx, truth, k):
    """RecBole's Recall / MRR / NDCG / Hit / Precision @k for one batch.  topk_idx: [B, k] item ids (best first);
    truth: list of B sets (or arrays) of ground-truth items.  Returns per-user arrays.  Vectorised: (row, item) pairs
    are compared as sorted 64-bit keys."""
    topk_idx = np.asarray(topk_idx, dtype=np.int64)
    b = topk_idx.shape[0]
    n_truth = np.fromiter((len(t) for t in truth), dtype=np.int64, count=b)
    stride = int(max(topk_idx.max(initial=0), max((max(t) for t in truth if len(t)), default=0))) + 2
    rows = np.repeat(np.arange(b, dtype=np.int64), n_truth)
    items = np.fromiter((i for t in truth for i in t), dtype=np.int64, count=int(n_truth.sum()))
    truth_keys = np.sort(rows * stride + items)
    cand = np.arange(b, dtype=np.int64)[:, None] * stride + topk_idx
    pos = np.searchsorted(truth_keys, cand.ravel())
    pos[pos >= len(truth_keys)] = max(len(truth_keys) - 1, 0)
    hit = (truth_keys[pos] == cand.ravel()).reshape(b, k) if len(truth_keys) else np.zeros((b, k), dtype=bool)
    hit &= topk_idx >= 0  # -1 = fewer than k items were rankable
    n_truth = n_truth.astype(np.float64)
    n_hit = hit.sum(1).astype(np.float64)
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    dcg = (hit * disc).sum(1)
    idcg = np.concatenate([[0.0], np.cumsum(disc)])[np.minimum(n_truth, k).astype(np.int64)]
    first = np.where(hit.any(1), hit.argmax(1) + 1, 0)
    return {
        "recall": n_hit / np.maximum(n_truth, 1),
        "precision": n_hit / k,
        "hit": (n_hit > 0).astype(np.float64),
        "ndcg": dcg / np.maximum(idcg, 1e-12),
        "mrr": np.where(first > 0, 1.0 / np.maximum(first, 1), 0.0),
    }


def topk_metric_sums(topk_idx, rows, truth_keys, n_truth, stride, k):
    """The same five metrics, summed over the batch, as tensor ops on whatever device the inputs live on (no host work per
    batch).  topk_idx [B, k] int64 (-1 = fewer than k rankable items); rows [B] = the batch users' ranks in the sorted list of
    evaluated users; truth_keys: sorted unique ``rank * stride + item``; n_truth [n_eval_users]."""
    cand = rows[:, None] * stride + topk_idx
    at = torch.searchsorted(truth_keys, cand.reshape(-1)).clamp_(max=max(truth_keys.numel() - 1, 0)).reshape(cand.shape)
    hit = (truth_keys[at] == cand) & (topk_idx >= 0) if truth_keys.numel() else torch.zeros_like(cand, dtype=torch.bool)
    nt = n_truth[rows].to(torch.float64)
    n_hit = hit.sum(1).to(torch.float64)
    disc = 1.0 / torch.log2(torch.arange(2, k + 2, dtype=torch.float64, device=cand.device))
    dcg = (hit.to(torch.float64) * disc).sum(1)
    idcg = torch.cat([torch.zeros(1, dtype=torch.float64, device=cand.device), torch.cumsum(disc, 0)])[nt.clamp(max=k).long()]
    first = torch.where(hit.any(1), hit.to(torch.int64).argmax(1) + 1, torch.zeros_like(rows))
    return torch.stack([(n_hit / nt.clamp(min=1)).sum(), (n_hit / k).sum(), (n_hit > 0).to(torch.float64).sum(),
                        (dcg / idcg.clamp(min=1e-12)).sum(), torch.where(first > 0, 1.0 / first.clamp(min=1).to(torch.float64), 0.0).sum()])


METRIC_NAMES = ("recall", "precision", "hit", "ndcg", "mrr")


@torch.no_grad()
def evaluate(model, eval_uid, eval_iid, k=10, batch_users=4096, history=None, device_metrics=True):
    """Full-sort evaluation (``mode: full``): every user with ground truth in (eval_uid, eval_iid) is ranked against all
    items, PAD and history masked, by the fused score/top-k kernel; metrics averaged over users.
    ``history``: GraphHandle whose user rows are the interactions to mask (RecBole's sampler ``used_ids``: the training
    set for the valid phase, training + valid for the test phase); None = the model's training graph.
    ``device_metrics``: ground truth as sorted keys in HBM and the metrics as tensor ops next to the top-k lists (one host
    read at the end); False = the per-batch numpy path (``topk_metrics``), which the device path is tested against."""
    model.eval()
    model.restore_user_e = model.restore_item_e = None
    if device_metrics and model.device.type == "cuda":
        dev = model.device
        eu = torch.as_tensor(np.asarray(eval_uid, dtype=np.int64), device=dev)
        ei = torch.as_tensor(np.asarray(eval_iid, dtype=np.int64), device=dev)
        users, rank = torch.unique(eu, return_inverse=True)  # (sorted)
        stride = int(model.n_items) + 1
        truth_keys = torch.unique(rank * stride + ei)
        n_truth = torch.bincount(truth_keys // stride, minlength=users.numel())
        sums = torch.zeros(len(METRIC_NAMES), dtype=torch.float64, device=dev)
        for s in range(0, users.numel(), batch_users):
            ub = users[s:s + batch_users]
            _, idx = model.full_sort_topk({"user_id": ub}, k, history=history)
            sums += topk_metric_sums(idx, torch.arange(s, s + ub.numel(), device=dev), truth_keys, n_truth, stride, k)
        out = (sums / max(int(users.numel()), 1)).cpu().tolist()
        return {f"{name}@{k}": v for name, v in zip(METRIC_NAMES, out)}
    truth = {}
    for u, i in zip(eval_uid.tolist(), eval_iid.tolist()):
        truth.setdefault(u, set()).add(i)
    users = np.asarray(sorted(truth), dtype=np.int64)
    sums, count = {}, 0
    for s in range(0, len(users), batch_users):
        ub = users[s:s + batch_users]
        _, idx = model.full_sort_topk({"user_id": torch.from_numpy(ub).to(model.device)}, k, history=history)
        m = topk_metrics(idx.cpu().numpy(), [truth[u] for u in ub.tolist()], k)
        for name, v in m.items():
            sums[name] = sums.get(name, 0.0) + float(v.sum())
        count += len(ub)
    return {f"{name}@{k}": v / max(count, 1) for name, v in sums.items()}


def fit(model, train_uid, train_iid, epochs=1, lr=1e-3, batch_size=2048, seed=2020, fused=None, log=None, graphed=True,
        device_sampler=True):
    """``Trainer._train_epoch`` x epochs: zero_grad -> calculate_loss -> backward -> Adam step per batch.  Plain LightGCN /
    NGCF / SGL / SimGCL / XSimGCL / NCL models use their autograd-free step (``train.fused_stepper``; ``fused=False`` forces the autograd path, whose
    gradients the fused steps are tested against); any other model goes through torch autograd + torch.optim.Adam, the
    whole step captured in a HIP graph and replayed (``graphed``; the odd-sized last batch of an epoch runs eagerly;
    SimGCL and XSimGCL are captured too since r04: their contrastive batches are masked, not ``unique``-d, on the device).  Batches come from ``BPRSampler`` on
    the model's GPU (``device_sampler``; False = the numpy sampler)."""
    on_gpu = next(model.parameters()).is_cuda
    sampler = BPRSampler(train_uid, train_iid, model.n_items, batch_size=batch_size, seed=seed,
                         device=model.device if (device_sampler and on_gpu) else None)
    # the autograd-free steps (train.py): plain LightGCN, NGCF, SGL, SimGCL, XSimGCL, NCL
    stepper = fused_stepper(model, lr=lr, graphed=graphed) if fused in (None, True) else None
    if fused and stepper is None:
        raise TypeError(f"no fused training step for {type(model).__name__} in this configuration")
    fused = stepper is not None
    graphed = graphed and not fused and next(model.parameters()).is_cuda and getattr(model, "graph_capturable", True)
    opt = None if (fused or graphed) else torch.optim.Adam(model.parameters(), lr=lr, fused=next(model.parameters()).is_cuda)
    gstep = None
    history = []
    warm_up = getattr(model, "warm_up_step", None)  # NCLTrainer._train_epoch (trainer.py:130-133): the last loss term
    for epoch in range(epochs):                       # (the prototype contrast) joins after warm_up_step epochs
        if hasattr(model, "e_step") and epoch % max(int(getattr(model, "m_step", 1) or 1), 1) == 0:
            model.e_step()  # NCLTrainer.fit (trainer.py:38-40)
        model.train()
        if fused and hasattr(stepper, "with_proto"):  # NCL: the prototype term joins after warm_up_step epochs (trainer.py:130-133)
            stepper.with_proto = not (warm_up is not None and epoch < warm_up)
        total = torch.zeros((), device=model.device)
        own_total = fused and hasattr(stepper, "loss_total")  # (the stepper sums its losses itself: no add launch per step)
        if own_total:
            stepper.loss_total.zero_()
        for batch in sampler:
            batch = {k: v.to(model.device) for k, v in batch.items() if fused or not k.startswith("_")}  # ("_triples": for the captured steps)
            if own_total:
                stepper.step(batch)
            elif fused:
                total += stepper.step(batch)
            elif graphed:
                reduce = total_without_last if (warm_up is not None and epoch < warm_up) else None
                if gstep is not None:
                    gstep.set_reduce(reduce)  # (NCL: the prototype term joins after warm_up_step epochs — one re-capture)
                if gstep is None and len(batch["user_id"]) == min(batch_size, len(train_uid)):
                    gstep = GraphedStep(model, batch, lr=lr, reduce=reduce)
                if gstep is not None and len(batch["user_id"]) == len(gstep.static["user_id"]):
                    total += gstep.step(batch).detach().reshape(())
                elif gstep is not None:
                    total += gstep.eager_step(batch).reshape(())
                else:  # a first batch that is not full-sized: nothing captured yet
                    gstep = GraphedStep(model, batch, lr=lr, reduce=reduce)
                    total += gstep.step(batch).detach().reshape(())
            else:
                opt.zero_grad(set_to_none=True)
                loss = model.calculate_loss(batch)
                if isinstance(loss, tuple):  # RecBole's trainer sums tuple losses
                    loss = sum(loss[:-1] if (warm_up is not None and epoch < warm_up) else loss)
                loss.backward()
                opt.step()
                total += loss.detach().reshape(())
        history.append(float(stepper.loss_total if own_total else total))
        if log:
            log(f"epoch {epoch}: train loss {history[-1]:.4f}")
    return history


def run(model_cls, uid, iid, n_users, n_items, config=None, epochs=1, seed=2020, k=10, log=None):
    """quick_start.run_recbole_gnn in miniature (quick_start.py:9-63): split -> dataset from the TRAINING part ->
    model -> fit -> evaluate on valid and test."""
    (tr_u, tr_i), (va_u, va_i), (te_u, te_i) = split_by_user(uid, iid, seed=seed)
    dataset = InteractionDataset(tr_u, tr_i, n_users, n_items)  # the model's graph is built from train_data.dataset
    cfg = {"device": "cuda", "enable_sparse": True, "embedding_size": 64, "require_pow": True}
    cfg.update(config or {})
    torch.manual_seed(seed)
    model = model_cls(cfg, dataset)
    losses = fit(model, tr_u, tr_i, epochs=epochs, lr=cfg.get("learning_rate", 1e-3), seed=seed, log=log)
    # the test phase masks training AND validation positives (RecBole's test sampler used_ids), the valid phase training only
    seen = GraphHandle.from_interactions(np.concatenate([tr_u, va_u]), np.concatenate([tr_i, va_i]), n_users, n_items,
                                         device=model.device) if len(va_u) else None
    return {"model": model, "train_loss": losses, "valid": evaluate(model, va_u, va_i, k=k) if len(va_u) else {},
            "test": evaluate(model, te_u, te_i, k=k, history=seen) if len(te_u) else {}}
